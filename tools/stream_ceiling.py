"""What HBM rate does a pure streaming kernel reach on this GPU for the step kernel's read/write mix?
(69 MB read + 437 MB written per launch at config 3.)  torch's elementwise kernels, HIP-event timed."""
import torch

def t(fn, n=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

dev = 'cuda'
W, R = 436_500_000 // 4, 69_500_000 // 4
out = torch.empty(W, dtype=torch.float32, device=dev)
src = torch.empty(R, dtype=torch.float32, device=dev).normal_()
big = torch.empty(W, dtype=torch.float32, device=dev).normal_()
ms = t(lambda: out.fill_(1.0));                 print(f'fill   437 MB            : {ms*1e3:7.1f} us  {W*4/ms/1e6:7.0f} GB/s')
ms = t(lambda: out.copy_(big));                 print(f'copy   437 MB -> 437 MB  : {ms*1e3:7.1f} us  {2*W*4/ms/1e6:7.0f} GB/s')
ms = t(lambda: torch.mul(big, 2.0, out=out));   print(f'scale  437 MB -> 437 MB  : {ms*1e3:7.1f} us  {2*W*4/ms/1e6:7.0f} GB/s')
def mix():
    out.fill_(1.0); src.mul_(1.0001)
ms = t(mix);                                    print(f'fill 437 + rmw 69.5 MB   : {ms*1e3:7.1f} us  {(W*4+2*R*4)/ms/1e6:7.0f} GB/s')
v = out[:R * 6].view(6, R)
ms = t(lambda: torch.add(src, 1.0, out=v[0]));  print(f'read 69.5 -> write 69.5  : {ms*1e3:7.1f} us  {2*R*4/ms/1e6:7.0f} GB/s')
ms = t(lambda: big.sum());                      print(f'read-only sum 437 MB     : {ms*1e3:7.1f} us  {W*4/ms/1e6:7.0f} GB/s')
