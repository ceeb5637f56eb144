// dcomp_big.hip -- instantiates the generic kernel of dcomp_big.h (any station count up to 64) for every lane-group width: one object,
// 144 kernels (step, reset, step with UE arrival / departure, fused rollout; each also as the instantiation that writes the compact record, and each of
// those again with the in-step heuristic policy), whatever the station count.
#include "dcomp_big.h"

namespace dcomp {
template <int UPAD, bool POL, bool COMPACT>
static void big_fill(BigKernelFn (&f)[4])
{
    f[0] = big_kernel<UPAD, false, false, COMPACT, POL>;
    f[1] = big_kernel<UPAD, true, false, COMPACT, POL>;
    f[2] = big_kernel<UPAD, false, true, COMPACT, POL>;
    f[3] = big_kernel<UPAD, false, false, COMPACT, POL, true>;
}

template <int UPAD>
static BigKernels big_make()
{
    BigKernels k{};
    big_fill<UPAD, false, false>(k.fn[0][0]); big_fill<UPAD, false, true>(k.fn[0][1]);
    big_fill<UPAD, true, false>(k.fn[1][0]); big_fill<UPAD, true, true>(k.fn[1][1]);
    k.gpb = big_block(UPAD) / UPAD; k.block = big_block(UPAD);
    return k;
}

BigKernels big_kernels_for_upad(int upad)
{
    switch (upad) {
    case 1: case 2: case 4: return big_make<4>();
    case 8: return big_make<8>();
    case 16: return big_make<16>();
    case 32: return big_make<32>();
    case 64: return big_make<64>();
    case 128: return big_make<128>();
    case 256: return big_make<256>();
    case 512: return big_make<512>();
    case 1024: return big_make<1024>();
    default: return BigKernels{};
    }
}
}  // namespace dcomp
