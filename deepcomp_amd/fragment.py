"""Compact rollout fragments for the learner hand-off (include/dcomp.h: dcomp_pack_fragment / dcomp_unpack_fragment).

The reference ships sample batches from its rollout workers to the trainer through Ray's object store (util/env_setup.py:266,
util/simulation.py:143); here a fragment of multi-agent observations [..., U, 4B+1] crosses xGMI in one RCCL all-gather
(deepcomp_amd/sharded.py).  Of the 4B+1 floats per UE, ``ues_at_bs`` / ``util_at_bs`` are per-ENV values replicated U times and
``connected`` is B bits (single_ue/variants.py:271-305): the compact record is 3.2-3.7x smaller and unpacks BIT-identically.
"""
import ctypes

import torch

from . import _lib


def fragment_words(num_ue, num_bs):
    n = _lib.load().dcomp_fragment_words(int(num_ue), int(num_bs))
    if n < 0:
        raise ValueError(f"num_ue={num_ue} / num_bs={num_bs} outside the library's limits")
    return n


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class FragmentCodec:
    """pack / unpack for observation tensors of one shape [..., U, 4B+1] on one device.  The flag word pack raises when its input
    is NOT an observation tensor (replicated columns that differ, `connected` entries that are not 0 / 1) is read by check()."""

    def __init__(self, num_ue, num_bs, device='cuda'):
        self.U, self.B = int(num_ue), int(num_bs)
        self.words = fragment_words(self.U, self.B)
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self._L = _lib.load()
        self.flags = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _n(self, t, inner):
        if t.device != self.device or not t.is_contiguous() or t.numel() % inner:
            raise ValueError(f"need a contiguous tensor on {self.device} whose size is a multiple of {inner}")
        return t.numel() // inner

    def pack(self, obs, out=None):
        """obs float32 [..., U, 4B+1] -> int32 [..., words] (the leading dimensions are kept)."""
        if obs.dtype != torch.float32 or obs.shape[-2:] != (self.U, 4 * self.B + 1):
            raise ValueError(f"obs must be float32 [..., {self.U}, {4 * self.B + 1}]")
        n = self._n(obs, self.U * (4 * self.B + 1))
        if out is None:
            out = torch.empty(tuple(obs.shape[:-2]) + (self.words,), dtype=torch.int32, device=self.device)
        elif out.dtype != torch.int32 or self._n(out, self.words) != n:
            raise ValueError(f"out must be int32 with {n} x {self.words} elements")
        with torch.cuda.device(self.device):
            _lib.check(self._L.dcomp_pack_fragment(obs.data_ptr(), n, self.U, self.B, out.data_ptr(), self.flags.data_ptr(), _stream(self.device)))
        return out

    def unpack(self, packed, out=None):
        """int32 [..., words] -> float32 [..., U, 4B+1], bit-identical to what pack() was given."""
        if packed.dtype != torch.int32 or packed.shape[-1] != self.words:
            raise ValueError(f"packed must be int32 [..., {self.words}]")
        n = self._n(packed, self.words)
        if out is None:
            out = torch.empty(tuple(packed.shape[:-1]) + (self.U, 4 * self.B + 1), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or self._n(out, self.U * (4 * self.B + 1)) != n:
            raise ValueError(f"out must be float32 with {n} x {self.U} x {4 * self.B + 1} elements")
        with torch.cuda.device(self.device):
            _lib.check(self._L.dcomp_unpack_fragment(packed.data_ptr(), n, self.U, self.B, out.data_ptr(), _stream(self.device)))
        return out

    def check(self):
        """Synchronises; raises if a pack() since the last check() was handed something that is not an observation tensor."""
        f = int(self.flags.item())
        if f:
            self.flags.zero_()
            raise ValueError("pack_fragment: input is not a multi-agent observation tensor (" +
                             ", ".join(m for b, m in ((1, "per-env columns differ between the rows of an env"), (2, "`connected` entry that is not 0 / 1")) if f & b) + ")")
