// dcomp_big.hip -- instantiates the generic kernel of dcomp_big.h (any station count up to 64) for every lane-group width: one object,
// 54 kernels (step, reset, step with UE arrival / departure; each also as the instantiation that writes the compact record), whatever the station count.
#include "dcomp_big.h"

namespace dcomp {
template <int UPAD>
static BigKernels big_make()
{
    return BigKernels{big_kernel<UPAD, false, false, false>, big_kernel<UPAD, true, false, false>, big_kernel<UPAD, false, true, false>,
                      big_kernel<UPAD, false, false, true>, big_kernel<UPAD, true, false, true>, big_kernel<UPAD, false, true, true>, big_block(UPAD) / UPAD, big_block(UPAD)};
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
    default: return BigKernels{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    }
}
}  // namespace dcomp
