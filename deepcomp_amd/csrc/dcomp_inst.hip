// dcomp_inst.hip -- compiled once per base-station count: hipcc -DDCOMP_B=<B> ... -o dcomp_inst_b<B>.o
// Instantiates step/reset kernels for every UE-group width (UPAD) at this B.
#include "dcomp_device.h"

#ifndef DCOMP_B
#error "compile with -DDCOMP_B=<number of base stations>"
#endif

#define DCOMP_CAT_(a, b) a##b
#define DCOMP_CAT(a, b) DCOMP_CAT_(a, b)

namespace dcomp {
KernelPair DCOMP_CAT(kernels_b, DCOMP_B)(int upad, int mp) { return kernels_for_upad<DCOMP_B>(upad, mp); }
}  // namespace dcomp
