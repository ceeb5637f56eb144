// Microbenchmark (GPU box): a synthetic "compute phase, then 33 KB of row stores per wave" kernel shaped like step_kernel_wide
// (256-thread workgroups, 40 KB of LDS -> four per CU, ~3 us of dependent VALU work per wave), to see which launch structure
// overlaps the compute of some waves with the stores of others.
//   hipcc --offload-arch=gfx950 -O3 -o overlap_model tools/micro/overlap_model.hip && ./overlap_model
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int ROW = 129;

__device__ __forceinline__ float compute(float x, int n)
{
#pragma unroll 8
    for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 0.25f);
    return x;
}
__device__ __forceinline__ void rows(float *base, int lane, float val)
{
    const int h = lane >> 5, j = lane & 31;
    for (int i = 0; i < 32; i++) {
        f4u v; v.x = v.y = v.z = v.w = val;
        *reinterpret_cast<f4u *>(base + (2 * i + h) * ROW + 4 * j) = v;
        if ((lane >> 1) == i) base[lane * ROW + 128] = val;
    }
}

__device__ __forceinline__ void rows_aligned_nt(float *base, int lane, float val, bool nt)
{
    for (int i = 0; i < 33; i++) {
        const int f = i * 256 + 4 * lane;
        if (f < 64 * ROW) {
            f4u v; v.x = v.y = v.z = v.w = val;
            f4u *dst = reinterpret_cast<f4u *>(base + f);
            if (nt) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
}

// 10 / 11: like 9 with the rows as line-aligned 1 KiB store instructions, non-temporal / plain.
// MODE 0: one workgroup per slot.  1: persistent (slot loop).  2: one per slot, stores at s_setprio 3.  3: compute only.  4: stores only.
// 5: persistent, and half of the workgroups start with a dummy compute phase (phase offset).
// 6: like 0, the compute starts from a 32-byte-per-lane state LOAD (as the step does).  7: like 0 with three workgroup barriers inside
// the compute phase.  8: 6 + 7.  9: 8 + the state is written back (32 B per lane) in the middle of the compute phase.
template <int MODE>
__global__ __launch_bounds__(256) void k(float *obs, int slots, int nfma, float *sink, float4 *state)
{
    __shared__ float lds[10000];                           // 40 KB: four workgroups per CU, like the wide kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    float acc = lds[(threadIdx.x * 7) & 255];
    if (MODE == 5 && (blockIdx.x & 1)) acc = compute(acc, nfma / 2);
    const int stride = (MODE == 1 || MODE == 5) ? gridDim.x : slots;
    for (int g = blockIdx.x; g < slots; g += stride) {
        if (MODE >= 6) {
            float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
            if (MODE != 7) { a4 = state[((size_t)g * 256 + threadIdx.x) * 2]; b4 = state[((size_t)g * 256 + threadIdx.x) * 2 + 1]; }
            acc += a4.x + b4.y;
            for (int q = 0; q < 4; q++) {
                acc = compute(acc + (float)g, nfma / 4);
                if (MODE >= 7 && q < 3) { lds[threadIdx.x] = acc; __syncthreads(); acc += lds[(threadIdx.x + 64) & 255]; }
                if (MODE >= 9 && q == 2) { state[((size_t)g * 256 + threadIdx.x) * 2] = make_float4(acc, a4.y, a4.z, a4.w); state[((size_t)g * 256 + threadIdx.x) * 2 + 1] = b4; }
            }
        } else
        if (MODE != 4) acc = compute(acc + (float)g, nfma);
        if (MODE == 2) __builtin_amdgcn_s_setprio(3);
        if (MODE == 10 || MODE == 11) rows_aligned_nt(obs + ((size_t)g * 4 + wave) * 64 * ROW, lane, acc, MODE == 10);
        else if (MODE != 3) rows(obs + ((size_t)g * 4 + wave) * 64 * ROW, lane, acc);
        if (MODE == 2) __builtin_amdgcn_s_setprio(0);
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
static void run(const char *name, float *obs, float *sink, float4 *state, int envs, int nfma, int grid_cap)
{
    const int slots = envs / 2;                             // a workgroup = 4 waves = 2 envs of 128 UEs
    const int grid = (MODE == 1 || MODE == 5) ? (grid_cap < slots ? grid_cap : slots) : slots;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, obs, slots, nfma, sink, state);
    (void)hipEventRecord(a);
    const int n = 50;
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, obs, slots, nfma, sink, state);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)slots * 4 * 64 * ROW * 4;
    printf("%-58s envs %6d: %8.1f us  (%5.0f GB/s of rows)\n", name, envs, ms / n * 1e3, MODE == 3 ? 0.0 : bytes / (ms / n * 1e-3) / 1e9);
}

int main()
{
    float *obs, *sink;
    (void)hipMalloc(&obs, (size_t)32768 * 128 * ROW * 4);
    (void)hipMalloc(&sink, 64);
    float4 *state;
    (void)hipMalloc(&state, (size_t)16384 * 256 * 32);
    (void)hipMemset(state, 0, (size_t)16384 * 256 * 32);
    const int nfma = 1500;
    for (int envs : {4096, 8192, 32768}) {
        run<3>("3 compute only", obs, sink, state, envs, nfma, 1024);
        run<4>("4 stores only", obs, sink, state, envs, nfma, 1024);
        run<0>("0 one workgroup per slot", obs, sink, state, envs, nfma, 1024);
        run<2>("2 one workgroup per slot, stores at s_setprio 3", obs, sink, state, envs, nfma, 1024);
        run<1>("1 persistent, 1024 workgroups", obs, sink, state, envs, nfma, 1024);
        run<1>("1 persistent, 512 workgroups", obs, sink, state, envs, nfma, 512);
        run<1>("1 persistent, 768 workgroups", obs, sink, state, envs, nfma, 768);
        run<5>("5 persistent 1024, odd workgroups offset by half a phase", obs, sink, state, envs, nfma, 1024);
        run<6>("6 one per slot, compute starts from a state load", obs, sink, state, envs, nfma, 1024);
        run<7>("7 one per slot, three workgroup barriers in the compute", obs, sink, state, envs, nfma, 1024);
        run<8>("8 state load + three barriers", obs, sink, state, envs, nfma, 1024);
        run<9>("9 ... + state written back mid-compute", obs, sink, state, envs, nfma, 1024);
        run<10>("10 like 9, rows line-aligned 1 KiB non-temporal", obs, sink, state, envs, nfma, 1024);
        run<11>("11 like 9, rows line-aligned 1 KiB plain", obs, sink, state, envs, nfma, 1024);
    }
    return 0;
}
