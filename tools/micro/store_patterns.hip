// Microbenchmark (GPU box): how fast do the observation-row store PATTERNS of the wide kernel stream, with no compute at all?
// One wavefront writes 64 rows of ROW = 129 floats (33 024 contiguous bytes), as step_kernel_wide does for 128 x 32.
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns tools/micro/store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int ROW = 129;

template <int PAT>
__global__ __launch_bounds__(256) void k(float *obs, int waves_total, int persistent_stride)
{
    const int lane = threadIdx.x & 63;
    int blk = blockIdx.x;
    if (PAT >= 7) {                          // XCD-aware: workgroup b runs on XCD b % 8; give every XCD one contiguous eighth of the buffer
        const int per = gridDim.x / 8;
        blk = (blockIdx.x % 8) * per + blockIdx.x / 8;
    }
    for (int w = blk * 4 + (threadIdx.x >> 6); w < waves_total; w += persistent_stride) {
        float *base = obs + (size_t)w * 64 * ROW;
        const float val = (float)(w + lane);
        if (PAT == 0) {                       // round 3: row by row, 4-byte columns (two store instructions per row + utility)
            for (int r = 0; r < 64; r++) {
                float *o = base + r * ROW;
                o[lane] = val; o[lane + 64] = val;
                if ((r & 3) == 3 && lane < 4) base[(r - 3 + lane) * ROW + 128] = val;
            }
        } else if (PAT == 1 || PAT == 2) {    // round 4: half-wave per row, 16-byte pieces at dword alignment (+ utility every 8 rows)
            const int h = lane >> 5, j = lane & 31;
            for (int i = 0; i < 32; i++) {
                f4u v; v.x = v.y = v.z = v.w = val;
                f4u *dst = reinterpret_cast<f4u *>(base + (2 * i + h) * ROW + 4 * j);
                if (PAT == 2) __builtin_nontemporal_store(v, dst); else *dst = v;
                if ((i & 3) == 3 && (lane >> 3) == (i >> 2)) base[lane * ROW + 128] = val;
            }
        } else if (PAT == 3 || PAT == 4 || PAT == 8 || PAT == 9) {    // line-aligned 1 KiB per store instruction (what LDS staging in memory layout would give)
            for (int i = 0; i < 33; i++) {
                const int f = i * 256 + 4 * lane;
                if (f < 64 * ROW) {
                    f4u v; v.x = v.y = v.z = v.w = val;
                    f4u *dst = reinterpret_cast<f4u *>(base + f);
                    if (PAT == 4 || PAT == 9) __builtin_nontemporal_store(v, dst); else *dst = v;
                }
            }
        } else if (PAT == 5 || PAT == 7) {                // like 1, utility float right with its rows (2 lanes per iteration)
            const int h = lane >> 5, j = lane & 31;
            for (int i = 0; i < 32; i++) {
                f4u v; v.x = v.y = v.z = v.w = val;
                *reinterpret_cast<f4u *>(base + (2 * i + h) * ROW + 4 * j) = v;
                if ((lane >> 1) == i) base[lane * ROW + 128] = val;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sweep(float *obs, size_t total_kib, int nt)
{
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (size_t k = w; k < total_kib; k += waves) {
        f4u v; v.x = v.y = v.z = v.w = (float)k;
        f4u *dst = reinterpret_cast<f4u *>(obs + k * 256 + 4 * lane);
        if (nt) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
}

template <int PAT>
static void run(const char *name, float *obs, int envs, int grid_cap)
{
    const int waves = envs * 2, blocks = (waves + 3) / 4;
    const int grid = grid_cap > 0 && grid_cap < blocks ? grid_cap : blocks;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, obs, waves, grid * 4);
    hipEventRecord(a);
    const int n = 100;
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, obs, waves, grid * 4);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)waves * 64 * ROW * 4;
    printf("%-46s envs %6d grid %5d: %8.1f us  %6.0f GB/s\n", name, envs, grid, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e9);
}

int main()
{
    const int sizes[] = {4096, 8192, 32768};
    float *obs;
    hipMalloc(&obs, (size_t)32768 * 128 * ROW * 4);
    for (int envs : sizes) {
        for (int cap : {0, 1024}) {
            run<0>("0 four-byte columns, row by row", obs, envs, cap);
            run<1>("1 16-byte pieces, half-wave per row", obs, envs, cap);
            run<5>("5 ... utility with its rows", obs, envs, cap);
            if (cap == 0) run<7>("7 ... and every XCD a contiguous eighth of the buffer", obs, envs, cap);
            run<2>("2 ... non-temporal", obs, envs, cap);
            run<3>("3 line-aligned 1 KiB per instruction", obs, envs, cap);
            run<4>("4 ... non-temporal", obs, envs, cap);
            if (cap == 0) run<8>("8 line-aligned 1 KiB, XCD-contiguous", obs, envs, cap);
            if (cap == 0) run<9>("9 ... non-temporal", obs, envs, cap);
        }
    }
    // How much of the sustained rate is DRAM locality?  The same bytes, line-aligned 1 KiB store instructions, but the resident waves
    // write ADJACENT KiBs at the same time (a persistent grid sweeping the buffer front to back) instead of each wave its own 33 KB.
    for (int envs : sizes) {
        const size_t total_kib = (size_t)envs * 2 * 64 * ROW * 4 / 1024;
        for (int nt = 0; nt < 2; nt++) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            for (int i = 0; i < 10; i++) hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, obs, total_kib, nt);
            (void)hipEventRecord(a);
            for (int i = 0; i < 50; i++) hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, obs, total_kib, nt);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            printf("6 front-to-back sweep by a persistent grid%s       envs %6d: %8.1f us  %6.0f GB/s\n", nt ? ", non-temporal" : "               ", envs,
                   ms / 50 * 1e3, (double)total_kib * 1024 / (ms / 50 * 1e-3) / 1e9);
        }
    }
    return 0;
}
