// dcomp_fragment.h (included by dcomp_api.hip) -- lossless compact form of multi-agent observation rows for the learner hand-off.
//
// What goes over xGMI when a rollout fragment is handed to a learner are the per-UE observations of
// RelNormEnv.get_ue_obs (deepcomp/env/single_ue/variants.py:271-305) in RLlib's sorted-key order:
//     connected[B] | dr[B] | ues_at_bs[B] | util_at_bs[B] | utility            (4B + 1 floats per UE)
// Of those, `ues_at_bs` and `util_at_bs` are properties of the ENV (variants.py:296-299: bs.num_conn_ues / num_ue,
// bs.avg_utility / MAX_UTILITY) replicated into every UE's row, and `connected` is B bits (variants.py:273).  The compact
// record of one env-step therefore is
//     U x { dr[B] f32 | utility f32 | connected bit mask u32 }   +   ues_at_bs[B] f32 | util_at_bs[B] f32
// (round 6: with more than 32 stations the mask is TWO words -- U (B + 3) + 2B; the generic kernel of dcomp_big.h writes that record too)
// = U (B + 2) + 2B words instead of U (4B + 1): 1 616 B instead of 5 248 B at 32 x 10 (3.25x fewer bytes on the links), 17 664 B
// instead of 66 048 B at 128 x 32 (3.7x).  unpack(pack(rows)) is BIT-IDENTICAL to the rows: floats are copied, never
// recomputed; `connected` entries are exactly 0.0f / 1.0f; rows of dead UE slots (UE arrival / departure: all zeros,
// central.py:46-55 style padding) are recognised by their all-zero `dr` block (a listed UE always has the entry 1.0 at its best
// station, variants.py:279-284).  pack VERIFIES what it drops -- every listed row's per-env columns against row 0's, every
// `connected` entry against {0, 1} -- and raises a flag word otherwise, so "lossless" is checked, not assumed.
//
// Both kernels are pure streaming kernels (HBM-bound: pack reads 4B + 1 floats per UE and writes B + 2, unpack the reverse):
// a wavefront moves a chunk of R <= 32 rows of one env through its own LDS slice, global accesses are 16-byte pieces of
// contiguous spans (global memory needs dword alignment only), the per-row work is a few LDS reads and selects.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dcomp_types.h"

namespace dcomp_frag {

constexpr int BLOCK = 256;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));        // 16-byte access at dword alignment
typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float4 as_f4(f4u v) { return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 as_u4(u4u v) { return make_uint4(v.x, v.y, v.z, v.w); }

struct FragParams {
    const float *obs_in;        // pack: rows in;   unpack: unused
    float *obs_out;             // unpack: rows out
    const uint32_t *packed_in;  // unpack
    uint32_t *packed_out;       // pack
    int32_t *flags;             // pack: bit 0 = a listed row's per-env columns differ from row 0's, bit 1 = `connected` entry not 0 / 1
    int32_t U, B, R, chunks;    // R rows per chunk (one wavefront each), chunks per env
    int64_t units;              // chunks in all
    int32_t step_r_cw, step_k_cw, step_r_row, step_c_row;       // 256 words / floats further on: rows and remainder (pack's / unpack's store walk)
    int32_t rows_words, lw_pack, cw_words, lw_unpack;   // LDS words per wave: the rows / all of pack's; the compact words / all of unpack's
    uint32_t magic_row;         // ceil(2^32 / (4B + 1)): f / (4B + 1) = umulhi(f, magic) for f < 2^16 (exact: f (4B + 1) < 2^32)
    uint32_t magic_cw;          // ceil(2^32 / (B + 2))
};

// words of one env-step record
__host__ __device__ inline int mask_words(int B) { return B > 32 ? 2 : 1; }                 // connection-set words per UE
__host__ __device__ inline int ue_words(int B) { return B + 1 + mask_words(B); }             // dr[B] | utility | mask word(s)
__host__ __device__ inline int env_words(int U, int B) { return U * ue_words(B) + 2 * B; }

// Workgroup b runs on XCD b mod 8: with this map every XCD streams ONE contiguous eighth of the fragment (see xcd_contiguous_block
// in dcomp_device.h: worth up to 19 % of the sustained write rate beyond the Infinity Cache).
__device__ __forceinline__ unsigned xcd_block()
{
    const unsigned b = blockIdx.x, n = gridDim.x, q = n >> 3, r = n & 7u, x = b & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}

// LDS ops of one wave execute in order; this only stops the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wave_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// One WAVEFRONT moves one chunk (R <= 32 rows of one env) and shares nothing with the other three waves of its workgroup: no
// workgroup barriers, every wave has its loads in flight on its own (a first version with one 256-thread workgroup per env and
// four __syncthreads-separated phases ran at 0.38 of the HBM peak at 32 x 10: latency-bound).
// Per-wave LDS (LW words): the chunk's rows as they lie in memory (R (4B + 1) floats, 16-byte aligned) | row 0's per-env columns (2B).
__global__ __launch_bounds__(BLOCK) void pack_kernel(const FragParams p)
{
    extern __shared__ float4 lds4[];
    const int B = p.B, U = p.U, ROW = 4 * B + 1, CW = ue_words(B);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *rows = reinterpret_cast<float *>(lds4) + (size_t)wave * p.lw_pack, *t0 = rows + p.rows_words;
    const int64_t unit = (int64_t)xcd_block() * (BLOCK / 64) + wave;             // every XCD a contiguous eighth of the units
    if (unit >= p.units) return;
    const int64_t env = unit / p.chunks;
    const int chunk = (int)(unit - env * p.chunks);
    const int r0 = chunk * p.R, nr = min(p.R, U - r0), nf = nr * ROW;
    const float *src = p.obs_in + ((size_t)env * U + r0) * ROW;
    // ALL loads of the unit are issued before the first one is used (registers, then LDS): a loop of "load 16 bytes, write them to
    // LDS" waits for every load in turn and has one piece per lane in flight -- the kernel is bound by load latency x bytes in
    // flight (this form: 95 -> 87 us at 65 536 x 32 x 10, 697 -> 484 us at 32 768 x 128 x 32).
    constexpr int NQ = 8;                                         // 16-byte pieces per lane: 64 lanes x 8 x 16 B = the 8 KB slice
    f4u q[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) {                                // UNCONDITIONAL (a piece beyond the unit re-reads piece 0 and is dropped):
        const int i = lane * 4 + k * 256;                         // behind a divergent `if` the compiler waits for each load at the join
        q[k] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(src + (i + 3 < nf ? i : 0)));     // read once, never again (nf >= 5)
    }
    const float tail = lane < (nf & 3) ? src[(nf & ~3) + lane] : 0.f;
    const float t0v = lane < 2 * B ? p.obs_in[(size_t)env * U * ROW + 2 * B + lane] : 0.f;     // row 0 of the env: the per-env columns
    const float t0w = lane + 64 < 2 * B ? p.obs_in[(size_t)env * U * ROW + 2 * B + lane + 64] : 0.f;       // (more than 32 stations: 2B > 64)
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        const int i = lane * 4 + k * 256;
        if (i + 3 < nf) *reinterpret_cast<float4 *>(rows + i) = as_f4(q[k]);
    }
    if (lane < (nf & 3)) rows[(nf & ~3) + lane] = tail;
    if (lane < 2 * B) t0[lane] = t0v;
    if (lane + 64 < 2 * B) t0[lane + 64] = t0w;
    wave_fence();
    // lane r < 32: row r's `connected` block -> bit mask (and: every entry is 0 or 1), then its ues_at_bs replicas against row 0's;
    // lane 32 + r: row r's dr block -> "listed" (some entry is non-zero), then its util_at_bs replicas.  Row stride 4B + 1 is odd:
    // the 32 lanes of a half walk their rows conflict-free.
    const int r = lane & 31, hi = lane >> 5;
    const bool mine = r < nr;
    const float *blk = rows + (mine ? r : 0) * ROW + hi * B;
    unsigned long long mask = 0;
    int bad = 0;
    for (int b = 0; b < B; b++) {
        const float c = blk[b];
        mask |= (c != 0.f ? 1ull : 0ull) << b;
        bad |= (!hi && c != 0.f && c != 1.f) ? 2 : 0;
    }
    const unsigned long long dmask = __shfl(mask, r + 32, 64);                               // the dr mask of my row (held by lane 32 + r)
    const bool listed = dmask != 0ull;
    {
        const float *rep = blk + 2 * B, *want = t0 + hi * B;                                   // ues_at_bs (lanes < 32) | util_at_bs (lanes >= 32)
        for (int b = 0; b < B; b++) bad |= (__float_as_uint(rep[b]) != (listed ? __float_as_uint(want[b]) : 0u)) ? 1 : 0;
    }
    if (mine && bad) atomicOr(p.flags, bad);
    // the connection words go where the per-row utility float sits in the compact record's neighbour: write them into the rows
    // buffer's `connected[0]` cell (dead from here on), so that word() below reads everything from one place
    wave_fence();
    if (mine && !hi) {
        rows[r * ROW] = __uint_as_float((uint32_t)mask);
        if (B > 32) rows[r * ROW + 1] = __uint_as_float((uint32_t)(mask >> 32));          // (`connected[1]`: dead from here on, too)
    }
    wave_fence();
    uint32_t *dst = p.packed_out + (size_t)env * env_words(U, B) + (size_t)r0 * CW;
    const int nw = nr * CW;
    auto word = [&](int w) -> uint32_t {                          // (generic form: the tail words)
        const int rr = (int)__umulhi((uint32_t)w, p.magic_cw), k = w - rr * CW;
        const int at = k < B ? B + k : k == B ? 4 * B : k - B - 1;                            // dr[k] | utility | connection mask word(s)
        return __float_as_uint(rows[rr * ROW + at]);
    };
    {
        // a lane walks (row, word-in-row) incrementally -- one division when it starts, none per word (v_mul_hi_u32 issues at a
        // quarter of the rate, and the profile had the kernels VALU-bound: 444 / 741 instructions per wave in pack / unpack)
        int rr = (int)__umulhi((uint32_t)(lane * 4), p.magic_cw), k = lane * 4 - rr * CW;
        for (int w = lane * 4; w + 3 < nw; w += 64 * 4) {
            uint32_t o[4];
            int r2 = rr, k2 = k;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int at = k2 < B ? B + k2 : k2 == B ? 4 * B : k2 - B - 1;
                o[j] = __float_as_uint(rows[__umul24(r2, ROW) + at]);
                k2++;
                if (k2 == CW) { k2 = 0; r2++; }
            }
            u4u v;
            v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
            *reinterpret_cast<u4u *>(dst + w) = v;
            rr += p.step_r_cw; k += p.step_k_cw;                  // 256 words on
            if (k >= CW) { k -= CW; rr++; }
        }
    }
    if (lane < (nw & 3)) dst[(nw & ~3) + lane] = word((nw & ~3) + lane);
    if (chunk == 0) for (int i = lane; i < 2 * B; i += 64) p.packed_out[(size_t)env * env_words(U, B) + (size_t)U * CW + i] = __float_as_uint(t0[i]);
}

// Per-wave LDS: the chunk's compact words (R (B + 2), 16-byte aligned) | per-env columns (2B) | listed[R]
__global__ __launch_bounds__(BLOCK) void unpack_kernel(const FragParams p)
{
    extern __shared__ float4 lds4[];
    const int B = p.B, U = p.U, ROW = 4 * B + 1, CW = ue_words(B);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t unit = (int64_t)xcd_block() * (BLOCK / 64) + wave;             // every XCD a contiguous eighth of the units
    if (unit >= p.units) return;
    const int64_t env = unit / p.chunks;
    const int chunk = (int)(unit - env * p.chunks);
    const int r0 = chunk * p.R, nr = min(p.R, U - r0);
    uint32_t *cw = reinterpret_cast<uint32_t *>(lds4) + (size_t)wave * p.lw_unpack, *t0 = cw + p.cw_words, *listed = t0 + 2 * B;
    const uint32_t *src = p.packed_in + (size_t)env * env_words(U, B) + (size_t)r0 * CW;
    const int nw = nr * CW;
    constexpr int NQ = 5;                                         // R (B + 2) <= 32 x 34 words = 4.25 KiB: at most 5 pieces per lane
    u4u q[NQ];
    if (nw >= 4) {                                                // (uniform; a unit of 1 UE x 1 station has 3 words: tail path only)
#pragma unroll
        for (int k = 0; k < NQ; k++) {                            // all loads in flight before the first LDS write, unconditional as in pack_kernel
            const int i = lane * 4 + k * 256;
            q[k] = __builtin_nontemporal_load(reinterpret_cast<const u4u *>(src + (i + 3 < nw ? i : 0)));
        }
    }
    const uint32_t tail = lane < (nw & 3) ? src[(nw & ~3) + lane] : 0u;
    const uint32_t t0v = lane < 2 * B ? p.packed_in[(size_t)env * env_words(U, B) + (size_t)U * CW + lane] : 0u;
    const uint32_t t0w = lane + 64 < 2 * B ? p.packed_in[(size_t)env * env_words(U, B) + (size_t)U * CW + lane + 64] : 0u;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        const int i = lane * 4 + k * 256;
        if (i + 3 < nw) *reinterpret_cast<uint4 *>(cw + i) = as_u4(q[k]);
    }
    if (lane < (nw & 3)) cw[(nw & ~3) + lane] = tail;
    if (lane < 2 * B) t0[lane] = t0v;
    if (lane + 64 < 2 * B) t0[lane + 64] = t0w;
    wave_fence();
    if (lane < nr) {                                                                          // listed <=> some dr entry is non-zero
        uint32_t any = 0;
        for (int b = 0; b < B; b++) any |= cw[lane * CW + b] & 0x7FFFFFFFu;                   // (+0 and -0 are both "zero")
        listed[lane] = any != 0u;
    }
    wave_fence();
    float *dst = p.obs_out + ((size_t)env * U + r0) * ROW;
    const int nf = nr * ROW;
    auto val = [&](int f) -> float {
        const int r = (int)__umulhi((uint32_t)f, p.magic_row), c = f - r * ROW;
        const uint32_t *q = cw + r * CW;
        if (c < B) return (float)((q[B + 1 + (c >> 5)] >> (c & 31)) & 1u);           // connected          variants.py:273
        if (c < 2 * B) return __uint_as_float(q[c - B]);                            // dr                 variants.py:279-284
        if (c < 4 * B) return listed[r] ? __uint_as_float(t0[c - 2 * B]) : 0.f;     // ues_at_bs | util_at_bs   variants.py:296-299
        return __uint_as_float(q[B]);                                               // utility            variants.py:287
    };
    const bool aligned = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;           // (uniform) non-temporal only for whole 16-byte pieces
    {
        int r = (int)__umulhi((uint32_t)(lane * 4), p.magic_row), c = lane * 4 - r * ROW;       // (row, column) of the lane's first float
        for (int f = lane * 4; f + 3 < nf; f += 64 * 4) {
            float o[4];
            int r2 = r, c2 = c;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t *q = cw + __umul24(r2, CW);
                const uint32_t w = q[c2 < B ? B + 1 + (c2 >> 5) : c2 < 2 * B ? c2 - B : B];   // connection mask word | dr[c - B] | utility
                const int e = c2 - 2 * B;
                const uint32_t env = t0[e >= 0 && e < 2 * B ? e : 0];                         // ues_at_bs | util_at_bs of the env
                const bool isenv = e >= 0 && e < 2 * B;
                const uint32_t bits = c2 < B ? __float_as_uint((float)((w >> (c2 & 31)) & 1u)) : w;
                o[j] = __uint_as_float(isenv ? (listed[r2] ? env : 0u) : bits);
                c2++;
                if (c2 == ROW) { c2 = 0; r2++; }
            }
            f4u v;
            v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
            if (aligned) __builtin_nontemporal_store(v, reinterpret_cast<f4u *>(dst + f));      // write-once stream for the learner
            else *reinterpret_cast<f4u *>(dst + f) = v;
            r += p.step_r_row; c += p.step_c_row;                 // 256 floats on
            if (c >= ROW) { c -= ROW; r++; }
        }
    }
    if (lane < (nf & 3)) dst[(nf & ~3) + lane] = val((nf & ~3) + lane);
}

static int rows_per_chunk(int U, int B)
{
    // <= 8 KB of rows per wave (five 4-wave workgroups per CU), <= 32 rows (one lane per row in each half-wave); whole envs where they fit
    const int cap = (8 * 1024) / ((4 * B + 1) * 4);
    int R = U < cap ? U : cap;
    if (R > 32) R = 32;
    if (R >= 4 && R < U) R &= ~3;          // chunks of a multiple of 4 rows start 16-byte aligned (a row is 4B + 1 floats): the row
    return R < 1 ? 1 : R;                  // stores of unpack are then ALIGNED non-temporal stores (misaligned ones stream at 2.9 TB/s)
}

static int fill(FragParams &p, int64_t n, int U, int B, int &grid, size_t &lds_pack, size_t &lds_unpack)
{
    if (n < 1 || U < 1 || U > DCOMP_MAX_UE || B < 1 || B > DCOMP_MAX_BS) return DCOMP_EINVAL;
    const int CW = ue_words(B);
    p.U = U; p.B = B;
    p.R = rows_per_chunk(U, B);
    p.chunks = (U + p.R - 1) / p.R;
    p.units = n * p.chunks;
    const int64_t blocks = (p.units + BLOCK / 64 - 1) / (BLOCK / 64);
    if (blocks > 0x7FFFFFFFll) return DCOMP_EINVAL;
    grid = (int)blocks;
    p.magic_row = (uint32_t)(0x100000000ull / (uint32_t)(4 * B + 1)) + 1u;
    p.magic_cw = (uint32_t)(0x100000000ull / (uint32_t)CW) + 1u;
    p.step_r_cw = 256 / CW; p.step_k_cw = 256 % CW;
    p.step_r_row = 256 / (4 * B + 1); p.step_c_row = 256 % (4 * B + 1);
    p.rows_words = (p.R * (4 * B + 1) + 3) & ~3;
    p.lw_pack = (p.rows_words + 2 * B + 3) & ~3;
    p.cw_words = (p.R * CW + 3) & ~3;
    p.lw_unpack = (p.cw_words + 2 * B + p.R + 3) & ~3;
    lds_pack = (size_t)p.lw_pack * 4 * (BLOCK / 64);
    lds_unpack = (size_t)p.lw_unpack * 4 * (BLOCK / 64);
    return DCOMP_OK;
}

}  // namespace dcomp_frag

