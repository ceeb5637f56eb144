// dcomp_big.h -- the env step for 33 ... 64 base stations (round 5; the reference has no limit: station.py:16-30, base.py:79-84).
//
// The kernels of dcomp_device.h / dcomp_wide.h unroll the station loop (one object file per B = 1 ... 32, per-station register arrays,
// a 32-bit connection mask): the right shape for the BASELINE configurations, not for arbitrary B.  This file is the GENERIC path:
// ONE instantiation per lane-group width serves every B up to 64 -- B is a run-time value, the per-station values of a UE live in
// its row of LDS (stride B + 1), the connection set is two 32-bit state words per UE (`conn` = stations 0-31 as before, `conn_hi` =
// stations 32-63: dcomp_state.conn_hi), the BS table is a device array staged in LDS per workgroup.  Same semantics, same
// numerics (FP64 positions / decisions in the reference's operation order, FP32 log2-domain rates via the same device functions),
// same outputs as the narrow kernels -- tests/test_bigb_gpu.py holds it to the oracle at B = 33 ... 64 and to the narrow kernels
// at B <= 32 (DCOMP_FORCE_BIG=1).  Three workgroup-wide reduction passes per step (sharing terms before and after the move, utility
// sums); 0.2-0.45 of the HBM peak where the specialised kernels reach 0.7-0.85.  What round 5 found it bound by (DESIGN_LOG R5.7): partial-line
// non-temporal stores (now plain), the LDS rows' cap on resident wavefronts (the carve below), per-mode loops in the reduction passes (now one).
//
// Mapping: one lane = one (env, UE); an env takes UPAD = next pow2 >= U lanes; a workgroup is ONE wavefront (64 / UPAD envs) or, for
// envs of more than 64 lanes, the env's own UPAD lanes.
// Per-station sums over an env's UEs (station.py:152-202, 63-83): thread t of the workgroup OWNS the (env, station) pairs
// t, t + BLK, ... and adds up the rows of that env in UE order -- deterministic, no atomics, conflict-free (consecutive threads own
// consecutive stations).  Not supported here (dcomp_create_v says so): UE arrival / departure, the fused rollout (dcomp_rollout_ex
// launches one step per launch), the in-step policy, the compact record.
#pragma once
#include "dcomp_device.h"

// A/B switches (tools/ab/ablate_big.sh, DESIGN_LOG R5.7); the defaults are the measured best
#ifndef DCOMP_BIG_WUNROLL
#define DCOMP_BIG_WUNROLL 4   // rows of the observation writer in flight per wavefront (1: -4 %)
#endif
#ifndef DCOMP_BIG_PUNROLL
#define DCOMP_BIG_PUNROLL 2   // stations of the post-move pair loop in flight per lane
#endif
#ifndef DCOMP_BIG_NT
#define DCOMP_BIG_NT 0        // 1: non-temporal stores like the narrow kernels (whose stores are whole 16-byte-per-lane lines); see big_store
#endif
#ifndef DCOMP_BIG_EARLY
#define DCOMP_BIG_EARLY 1     // the connected | dr blocks of the observation leave right after the post-move pairs, under the rest of the step (0: +1-3 %)
#endif
#ifndef DCOMP_BIG_PARK
#define DCOMP_BIG_PARK 1      // post-move pass: rates parked in the rows by the UEs' own lanes (needs DCOMP_BIG_EARLY or no observation; 0: +3-5 % in mixed sharing)
#endif
#ifndef DCOMP_BIG_ABL
#define DCOMP_BIG_ABL 0       // timing-only ablation (results WRONG): 1 max-cap winner, 2 utility aggregates, 4 rows, 8 pairs, 16 sharing aggregates, 32 move
#endif

namespace dcomp {

struct BigParams {
    uint32_t *conn_hi;                 // [E*U] stations 32 ... 63 of the connection set
    const double2 *bs;                 // [B] station positions
    const int32_t *mode;               // [B] DCOMP_*_FAIR / MAX_CAP
    int32_t B;
    unsigned long long maxcap_mask;    // bit b: station b is max-cap
};

// Workgroup size: ONE wavefront, or as many as an env needs.  The rows cost (B + 1) * 4 bytes of LDS per lane (260 at B = 64), so a CU holds
// 7-8 wavefronts of this kernel at most; 256-lane workgroups (the first version) fit ONCE per CU at B = 64 -- 4 wavefronts, one per SIMD,
// nothing to hide a barrier or an LDS round trip behind (8 192 x 32 x 64: 181 us, SQ_WAIT_ANY 44 % of the wave cycles).
__host__ __device__ constexpr int big_block(int upad) { return upad < 64 ? 64 : upad; }
// LDS one workgroup carves (host and device use the same function).  Per lane: its row (B stations + one pad column that also holds the row
// maximum), its connection set, ewma / utility (one slot: the utility is written after the last reader of the ewma) and reward_before; per
// (env, station): three aggregate slots (count | sum of the sharing terms, later of the utilities | max-cap winner, later the minimum
// utility).  Only where they are used: the partial sums of split pairs, the FP64 positions (max-cap stations only).
// 64 stations, 32 UEs, no max-cap: 20 480 bytes = EIGHT wavefronts per CU (the first carve, 24 064 bytes, held six).
struct BigCarve { int mask, row, ewma, rb, agg, mode, part, pos, total; };
__host__ __device__ inline BigCarve big_carve(int B, int gpb, int blk, bool maxcap)
{
    BigCarve c;
    int o = B * (int)sizeof(double2);
    c.mask = o; o += blk * 8;
    c.row = o; o += blk * (B + 1) * 4;
    c.ewma = o; o += blk * 4;
    c.rb = o; o += blk * 4;
    c.agg = o; o += 3 * gpb * B * 4;
    c.mode = o; o += B * 4;
    o = (o + 15) & ~15;
    c.part = o; if (2 * gpb * B <= blk) o += blk * 16;          // aggregate() splits a pair's rows over >= 2 threads
    c.pos = o; if (maxcap) o += blk * 16;
    c.total = o;
    return c;
}
// what dcomp_create_v checks against the 160 KB of a CU (before it knows the sharing modes; the documented bound of include/dcomp_types.h)
__host__ __device__ inline size_t big_lds_bound(int B, int blk)
{
    return 64 * sizeof(double2) + (size_t)blk * sizeof(double2) + (size_t)blk * sizeof(unsigned long long) + (size_t)blk * (B + 1) * 4 + (size_t)4 * blk * 4 +
           (size_t)5 * B * 4 + 64 * 4 + (size_t)blk * 16;
}

__device__ __forceinline__ void big_pair(double px, double py, const double2 bp, const KParams &p, bool &in_range, float &l2)
{
    bool near;
    pair_eval(px, py, bp.x, bp.y, p, in_range, l2, near);
    if (near) {                                                    // rare, per lane: within 1.26 m of the station
        const double dx = bp.x - px, dy = bp.y - py;
        if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2 = pair_eval_tiny(px, py, bp.x, bp.y, p);
    }
}
// Every output store of this kernel.  PLAIN stores: the rows leave as 4-byte-per-lane pieces of B floats at a stride of 4B + 1 floats, i.e. as
// partial cache lines, and a non-temporal partial line does not wait in the L2 for its other half (8 192 x 32 x 64, resource-fair, same box:
// 134 us with non-temporal, 81 us with plain stores; staging 4 rows in LDS for 16-byte-per-lane whole lines was SLOWER than either, R5.7).
__device__ __forceinline__ void big_store(float *ptr, float v)
{
#if DCOMP_BIG_NT
    stream_store(ptr, v);
#else
    *ptr = v;
#endif
}
__device__ __forceinline__ float big_rate(float l2)                // bw * log2(1 + snr), station.py:129-138
{
    bool big;
    float r = rate_unshared_small(l2, big);
    if (big) r = rate_unshared_any(l2);
    return r;
}

template <int UPAD, bool RESET>
__global__ __launch_bounds__(big_block(UPAD)) void big_kernel(const KParams p, const BigParams x)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    constexpr int BLK = big_block(UPAD), NWAVE = BLK / 64, GPB = BLK / UPAD;
    const int B = x.B, BR = B + 1, U = p.U;
    const bool any_maxcap = x.maxcap_mask != 0ull;
    const BigCarve cv = big_carve(B, GPB, BLK, any_maxcap);
    double2 *const bs_s = reinterpret_cast<double2 *>(big_smem);
    unsigned long long *const mask_s = reinterpret_cast<unsigned long long *>(big_smem + cv.mask);
    float *const row = reinterpret_cast<float *>(big_smem + cv.row);
    float *const ewma_s = reinterpret_cast<float *>(big_smem + cv.ewma), *const util_s = ewma_s, *const rb_s = reinterpret_cast<float *>(big_smem + cv.rb);
    float *const agg_n = reinterpret_cast<float *>(big_smem + cv.agg), *const agg_s = agg_n + GPB * B, *const agg_u = agg_s;
    uint32_t *const mc_win = reinterpret_cast<uint32_t *>(agg_s + GPB * B);
    float *const agg_m = reinterpret_cast<float *>(mc_win);
    int *const mode_s = reinterpret_cast<int *>(big_smem + cv.mode);
    float4 *const part_s = reinterpret_cast<float4 *>(big_smem + cv.part);     // partial sums of split pairs (aggregate)
    double2 *const pos_s = reinterpret_cast<double2 *>(big_smem + cv.pos);     // max-cap stations only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env_local = tid / UPAD, u = tid % UPAD;
    const int env0 = blockIdx.x * GPB, env = env0 + env_local;
    const bool active = env < p.E && u < U;
    const bool alive = active && (!RESET || u < p.U0);
    const int idx = env * U + u;
    float *const myrow = row + tid * BR;
    for (int i = tid; i < B; i += BLK) { bs_s[i] = x.bs[i]; mode_s[i] = x.mode[i]; }

    double px = 0.0, py = 0.0;
    unsigned long long mv = 0, conn = 0;
    uint32_t act = 0;
    float ewma = 0.f, dr_req = 1.f;
    bool step_util = false;
    int vrange = MV_CFG_ARRIVED;
    if (alive) {
        const UeCfg c = p.ue_cfg[u];
        step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req;
        vrange = mv_cfg_pack(c.vel_lo, c.vel_hi, c.pause, c.border);
    }
    if (RESET) {                                                   // MobileEnv.reset (base.py:169-189): user.py:98-116 + movement.py:110-122
        if (alive) reset_ue(p, env, u, p.episode, px, py, mv);
        if (active) {
            p.pos[idx] = make_double2(px, py);
            p.mv[idx] = mv;
            p.conn[idx] = 0u;
            x.conn_hi[idx] = 0u;
            p.ewma[idx] = 0.f;
        }
    } else if (active) {
        const double2 q = p.pos[idx];
        px = q.x; py = q.y;
        mv = p.mv[idx];
        conn = (unsigned long long)p.conn[idx] | ((unsigned long long)x.conn_hi[idx] << 32);
        ewma = p.ewma[idx];
        act = p.action[idx];
        if (act > (uint32_t)B) { atomicOr(p.flags, DCOMP_FLAG_BAD_ACTION); act = 0; }
    }
    __syncthreads();                                               // the BS table

    // (env, station) sums over the UEs of an env.  WHAT: 0 = sharing terms from the l2snr values parked in the rows (count, sum of
    // 1 / rate resp. rate / (ewma + eps), the max-cap winner), 1 = utility sums (count, sum, min over the connected UEs).
    // Envs of many lanes have few pairs per workgroup (512 lanes, 10 stations: 10 owners walking 512 rows each while 502 lanes wait: 4 096 x 512 x 10
    // took 1.39 ms): there the rows of a pair are split over NSPLIT threads -- thread q * NSPLIT + k sums rows k, k + NSPLIT, ... -- and the owner
    // adds the partial sums in the order k = 0 ... NSPLIT - 1 (still deterministic).
    auto aggregate = [&](int what, bool parked) {
        if ((DCOMP_BIG_ABL & 2) && what == 1) return;
        if ((DCOMP_BIG_ABL & 16) && what == 0) return;
        const int P = GPB * B;
        int nsplit = P >= BLK ? 1 : BLK / P;
        nsplit = nsplit > 32 ? 32 : nsplit;
        // Branch-free and unrolled (a branch per row saved little -- SOME lane of the 64 is connected in most trips -- and put an LDS round trip + a
        // rate evaluation on one dependent chain per row), and ONE loop for rate-fair and proportional-fair owners (the threads of a wavefront own
        // stations of every mode: a loop per mode ran with a quarter of the lanes each).  Same sums: a row that is not connected adds +0.
        // parked: the rows hold the RATES of the connected stations (pre-move pass), not log2 snr.
        auto rows_of = [&](int el, int b, int first, int stride, float &n, float &s, float &mn) {
            const int base = el * UPAD, mode = mode_s[b];
            const bool sums = what == 0 && (mode == DCOMP_RATE_FAIR || mode == DCOMP_PROP_FAIR), rate_fair = mode == DCOMP_RATE_FAIR;
            if (what == 1) {
#pragma unroll 4
                for (int v = first; v < U; v += stride) {
                    const bool c = (mask_s[base + v] >> b) & 1ull;
                    const float uv = util_s[base + v];
                    n += c ? 1.f : 0.f; s += c ? uv : 0.f; mn = c ? fminf(mn, uv) : mn;
                }
            } else if (__builtin_amdgcn_ballot_w64(sums) != 0ull) {            // station.py:177-180 | station.py:150, 192-195
#pragma unroll 4
                for (int v = first; v < U; v += stride) {
                    const bool c = ((mask_s[base + v] >> b) & 1ull) != 0ull, cs = c && sums;
                    const float l = cs ? row[(base + v) * BR + b] : 0.f;          // (a row without this station holds leftovers)
                    const float r = parked ? l : big_rate(l);
                    const float y = fast_rcp(rate_fair ? r : ewma_s[base + v] + EPS);
                    n += c ? 1.f : 0.f; s += cs ? (rate_fair ? y : r * y) : 0.f;
                }
            } else {
#pragma unroll 4
                for (int v = first; v < U; v += stride) n += ((mask_s[base + v] >> b) & 1ull) ? 1.f : 0.f;
            }
        };
        if (nsplit > 1) {
            float n = 0.f, s = 0.f, mn = MAX_UTIL;
            if (tid < P * nsplit) {
                const int q = tid / nsplit, k = tid - q * nsplit, el = q / B, b = q - el * B;
                if (env0 + el < p.E) rows_of(el, b, k, nsplit, n, s, mn);
            }
            part_s[tid] = make_float4(n, s, mn, 0.f);
            __syncthreads();
        }
        for (int q = tid; q < P; q += BLK) {
            const int el = q / B, b = q - el * B, base = el * UPAD;
            float n = 0.f, s = 0.f, mn = MAX_UTIL;
            uint32_t win = 0xFFFFFFFFu;
            if (env0 + el < p.E) {
                const int mode = mode_s[b];
                if (nsplit > 1) {
                    for (int k = 0; k < nsplit; k++) { const float4 t = part_s[q * nsplit + k]; n += t.x; s += t.y; mn = fminf(mn, t.z); }
                } else rows_of(el, b, 0, 1, n, s, mn);
                if (!(DCOMP_BIG_ABL & 1) && what == 0 && mode == DCOMP_MAX_CAP && n > 0.f) {
                    // station.py:183-187: the UE with the highest FP64 rate is served; equal rates -> the oldest connection, then the lowest UE
                    // index (dcomp_device.h shared_rates has the derivation: nearest UE, contenders within 1e-7, the collapsing FP64 key)
                    const double2 bp = bs_s[b];
                    double dmin = 1e300;
                    for (int v = 0; v < U; v++) if ((mask_s[base + v] >> b) & 1ull) {
                        const double dx = bp.x - pos_s[base + v].x, dy = bp.y - pos_s[base + v].y;
                        dmin = fmin(dmin, __builtin_fma(dy, dy, dx * dx));
                    }
                    int ncand = 0, only = 0;
                    for (int v = 0; v < U; v++) if ((mask_s[base + v] >> b) & 1ull) {
                        const double dx = bp.x - pos_s[base + v].x, dy = bp.y - pos_s[base + v].y;
                        if (__builtin_fma(dy, dy, dx * dx) <= dmin * (1.0 + 1e-7)) { ncand++; only = v; }
                    }
                    if (ncand == 1) win = (uint32_t)only;
                    else {
                        unsigned long long best = 0ull;
                        uint32_t bestw = 0xFFFFFFFFu;
                        for (int v = 0; v < U; v++) if ((mask_s[base + v] >> b) & 1ull) {
                            const double dx = bp.x - pos_s[base + v].x, dy = bp.y - pos_s[base + v].y;
                            if (__builtin_fma(dy, dy, dx * dx) > dmin * (1.0 + 1e-7)) continue;
                            const unsigned long long key = maxcap_rate_key(p.pl_c1, p.pl_c2, pos_s[base + v].x, pos_s[base + v].y, bp.x, bp.y);
                            const uint32_t w = ((uint32_t)p.conn_since[((size_t)(env0 + el) * U + v) * B + b] << 10) | (uint32_t)v;     // (16-bit step, UE index < 1 024)
                            if (key > best || (key == best && w < bestw)) { best = key; bestw = w; }
                        }
                        win = bestw & 0x3FFu;
                    }
                }
            }
            if (what == 1) { agg_n[q] = n; agg_u[q] = s; agg_m[q] = mn; }
            else { agg_n[q] = n; agg_s[q] = s; mc_win[q] = win; }
        }
    };
    // this UE's share of every station it is connected to (station.py:152-202); keep = park it in the row (the stale rates of user.py:148-157)
    auto shared = [&](bool keep, bool parked) -> float {
        const float inv_ewma = fast_rcp(ewma + EPS);
        float curr = 0.f;
        for (unsigned long long m = conn; m; m &= m - 1ull) {
            const int b = __ffsll((long long)m) - 1, q = env_local * B + b, mode = mode_s[b];
            const float dru = parked ? myrow[b] : big_rate(myrow[b]);
            float out;
            if (mode == DCOMP_RES_FAIR) out = dru * fast_rcp(fmaxf(agg_n[q], 1.f));                         // station.py:171-173
            else if (mode == DCOMP_RATE_FAIR) out = fast_rcp(agg_s[q]);                                     // station.py:180
            else if (mode == DCOMP_PROP_FAIR) out = (dru * inv_ewma) * fast_rcp(agg_s[q] + EPS) * dru;      // station.py:194-195
            else out = mc_win[q] == (uint32_t)u ? dru : 0.f;                                                // station.py:183-187
            curr += out;
            if (keep) myrow[b] = out;
        }
        return curr;
    };

    float reward_before = 0.f;
    if (!RESET) {
        // 1. the pre-move position matters where the UE is connected and at the station it acts on (base.py:247-263 -> user.py:190-222)
        const unsigned long long act_bit = act ? 1ull << (act - 1u) : 0ull;
        unsigned long long inr_old = 0ull;
        for (unsigned long long need = active ? (conn | act_bit) : 0ull; need; need &= need - 1ull) {
            const int b = __ffsll((long long)need) - 1;
            bool ir;
            float l;
            big_pair(px, py, bs_s[b], p, ir, l);
            inr_old |= (unsigned long long)ir << b;
            myrow[b] = big_rate(l);                                // parked: rates, not log2 snr (nothing after this pass needs the latter)
        }
        if (act_bit) {
            if (conn & act_bit) conn &= ~act_bit;
            else if (inr_old & act_bit) {
                conn |= act_bit;
                if (x.maxcap_mask & act_bit) p.conn_since[(size_t)idx * B + (act - 1u)] = (uint16_t)p.time;
            }
        }
        mask_s[tid] = active ? conn : 0ull;
        ewma_s[tid] = ewma;
        if (any_maxcap) pos_s[tid] = make_double2(px, py);
        __threadfence_block();
        __syncthreads();
        // 2. rates before the move (base.py:446) -> reward_before (base.py:158-167)
        aggregate(0, true);
        __syncthreads();
        const float curr_pre = shared(true, true);
        reward_before = clamp_med3(ue_utility(curr_pre, step_util, dr_req), MIN_UTIL, MAX_UTIL) * (1.0f / MAX_UTIL);
        // 3. move (base.py:447 -> user.py:159-173)
        if (active && !(DCOMP_BIG_ABL & 32)) {
            move_ue<false>(p, env, (uint32_t)u + 1u, p.episode, px, py, mv, vrange);
            if (px < 0.0 || py < 0.0 || px > (double)p.map_w || py > (double)p.map_h) atomicOr(p.flags, DCOMP_FLAG_OUTSIDE_MAP);
        }
        // 4. drop what is out of range at the new position (user.py:175-188); EWMA from the STALE rates of what stays (user.py:148-157)
        float stale = 0.f;
        for (unsigned long long m = conn; m; m &= m - 1ull) {
            const int b = __ffsll((long long)m) - 1;
            const double dx = bs_s[b].x - px, dy = bs_s[b].y - py;
            if (dist_sq_ref(dx, dy) < p.dt2) stale += myrow[b];
            else conn &= ~(1ull << b);
        }
        ewma = __builtin_fmaf(0.9f, stale, 0.1f * ewma);
        __syncthreads();                                           // every owner thread is done with the pre-move rows / masks
    }
    // 5. every station at the (new) position: log2 snr into the row, the in-range mask, the row maximum
    unsigned long long in_range = 0ull;
    float l2max = -3.0e38f;
#pragma unroll DCOMP_BIG_PUNROLL
    for (int b = 0; b < B; b++) {
        bool ir = true;
        float l = 0.f;
        if (!(DCOMP_BIG_ABL & 8)) big_pair(px, py, bs_s[b], p, ir, l);
        in_range |= (unsigned long long)ir << b;
        myrow[b] = l;
        l2max = fmaxf(l2max, l);
    }
    mask_s[tid] = alive ? conn : 0ull;
    ewma_s[tid] = ewma;
    if (any_maxcap) pos_s[tid] = make_double2(px, py);
    myrow[B] = l2max;                                              // the pad column
    __syncthreads();
    // Observation, first half (variants.py:271-284): `connected` and the relative snr are final here -- their stores drain under the rest of the
    // step instead of queueing behind it (a wavefront waiting on its stores holds its rows of LDS and keeps the next one out).
    // Multi-agent rows: a wavefront per UE row, lanes along the stations (B <= 64: one trip), uniform control flow.
    const int w0 = __builtin_amdgcn_readfirstlane(wave);
    const int ROW = 4 * B + 1, UB = U * B, kind = p.kind;
    auto rows_early = [&]() {
        for (int el = 0; el < GPB && kind == DCOMP_MULTI; el++) {
            if (env0 + el >= p.E) break;                           // (uniform)
            float *const dst0 = p.obs + (size_t)(env0 + el) * U * ROW;
#pragma unroll DCOMP_BIG_WUNROLL
            for (int uu = w0; uu < U; uu += NWAVE) {
                const int r = el * UPAD + uu;
                const bool live = !RESET || uu < p.U0;
                float *const dst = dst0 + (size_t)uu * ROW;
                if (lane < B) {
                    big_store(dst + lane, live ? (float)((mask_s[r] >> lane) & 1ull) : 0.f);
                    big_store(dst + B + lane, live ? fast_exp2(row[r * BR + lane] - row[r * BR + B]) : 0.f);                               // variants.py:276-284
                }
            }
        }
        if (kind == DCOMP_CENTRAL && env < p.E) {
            // central rows are short (U (2B+1) floats per ENV): the lanes of an env walk its connected / dr blocks, (UE, station) advanced
            // incrementally (no division per element)
            float *const dst = p.obs + (size_t)env * U * (2 * B + 1);
            const int base = env_local * UPAD;
            int uu = u / B, b = u - uu * B;
            for (int c = u; c < UB; c += UPAD) {
                const bool live = !RESET || uu < p.U0;
                big_store(dst + c, live ? (float)((mask_s[base + uu] >> b) & 1ull) : 0.f);
                big_store(dst + UB + c, live ? fast_exp2(row[(base + uu) * BR + b] - row[(base + uu) * BR + B]) : 0.f);
                b += UPAD;
                while (b >= B) { b -= B; uu++; }
            }
        }
    };
    if (DCOMP_BIG_EARLY && p.obs && !(DCOMP_BIG_ABL & 4)) rows_early();
    // 6. rates after the move (base.py:451)
    float curr = 0.f;
    if (!RESET) {
        // With the first half of the rows gone (or no observation asked for) nothing needs log2 snr any more: every lane parks the RATES of its
        // connected stations, like the pre-move pass, and the owner threads evaluate none (mixed sharing: they walked 64 rows x 2 pairs each).
        const bool park = DCOMP_BIG_PARK && NWAVE <= 4 && (DCOMP_BIG_EARLY || !p.obs);     // (two more barriers: 16-wave workgroups lose 3 %)
        if (park) {
            __syncthreads();                                       // the readers of the log2 snr rows (rows_early) are done
            for (unsigned long long m = conn; m; m &= m - 1ull) { const int b = __ffsll((long long)m) - 1; myrow[b] = big_rate(myrow[b]); }
            __syncthreads();
        }
        aggregate(0, park);
        __syncthreads();
        curr = shared(false, park);
    }
    const float util = ue_utility(curr, step_util, dr_req);
    if (!RESET && active) {
        p.pos[idx] = make_double2(px, py);
        p.mv[idx] = mv;
        p.conn[idx] = (uint32_t)conn;
        x.conn_hi[idx] = (uint32_t)(conn >> 32);
        p.ewma[idx] = ewma;
    }
    util_s[tid] = alive ? util : 0.f;
    rb_s[tid] = alive ? reward_before : 0.f;
    __syncthreads();
    // 7. per-station utility aggregates (station.py:63-83), reward, info, observation
    aggregate(1, false);
    __syncthreads();
    const int n_eff = RESET ? p.U0 : U;
    if (kind == DCOMP_CENTRAL) {                                   // central.py:65-73: over the UEs' rewards_before
        if (active && u == 0) {
            const int base = env_local * UPAD;
            float r = p.reward_agg == DCOMP_REWARD_MIN ? 1.f : 0.f, su = 0.f;
            for (int v = 0; v < n_eff; v++) {
                r = p.reward_agg == DCOMP_REWARD_MIN ? fminf(r, rb_s[base + v]) : r + rb_s[base + v];
                su += util_s[base + v];
            }
            if (p.reward_agg == DCOMP_REWARD_AVG) r = r / (float)n_eff;
            if (p.reward) p.reward[env] = RESET ? 0.f : r;
            if (p.sum_util) p.sum_util[env] = su;
        }
    } else {
        float reward = 0.f;
        if (!RESET) {
            reward = util;                                         // multi_agent.py:52 (own utility, NOT normalised)
            if (p.reward_agg == DCOMP_REWARD_SUM) {                // multi_agent.py:73-79
                if (in_range != 0ull) {
                    const int base = env_local * UPAD;
                    float s = 0.f;
                    for (int v = 0; v < U; v++) if (mask_s[base + v] & conn) s += rb_s[base + v];
                    reward = s;
                }
            } else if (p.reward_agg == DCOMP_REWARD_AVG) {         // multi_agent.py:60-71
                float n = 0.f, t = 0.f;
                for (unsigned long long m = in_range; m; m &= m - 1ull) { const int q = env_local * B + __ffsll((long long)m) - 1; n += agg_n[q]; t += agg_u[q]; }
                if (n > 0.f) reward = conn == 0ull ? (t + util) / (n + 1.f) : t / n;
            } else {                                               // multi_agent.py:81-85, station.py:78-83
                float m_ = util;
                for (unsigned long long m = in_range; m; m &= m - 1ull) { const int q = env_local * B + __ffsll((long long)m) - 1; m_ = fminf(m_, agg_n[q] > 0.f ? agg_m[q] : MAX_UTIL); }
                reward = in_range != 0ull ? m_ : util;
            }
        }
        if (active && p.reward) big_store(&p.reward[idx], alive ? reward : 0.f);
        if (active && u == 0 && p.sum_util) {
            const int base = env_local * UPAD;
            float su = 0.f;
            for (int v = 0; v < n_eff; v++) su += util_s[base + v];
            p.sum_util[env] = su;
        }
    }
    if (active) {                                                  // base.py:383-411
        if (p.ue_dr) big_store(&p.ue_dr[idx], alive ? curr : 0.f);
        if (p.ue_util) big_store(&p.ue_util[idx], alive ? util : 0.f);
        if (p.rb_out) big_store(&p.rb_out[idx], alive ? reward_before : 0.f);
    }
    if (!p.obs || (DCOMP_BIG_ABL & 4)) return;
    if (!DCOMP_BIG_EARLY) rows_early();
    const float inv_u = 1.0f / (float)n_eff;
    // second half (variants.py:286-305): ues_at_bs | util_at_bs (the same in every row of an env: variants.py:296, 299) + the row's utility
    for (int el = 0; el < GPB && kind == DCOMP_MULTI; el++) {
        if (env0 + el >= p.E) break;                               // (uniform)
        float n_col = 0.f, u_col = 0.f;
        if (lane < B) {
            const int q = el * B + lane;
            const float n = agg_n[q];
            n_col = n * inv_u;
            u_col = agg_u[q] * fast_rcp(fmaxf(n, 1.f)) * (1.0f / MAX_UTIL);
        }
        float *const dst0 = p.obs + (size_t)(env0 + el) * U * ROW;
#pragma unroll DCOMP_BIG_WUNROLL
        for (int uu = w0; uu < U; uu += NWAVE) {
            const int r = el * UPAD + uu;
            const bool live = !RESET || uu < p.U0;
            float *const dst = dst0 + (size_t)uu * ROW;
            if (lane < B) {
                big_store(dst + 2 * B + lane, live ? n_col : 0.f);
                big_store(dst + 3 * B + lane, live ? u_col : 0.f);
            }
            if (lane == 0) big_store(dst + 4 * B, live ? util_s[r] * (1.0f / MAX_UTIL) : 0.f);
        }
    }
    if (kind == DCOMP_CENTRAL && env < p.E) {
        float *const dst = p.obs + (size_t)env * U * (2 * B + 1);
        const int base = env_local * UPAD;
        for (int c = u; c < U; c += UPAD) big_store(dst + 2 * UB + c, util_s[base + c] * (1.0f / MAX_UTIL));
    }
}

using BigKernelFn = void (*)(const KParams, const BigParams);
struct BigKernels { BigKernelFn step, reset; int gpb, block; };
BigKernels big_kernels_for_upad(int upad);

}  // namespace dcomp
