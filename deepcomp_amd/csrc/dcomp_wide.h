// dcomp_wide.h -- step kernel for wide envs (U > 32 lanes per env: UPAD = 64 / 128 / 256), e.g. BASELINE config 5
// (128 UE x 32 BS).  Included by dcomp_device.h.
//
// Same semantics as step_kernel; different organisation, because with B = 32 fully unrolled register arrays
// (log2 snr, rates, counts, sums: 5 x 32 VGPRs) cap occupancy at 2 waves/SIMD:
//   * one wavefront holds UEs of ONE env, so everything per (env, BS) -- connected-UE count, sum 1/rate, sum
//     priority, sum utility -- is wave-uniform per station: it lives in small LDS tables, not in per-lane arrays;
//   * every lane owns a ROW of B floats in LDS (`strow`): first the unshared rates at the stations it is connected to
//     before the move, then the post-move log2 snr of all B stations, in between the unshared rates after the move, at the
//     end the normalised `dr` observation entries, which the row-by-row observation stores read column-wise (the transpose);
//   * rates are SPARSE: a UE is connected to 1-4 of the 32 stations, and only there does a rate exist.  Each lane walks its own
//     set bits (per-lane station index; BS position, per-station totals from LDS tables) for the pair / rate evaluation at the
//     pre-move position, for the rate series after the move and for the final shared rates;
//   * per-station sums over an env's UEs are TRANSPOSED: lane l owns station l & 31 and adds up the rows of its own half-wave
//     (conflict-free column reads of the rows, broadcast reads of the row's connection mask); halves are combined with one
//     cross-lane move, the waves of an env through a per-wave table and ONE workgroup barrier per sum (three per step: rates
//     before the move, rates after the move, utilities).  Round 2 swept the stations in chunks of 6 with a 64-lane butterfly per
//     station and sum (12 barriers, ~1 400 of the kernel's 2 800 VALU instructions per wave);
//   * the only dense per-station work left is what the observation format makes irreducible: the post-move pair evaluation of
//     all B stations (every station's relative SNR is an observation entry) and the row stores;
//   * observation rows (4B+1 floats, 516 B at B = 32) are written row by row: lane c owns column c, c+64, c+128; the
//     per-env columns (ues_at_bs, util_at_bs) are preloaded once per lane, the per-UE `dr` columns come from the rows in LDS,
//     `connected` / `utility` of row r are wave-uniform (v_readlane).  Every store instruction writes 256 contiguous bytes.
#pragma once
#include <type_traits>

namespace dcomp {

#ifndef DCOMP_WIDE_PC
#define DCOMP_WIDE_PC 8          // stations per trip of the dense post-move pair evaluation (BS positions: one s_load burst per trip)
#endif
#ifndef DCOMP_WIDE_QUAD_ROWS
#define DCOMP_WIDE_QUAD_ROWS 1   // B % 4 == 0: observation rows leave as 16-byte pieces, a half-wave per row (0: round-3 form, 4-byte columns; A/B)
#endif
#ifndef DCOMP_WIDE_NT_ROWS
#define DCOMP_WIDE_NT_ROWS 0     // non-temporal 16-byte row stores (A/B)
#endif
#ifndef DCOMP_WIDE_UTIL_EACH
#define DCOMP_WIDE_UTIL_EACH 1   // the utility float of a row (its last) is stored in the trip that stores the row, not every other trip: the
#endif                           // line it completes leaves L2 whole (32 768 envs: -3 %; tools/micro/store_patterns.hip pattern 5 vs 1)
#ifndef DCOMP_WIDE_PERSIST
#define DCOMP_WIDE_PERSIST 0     // experiment (round 4): persistent workgroups that request the next slot's state before they store this one's rows
#endif
#ifndef DCOMP_WIDE_STORE_PRIO
#define DCOMP_WIDE_STORE_PRIO 0  // A/B: the row-store loop (1) / everything behind the step's last barrier (2) at raised issue priority: +-0.2 %
#endif
#ifndef DCOMP_WIDE_K
#define DCOMP_WIDE_K 4           // connections per UE the register fast paths hold; a wave with a busier UE takes the LDS detours
#endif

template <int B, int UPAD>
struct alignas(16) WideShared {
    // per-wave rows, one per lane: B floats at stride B + 1 (odd for B = 32: a lane walking its own row and 32 lanes reading one
    // column are both conflict-free, and every cell address is `base + constant`: no per-access arithmetic)
    float drst[4][64 * (B + 1)];
    uint2 row_ci[256];                    // per UE row: {connection mask, 2nd word}: nothing the rate sums need; the UE's utility (bits) for the utility sums
    float4 part_a[4][32];                 // per wave and station: rate sums before the move {count, sum}, later the utility sums
                                          // {count, sum utility, min utility, sum of all utilities}
    union {
        float2 part_b[4][32];             // per wave and station: rate sums after the move {count, sum}
        float nb_rb[256];                 // 'sum' reward: reward_before of the block's UEs (after part_b's last reader)
    };
    double2 bs[32];                       // BS positions, indexed PER LANE in the sparse passes
    float4 nib[16];                       // `connected` pieces of the observation rows: entry n = the four floats of nibble n
    float xw[2][4];                       // central reward: per-wave partials
};
// (39.8 KB at B = 32: four workgroups per CU, as before -- the limit is 40 KB)

template <int B>
__device__ __forceinline__ int wide_col(int row, int c) { return row * (B + 1) + c; }

// Sharing model of station b for a PER-LANE b (station.py:152-202): compile-time patterns need no table.
template <int MP>
__device__ __forceinline__ int wide_mode_of(const KParams &p, const int32_t *lds_modes, int b)
{
    if (MP == MP_RES_FAIR) return DCOMP_RES_FAIR;
    if (MP == MP_MIXED) { const int m3 = b - 3 * ((b * 11) >> 5); return m3 == 0 ? DCOMP_RES_FAIR : m3 == 1 ? DCOMP_RATE_FAIR : DCOMP_PROP_FAIR; }   // b % 3 for b < 32
    return lds_modes[b];
}

// What a UE parks in its row for a station it is connected to: the TERM of that station's sum over its UEs -- rate-fair: 1 / rate
// (station.py:177-180), proportional-fair: rate / (ewma + eps) (station.py:150,192-195), resource-fair: nothing is summed, the
// rate itself waits there.  The summing lanes then only mask and add.
__device__ __forceinline__ float wide_park(int mode, float dru, float inv_e)
{
    return mode == DCOMP_RATE_FAIR ? fast_rcp(dru) : mode == DCOMP_PROP_FAIR ? dru * inv_e : dru;
}

// Transposed per-station sums over the UEs of this wave: lane l owns station l & 31 and the 32 rows of its half-wave.
// Per row: the station's bit of the row's connection mask as 0 / -1 (v_bfe_i32), AND, add -- 4 VALU.  Returns {number of
// connected UEs, sum of their terms} of the lane's station over the WAVE (both halves combined).
template <int B, int MP, class SH>
__device__ __forceinline__ float2 wide_rate_sums(SH &sh, int wave, int lane)
{
    const int sb = lane & 31, sbc = sb < B ? sb : 0;              // (lanes of stations that do not exist add up station 0, unused)
    const int half0 = (lane & 32);                                // first row of my half-wave within the wave
    const float *col = sh.drst[wave] + wide_col<B>(half0, sbc);   // my station's column, first row of my half
    const uint2 *ci = sh.row_ci + wave * 64 + half0;
    int n = 0;
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; r++) {
        const int m = __builtin_amdgcn_sbfe((int)ci[r].x, sbc, 1);            // 0 / -1; the mask read is a broadcast within the half-wave
        n -= m;
        if (MP != MP_RES_FAIR)                                                 // (resource-fair everywhere: counts only, station.py:171-173)
            a += __int_as_float(__float_as_int(col[r * (B + 1)]) & m);        // consecutive banks within the half-wave
    }
    float nf = (float)n;
    nf += __shfl_xor(nf, 32, 64);
    a += __shfl_xor(a, 32, 64);
    return make_float2(nf, a);
}

// State of one UE as the step reads it, and the per-UE configuration (the same in every env).
struct WideIn { double px, py; unsigned long long mv; uint32_t conn; float ewma; uint32_t act; };
struct WideCfg { bool step_util; float dr_req; int vrange; };

// The state of this lane's UE in the envs of workgroup-slot g (the loads are issued here; whoever reads the fields waits for them).
template <int UPAD>
__device__ __forceinline__ WideIn wide_load(const KParams &p, int g)
{
    constexpr int NW = UPAD / 64, GPB = 256 / UPAD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int env = g * GPB + wave / NW, u = (wave % NW) * 64 + lane;
    WideIn in{0.0, 0.0, 0ull, 0u, 0.f, 0u};
    if (env < p.E && u < p.U) {
        const int idx = env * p.U + u;
        const double2 q = p.pos[idx];
        in.px = q.x; in.py = q.y;
        in.mv = p.mv[idx];
        in.conn = p.conn[idx];
        in.ewma = p.ewma[idx];
        in.act = p.action[idx];
    }
    return in;
}

// One MobileEnv.step of the GPB envs of workgroup-slot g.  `in`: their state (already requested); on return it holds the state of
// slot g_next when has_next -- requested before this slot's observation rows are stored, see step_kernel_wide.
template <int B, int UPAD, int MP>
__device__ __forceinline__ void wide_step_one(const KParams &p, WideShared<B, UPAD> &sh, const int32_t *lds_modes, const WideCfg cfg,
                                              const int g, WideIn &in, const bool has_next, const int g_next)
{
    constexpr int NW = UPAD / 64, GPB = 256 / UPAD, ROW = 4 * B + 1, K = DCOMP_WIDE_K, PC = DCOMP_WIDE_PC;
    // (the thread index is hidden from the optimiser once per slot: everything derived from it -- lane constants, table addresses,
    // output offsets -- would otherwise be hoisted out of step_kernel_wide's slot loop and live in ~100 extra VGPRs across it)
    int tid = threadIdx.x;
    if (DCOMP_WIDE_PERSIST) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int env_local = wave / NW, u = (wave % NW) * 64 + lane;
    const int env = g * GPB + env_local;
    const bool active = (env < p.E) && (u < p.U);
    const int idx = env * p.U + u;
    const int w0 = (wave / NW) * NW;                               // first wave of my env

    double px = in.px, py = in.py;
    unsigned long long mv = in.mv;
    uint32_t conn = in.conn, act = in.act;
    float ewma = in.ewma;
    const bool step_util = cfg.step_util;
    const float dr_req = cfg.dr_req;
    const int vrange = cfg.vrange;
    if (act > (uint32_t)B) { atomicOr(p.flags, DCOMP_FLAG_BAD_ACTION); act = 0; }

    float *const st = sh.drst[wave];
    float *const myrow = st + wide_col<B>(lane, 0);
    auto cell = [&](int b) -> float & { return myrow[b]; };                       // my row, station b
    // log2 snr (+ offset) and unshared rate of this UE at station b, at the position it stands on (per-lane b)
    auto pair_at = [&](int b, bool &ir) -> float {
        const double2 bp = sh.bs[b];
        bool near;
        float l2;
        pair_eval(px, py, bp.x, bp.y, p, ir, l2, near);
        if (near) {                                            // rare, per lane: within 1.26 m of the station
            const double dx = bp.x - px, dy = bp.y - py;
            if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2 = pair_eval_tiny(px, py, bp.x, bp.y, p);
        }
        return l2;
    };
    auto rate_of = [&](float l2) -> float {
        bool big;
        float dru = rate_unshared_small(l2, big);
        if (big) dru = rate_unshared_any(l2);                  // rare: snr > 1/64
        return dru;
    };

    // ---- 1. pre-move pass, SPARSE: the pre-move position only matters where this UE is connected (its rate before the move:
    // base.py:446, and the stale-rate EWMA term) and at the station it acts on (in range -> may connect, user.py:203-222):
    // typically 1-4 of the B stations.  The station's sum term (wide_park) is parked in the lane's row, the unshared rate itself
    // waits in one of K registers (a wave with a UE of more than K stations to look at evaluates it again in step 3).
    const uint32_t act_bit = act ? 1u << (act - 1u) : 0u;
    const float inv_ewma_old = fast_rcp(ewma + EPS);
    uint32_t inr_old = 0;                                      // in range at the OLD position (only bits of `need` are set)
    const uint32_t need0 = active ? (conn | act_bit) : 0u;
    const bool many_pre = __ballot(__builtin_popcount(need0) > K) != 0ull;        // wave-uniform, rare
    float druk[K];
#pragma unroll
    for (int k = 0; k < K; k++) druk[k] = 0.f;
    {
        uint32_t need = need0;
        int k = 0;
        while (__ballot(need != 0u) != 0ull) {
            if (need != 0u) {
                const int b = __ffs((int)need) - 1;
                need &= need - 1u;
                bool ir;
                const float dru = rate_of(pair_at(b, ir));
                inr_old |= (uint32_t)ir << b;
#pragma unroll
                for (int j = 0; j < K; j++) if (j == k) druk[j] = dru;
                cell(b) = wide_park(wide_mode_of<MP>(p, lds_modes, b), dru, inv_ewma_old);
            }
            k++;
        }
    }
    // toggle (base.py:259-263 -> user.py:190-222): connected -> disconnect; not connected and in range at the pre-move position -> connect
    conn ^= act_bit & (conn | inr_old);
    const uint32_t conn_pre = active ? conn : 0u;
    sh.row_ci[tid] = make_uint2(conn_pre, 0u);
    wave_lds_fence();
    // ---- 2. per-station {|S_b|, sum} before the move (station.py:152-202), transposed
    {
        const float2 s = wide_rate_sums<B, MP>(sh, wave, lane);
        if (lane < 32) sh.part_a[wave][lane] = make_float4(s.x, s.y, 0.f, 0.f);
    }
    if (NW > 1) __syncthreads(); else wave_lds_fence();
    // ---- 3. shared rates before the move, where connected (sparse) -> reward_before (base.py:158-167); the rates stay in K
    // registers for the stale-rate EWMA term (user.py:148-157) -- a wave with a UE of more than K connections parks them in the
    // rows instead and the pair loop below picks them up
    float curr = 0.f, outk[K];
#pragma unroll
    for (int k = 0; k < K; k++) outk[k] = 0.f;
    auto shared_rate = [&](int b, float parked, float dru, auto part) -> float {
        float n = 0.f, a = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) { n += part[w0 + w][b].x; a += part[w0 + w][b].y; }
        const int mode = wide_mode_of<MP>(p, lds_modes, b);
        if (mode == DCOMP_RES_FAIR) return dru * fast_rcp(fmaxf(n, 1.f));                          // station.py:171-173
        if (mode == DCOMP_RATE_FAIR) return fast_rcp(a);                                           // station.py:180
        return parked * fast_rcp(a + EPS) * dru;                                                   // station.py:194-195 (parked = rate / ewma)
    };
    {
        uint32_t todo = need0;                                 // the stations of step 1, in the same order: slot k of druk
        int k = 0;
        while (__ballot(todo != 0u) != 0ull) {
            if (todo != 0u) {
                const int b = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                if ((conn_pre >> b) & 1u) {
                    float dru = 0.f;
                    if (many_pre) { bool ir; dru = rate_of(pair_at(b, ir)); }
                    else {
#pragma unroll
                        for (int j = 0; j < K; j++) if (j == k) dru = druk[j];
                    }
                    const float out = shared_rate(b, cell(b), dru, sh.part_a);
                    curr += out;
                    if (many_pre) cell(b) = out;
                    else {
#pragma unroll
                        for (int j = 0; j < K; j++) if (j == k) outk[j] = out;
                    }
                }
            }
            k++;
        }
    }
    const float util_pre = ue_utility(curr, step_util, dr_req);
    const float reward_before = clamp_med3(util_pre, MIN_UTIL, MAX_UTIL) * (1.0f / MAX_UTIL);
    // ---- 4. move (base.py:447 -> user.py:159-173)
    if (active) {
        move_ue(p, env, (uint32_t)u + 1u, p.episode, px, py, mv, vrange);
        if (px < 0.0 || py < 0.0 || px > (double)p.map_w || py > (double)p.map_h) atomicOr(p.flags, DCOMP_FLAG_OUTSIDE_MAP);
    }
    // ---- 5. pairs at the new position, ALL stations (every station's relative SNR is an observation entry): log2 snr into the row
    uint32_t inr_new = 0;
    float stale = 0.f, l2max = -1e30f;
    auto pairs = [&](const int c0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;       // every slot of the chunk is a station: no bounds tests
        float l2n[PC];
        bool anytiny = false;
#if DCOMP_EDGE_MODE == 0
        bool edge = false;
#else
        float emin = 3.0e38f;                                    // (the wide kernel runs two to four waves per SIMD: the vector-only form, dcomp_device.h DCOMP_EDGE_MODE)
#endif
#pragma unroll
        for (int j = 0; j < PC; j++) {
            const int b = c0 + j;
            l2n[j] = -30.f;
            if (FULL || b < B) {
                bool inr_n;
                float q;
                pair_eval_q(px, py, p.bs_x[b], p.bs_y[b], p, inr_n, l2n[j], q);
                anytiny |= q < NEAR_D2;
#if DCOMP_EDGE_MODE == 0
                edge |= q == p.dt2f;                             // the fused d^2 cannot decide this pair (dcomp_device.h, dist_sq_ref)
#else
                emin = min_med3(emin, __builtin_fabsf(q - p.dt2f));
#endif
                inr_new |= (uint32_t)inr_n << b;
            }
        }
#if DCOMP_EDGE_MODE != 0
        const bool edge = emin == 0.f;
#endif
#if !DCOMP_DSQ_FUSED
        if (__ballot(edge) != 0ull || p.dsq_exact) {            // rare (~1e-7 per pair), wave-uniform: this chunk's decisions in the reference's form
#pragma unroll
            for (int j = 0; j < PC; j++) {
                const int b = c0 + j;
                if (FULL || b < B) inr_new = (inr_new & ~(1u << b)) | ((uint32_t)in_range_exact(px, py, p.bs_x[b], p.bs_y[b], p.dt2) << b);
            }
        }
#endif
        if (__ballot(anytiny) != 0ull) {                       // a lane within 1.26 m of one of these stations (new position)
#pragma unroll
            for (int j = 0; j < PC; j++) {
                const int b = c0 + j;
                if (FULL || b < B) {
                    const double dx = p.bs_x[b] - px, dy = p.bs_y[b] - py;
                    if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2n[j] = pair_eval_tiny(px, py, p.bs_x[b], p.bs_y[b], p);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PC; j++) {
            const int b = c0 + j;
            if (FULL || b < B) {
                if (many_pre) {                                // the pre-move shared rate waits in the row: stale-rate EWMA term
                    const float old = myrow[c0 + j];
                    stale += (((conn_pre & inr_new) >> b) & 1u) ? old : 0.f;
                }
                l2max = fmaxf(l2max, l2n[j]);
                myrow[c0 + j] = l2n[j];
            }
        }
    };
    constexpr int NFULL = B / PC * PC;
#pragma unroll 1
    for (int c0 = 0; c0 < NFULL; c0 += PC) pairs(c0, std::true_type{});
    if constexpr (NFULL < B) pairs(NFULL, std::false_type{});
    if (!many_pre) {                                           // the usual case: the rates are in registers, in the order of `need0`
        uint32_t todo = need0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int b = todo ? __ffs((int)todo) - 1 : 0;
            stale += (todo && (((conn_pre & inr_new) >> b) & 1u)) ? outk[k] : 0.f;
            todo &= todo - 1u;
        }
    }
    conn &= inr_new;                                                              // user.py:175-188
    ewma = __builtin_fmaf(0.9f, stale, 0.1f * ewma);   // one explicit contraction: every kernel variant rounds alike   // user.py:148-157

    // ---- 6. rates after the move (base.py:451), sparse: the sum term of the stations the UE is (still) connected to goes into
    // the row in place of the log2 snr, which waits in K registers together with the unshared rate (a wave with a UE of more
    // than K connections evaluates the pair again instead)
    const float inv_ewma = fast_rcp(ewma + EPS);
    const uint32_t conn_post = active ? conn : 0u;
    const bool many_post = __ballot(__builtin_popcount(conn_post) > K) != 0ull;   // wave-uniform, rare
    float sv[K];
#pragma unroll
    for (int k = 0; k < K; k++) { sv[k] = 0.f; druk[k] = 0.f; }
    {
        uint32_t todo = conn_post;
        int k = 0;
        while (__ballot(todo != 0u) != 0ull) {
            if (todo != 0u) {
                const int b = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                const float l2 = cell(b);
                const float dru = rate_of(l2);
#pragma unroll
                for (int j = 0; j < K; j++) if (j == k) { sv[j] = l2; druk[j] = dru; }
                cell(b) = wide_park(wide_mode_of<MP>(p, lds_modes, b), dru, inv_ewma);
            }
            k++;
        }
    }
    sh.row_ci[tid] = make_uint2(conn_post, 0u);
    wave_lds_fence();
    float my_cnt;                                              // |S_b| after the move of the lane's station, over my wave (the utility sums reuse it)
    {
        const float2 s = wide_rate_sums<B, MP>(sh, wave, lane);
        my_cnt = s.x;
        if (lane < 32) sh.part_b[wave][lane] = s;
    }
    if (NW > 1) __syncthreads(); else wave_lds_fence();
    curr = 0.f;
    {
        uint32_t todo = conn_post;
        int k = 0;
        while (__ballot(todo != 0u) != 0ull) {
            if (todo != 0u) {
                const int b = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                float l2 = 0.f, dru = 0.f;
                if (many_post) { bool ir; l2 = pair_at(b, ir); dru = rate_of(l2); }       // the pair again (same code, same bits)
                else {
#pragma unroll
                    for (int j = 0; j < K; j++) if (j == k) { l2 = sv[j]; dru = druk[j]; }
                }
                curr += shared_rate(b, cell(b), dru, sh.part_b);
                cell(b) = l2;                                      // put the log2 snr back
            }
            k++;
        }
    }
    const float util = ue_utility(curr, step_util, dr_req);
    if (active) {
        p.pos[idx] = make_double2(px, py);
        p.mv[idx] = mv;
        p.conn[idx] = conn;
        p.ewma[idx] = ewma;
    }
    // The next slot's state is requested HERE, before this slot's observation rows are stored, and its arrival is awaited right
    // before the first row store (wide_arrived): vmcnt counts loads and stores in one in-order counter, so a load issued after
    // the 40 row stores could only be waited for together with them -- the wave would sit out the drain of its own stores before
    // it could compute again.  This way the rows of slot g drain while the wave computes slot g + grid (step_kernel_wide).
    // (Unconditional on purpose -- the last slot of a workgroup requests its own state again and drops it: the compiler does not
    // see that a conditional request and a conditional wait share their condition, and would guard every later write of these
    // registers, and the top of the slot loop, with a wait for "the request that was never awaited".)
    WideIn nx{0.0, 0.0, 0ull, 0u, 0.f, 0u};
    if (DCOMP_WIDE_PERSIST) nx = wide_load<UPAD>(p, has_next ? g_next : g);
    auto wide_arrived = [&]() {
        if (DCOMP_WIDE_PERSIST) {
            asm volatile("" : "+v"(nx.px), "+v"(nx.py), "+v"(nx.mv), "+v"(nx.conn), "+v"(nx.ewma), "+v"(nx.act));   // a use: the compiler's s_waitcnt goes here
            in = nx;
        }
    };

    // ---- 7. per-BS utility aggregates (station.py:63-83), transposed like the rate sums: what a station needs -- |S_b|, sum and
    // min of the utilities of its UEs -- is a function of just TWO words per UE (connection mask, utility).
    const bool multi = p.kind == DCOMP_MULTI;
    sh.row_ci[tid].y = __float_as_uint(active ? util : 0.f);      // (.x already holds the post-move masks)
    wave_lds_fence();
    {
        const int sb = lane & 31, sbc = sb < B ? sb : 0;          // my station (B <= 32)
        const uint2 *ci = sh.row_ci + (tid & ~31);                // the rows of my half-wave
        float t = 0.f, mn = MAX_UTIL, ta = 0.f;                   // ta: sum of ALL utilities (info sum_utility, base.py:383-411)
        if (multi && p.reward_agg == DCOMP_REWARD_MIN) {          // the minimum is only needed by the 'min' reward (multi_agent.py:81-85)
#pragma unroll 8
            for (int r = 0; r < 32; r++) {
                const uint2 c = ci[r];
                const int m = __builtin_amdgcn_sbfe((int)c.x, sbc, 1);
                const float uj = __uint_as_float(c.y);
                t += __int_as_float((int)c.y & m);
                mn = m ? min_med3(mn, uj) : mn;
                ta += uj;
            }
        } else {
#pragma unroll 8
            for (int r = 0; r < 32; r++) {
                const uint2 c = ci[r];
                const int m = __builtin_amdgcn_sbfe((int)c.x, sbc, 1);
                t += __int_as_float((int)c.y & m);
                ta += __uint_as_float(c.y);
            }
        }
        t += __shfl_xor(t, 32, 64); mn = min_med3(mn, __shfl_xor(mn, 32, 64)); ta += __shfl_xor(ta, 32, 64);
        if (lane < 32) sh.part_a[wave][sb] = make_float4(my_cnt, t, mn, ta);     // (the pre-move sums in part_a were consumed before the barrier of step 6)
    }
    __syncthreads();                                                              // per-wave tables (and nb_*) visible to the block
#if DCOMP_WIDE_STORE_PRIO == 2
    __builtin_amdgcn_s_setprio(3);                 // (A/B: everything behind the last barrier of the step at raised priority)
#endif
    auto station = [&](int b) -> float4 {                                         // totals of station b over the env's waves
        float4 a = sh.part_a[w0][b];
#pragma unroll
        for (int k = 1; k < NW; k++) { const float4 q = sh.part_a[w0 + k][b]; a.x += q.x; a.y += q.y; a.z = min_med3(a.z, q.z); a.w += q.w; }
        return a;
    };
    float rn = 0.f, rt = 0.f, rmin = util;
    const float inv_u = fast_rcp((float)p.U);                      // (v_rcp_f32: the IEEE division sequence is 11 instructions)
    if (multi && p.reward_agg != DCOMP_REWARD_SUM) {                              // multi_agent.py:60-71, 81-85
        // over the stations in range of this UE (1-2 of the B on the grid layouts): sparse again, per-lane station index
        uint32_t todo = active ? inr_new : 0u;
        while (__ballot(todo != 0u) != 0ull) {
            if (todo != 0u) {
                const int b = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                const float4 q = station(b);
                rn += q.x; rt += q.y; rmin = fminf(rmin, q.x > 0.f ? q.z : MAX_UTIL);
            }
        }
    }
    float reward = 0.f;
    if (!multi) {                                                                 // central.py:65-73
        float r;
        const bool is_min = p.reward_agg == DCOMP_REWARD_MIN;
        if (is_min) r = group_reduce<64, OpMin>(active ? reward_before : 1.f);
        else r = group_reduce<64, OpSum>(active ? reward_before : 0.f);
        if (NW > 1) {
            if (lane == 0) sh.xw[0][wave] = r;
            __syncthreads();
            r = sh.xw[0][w0];
#pragma unroll
            for (int k = 1; k < NW; k++) r = is_min ? OpMin::f(r, sh.xw[0][w0 + k]) : r + sh.xw[0][w0 + k];
        }
        if (p.reward_agg == DCOMP_REWARD_AVG) r = r * fast_rcp((float)p.U);
        reward = r;
    } else {                                                                      // multi_agent.py:39-95
        reward = util;
        if (p.reward_agg == DCOMP_REWARD_SUM) {
            sh.nb_rb[tid] = reward_before;                                        // (row_ci[].x: the post-move masks)
            __syncthreads();
            if (inr_new != 0) {
                float s = 0.f;
                const int base = env_local * UPAD;
                for (int v = 0; v < p.U; v++) if (sh.row_ci[base + v].x & conn) s += sh.nb_rb[base + v];
                reward = s;
            }
        } else if (p.reward_agg == DCOMP_REWARD_AVG) {
            if (rn > 0.f) { const bool lone = conn == 0u; reward = (lone ? rt + util : rt) * fast_rcp(lone ? rn + 1.f : rn); }   // one reciprocal, not two IEEE divisions
        } else {
            reward = rmin;
        }
    }
    if (p.sum_util && active && u == 0) p.sum_util[env] = station(0).w;           // base.py:383-411
    if (active) {
        if (p.ue_dr) p.ue_dr[idx] = curr;
        if (p.ue_util) p.ue_util[idx] = util;
        if (p.rb_out) p.rb_out[idx] = reward_before;
        if (p.reward) { if (multi) p.reward[idx] = reward; else if (u == 0) p.reward[env] = reward; }
    }
    const float util_n = util * (1.0f / MAX_UTIL);

    // ---- observation
    wide_arrived();
    if (!multi) {                                                                 // central.py:31-57: connected | dr | utility blocks
        if (active) {
            float *base = p.obs + (size_t)env * p.U * (2 * B + 1);
#pragma unroll 4
            for (int b = 0; b < B; b++) {
                base[u * B + b] = (float)((conn >> b) & 1u);
                base[p.U * B + u * B + b] = fast_exp2(cell(b) - l2max);
            }
            base[2 * p.U * B + u] = util_n;
            if (p.next_act)                                                       // dcomp_set_policy: the rules on the entries just stored
                p.next_act[idx] = (uint8_t)policy_action_fn<B>(p, conn, [&](int b) { return fast_exp2(cell(b) - l2max); });
        }
        return;
    }
    if (p.obs_compact) {                                                          // (uniform)
        // ---- the compact record instead of the rows (dcomp_out.obs_compact; layout in dcomp_fragment.h): per UE dr[B] | utility |
        // connection mask, then the per-env columns once.  A wave's records are ONE contiguous span of 64 (B + 2) words.  My row of the
        // LDS table (stride B + 1) gets the utility in its spare word, so [dr | utility] are B + 1 consecutive words per row and the
        // masks wait in row_ci: a lane walks (row, word) of its 16-byte pieces incrementally, one LDS read per word.
        constexpr int CW = B + 2;
#pragma unroll 4
        for (int b = 0; b < B; b++) cell(b) = fast_exp2(cell(b) - l2max);                     // variants.py:276-284
        myrow[B] = util_n;
        if (p.next_act) {                                                         // dcomp_set_policy (uniform): the rules on this UE's dr row
            const int a = policy_action_fn<B>(p, conn, [&](int b) { return cell(b); });
            if (active) p.next_act[idx] = (uint8_t)a;
        }
        wave_lds_fence();
        const unsigned long long am = __ballot(active);
        const int nrows = group_popcount<64>(am, 0);                              // active lanes are lanes [0, nrows)
        const size_t rec0 = ((size_t)env * p.U + (size_t)(wave % NW) * 64) * CW + (size_t)env * (2 * B);
        uint32_t *const dst = reinterpret_cast<uint32_t *>(p.obs) + rec0;
        const uint32_t *const stw = reinterpret_cast<const uint32_t *>(st);
        const uint2 *const ci = sh.row_ci + wave * 64;
        const int nw = nrows * CW;
        typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
        int r = (lane * 4) / CW, k = lane * 4 - r * CW;
        for (int w = lane * 4; w < nw; w += 256) {
            uint32_t o4[4];
            int r2 = r, k2 = k;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int rr = min(r2, 63);
                o4[j] = k2 == B + 1 ? ci[rr].x : stw[wide_col<B>(rr, k2)];
                k2++;
                if (k2 == CW) { k2 = 0; r2++; }
            }
            if (w + 3 < nw) { u4u v; v.x = o4[0]; v.y = o4[1]; v.z = o4[2]; v.w = o4[3]; *reinterpret_cast<u4u *>(dst + w) = v; }
            else for (int j = 0; j < 4; j++) if (w + j < nw) dst[w + j] = o4[j];
            r += 256 / CW; k += 256 % CW;
            if (k >= CW) { k -= CW; r++; }
        }
        if ((wave % NW) == 0 && nrows > 0 && lane < 2 * B) {                      // ues_at_bs | util_at_bs, once per env (variants.py:296-299)
            const float4 t = station(lane < B ? lane : lane - B);
            const float e = lane < B ? t.x * inv_u : (t.x > 0.f ? t.y * fast_rcp(t.x) * (1.0f / MAX_UTIL) : 0.f);
            reinterpret_cast<float *>(p.obs)[((size_t)env * p.U + p.U) * CW + (size_t)env * (2 * B) + lane] = e;
        }
        return;
    }
    if constexpr (B % 4 == 0 && DCOMP_WIDE_QUAD_ROWS) {
        // ---- rows leave as 16-BYTE pieces, a half-wave per row (round 4).  With B a multiple of 4 the four blocks of a row start at
        // multiples of 4 floats, so piece j (floats 4j .. 4j+3, j < B) of a row lies inside ONE block: lanes 0 .. B-1 of a half-wave
        // take the B pieces of one row -- a lane's kind of column is a constant of the whole loop -- and one store instruction
        // covers two whole rows (2 x 4B floats contiguous but for the utility float in between): 32 + 8 store instructions per
        // wave instead of 128 + 16 four-byte ones.  Global memory needs dword alignment only.  EVERY piece comes out of LDS with one
        // ds_read_b128, so the loop has no per-kind branches: `dr` pieces from the re-laid rows, the per-env columns from a 16-piece
        // table the wave builds once, `connected` from a constant table of the 16 possible nibbles (sh.nib).
        constexpr int QPR = B / 4;                                                // 16-byte pieces per block of a row
        float drv[B];
#pragma unroll
        for (int b = 0; b < B; b++) drv[b] = fast_exp2(cell(b) - l2max);                          // variants.py:276-284
        if (p.next_act) {                                                         // dcomp_set_policy (uniform): the rules on this UE's dr row
            const int a = policy_action<B>(p, conn, drv);
            if (active) p.next_act[idx] = (uint8_t)a;
        }
        // The dr entries are read back by OTHER lanes, 16 bytes at a time: rows at stride B words, the pieces of row l rotated by l
        // (piece q in slot (q + l) mod QPR): eight consecutive lanes writing their piece q -- and the QPR lanes reading one row --
        // touch every bank once.  The rows overlap the old stride-(B + 1) layout: every lane has read its row before any is rewritten.
        wave_lds_fence();
        float4 *const qst = reinterpret_cast<float4 *>(st);
        {
            const int rot = lane % QPR;
#pragma unroll
            for (int q = 0; q < QPR; q++) {
                int s = q + rot;
                s -= s >= QPR ? QPR : 0;
                qst[lane * QPR + s] = make_float4(drv[4 * q], drv[4 * q + 1], drv[4 * q + 2], drv[4 * q + 3]);
            }
        }
        const int h = lane >> 5, j = lane & 31;
        const int kind = j / QPR, q = j - kind * QPR;                             // 0 connected | 1 dr | 2 ues_at_bs | 3 util_at_bs (j < B)
        // the per-env columns, 2 QPR pieces, in the 64 words the re-laid rows leave free at the end of the wave's region
        float4 *const envq = qst + 64 * QPR;
        static_assert(64 * (B + 1) - 64 * B >= 8 * QPR, "the per-env pieces live behind the re-laid rows");
        if (kind >= 2 && j < B && h == 0) {
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 t = station(4 * q + k);
                e[k] = kind == 2 ? t.x * inv_u                                                          // variants.py:296
                                 : (t.x > 0.f ? t.y * fast_rcp(t.x) * (1.0f / MAX_UTIL) : 0.f);         // variants.py:299
            }
            envq[j - 2 * QPR] = make_float4(e[0], e[1], e[2], e[3]);
        }
        wave_lds_fence();                                                         // the re-laid rows and the table are this wave's own
        const unsigned long long am = __ballot(active);
        const int nrows = group_popcount<64>(am, 0);                              // active lanes are lanes [0, nrows)
        const size_t row0 = (size_t)env * p.U + (size_t)(wave % NW) * 64;
        float *const out0 = p.obs + (row0 + h) * ROW + 4 * j;                     // this lane's piece of row 2i + h, at i = 0
        const int npairs = (DCOMP_ABLATE & 8) ? 0 : (nrows + 1) >> 1;
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        // where this lane's piece of row 2i + h waits
        auto piece = [&](int i) -> const float4 * {
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)conn, 2 * i);
            const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)conn, 2 * i + 1);
            const uint32_t nib = ((h ? c1 : c0) >> (4 * q)) & 15u;
            int slot = (q + h + 2 * i) % QPR;                                     // (q + r) mod QPR, r = 2i + h   (QPR = 8: an AND)
            const float4 *a = kind == 1 ? qst + (2 * i + h) * QPR + slot : envq + (j - 2 * QPR);
            return kind == 0 ? sh.nib + nib : a;
        };
        auto put = [&](int i, const float4 v) {
            if (j < B && 2 * i + h < nrows) {
                f4u w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
                f4u *dst = reinterpret_cast<f4u *>(out0 + (size_t)i * (2 * ROW));
#if DCOMP_WIDE_NT_ROWS
                __builtin_nontemporal_store(w, dst);
#else
                *dst = w;
#endif
            }
        };
#if DCOMP_WIDE_STORE_PRIO == 1
        __builtin_amdgcn_s_setprio(3);             // see step_kernel_wide: a wave that has rows to store goes first
#endif
        for (int i = 0; i < npairs; i += 2) {                                     // two row pairs per trip: both LDS reads in flight
            const int i1 = min(i + 1, 31);
            const float4 *a0 = piece(i), *a1 = piece(i1);
            const float4 v0 = *a0, v1 = *a1;
            put(i, v0);
            if (i + 1 < npairs) put(i + 1, v1);
            // the utility column (the row's last float) of the eight rows just completed: their own lanes store it, right behind
#if DCOMP_WIDE_UTIL_EACH
            if ((lane >> 2) == (i >> 1) && active) p.obs[(row0 + lane) * ROW + 4 * B] = util_n;      // the four rows of this trip, at once
#else
            if (((i & 2) || i + 2 >= npairs) && (lane >> 3) == (i >> 2) && active)
                p.obs[(row0 + lane) * ROW + 4 * B] = util_n;
#endif
        }
#if DCOMP_WIDE_STORE_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        return;
    }
    // the per-UE dr columns: normalise in place, the row loop reads them column-wise (lane r's row -> output row r)
#pragma unroll 4
    for (int b = 0; b < B; b++) cell(b) = fast_exp2(cell(b) - l2max);                         // variants.py:276-284
    if (p.next_act) {                                                             // dcomp_set_policy (uniform): the rules on this UE's dr row
        const int a = policy_action_fn<B>(p, conn, [&](int b) { return cell(b); });
        if (active) p.next_act[idx] = (uint8_t)a;
    }
    wave_lds_fence();                                                             // the rows are this wave's own; the tables were fenced above
    // column slots of this lane: c = lane + 64 k.  Row layout: connected[B] | dr[B] | ues_at_bs[B] | util_at_bs[B] | utility
    constexpr int NSLOT = (4 * B + 63) / 64;                                      // the utility column (4B) goes separately, below
    float pre[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
        const int c = lane + 64 * k;
        pre[k] = 0.f;
        if (c >= 2 * B && c < 3 * B) pre[k] = station(c - 2 * B).x * inv_u;                               // variants.py:296
        else if (c >= 3 * B && c < 4 * B) {                                                                 // variants.py:299
            const float4 q = station(c - 3 * B);
            pre[k] = q.x > 0.f ? q.y * fast_rcp(q.x) * (1.0f / MAX_UTIL) : 0.f;
        }
    }
    const unsigned long long am = __ballot(active);
    const int nrows = group_popcount<64>(am, 0);                                              // active lanes are lanes [0, nrows)
    const size_t row0 = (size_t)env * p.U + (size_t)(wave % NW) * 64;
    // Rows go out four at a time; the utility column (one float per row, the row's last) is stored by lanes 0-3 right
    // behind its rows: one store instruction per four rows instead of a single-lane store per row -- and close in time to
    // the rest of the row.  (All 64 utility entries in ONE scattered store after the loop is faster while the observation
    // buffer fits the 256 MB Infinity Cache -- 4 096 envs -- and 45 % slower beyond: lines leave L2 partially written.)
    // Straight-line per row: every lane reads "its" dr cell of the row (lanes that own another kind of column read cell 0 and
    // drop it), the selects are per-lane constants -- no exec-mask branches, and the four LDS reads of a group of rows are in
    // flight together (the branchy form waited for each row's read: 64 exposed LDS round trips per wave).
    bool is_conn[NSLOT], is_dr[NSLOT];
    int dcol[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
        const int c = lane + 64 * k;
        is_conn[k] = c < B;
        is_dr[k] = c >= B && c < 2 * B;
        dcol[k] = is_dr[k] ? c - B : 0;
    }
    for (int r0 = 0; r0 < ((DCOMP_ABLATE & 8) ? 0 : nrows); r0 += 4) {
        float d[4][NSLOT];
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const int r = min(r0 + k4, 63);
#pragma unroll
            for (int k = 0; k < NSLOT; k++)
                if (64 * k < 2 * B && 64 * k + 64 > B) d[k4][k] = st[wide_col<B>(r, dcol[k])];     // (slot holds dr columns: compile-time)
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const int r = r0 + k4;
            if (r < nrows) {                                                      // uniform
                const uint32_t conn_r = (uint32_t)__builtin_amdgcn_readlane((int)conn, r);
                float *orow = p.obs + (row0 + r) * ROW;
#pragma unroll
                for (int k = 0; k < NSLOT; k++) {
                    const int c = lane + 64 * k;
                    float v = pre[k];
                    if (64 * k < 2 * B && 64 * k + 64 > B) v = is_dr[k] ? d[k4][k] : v;
                    if (64 * k < B) v = is_conn[k] ? (float)((conn_r >> (c & 31)) & 1u) : v;
                    if (4 * B - 64 * k >= 64 || c < 4 * B) orow[c] = v;  // plain store: non-temporal 4-byte stores bypass L2 write-combining (measured slower)
                }
            }
        }
        if (lane < 4 && r0 + lane < nrows)
            p.obs[(row0 + r0 + lane) * ROW + 4 * B] = __uint_as_float(sh.row_ci[wave * 64 + r0 + lane].y) * (1.0f / MAX_UTIL);
    }
}

// One workgroup per slot (GPB envs).  DCOMP_WIDE_PERSIST = 1 is the round-4 experiment that did NOT pay: persistent workgroups
// (at most as many as the GPU holds at once, each walking the slots g, g + grid, ...) that request the next slot's state before
// they store this slot's rows, so that a wave computes while its own rows drain -- vmcnt counts loads and stores in one in-order
// counter, so every wait for a load inside the slot loop had to go (VM_ARRIVED in the rare branches, the unconditional request).
// Same-box A/B: 4 096 x 128 x 32 54.5 -> 60.2 us, 32 768 envs 515 -> 541 us.  The slot loop costs 67 scalar spills and the
// 128-VGPR cap, and the overlap it was built for does not come from a wave's own stores draining: a wave in its store loop is
// held at ISSUE while the CU's memory pipe is backed up (tools/micro/store_patterns.hip: the rows alone take 39.5 us at 4 096
// envs, 410-450 us at 32 768), and while it is held its SIMD can only run the OTHER waves.  What helps is that the storing wave
// never waits for an issue slot behind computing waves: DCOMP_WIDE_STORE_PRIO.
template <int B, int UPAD, int MP>
__global__ __launch_bounds__(256, DCOMP_WIDE_PERSIST ? 4 : 1) void step_kernel_wide(const KParams p)
{
    static_assert(UPAD >= 64, "wide kernel: one wavefront holds UEs of a single env");
    constexpr int NW = UPAD / 64, GPB = 256 / UPAD;
    using SH = WideShared<B, UPAD>;
    __shared__ SH sh;
    __shared__ int32_t lds_modes[32];
    const int tid = threadIdx.x;
    const int total = (p.E + GPB - 1) / GPB;
    const int slot0 = DCOMP_WIDE_PERSIST ? (int)blockIdx.x : xcd_contiguous_block();     // every XCD a contiguous eighth of the slots
    WideIn in = wide_load<UPAD>(p, slot0);
    WideCfg cfg{false, 1.f, MV_CFG_ARRIVED};
    {
        const int u = ((tid >> 6) % NW) * 64 + (tid & 63);
        if (u < p.U) {
            const UeCfg c = p.ue_cfg[u];
            cfg.step_util = c.util == DCOMP_UTIL_STEP; cfg.dr_req = c.dr_req;
            cfg.vrange = mv_cfg_pack(c.vel_lo, c.vel_hi, c.pause, c.border);
        }
    }
    if (DCOMP_WIDE_PERSIST) {
        // the first slot's state has ARRIVED before the slot loop: inside it no wait for a load may remain
        asm volatile("" : "+v"(in.px), "+v"(in.py), "+v"(in.mv), "+v"(in.conn), "+v"(in.ewma), "+v"(in.act));
        asm volatile("" : "+v"(cfg.dr_req), "+v"(cfg.vrange));
    }
    if (tid < B) { sh.bs[tid] = make_double2(p.bs_x[tid], p.bs_y[tid]); lds_modes[tid] = p.bs_mode[tid]; }
    if (tid >= 64 && tid < 80) { const int n = tid - 64; sh.nib[n] = make_float4((float)(n & 1), (float)((n >> 1) & 1), (float)((n >> 2) & 1), (float)(n >> 3)); }
    __syncthreads();                                                              // BS table, nibble table
    if (!DCOMP_WIDE_PERSIST) {
        wide_step_one<B, UPAD, MP>(p, sh, lds_modes, cfg, slot0, in, false, 0);
        return;
    }
#pragma unroll 1
    for (int g = blockIdx.x; g < total; g += gridDim.x) {
        const int gn = g + gridDim.x;
        wide_step_one<B, UPAD, MP>(p, sh, lds_modes, cfg, g, in, gn < total, gn);
        if (gn < total) __syncthreads();                  // this slot's per-station tables are read until its waves have built their pieces
    }
}

}  // namespace dcomp
