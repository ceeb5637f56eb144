// dcomp_wide.h -- step kernel for wide envs (U > 32 lanes per env: UPAD = 64 / 128 / 256), e.g. BASELINE config 5
// (128 UE x 32 BS).  Included by dcomp_device.h.
//
// Same semantics as step_kernel; different organisation, because with B = 32 fully unrolled register arrays
// (log2 snr, rates, counts, sums: 5 x 32 VGPRs) cap occupancy at 2 waves/SIMD:
//   * one wavefront holds UEs of ONE env, so everything per (env, BS) -- connected-UE count, sum 1/rate, sum
//     priority, sum utility -- is wave-uniform: it lives in scalar registers / LDS tables, not in per-lane arrays;
//   * the BS loop is a real loop over chunks of BC = 3 or 4 stations (not unrolled over B: bounded registers and code size); the
//     post-move log2 snr of all B stations is parked in the lane's own LDS row, which later becomes the `dr` transpose;
//   * the move is done FIRST (it does not depend on rates), so one sweep over the BS chunks can evaluate the pre-move
//     pair, the post-move pair, the toggle, the pre-move rate, the drop and the stale-rate EWMA term of a station
//     together (user.py:148-188), without keeping B pre-move rates alive;
//   * observation rows (4B+1 floats, 516 B at B = 32) are written row by row: lane c owns column c, c+64, c+128; the
//     per-env columns (ues_at_bs, util_at_bs) are preloaded once per lane from an LDS table, the per-UE `dr` columns
//     come from an LDS transpose, `connected` / `utility` of row r are wave-uniform (v_readlane).  Every store
//     instruction writes 256 contiguous bytes.
#pragma once
#include <type_traits>

namespace dcomp {

#ifndef DCOMP_WIDE_BC
#define DCOMP_WIDE_BC 4
#endif
#ifndef DCOMP_WIDE_BC_MIXED
#define DCOMP_WIDE_BC_MIXED 6
#endif
constexpr int WIDE_BC = DCOMP_WIDE_BC;
constexpr int WIDE_BC_MAX = DCOMP_WIDE_BC_MIXED > DCOMP_WIDE_BC ? DCOMP_WIDE_BC_MIXED : DCOMP_WIDE_BC;   // BSs per chunk: 4 -> ~100 VGPRs (4-5 waves/SIMD), 8 -> ~145 (3 waves/SIMD)
// The CLI-default 'mixed' pattern cycles resource- / rate- / proportional-fair with the station index: chunks of THREE make the
// model of every chunk slot a compile-time constant (no scalar mode tests and branches per station): 0.1021 -> 0.0926 ms at config
// 5's per-GPU share.  The other patterns are faster with 4 (resource-fair 0.077 vs 0.086 ms, generic 0.105 vs 0.115 ms).
constexpr int wide_bc(int mp) { return mp == MP_MIXED ? DCOMP_WIDE_BC_MIXED : WIDE_BC; }

template <int B, int UPAD>
struct alignas(16) WideShared {
    float drst[4][64 * (B + 1)];          // per-wave transpose of the per-UE `dr` observation (row stride B+1: conflict-free)
    float xw[2][4][2 * WIDE_BC_MAX];          // double-buffered per-wave partials of the cross-wave exchange
    // (sized to the shape: 4 workgroups of this kernel must keep fitting into the CU's 160 KB of LDS)
    float4 tab[256 / UPAD][32];           // per env in this block and station: {|S_b|, sum utility, min utility, sum of all utilities}
    float4 part[UPAD > 64 ? 4 : 1][32];   // the same per wave (cross-wave combine when an env spans several waves)
    double2 bs[32];                       // BS positions, indexed PER LANE in the sparse pre-move pass
    uint32_t nb_conn[256];                // conn' of the block's UEs (column sums below; 'sum' reward)
    float nb_util[256];                   // utility of the block's UEs
    float nb_rb[256];                     // 'sum' reward: reward_before of the block's UEs
};

// Combine N wave-uniform partials over the NW waves of an env; one barrier per call (buffers alternate).
template <int N, int NW, class Op, class SH>
__device__ __forceinline__ void wide_xchg(float (&v)[N], SH &sh, int &buf, int wave, int lane)
{
    if (NW == 1) return;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) sh.xw[buf][wave][i] = v[i];
    }
    if (!(DCOMP_ABLATE & 256)) __syncthreads();          // (ablation bit 256: timing without the cross-wave barriers; results are wrong)
    const int w0 = (wave / NW) * NW;
#pragma unroll
    for (int i = 0; i < N; i++) {
        float a = sh.xw[buf][w0][i];
#pragma unroll
        for (int k = 1; k < NW; k++) a = Op::f(a, sh.xw[buf][w0 + k][i]);
        v[i] = a;
    }
    buf ^= 1;
}

// Shared rates of one chunk of base stations (station.py:152-220).  c[j]: connected to BS c0+j; l2[j]: log2 snr.
// Returns the shared rate per station in dr[j] (0 where not connected) and |S_b| in cnt[j].
// PRE: l2[j] already holds the UNSHARED rate of the connected pairs (the sparse pre-move pass of step_kernel_wide).
template <int B, int NW, int MP, int BCC, bool FULL, bool PRE = false, class SH>
__device__ __forceinline__ void wide_chunk_rates(const KParams &p, SH &sh, int &buf, int c0, const bool (&c)[BCC],
                                                 const float (&l2)[BCC], float inv_ewma, int wave, int lane,
                                                 float (&dr)[BCC], float (&cnt)[BCC], bool near_hint)
{
    // near_hint (wave-uniform): some lane of this wave may be within 1.26 m of a BS (pair_eval's flag); only then can a pair
    // have snr > 1/64, which the short series does not cover (see shared_rates)
    float ex[2 * BCC];               // [0,BC): counts, [BC,2BC): sums
#pragma unroll
    for (int j = 0; j < BCC; j++) {
        dr[j] = 0.f; ex[j] = 0.f; ex[BCC + j] = 0.f;
        if (FULL || c0 + j < B) {
            const unsigned long long m = __ballot(c[j]);
            bool f = false;                                    // straight-line: with 64 UEs of one env per wave a station is
            const float t = PRE ? l2[j] : rate_unshared_small(l2[j], f);   // rarely empty, and a skip branch per station costs more
            dr[j] = c[j] ? t : 0.f;
            ex[j] = (float)group_popcount<64>(m, 0);
        }
    }
    if (!PRE && near_hint) {
#pragma unroll
        for (int j = 0; j < BCC; j++) if ((FULL || c0 + j < B) && c[j] && l2[j] > RATE_SMALL_L2) dr[j] = rate_unshared_any(l2[j]);
    }
    bool any_sum = false;
#pragma unroll
    for (int j = 0; j < BCC; j++) {
        if (FULL || c0 + j < B) {
            const int mode = bs_mode_of<MP>(p, c0 + j);
            if (mode == DCOMP_RATE_FAIR) { ex[BCC + j] = c[j] ? fast_rcp(dr[j]) : 0.f; any_sum = true; }
            else if (mode == DCOMP_PROP_FAIR) { ex[BCC + j] = dr[j] * inv_ewma; any_sum = true; }
        }
    }
    if (MP == MP_MIXED && BCC % 3 == 0) {
        // chunks start at a multiple of 3: slots 0, 3, ... are resource-fair (bs_mode_of) and need no sum over the UEs
        constexpr int NS = BCC - BCC / 3;
        float sv[NS];
#pragma unroll
        for (int j = 0, k = 0; j < BCC; j++) if (j % 3 != 0) sv[k++] = ex[BCC + j];
        group_reduce_vec<64, OpSum, NS>(sv);
#pragma unroll
        for (int j = 0, k = 0; j < BCC; j++) if (j % 3 != 0) ex[BCC + j] = sv[k++];
    } else if (any_sum) {                // uniform (modes are uniform)
        float sv[BCC];
#pragma unroll
        for (int j = 0; j < BCC; j++) sv[j] = ex[BCC + j];
        group_reduce_vec<64, OpSum, BCC>(sv);
#pragma unroll
        for (int j = 0; j < BCC; j++) ex[BCC + j] = sv[j];
    }
    wide_xchg<2 * BCC, NW, OpSum>(ex, sh, buf, wave, lane);
#pragma unroll
    for (int j = 0; j < BCC; j++) {
        cnt[j] = ex[j];
        if (FULL || c0 + j < B) {
            const int mode = bs_mode_of<MP>(p, c0 + j);
            const float dru = dr[j], agg = ex[BCC + j];
            float out;
            if (mode == DCOMP_RES_FAIR) out = dru * fast_rcp(fmaxf(cnt[j], 1.f));
            else if (mode == DCOMP_RATE_FAIR) out = fast_rcp(agg);
            else out = (dru * inv_ewma) * fast_rcp(agg + EPS) * dru;       // proportional-fair (max-cap never gets here)
            dr[j] = c[j] ? out : 0.f;
        }
    }
}

template <int B, int UPAD, int MP>
__global__ __launch_bounds__(256) void step_kernel_wide(const KParams p)
{
    static_assert(UPAD >= 64, "wide kernel: one wavefront holds UEs of a single env");
    constexpr int NW = UPAD / 64, GPB = 256 / UPAD, BC = wide_bc(MP), ROW = 4 * B + 1;
    __shared__ WideShared<B, UPAD> sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env_local = wave / NW, u = (wave % NW) * 64 + lane;
    const int env = blockIdx.x * GPB + env_local;
    const bool active = (env < p.E) && (u < p.U);
    const int idx = env * p.U + u;
    int buf = 0;
    if (tid < B) sh.bs[tid] = make_double2(p.bs_x[tid], p.bs_y[tid]);          // for the per-lane station index of the sparse pass

    double px = 0.0, py = 0.0;
    unsigned long long mv = 0;
    uint32_t conn = 0, act = 0;
    float ewma = 0.f;
    bool step_util = false;
    float dr_req = 1.f;
    int vrange = MV_CFG_ARRIVED;
    if (active) {
        double2 q = p.pos[idx];
        px = q.x; py = q.y;
        mv = p.mv[idx];
        conn = p.conn[idx];
        ewma = p.ewma[idx];
        act = p.action[idx];
        {                                                            // loaded here, next to the state, not in the middle of the move
            const UeCfg c = p.ue_cfg[u];
            step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req;
            vrange = mv_cfg_pack(c.vel_lo, c.vel_hi, c.pause, c.border);
        }
    }
    if (act > (uint32_t)B) { atomicOr(p.flags, DCOMP_FLAG_BAD_ACTION); act = 0; }

    __syncthreads();                                                              // BS table

    // ---- pre-move pass, SPARSE: the pre-move position only matters where this UE is connected (its rate before the move:
    // base.py:446, and the stale-rate EWMA term) and at the station it acts on (in range -> may connect, user.py:203-222):
    // typically 1-4 of the B stations.  Each lane walks ITS OWN set bits -- per-lane station index, BS position from the LDS
    // table -- and parks the unshared rate of station b in strow[b]; the station sweep below picks it up where the UE is
    // connected (other entries are never read) before it overwrites strow[b] with the post-move log2 snr.  The trip count is
    // the largest need-set in the wave (~5) instead of B = 32 dense evaluations of pair, log2 and rate series per lane.
    float *const strow = sh.drst[wave] + lane * (B + 1);      // this lane's row: pre-move rates -> log2 snr' -> normalised dr
    const uint32_t act_bit = act ? 1u << (act - 1u) : 0u;
    uint32_t inr_old = 0;                                      // in range at the OLD position (only bits of `need` are set)
    {
        uint32_t need = active ? (conn | act_bit) : 0u;
        while (__ballot(need != 0u) != 0ull) {
            if (need != 0u) {
                const int b = __ffs((int)need) - 1;
                need &= need - 1u;
                const double2 bp = sh.bs[b];
                bool ir, near;
                float l2;
                pair_eval(px, py, bp.x, bp.y, p, ir, l2, near);
                if (near) {                                        // rare, per lane: within 1.26 m of the station
                    const double dx = bp.x - px, dy = bp.y - py;
                    if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2 = pair_eval_tiny(px, py, bp.x, bp.y, p);
                }
                inr_old |= (uint32_t)ir << b;
                bool big;
                float dru = rate_unshared_small(l2, big);
                if (big) dru = rate_unshared_any(l2);              // rare: snr > 1/64
                strow[b] = dru;
            }
        }
    }
    // toggle (base.py:259-263 -> user.py:190-222): connected -> disconnect; not connected and in range at the pre-move position -> connect
    conn ^= act_bit & (conn | inr_old);
    // move (base.py:447 -> user.py:159-173)
    if (active) {
        move_ue(p, env, (uint32_t)u + 1u, p.episode, px, py, mv, vrange);
        if (px < 0.0 || py < 0.0 || px > (double)p.map_w || py > (double)p.map_h) atomicOr(p.flags, DCOMP_FLAG_OUTSIDE_MAP);
    }

    // ---- sweep 1: pre-move shared rates, post-move pairs, drop, stale-rate EWMA term; post-move log2 snr parked in strow[]
    uint32_t inr_new = 0;
    bool near_any = false;                                     // wave-uniform: a lane is within 1.26 m of some BS at the NEW position
    float curr = 0.f, stale = 0.f, l2max = -1e30f;
    const float inv_ewma_old = fast_rcp(ewma + EPS);
    auto sweep1 = [&](const int c0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;       // every slot of the chunk is a station: no bounds tests
        bool c[BC];
        float dru[BC], l2n[BC], dr[BC], cnt[BC];
        bool anytiny = false;
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const int b = c0 + j;
            c[j] = (FULL || b < B) && ((conn >> b) & 1u);
            dru[j] = 0.f; l2n[j] = -30.f;
            if (FULL || b < B) {
                bool inr_n, t1;
                pair_eval(px, py, p.bs_x[b], p.bs_y[b], p, inr_n, l2n[j], t1);
                anytiny |= t1;
                inr_new |= (uint32_t)inr_n << b;
                dru[j] = strow[b];                             // the sparse pass's rate where connected; unused otherwise
            }
        }
        const bool near_chunk = __ballot(anytiny) != 0ull;       // a lane within 1.26 m of one of these stations (new position)
        near_any |= near_chunk;
        if (near_chunk) {
#pragma unroll
            for (int j = 0; j < BC; j++) {
                const int b = c0 + j;
                if (FULL || b < B) {
                    const double dx = p.bs_x[b] - px, dy = p.bs_y[b] - py;
                    if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2n[j] = pair_eval_tiny(px, py, p.bs_x[b], p.bs_y[b], p);
                }
            }
        }
        wide_chunk_rates<B, NW, MP, BC, FULL, true>(p, sh, buf, c0, c, dru, inv_ewma_old, wave, lane, dr, cnt, false);
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const int b = c0 + j;
            if (FULL || b < B) {
                curr += dr[j];
                stale += ((inr_new >> b) & 1u) ? dr[j] : 0.f;                     // dr[j] is 0 unless connected
                l2max = fmaxf(l2max, l2n[j]);
                strow[b] = l2n[j];
            }
        }
    };
    constexpr int NFULL = B / BC * BC;
#pragma unroll 1
    for (int c0 = 0; c0 < NFULL; c0 += BC) sweep1(c0, std::true_type{});
    if constexpr (NFULL < B) sweep1(NFULL, std::false_type{});
    const float util_pre = ue_utility(curr, step_util, dr_req);
    const float reward_before = clamp_med3(util_pre, MIN_UTIL, MAX_UTIL) * (1.0f / MAX_UTIL);
    conn &= inr_new;                                                              // user.py:175-188
    ewma = __builtin_fmaf(0.9f, stale, 0.1f * ewma);   // one explicit contraction: every kernel variant rounds alike                                            // user.py:148-157

    // ---- sweep 2: rates after the move (base.py:451).  The unshared rate is only needed where the UE is (still) connected:
    // a sparse pass like the pre-move one computes it for the lane's own set bits and puts it into strow[b]; the log2 snr it
    // displaces there (needed again for the `dr` observation) waits in four registers and is put back after the sweep.  A
    // wavefront in which some UE holds more than four connections takes the dense path (rate series for every station).
    curr = 0.f;
    const float inv_ewma = fast_rcp(ewma + EPS);
    const uint32_t conn_post = active ? conn : 0u;
    const bool dense2 = __ballot(__builtin_popcount(conn_post) > 4) != 0ull;      // wave-uniform, rare
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    if (!dense2) {
        uint32_t todo = conn_post;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (__ballot(todo != 0u) == 0ull) break;
            if (todo != 0u) {
                const int b = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                const float l2 = strow[b];
                sv[k] = l2;
                bool big;
                float dru = rate_unshared_small(l2, big);
                if (big) dru = rate_unshared_any(l2);              // rare: snr > 1/64
                strow[b] = dru;
            }
        }
    }
    auto sweep2 = [&](const int c0, auto full_tag, auto pre_tag) {
        constexpr bool FULL = decltype(full_tag)::value, PRE = decltype(pre_tag)::value;
        bool c[BC];
        float l2c[BC], dr[BC], cnt[BC];
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const int b = c0 + j;
            c[j] = (FULL || b < B) ? (bool)((conn >> b) & 1u) : false;
            l2c[j] = (FULL || b < B) ? strow[b] : -30.f;           // PRE: the unshared rate where connected (unused elsewhere)
        }
        wide_chunk_rates<B, NW, MP, BC, FULL, PRE>(p, sh, buf, c0, c, l2c, inv_ewma, wave, lane, dr, cnt, near_any);
#pragma unroll
        for (int j = 0; j < BC; j++) if (FULL || c0 + j < B) curr += dr[j];
    };
    if (!dense2) {
#pragma unroll 1
        for (int c0 = 0; c0 < NFULL; c0 += BC) sweep2(c0, std::true_type{}, std::true_type{});
        if constexpr (NFULL < B) sweep2(NFULL, std::false_type{}, std::true_type{});
        uint32_t todo = conn_post;                                 // put the log2 snr back
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (todo != 0u) {
                const int b = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                strow[b] = sv[k];
            }
        }
    } else {
#pragma unroll 1
        for (int c0 = 0; c0 < NFULL; c0 += BC) sweep2(c0, std::true_type{}, std::false_type{});
        if constexpr (NFULL < B) sweep2(NFULL, std::false_type{}, std::false_type{});
    }
    const float util = ue_utility(curr, step_util, dr_req);
    if (active) {
        p.pos[idx] = make_double2(px, py);
        p.mv[idx] = mv;
        p.conn[idx] = conn;
        p.ewma[idx] = ewma;
    }

    // ---- sweep 3: per-BS utility aggregates (station.py:63-83), reward, per-env observation tables
    // Transposed: what a station needs -- |S_b|, sum and min of the utilities of its UEs -- is a function of just TWO words
    // per UE (connection mask, utility).  Every lane parks those two words in LDS; lane l then owns station l & 31 and adds up
    // the 32 UEs of its own half-wave (rows it reads are broadcast within the half), the two halves and the env's waves are
    // combined, and the totals go to a per-env table.  ~5 VALU per (station, UE-row) pair of a half-wave instead of a 64-lane
    // butterfly per station and sum kind, and 2 workgroup barriers per step instead of one or two per chunk of stations.
    const bool multi = p.kind == DCOMP_MULTI;
    sh.nb_conn[tid] = active ? conn : 0u;
    sh.nb_util[tid] = active ? util : 0.f;
    wave_lds_fence();
    {
        const int sb = lane & 31;                                 // my station (B <= 32)
        const int r0 = tid & ~31;                                 // first UE row of my half-wave
        float n = 0.f, t = 0.f, mn = MAX_UTIL, ta = 0.f;          // ta: sum of ALL utilities (info sum_utility, base.py:383-411)
        if (multi && p.reward_agg == DCOMP_REWARD_MIN) {          // the minimum is only needed by the 'min' reward (multi_agent.py:81-85)
#pragma unroll 8
            for (int r = 0; r < 32; r++) {
                const uint32_t cj = sh.nb_conn[r0 + r];
                const float uj = sh.nb_util[r0 + r];
                const bool bit = (cj >> sb) & 1u;
                n += bit ? 1.f : 0.f;
                t += bit ? uj : 0.f;
                mn = bit ? min_med3(mn, uj) : mn;
                ta += uj;
            }
        } else {
#pragma unroll 8
            for (int r = 0; r < 32; r++) {
                const uint32_t cj = sh.nb_conn[r0 + r];
                const float uj = sh.nb_util[r0 + r];
                const bool bit = (cj >> sb) & 1u;
                n += bit ? 1.f : 0.f;
                t += bit ? uj : 0.f;
                ta += uj;
            }
        }
        n += __shfl_xor(n, 32, 64); t += __shfl_xor(t, 32, 64); mn = min_med3(mn, __shfl_xor(mn, 32, 64)); ta += __shfl_xor(ta, 32, 64);
        if (NW > 1) {
            if (lane < 32) sh.part[wave][sb] = make_float4(n, t, mn, ta);
            __syncthreads();
            const int w0 = (wave / NW) * NW;
            float4 a = sh.part[w0][sb];
#pragma unroll
            for (int k = 1; k < NW; k++) { const float4 q = sh.part[w0 + k][sb]; a.x += q.x; a.y += q.y; a.z = min_med3(a.z, q.z); a.w += q.w; }
            n = a.x; t = a.y; mn = a.z; ta = a.w;
        }
        if ((wave % NW) == 0 && lane < 32) sh.tab[env_local][sb] = make_float4(n, t, mn, ta);
    }
    __syncthreads();                                                              // tables (and nb_*) visible to the block
    float rn = 0.f, rt = 0.f, rmin = util;
    const float inv_u = 1.0f / (float)p.U;
    if (multi && p.reward_agg != DCOMP_REWARD_SUM) {                              // multi_agent.py:60-71, 81-85
        // over the stations in range of this UE (1-2 of the B on the grid layouts): sparse again, per-lane station index
        uint32_t todo = active ? inr_new : 0u;
        while (__ballot(todo != 0u) != 0ull) {
            if (todo != 0u) {
                const int b = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                const float4 q = sh.tab[env_local][b];
                rn += q.x; rt += q.y; rmin = fminf(rmin, q.x > 0.f ? q.z : MAX_UTIL);
            }
        }
    }
    float reward = 0.f;
    if (!multi) {                                                                 // central.py:65-73
        float r[1];
        if (p.reward_agg == DCOMP_REWARD_MIN) {
            r[0] = group_reduce<64, OpMin>(active ? reward_before : 1.f);
            wide_xchg<1, NW, OpMin>(r, sh, buf, wave, lane);
        } else {
            r[0] = group_reduce<64, OpSum>(active ? reward_before : 0.f);
            wide_xchg<1, NW, OpSum>(r, sh, buf, wave, lane);
            if (p.reward_agg == DCOMP_REWARD_AVG) r[0] = r[0] / (float)p.U;
        }
        reward = r[0];
    } else {                                                                      // multi_agent.py:39-95
        reward = util;
        if (p.reward_agg == DCOMP_REWARD_SUM) {
            sh.nb_rb[tid] = reward_before;                                        // (nb_conn was written in sweep 3)
            __syncthreads();
            if (inr_new != 0) {
                float s = 0.f;
                const int base = env_local * UPAD;
                for (int v = 0; v < p.U; v++) if (sh.nb_conn[base + v] & conn) s += sh.nb_rb[base + v];
                reward = s;
            }
        } else if (p.reward_agg == DCOMP_REWARD_AVG) {
            if (rn > 0.f) reward = (conn == 0u) ? (rt + util) / (rn + 1.f) : rt / rn;
        } else {
            reward = rmin;
        }
    }
    if (p.sum_util && active && u == 0) p.sum_util[env] = sh.tab[env_local][0].w;  // base.py:383-411
    if (active) {
        if (p.ue_dr) p.ue_dr[idx] = curr;
        if (p.ue_util) p.ue_util[idx] = util;
        if (p.rb_out) p.rb_out[idx] = reward_before;
        if (p.reward) { if (multi) p.reward[idx] = reward; else if (u == 0) p.reward[env] = reward; }
    }
    const float util_n = util * (1.0f / MAX_UTIL);

    // ---- observation
    if (!multi) {                                                                 // central.py:31-57: connected | dr | utility blocks
        if (active) {
            float *base = p.obs + (size_t)env * p.U * (2 * B + 1);
#pragma unroll 4
            for (int b = 0; b < B; b++) {
                base[u * B + b] = (float)((conn >> b) & 1u);
                base[p.U * B + u * B + b] = fast_exp2(strow[b] - l2max);
            }
            base[2 * p.U * B + u] = util_n;
            if (p.next_act)                                                       // dcomp_set_policy: the rules on the entries just stored
                p.next_act[idx] = (uint8_t)policy_action_fn<B>(p, conn, [&](int b) { return fast_exp2(strow[b] - l2max); });
        }
        return;
    }
    // transpose the per-UE dr columns through LDS (lane r -> row r)
    float *st = sh.drst[wave];
#pragma unroll 4
    for (int b = 0; b < B; b++) strow[b] = fast_exp2(strow[b] - l2max);                       // variants.py:276-284
    if (p.next_act) {                                                             // dcomp_set_policy (uniform): the rules on this UE's dr row
        const int a = policy_action_fn<B>(p, conn, [&](int b) { return strow[b]; });
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
        if (c >= 2 * B && c < 3 * B) pre[k] = sh.tab[env_local][c - 2 * B].x * inv_u;                    // variants.py:296
        else if (c >= 3 * B && c < 4 * B) {                                                                 // variants.py:299
            const float4 q = sh.tab[env_local][c - 3 * B];
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
    for (int r0 = 0; r0 < ((DCOMP_ABLATE & 8) ? 0 : nrows); r0 += 4) {
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const int r = r0 + k4;
            if (r < nrows) {
                const uint32_t conn_r = (uint32_t)__builtin_amdgcn_readlane((int)conn, r);
                float *orow = p.obs + (row0 + r) * ROW;
#pragma unroll
                for (int k = 0; k < NSLOT; k++) {
                    const int c = lane + 64 * k;
                    if (c < 4 * B) {
                        float v = pre[k];
                        if (c < B) v = (float)((conn_r >> c) & 1u);
                        else if (c < 2 * B) v = st[r * (B + 1) + (c - B)];
                        orow[c] = v;         // plain store: non-temporal 4-byte stores bypass L2 write-combining (measured slower)
                    }
                }
            }
        }
        if (lane < 4 && r0 + lane < nrows)
            p.obs[(row0 + r0 + lane) * ROW + 4 * B] = sh.nb_util[wave * 64 + r0 + lane] * (1.0f / MAX_UTIL);
    }
}

}  // namespace dcomp
