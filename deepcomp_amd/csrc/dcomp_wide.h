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
constexpr int WIDE_BC = DCOMP_WIDE_BC;   // BSs per chunk: 4 -> ~100 VGPRs (4-5 waves/SIMD), 8 -> ~145 (3 waves/SIMD)
// The CLI-default 'mixed' pattern cycles resource- / rate- / proportional-fair with the station index: chunks of THREE make the
// model of every chunk slot a compile-time constant (no scalar mode tests and branches per station): 0.1021 -> 0.0926 ms at config
// 5's per-GPU share.  The other patterns are faster with 4 (resource-fair 0.077 vs 0.086 ms, generic 0.105 vs 0.115 ms).
constexpr int wide_bc(int mp) { return mp == MP_MIXED ? 3 : WIDE_BC; }

template <int B, int UPAD>
struct alignas(16) WideShared {
    float drst[4][64 * (B + 1)];          // per-wave transpose of the per-UE `dr` observation (row stride B+1: conflict-free)
    float xw[2][4][2 * WIDE_BC];          // double-buffered per-wave partials of the cross-wave exchange
    float tab_cnt[4][B];                  // per env in this block: |S_b| / U          (variants.py:296)
    float tab_ub[4][B];                   //                        avg utility at b / 20 (variants.py:299)
    uint32_t nb_conn[256];                // 'sum' reward: conn' and reward_before of the block's UEs
    float nb_rb[256];
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
    __syncthreads();
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
template <int B, int NW, int MP, int BCC, bool FULL, class SH>
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
            bool f;                                            // straight-line: with 64 UEs of one env per wave a station is
            const float t = rate_unshared_small(l2[j], f);     // rarely empty, and a skip branch per station costs more
            dr[j] = c[j] ? t : 0.f;
            ex[j] = (float)group_popcount<64>(m, 0);
        }
    }
    if (near_hint) {
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
    if (any_sum) {                       // uniform (modes are uniform)
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

    double px = 0.0, py = 0.0;
    unsigned long long mv = 0;
    uint32_t conn = 0, act = 0;
    float ewma = 0.f;
    bool step_util = false;
    float dr_req = 1.f;
    int vrange = -1;
    if (active) {
        double2 q = p.pos[idx];
        px = q.x; py = q.y;
        mv = p.mv[idx];
        conn = p.conn[idx];
        ewma = p.ewma[idx];
        act = p.action[idx];
        if (p.rng_mode != DCOMP_RNG_TAPE || !p.all_log_util) {       // loaded here, next to the state, not in the middle of the move
            const UeCfg c = p.ue_cfg[u];
            step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req;
            vrange = (int)c.vel_lo | ((int)c.vel_hi << 8);
        }
    }
    if (act > (uint32_t)B) { atomicOr(p.flags, DCOMP_FLAG_BAD_ACTION); act = 0; }

    // move first (base.py:447 -> user.py:159-173); keep the old position for the pre-move pairs
    const double ox = px, oy = py;
    if (active) {
        move_ue(p, env, (uint32_t)u + 1u, p.episode, px, py, mv, vrange);
        if (px < 0.0 || py < 0.0 || px > (double)p.map_w || py > (double)p.map_h) atomicOr(p.flags, DCOMP_FLAG_OUTSIDE_MAP);
    }

    // ---- sweep 1: toggle, pre-move rates, drop, stale-rate EWMA term; post-move log2 snr kept in l2n[]
    float *const strow = sh.drst[wave] + lane * (B + 1);      // this lane's row: log2 snr' now, normalised dr later
    uint32_t inr_new = 0;
    bool near_any = false;                                     // wave-uniform: a lane came within 1.26 m of some BS this step
    float curr = 0.f, stale = 0.f, l2max = -1e30f;
    const float inv_ewma_old = fast_rcp(ewma + EPS);
    const uint32_t act_bit = act ? 1u << (act - 1u) : 0u;
    auto sweep1 = [&](const int c0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;       // every slot of the chunk is a station: no bounds tests
        bool c[BC];
        float l2o[BC], l2n[BC], dr[BC], cnt[BC];
        bool anytiny = false;
        uint32_t inr_old = 0;
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const int b = c0 + j;
            l2o[j] = -30.f;
            if (FULL || b < B) {
                bool inr_o, inr_n, t0, t1;
                pair_eval(ox, oy, p.bs_x[b], p.bs_y[b], p, inr_o, l2o[j], t0);
                pair_eval(px, py, p.bs_x[b], p.bs_y[b], p, inr_n, l2n[j], t1);
                anytiny |= t0 | t1;
                inr_new |= (uint32_t)inr_n << b;
                inr_old |= (uint32_t)inr_o << b;
            }
        }
        // toggle of the acted-on station if it is in this chunk, branch-free (base.py:259-263 -> user.py:190-222):
        // connected -> disconnect; not connected and in range at the pre-move position -> connect
        conn ^= act_bit & (conn | inr_old) & (((BC >= 32 ? 0u : (1u << BC)) - 1u) << c0);
#pragma unroll
        for (int j = 0; j < BC; j++) c[j] = (FULL || c0 + j < B) && ((conn >> (c0 + j)) & 1u);
        const bool near_chunk = __ballot(anytiny) != 0ull;       // a lane within 1.26 m of one of these stations (old or new position)
        near_any |= near_chunk;
        if (near_chunk) {
#pragma unroll
            for (int j = 0; j < BC; j++) {
                const int b = c0 + j;
                if (FULL || b < B) {
                    double dx = p.bs_x[b] - ox, dy = p.bs_y[b] - oy;
                    if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2o[j] = pair_eval_tiny(ox, oy, p.bs_x[b], p.bs_y[b], p);
                    dx = p.bs_x[b] - px; dy = p.bs_y[b] - py;
                    if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2n[j] = pair_eval_tiny(px, py, p.bs_x[b], p.bs_y[b], p);
                }
            }
        }
        wide_chunk_rates<B, NW, MP, BC, FULL>(p, sh, buf, c0, c, l2o, inv_ewma_old, wave, lane, dr, cnt, near_chunk);
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
    ewma = 0.9f * stale + 0.1f * ewma;                                            // user.py:148-157

    // ---- sweep 2: rates after the move (base.py:451)
    curr = 0.f;
    const float inv_ewma = fast_rcp(ewma + EPS);
    auto sweep2 = [&](const int c0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        bool c[BC];
        float l2c[BC], dr[BC], cnt[BC];
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const int b = c0 + j;
            c[j] = (FULL || b < B) ? (bool)((conn >> b) & 1u) : false;
            l2c[j] = (FULL || b < B) ? strow[b] : -30.f;
        }
        wide_chunk_rates<B, NW, MP, BC, FULL>(p, sh, buf, c0, c, l2c, inv_ewma, wave, lane, dr, cnt, near_any);
#pragma unroll
        for (int j = 0; j < BC; j++) if (FULL || c0 + j < B) curr += dr[j];
    };
#pragma unroll 1
    for (int c0 = 0; c0 < NFULL; c0 += BC) sweep2(c0, std::true_type{});
    if constexpr (NFULL < B) sweep2(NFULL, std::false_type{});
    const float util = ue_utility(curr, step_util, dr_req);
    if (active) {
        p.pos[idx] = make_double2(px, py);
        p.mv[idx] = mv;
        p.conn[idx] = conn;
        p.ewma[idx] = ewma;
    }

    // ---- sweep 3: per-BS utility aggregates (station.py:63-83), reward, per-env observation tables
    const bool multi = p.kind == DCOMP_MULTI;
    const bool need_min = multi && p.reward_agg == DCOMP_REWARD_MIN;
    float rn = 0.f, rt = 0.f, rmin = util;
    const float inv_u = 1.0f / (float)p.U;
#pragma unroll 1
    for (int c0 = 0; c0 < B; c0 += BC) {
        float ex[2 * BC], mn[BC];
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const int b = c0 + j;
            const bool cb = b < B ? (bool)((conn >> b) & 1u) : false;
            ex[j] = b < B ? (float)group_popcount<64>(__ballot(cb), 0) : 0.f;
            ex[BC + j] = cb ? util : 0.f;
            mn[j] = cb ? util : MAX_UTIL;
        }
        {
            float sv[BC];
#pragma unroll
            for (int j = 0; j < BC; j++) sv[j] = ex[BC + j];
            group_reduce_vec<64, OpSum, BC>(sv);
#pragma unroll
            for (int j = 0; j < BC; j++) ex[BC + j] = sv[j];
        }
        wide_xchg<2 * BC, NW, OpSum>(ex, sh, buf, wave, lane);
        if (need_min) {
            group_reduce_vec<64, OpMin, BC>(mn);
            wide_xchg<BC, NW, OpMin>(mn, sh, buf, wave, lane);
        }
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const int b = c0 + j;
            if (b < B) {
                const float n = ex[j], t = ex[BC + j];
                if ((inr_new >> b) & 1u) { rn += n; rt += t; rmin = fminf(rmin, n > 0.f ? mn[j] : MAX_UTIL); }
                if (lane == 0 && (wave % NW) == 0) {
                    sh.tab_cnt[env_local][b] = n * inv_u;
                    sh.tab_ub[env_local][b] = n > 0.f ? t * fast_rcp(n) * (1.0f / MAX_UTIL) : 0.f;
                }
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
            sh.nb_conn[tid] = active ? conn : 0u;
            sh.nb_rb[tid] = reward_before;
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
    if (p.sum_util) {                                                             // base.py:383-411
        float s[1];
        s[0] = group_reduce<64, OpSum>(active ? util : 0.f);
        wide_xchg<1, NW, OpSum>(s, sh, buf, wave, lane);
        if (active && u == 0) p.sum_util[env] = s[0];
    }
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
        }
        return;
    }
    // transpose the per-UE dr columns through LDS (lane r -> row r)
    float *st = sh.drst[wave];
#pragma unroll 4
    for (int b = 0; b < B; b++) strow[b] = fast_exp2(strow[b] - l2max);                       // variants.py:276-284
    __syncthreads();                                                              // tables of sweep 3 + staging visible
    // column slots of this lane: c = lane + 64 k.  Row layout: connected[B] | dr[B] | ues_at_bs[B] | util_at_bs[B] | utility
    constexpr int NSLOT = (ROW + 63) / 64;
    float pre[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
        const int c = lane + 64 * k;
        pre[k] = 0.f;
        if (c >= 2 * B && c < 3 * B) pre[k] = sh.tab_cnt[env_local][c - 2 * B];
        else if (c >= 3 * B && c < 4 * B) pre[k] = sh.tab_ub[env_local][c - 3 * B];
    }
    const unsigned long long am = __ballot(active);
    const int nrows = group_popcount<64>(am, 0);                                              // active lanes are lanes [0, nrows)
    const size_t row0 = (size_t)env * p.U + (size_t)(wave % NW) * 64;
    for (int r = 0; r < ((DCOMP_ABLATE & 8) ? 0 : nrows); r++) {
        const uint32_t conn_r = (uint32_t)__builtin_amdgcn_readlane((int)conn, r);
        const float util_r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(util_n), r));
        float *orow = p.obs + (row0 + r) * ROW;
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            const int c = lane + 64 * k;
            if (c < ROW) {
                float v = pre[k];
                if (c < B) v = (float)((conn_r >> c) & 1u);
                else if (c < 2 * B) v = st[r * (B + 1) + (c - B)];
                else if (c == 4 * B) v = util_r;
                orow[c] = v;                 // plain store: non-temporal 4-byte stores bypass L2 write-combining (measured slower)
            }
        }
    }
}

}  // namespace dcomp
