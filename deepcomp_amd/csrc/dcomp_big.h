// dcomp_big.h -- the GENERIC env step: any station count up to 64, up to 1 024 UE slots per env, UE arrival / departure included
// (the reference limits neither: station.py:16-30, base.py:79-84, 433-443, 592-618).
//
// The kernels of dcomp_device.h / dcomp_wide.h unroll the station loop (one object file per B = 1 ... 32, per-station register arrays,
// a 32-bit connection mask): the right shape for the BASELINE configurations, not for arbitrary B.  Here ONE instantiation per
// lane-group width serves every B <= 64: B is a run-time value, the connection set is two 32-bit state words per UE (`conn` = stations
// 0-31, `conn_hi` = 32-63: dcomp_state.conn_hi), the BS table is a device array.  Same semantics and numerics as the specialised kernels
// (FP64 positions / decisions in the reference's form, FP32 log2-domain rates through the same device functions); tests/test_bigb_gpu.py
// holds it to the oracle at B = 33 ... 64 and to the specialised kernels at B <= 32 (DCOMP_FORCE_BIG=1).
//
// Round 6 rewrite.  Round 5 kept every UE's per-station values in a row of LDS (20.5 KB per wavefront at B = 64:
// eight wavefronts per CU) and had owner threads walk those rows: 6 190 VALU + 3 380 SALU + 880 LDS instructions per wave-step, 0.17-0.44 of
// the HBM peak.  Now NOTHING per (UE, station) lives in LDS:
//   * one lane = one (env, UE slot) for the state, the movement and the SPARSE work (a UE walks the bits of its own connection set: its shares,
//     the drop, the stale-rate EWMA -- a handful of stations, each pair re-evaluated where it is needed instead of parked);
//   * one lane = one STATION for everything that is per (env, station) or dense over the stations: the wave walks the UE rows of its envs,
//     a row's position and connection set arrive as ONE broadcast LDS read, and lane b evaluates (row, station b) -- the in-range compare's
//     lane mask IS the row's in-range set, a `connected` entry is one bit-field extract of the row's set, the row maximum of the relative-snr
//     block is one wave reduction, and the row leaves as four coalesced stores (B floats each) straight from registers: no transposition buffer;
//   * per-station utility sums: lane b adds bit b of every row's mask (five instructions per row);
//   * per-station connected counts and sums of the rate-fair / proportional-fair terms: the UE lanes add them into the per-env arrays in LDS
//     themselves, each for the few stations of its own set -- `ds_add_f32` of 1.0 for the counts (exact, any order), the terms UE by UE (a
//     uniform loop enables the lanes of UE index uu; LDS operations of a wavefront complete in program order): deterministic;
//   * envs wider than a wavefront combine per-wave partial sums in wave order.
// LDS per wavefront: 2.6 KB of per-lane slots + the tables -- the kernel is bound by registers (no per-B arrays), not by LDS; 513 ... 1 024 UE
// slots no longer limit the station count.  UE arrival / departure (DYN): slots shift inside the env's lane group exactly as in dcomp_dyn.h.
// The compact record (dcomp_out.obs_compact; two set words per UE above 32 stations) is written here too.  Not here: the fused rollout
// (dcomp_rollout_ex launches one step per launch) and the in-step policy.
#pragma once
#include "dcomp_device.h"

#ifndef DCOMP_BIG_RUNROLL
#define DCOMP_BIG_RUNROLL 2   // UE rows of the observation loop in flight per wavefront
#endif
#ifndef DCOMP_BIG_NT
#define DCOMP_BIG_NT 0        // 1: non-temporal row stores (partial lines: measured slower in round 5, R5.7)
#endif
#ifndef DCOMP_BIG_ABL
#define DCOMP_BIG_ABL 0       // timing-only ablation (results WRONG): 1 max-cap winner, 2 utility aggregates, 4 rows, 8 row pairs, 16 sharing aggregates, 32 move, 64 sparse share loops
#endif

namespace dcomp {

struct BigParams {
    uint32_t *conn_hi;                 // [E*U] stations 32 ... 63 of the connection set
    const double2 *bs;                 // [B] station positions
    const int32_t *mode;               // [B] DCOMP_*_FAIR / MAX_CAP
    int32_t B;
    unsigned long long maxcap_mask;    // bit b: station b is max-cap
    unsigned long long summode_mask;   // bit b: station b is rate-fair or proportional-fair (its share needs a sum of per-pair terms)
    int32_t row_x4;                    // multi-agent rows of more than 32 stations leave as ONE 16-byte store per lane and row (see the row loop)
};

// Workgroup: ONE wavefront (64 / UPAD envs), or the env's own UPAD lanes above 64.
__host__ __device__ constexpr int big_block(int upad) { return upad < 64 ? 64 : upad; }
// LDS one workgroup carves (host and device use the same function): the BS table, then per LANE 16 B position + 16 B {mask lo, mask hi,
// ewma | utility, reward_before} + 8 B in-range set, per (env, station) three aggregates, per (wave, station) three partial sums where an
// env spans waves.  64 stations, 32 UEs: 5.4 KB per wavefront.
struct BigCarve { int mode, pos, slot, inr, agg, part, stage, total; };
__host__ __device__ inline BigCarve big_carve(int B, int gpb, int blk)
{
    BigCarve c;
    int o = B * (int)sizeof(double2);
    c.mode = o; o += B * 4; o = (o + 15) & ~15;
    c.pos = o; o += blk * 16;
    c.slot = o; o += blk * 16;
    c.inr = o; o += blk * 8;
    c.agg = o; o += 3 * gpb * B * 4 + 16; o = (o + 15) & ~15;      // (+ one scratch word: where the empty term slots add their 0.0)
    c.part = o; o += blk > 64 ? 3 * (blk / 64) * B * 4 : 0; o = (o + 15) & ~15;
    c.stage = o; o += B > 32 ? (blk / 64) * 4 * B * 4 : 0;           // one row (4B words) per wavefront: the 16-byte form of the multi-agent rows (row_x4)
    c.total = (o + 15) & ~15;
    return c;
}

__device__ __forceinline__ void big_pair(double px, double py, const double2 bp, const KParams &p, bool &in_range, float &l2)
{
    bool near;
    pair_eval(px, py, bp.x, bp.y, p, in_range, l2, near);          // (incl. the reference-form re-check at the range boundary)
    if (near) {                                                    // rare, per lane: within 1.26 m of the station
        const double dx = bp.x - px, dy = bp.y - py;
        if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2 = pair_eval_tiny(px, py, bp.x, bp.y, p);
    }
}
// Every row store: PLAIN.  The rows leave as B floats per block at a stride of 4B + 1 floats, i.e. as partial cache lines, and a non-temporal
// partial line does not wait in the L2 for its other half (round 5, 8 192 x 32 x 64: 134 us non-temporal, 81 us plain).
__device__ __forceinline__ void big_store(float *ptr, float v)
{
#if DCOMP_BIG_NT
    stream_store(ptr, v);
#else
    *ptr = v;
#endif
}
// The observation rows leave through BUFFER stores: resource = the env's rows (uniform, four SGPRs), scalar offset = the row / the block of the row,
// vector offset = the lane's station (loop-invariant).  As plain global stores every row cost ten 64-bit vector adds on per-lane pointers (one
// v_lshl_add_u64 per store + one per pointer to step to the next row): a quarter of the row loop's vector instructions.
using BigRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BigRsrc big_rsrc(float *base) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ void big_bstore(BigRsrc r, uint32_t lane_bytes, uint32_t row_bytes, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)lane_bytes, (int)row_bytes, 0);
}
__device__ __forceinline__ void big_bstore4(BigRsrc r, uint32_t lane_bytes, uint32_t row_bytes, const float (&v)[4])
{
    using u4 = unsigned int __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, r, (int)lane_bytes, (int)row_bytes, 0);
}
__device__ __forceinline__ float big_rate(float l2)                // bw * log2(1 + snr), station.py:129-138
{
    bool big;
    float r = rate_unshared_small(l2, big);
    if (big) r = rate_unshared_any(l2);
    return r;
}
// max over the 64 lanes of a wavefront, result uniform (an SGPR): six v_max_f32 with DPP operands + one v_readlane.  (The compiler's form of the
// same butterfly -- group_reduce<64, OpMax> -- is a v_mov_b32_dpp + v_med3_f32 pair per stage plus a ds_swizzle and a ds_bpermute: 14
// instructions on the critical path of every observation row.)  Needs every lane active; no NaN reaches it.
__device__ __forceinline__ float wave_max_f32(float v)
{
    int s;
    asm volatile("s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_readlane_b32 %1, %0, 63\n"
                 "s_nop 3\n"
                 : "+v"(v), "=s"(s));
    return __int_as_float(s);
}
// (row, station) for a lane in its STATION role: log2 snr and the in-range decision.  The two rare cases -- the fused d^2 cannot decide
// (pair_eval_q), a UE within 1.26 m of the station -- sit behind ONE wave-uniform branch (a per-lane `if` here is if-converted: v_sqrt + a
// second v_log in every row).  ok: the lane holds a station and the row exists.
__device__ __forceinline__ void big_row_pair(double qx, double qy, const double2 bp, const KParams &p, bool ok, bool &in_range, float &l2)
{
    float qf;
    pair_eval_q(qx, qy, bp.x, bp.y, p, in_range, l2, qf);
    const bool special = ok && (qf == p.dt2f || qf < NEAR_D2);
    if (__builtin_amdgcn_ballot_w64(special) != 0ull || p.dsq_exact) {
        if (ok && (qf == p.dt2f || p.dsq_exact)) in_range = in_range_exact(qx, qy, bp.x, bp.y, p.dt2);
        if (ok && qf < NEAR_D2) {
            const double dx = bp.x - qx, dy = bp.y - qy;
            if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2 = pair_eval_tiny(qx, qy, bp.x, bp.y, p);
        }
    }
    in_range = in_range && ok;
    l2 = ok ? l2 : -3.0e38f;
}
// a UE's share of one station (station.py:152-202): n = connected UEs, s = the station's sum of sharing terms
__device__ __forceinline__ float big_share(int mode, float dru, float n, float s, float inv_ewma, bool winner)
{
    if (mode == DCOMP_RES_FAIR) return dru * fast_rcp(fmaxf(n, 1.f));                        // station.py:171-173
    if (mode == DCOMP_RATE_FAIR) return fast_rcp(s);                                         // station.py:180
    if (mode == DCOMP_PROP_FAIR) return (dru * inv_ewma) * fast_rcp(s + EPS) * dru;          // station.py:194-195
    return winner ? dru : 0.f;                                                               // station.py:183-187
}

// The in-step heuristic policy (dcomp_set_policy; agent/heuristics.py:13-187) on one observation row held by the lanes in their STATION role --
// the rules, first-maximum ties and index order of heuristic_kernel (dcomp_api.hip) / policy_action (dcomp_device.h), on the dr values the row
// loop is about to store.  A row over the whole wavefront (more than 32 stations): argmax, sets and candidates are lane masks of compares.
// d: the lane's dr entry; ok: the lane holds a station (ok_mask: those lanes); conn: the row's connection set (uniform).  Every lane must be here.
__device__ __forceinline__ int big_policy_wave(const KParams &p, unsigned long long conn, float d, bool ok, unsigned long long ok_mask, int lane, int B)
{
    d = ok ? d : -3.0e38f;
    const float mx = wave_max_f32(d);
    const int best = __ffsll((long long)(__builtin_amdgcn_ballot_w64(d == mx) & ok_mask)) - 1;      // np.argmax: the first maximum (heuristics.py:27)
    if (p.policy == DCOMP_POLICY_3GPP)                                                           // heuristics.py:30-38
        return ((conn >> best) & 1ull) ? 0 : conn ? __ffsll((long long)conn) : best + 1;
    unsigned long long sel = ok_mask;                                                            // FullCoMP: every cell
    if (p.policy == DCOMP_POLICY_DYNAMIC) sel = __builtin_amdgcn_ballot_w64(d >= mx * p.policy_eps) & ok_mask;      // heuristics.py:87-90
    else if (p.policy == DCOMP_POLICY_CLUSTER)                                                   // :172-176; two words (lo, hi) per station beyond 32 stations
        sel = B <= 32 ? (unsigned long long)p.policy_cluster[best] : ((unsigned long long)p.policy_cluster[2 * best] | ((unsigned long long)p.policy_cluster[2 * best + 1] << 32));
    const unsigned long long drop = conn & ~sel;
    const unsigned long long cand = sel & ~conn & ok_mask;
    const bool c = (cand >> lane) & 1ull;
    const float m2 = wave_max_f32(c ? d : -3.0e38f);                                             // strongest candidate, first of equals (:57-63, :101-106)
    const int a = __ffsll((long long)__builtin_amdgcn_ballot_w64(c && d == m2));
    return drop ? __ffsll((long long)drop) : cand ? a : 0;                                       // cells outside the set first, index order (:96-99, :178-181)
}
// The same for the rows of a trip that holds SEVERAL rows (up to 32 stations): lane = (row `sub` of the trip, station sb), BP lanes per row.
template <int BP>
__device__ __forceinline__ int big_policy_group(const KParams &p, uint32_t conn, float d, bool ok, int sub, int sb, int B)
{
    const uint32_t all = B == 32 ? ~0u : (1u << (B & 31)) - 1u;
    auto mine = [&](unsigned long long bal) { return (uint32_t)(bal >> (sub * BP)) & all; };      // this row's lanes of a wave-wide mask
    d = ok ? d : -3.0e38f;
    const float mx = group_reduce<BP, OpMax>(d);
    const uint32_t ism = mine(__builtin_amdgcn_ballot_w64(d == mx));
    const int best = ism ? __builtin_ffs((int)ism) - 1 : 0;
    uint32_t sel = all;
    if (p.policy == DCOMP_POLICY_DYNAMIC) sel = mine(__builtin_amdgcn_ballot_w64(d >= mx * p.policy_eps));
    else if (p.policy == DCOMP_POLICY_CLUSTER) sel = p.policy_cluster[best];
    const uint32_t drop = conn & ~sel, cand = sel & ~conn & all;
    const bool c = ok && ((cand >> sb) & 1u);
    const float m2 = group_reduce<BP, OpMax>(c ? d : -3.0e38f);
    const int a = __builtin_ffs((int)mine(__builtin_amdgcn_ballot_w64(c && d == m2)));
    if (p.policy == DCOMP_POLICY_3GPP) return ((conn >> best) & 1u) ? 0 : conn ? __builtin_ffs((int)conn) : best + 1;
    return drop ? __builtin_ffs((int)drop) : cand ? a : 0;
}

template <int UPAD, bool RESET, bool DYN, bool COMPACT, bool POL, bool ROLL = false>
__global__ __launch_bounds__(big_block(UPAD)) void big_kernel(const KParams p, const BigParams x)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    constexpr int BLK = big_block(UPAD), NWAVE = BLK / 64, GPB = BLK / UPAD;
    constexpr int EPW = UPAD >= 64 ? 1 : 64 / UPAD;                // env pieces a wavefront holds
    constexpr int PL = UPAD >= 64 ? 64 : UPAD;                     // lane slots per piece
    {                                                              // the BS table, once per launch
        const int B = x.B;
        const BigCarve cv = big_carve(B, GPB, BLK);
        double2 *const bs_s = reinterpret_cast<double2 *>(big_smem);
        int *const mode_s = reinterpret_cast<int *>(big_smem + cv.mode);
        for (int i = threadIdx.x; i < B; i += BLK) { bs_s[i] = x.bs[i]; mode_s[i] = x.mode[i]; }
    }
    // ROLL: the fused rollout (dcomp_rollout on the generic kernel, round 6) -- p.num_steps consecutive steps in ONE launch: the UE state stays in
    // these registers from step to step (it is still written back every step: 36 bytes per UE against a row of 4B + 1 floats), step t's outputs go
    // to slice t of the caller's [T][...] buffers (out_every_step) or only the last step's are written, the actions come from the tape -- or, in
    // the closed loop (policy_loop), from the next_action buffer the step before wrote.  Resets at the horizon are the host's (one launch per
    // stretch of an episode, the reset kernel in between).
    static_assert(!ROLL || (!RESET && !DYN), "the fused rollout steps a fixed UE list");
    double px = 0.0, py = 0.0;
    unsigned long long mv = 0, conn = 0;
    float ewma = 0.f;
    const int nsteps = ROLL ? p.num_steps : 1;
    const bool ploop = ROLL && POL && p.policy_loop != 0;
    // (one step as a lambda, called once or from the step loop: a `for` around the body -- even one of a single trip -- cost the plain step 11 VGPRs and a wave of occupancy)
    auto one_step = [&](const int t) __attribute__((always_inline)) {
    int B = x.B, U = p.U;
    if (ROLL) asm volatile("" : "+s"(B), "+s"(U));               // (fused rollout: opaque per step, like tid below -- or every derived offset is carried across the loop in spilled SGPRs)
    const bool any_maxcap = x.maxcap_mask != 0ull, any_sum = x.summode_mask != 0ull;
    const BigCarve cv = big_carve(B, GPB, BLK);
    double2 *const bs_s = reinterpret_cast<double2 *>(big_smem);
    int *const mode_s = reinterpret_cast<int *>(big_smem + cv.mode);
    double2 *const pos_s = reinterpret_cast<double2 *>(big_smem + cv.pos);
    uint4 *const slot_s = reinterpret_cast<uint4 *>(big_smem + cv.slot);               // {mask lo, mask hi, ewma | utility, reward_before}
    unsigned long long *const inr_s = reinterpret_cast<unsigned long long *>(big_smem + cv.inr);
    float *const agg_n = reinterpret_cast<float *>(big_smem + cv.agg), *const agg_s = agg_n + GPB * B, *const agg_u = agg_s;
    uint32_t *const mc_win = reinterpret_cast<uint32_t *>(agg_s + GPB * B);
    float *const agg_m = reinterpret_cast<float *>(mc_win);
    float *const part_s = reinterpret_cast<float *>(big_smem + cv.part);               // [3][NWAVE][B] (envs wider than a wavefront)

    int tid = threadIdx.x;
    if (ROLL) asm volatile("" : "+v"(tid));                        // (fused rollout: opaque per step, or the compiler carries every per-lane address and
                                                                   //  role value across the step loop -- 172 VGPRs, two waves per SIMD)
    const int lane = tid & 63;
    const int wave = NWAVE > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;     // (uniform, and the compiler must know it: every row loop below is scalar control flow)
    const int env_local = tid / UPAD, u = tid % UPAD;
    const int env0 = xcd_contiguous_block() * GPB, env = env0 + env_local;
    const bool active = env < p.E && u < U;
    int cur = DYN ? p.cur_ue : (RESET ? p.U0 : U);                 // UEs in the list (the same in every env: the schedule is configuration)
    bool alive = active && u < cur;
    const int idx = env * U + u;
    // the lane's STATION role.  Up to 32 stations a wavefront takes SEVERAL rows per trip: lane = (row of the trip, station) with BP = 8 / 16 / 32 / 64
    // lanes per row (the next power of two >= B), SUB = 64 / BP rows per trip (10 stations: four rows at once instead of 10 busy lanes of 64);
    // per-station partial sums are then added across the SUB lane groups (ds_bpermute butterflies), always in the same order.
    const int LBP = B > 32 ? 6 : B > 16 ? 5 : B > 8 ? 4 : 3, BP = 1 << LBP, SUB = 64 >> LBP;
    const int sb = lane & (BP - 1), sub = lane >> LBP;
    const bool st_ok = sb < B;
    const double2 mybs = st_ok ? x.bs[sb] : make_double2(0.0, 0.0);
    const bool hi_lane = sb >= 32;
    const uint32_t sh = (uint32_t)sb & 31u;
    auto sub_sum_i = [&](int v) { for (int off = BP; off < 64; off <<= 1) v += __shfl_xor(v, off, 64); return v; };
    auto sub_sum_f = [&](float v) { for (int off = BP; off < 64; off <<= 1) v += __shfl_xor(v, off, 64); return v; };
    auto sub_min_f = [&](float v) { for (int off = BP; off < 64; off <<= 1) v = fminf(v, __shfl_xor(v, off, 64)); return v; };
    const bool st_writer = st_ok && sub == 0;                      // the lane that publishes station sb's aggregate

    const size_t obs_stride = (size_t)p.E * (COMPACT ? (size_t)(U * (B + 1 + (B > 32 ? 2 : 1)) + 2 * B) : (size_t)U * (p.kind == DCOMP_MULTI ? 4 * B + 1 : 2 * B + 1));
    const bool emit = !ROLL || p.out_every_step || t == nsteps - 1;        // (uniform) whether this step's outputs are written
    const size_t ts = (ROLL && p.out_every_step) ? (size_t)t : 0;
    const size_t EUs = (size_t)p.E * U;
    float *const o_obs = (emit && p.obs) ? p.obs + ts * obs_stride : nullptr;
    float *const o_reward = (emit && p.reward) ? p.reward + ts * (p.kind == DCOMP_MULTI ? EUs : (size_t)p.E) : nullptr;
    float *const o_sum_util = (emit && p.sum_util) ? p.sum_util + ts * p.E : nullptr;
    float *const o_ue_dr = (emit && p.ue_dr) ? p.ue_dr + ts * EUs : nullptr;
    float *const o_ue_util = (emit && p.ue_util) ? p.ue_util + ts * EUs : nullptr;
    float *const o_rb = (emit && p.rb_out) ? p.rb_out + ts * EUs : nullptr;
    const uint8_t *const action = !ROLL ? p.action : ploop ? (t ? p.next_act : p.action) : p.action + (size_t)t * EUs;
    const uint32_t time = p.time + (uint32_t)t;
    uint32_t act = 0, uidw = (uint32_t)u + 1u;
    float dr_req = 1.f;
    bool step_util = false;
    int vrange = MV_CFG_ARRIVED;
    if (RESET) {                                                   // MobileEnv.reset (base.py:169-189): user.py:98-116 + movement.py:110-122
        if (alive) {
            const UeCfg c = p.ue_cfg[u];
            step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req;
            reset_ue(p, env, u, p.episode, px, py, mv);
        }
        if (active) {
            p.pos[idx] = make_double2(px, py);
            p.mv[idx] = mv;
            p.conn[idx] = 0u;
            x.conn_hi[idx] = 0u;
            p.ewma[idx] = 0.f;
            if (p.uid) p.uid[idx] = alive ? (uint16_t)(u + 1) : (uint16_t)0;
            if (p.orig_consumed && alive) p.orig_consumed[(size_t)env * p.U0 + u] = 0xFFFFu;
        }
    } else if (alive) {
        if (!ROLL || t == 0) {
            const double2 q = p.pos[idx];
            px = q.x; py = q.y;
            mv = p.mv[idx];
            conn = (unsigned long long)p.conn[idx] | ((unsigned long long)x.conn_hi[idx] << 32);
            ewma = p.ewma[idx];
        }
        act = action[idx];
        if (DYN) uidw = p.uid[idx];
        if (act > (uint32_t)B) { atomicOr(p.flags, DCOMP_FLAG_BAD_ACTION); act = 0; }
    }
    __syncthreads();                                               // the BS table

    if (!RESET) {
        // 1. toggle (base.py:247-263 -> user.py:190-222): only the station the UE acts on matters at the pre-move position
        if (act > 0u) {
            const unsigned long long bit = 1ull << (act - 1u);
            if (conn & bit) conn &= ~bit;
            else {
                bool ir;
                float l;
                big_pair(px, py, bs_s[act - 1u], p, ir, l);
                if (ir) {
                    conn |= bit;
                    if (x.maxcap_mask & bit) p.conn_since[(size_t)idx * B + (act - 1u)] = (uint16_t)time;
                }
            }
        }
        // 1b. UE departure / arrival (base.py:433-443), after the actions and before the rates: dcomp_dyn.h's event phase with a 64-bit set
        if (DYN && (p.n_remove > 0 || p.n_add > 0)) {
            uint32_t *const xw = reinterpret_cast<uint32_t *>(big_smem + cv.pos);         // exchange words (envs wider than a wavefront): pos | slot | inr areas
            for (int k = 0; k < p.n_remove; k++) {                 // base.py:608-618: pop(idx) + disconnect_from_all
                int r;
                if (p.rng_mode == DCOMP_RNG_TAPE) r = (env < p.E) ? p.ev_remove[(size_t)env * p.n_remove + k] : 0;
                else {
                    uint32_t d[4];
                    philox4x32_10(p.env_base + (uint32_t)env, 0xFFFE0000u + p.ev_rem_base + (uint32_t)k, p.episode, 0u, p.seed_lo, p.seed_hi, d);
                    r = (int)__umulhi(d[0], (uint32_t)cur);
                }
                if (alive && u == r && !(uidw & UID_BORN) && p.orig_consumed)             // host bookkeeping of the initial UEs' streams
                    p.orig_consumed[(size_t)env * p.U0 + ((uidw & 0x7FFFu) - 1u)] = (uint16_t)(mv >> 48);
                const bool take = active && u >= r && u + 1 < cur;                        // slots behind the leaver move up
                if (any_maxcap) {                                  // the step-of-connection rows travel with their UEs (rare: an event step with max-cap stations)
                    for (int b = 0; b < B; b++) {
                        uint16_t t = 0;
                        if (take) t = p.conn_since[(size_t)(idx + 1) * B + b];
                        __syncthreads();
                        if (take) p.conn_since[(size_t)idx * B + b] = t;
                        __syncthreads();
                    }
                }
                if constexpr (UPAD <= 64) {
                    const double nx = __shfl_down(px, 1, UPAD), ny = __shfl_down(py, 1, UPAD);
                    const unsigned long long nmv = __shfl_down(mv, 1, UPAD), nconn = __shfl_down(conn, 1, UPAD);
                    const uint32_t nuid = __shfl_down(uidw, 1, UPAD);
                    const float newma = __shfl_down(ewma, 1, UPAD);
                    if (take) { px = nx; py = ny; mv = nmv; conn = nconn; uidw = nuid; ewma = newma; }
                } else {
                    const unsigned long long bx = (unsigned long long)__double_as_longlong(px), by = (unsigned long long)__double_as_longlong(py);
                    uint32_t wd[10] = {(uint32_t)bx, (uint32_t)(bx >> 32), (uint32_t)by, (uint32_t)(by >> 32), (uint32_t)mv, (uint32_t)(mv >> 32),
                                       (uint32_t)conn, (uint32_t)(conn >> 32), uidw, __float_as_uint(ewma)};
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < 10; w++) xw[w * BLK + tid] = wd[w];
                    __syncthreads();
                    if (take) {
#pragma unroll
                        for (int w = 0; w < 10; w++) wd[w] = xw[w * BLK + tid + 1];     // (take: the next slot belongs to the same env)
                    }
                    __syncthreads();
                    px = __longlong_as_double((long long)(((unsigned long long)wd[1] << 32) | wd[0]));
                    py = __longlong_as_double((long long)(((unsigned long long)wd[3] << 32) | wd[2]));
                    mv = ((unsigned long long)wd[5] << 32) | wd[4];
                    conn = ((unsigned long long)wd[7] << 32) | wd[6];
                    uidw = wd[8]; ewma = __uint_as_float(wd[9]);
                }
                cur -= 1;
                if (u == cur) {
                    conn = 0; uidw = 0; ewma = 0.f; mv = 0; px = 0.0; py = 0.0;
                    if (any_maxcap && active) for (int b = 0; b < B; b++) p.conn_since[(size_t)idx * B + b] = 0;
                }
                alive = active && u < cur;
            }
            for (int k = 0; k < p.n_add; k++) {                    // base.py:592-606
                uint32_t last;                                     // id of ue_list[-1]
                if constexpr (UPAD <= 64) last = __shfl(uidw, (lane & ~(UPAD - 1)) + (cur - 1), 64);
                else {
                    __syncthreads();
                    xw[tid] = uidw;
                    __syncthreads();
                    last = xw[cur - 1];
                    __syncthreads();
                }
                if (active && u == cur) {
                    int ax, ay;
                    if (p.rng_mode == DCOMP_RNG_TAPE) { const size_t t = ((size_t)env * p.n_add + k) * 2; ax = p.ev_add_xy[t]; ay = p.ev_add_xy[t + 1]; }
                    else {                                         // map.rand_border_point (map.py:52-65)
                        uint32_t d[4];
                        philox4x32_10(p.env_base + (uint32_t)env, 0xFFFF0000u + p.ev_add_base + (uint32_t)k, p.episode, 0u, p.seed_lo, p.seed_hi, d);
                        const int rx = (int)__umulhi(d[0], (uint32_t)p.map_w + 1u), ry = (int)__umulhi(d[1], (uint32_t)p.map_h + 1u);
                        const int border = (int)__umulhi(d[2], 4u);                      // left, right, top, bottom
                        ax = border == 0 ? 0 : border == 1 ? p.map_w - 1 : rx;
                        ay = border == 2 ? p.map_h - 1 : border == 3 ? 0 : ry;
                    }
                    uidw = ((last & 0x7FFFu) + 1u) | UID_BORN;
                    px = (double)ax; py = (double)ay;
                    uint32_t vel, wx, wy;
                    draw_triple(p, env, uidw, 0u, p.episode, vel, wx, wy, load_mv_cfg(p, uidw));
                    mv = mv_pack(wx, wy, vel, 0u, 0u, 1u);
                    conn = 0; ewma = 0.f;
                    if (any_maxcap) for (int b = 0; b < B; b++) p.conn_since[(size_t)idx * B + b] = 0;
                }
                cur += 1;
            }
            alive = active && u < cur;
            __syncthreads();                                       // the exchange words are free again
        }
        if (alive) {                                               // per-UE configuration (by id where the list changes)
            if (DYN) {
                vrange = load_mv_cfg(p, uidw);
                if (!(uidw & UID_BORN) && !p.all_log_util) { const UeCfg c = p.ue_cfg[(uidw & 0x7FFFu) - 1u]; step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req; }
            } else {
                const UeCfg c = p.ue_cfg[u];
                step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req;
                vrange = mv_cfg_pack(c.vel_lo, c.vel_hi, c.pause, c.border);
            }
        }
    }

    // ---- per-(env, station) aggregates of the rows this workgroup holds: connected count n, sum of the sharing terms s, max-cap winner.
    // qx / qy: the position the rates are taken at.  Slots and positions must be published (and a barrier passed) before the call.
    auto piece_rows = [&](int el, int &r0, int &envl, int &uu0) -> int {              // rows (lane slots) of env piece el of this wave
        r0 = wave * 64 + el * PL;
        envl = r0 / UPAD;
        uu0 = UPAD > 64 ? wave * 64 : 0;
        if (env0 + envl >= p.E) return 0;
        const int n = U - uu0;
        return n < 0 ? 0 : (n > PL ? PL : n);
    };
    auto sharing_aggregates = [&](double qx, double qy, float ewma_v) {
        if (DCOMP_BIG_ABL & 16) return;
        // Both aggregates are built by the UE lanes themselves, each walking the few bits of its OWN connection set (round 6, second pass: the
        // station lanes used to scan every row -- 40 rows x 13 instructions per pass at 10 x 40 central, 47 of its 130 us):
        //  * the COUNT n_b: `ds_add_f32 1.0` per connected station -- sums of ones below 2^24 are exact, so the order is irrelevant;
        //  * the SUM of the rate-fair / proportional-fair terms (station.py:177-180 | 150, 192-195): the order matters in the last bit, so the
        //    adds are issued UE by UE -- trip uu of a uniform loop enables the lanes whose UE index is uu (one per env of the wavefront: different
        //    envs, different addresses), four terms per UE and round; LDS operations of a wavefront complete in program order.  Envs wider than
        //    a wavefront add into per-wave partial sums first and combine them in wave order.  Deterministic, no compare-and-swap loops.
        for (int i = tid; i < 2 * GPB * B; i += BLK) agg_n[i] = 0.f;              // (agg_s follows agg_n)
        if (NWAVE > 1 && lane < B) part_s[(NWAVE + wave) * B + lane] = 0.f;
        __syncthreads();
        float *const cnt_row = agg_n + env_local * B;
        for (unsigned long long m = alive ? conn : 0ull; m; m &= m - 1ull) __hip_atomic_fetch_add(cnt_row + (__ffsll((long long)m) - 1), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (any_sum) {
            float *const sum_row = NWAVE > 1 ? part_s + (NWAVE + wave) * B : agg_s + env_local * B;
            const int my_uu = NWAVE > 1 ? u - wave * 64 : u;            // this lane's row inside its wavefront's piece
            const int piece_len = NWAVE > 1 ? (U - wave * 64 < 64 ? U - wave * 64 : 64) : U;
            unsigned long long todo = alive ? (conn & x.summode_mask) : 0ull;
            const float inv_e = fast_rcp(ewma_v + EPS);
            while (true) {
                uint32_t tb[4];
                float tv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    tb[k] = 0xFFu; tv[k] = 0.f;
                    if (todo) {
                        const int b = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        bool ir;
                        float l;
                        big_pair(qx, qy, bs_s[b], p, ir, l);
                        const float r = big_rate(l);
                        tb[k] = (uint32_t)b;
                        tv[k] = mode_s[b] == DCOMP_RATE_FAIR ? fast_rcp(r) : r * inv_e;
                    }
                }
                // (empty slots add 0.0 to a scratch word, slots no lane of the wavefront uses are skipped: the loop body is one compare, the
                //  exec switch and one ds_add_f32 per slot in use)
                float *ta[4];
                bool use[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    ta[k] = tb[k] != 0xFFu ? sum_row + tb[k] : agg_n + 3 * GPB * B;
                    use[k] = __builtin_amdgcn_ballot_w64(tb[k] != 0xFFu) != 0ull;
                }
                for (int uu = 0; uu < piece_len; uu++) {                // UE order
                    if (my_uu == uu) {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (use[k]) __hip_atomic_fetch_add(ta[k], tv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (!__syncthreads_or(todo != 0ull)) break;             // further rounds: UEs with more than four such connections
            }
        }
        __syncthreads();
        if (NWAVE > 1 && any_sum) {                                // one env over several waves: the partial sums, combined in wave order
            if (wave == 0 && lane < B) {
                float sacc = 0.f;
                for (int w = 0; w < NWAVE; w++) sacc += part_s[(NWAVE + w) * B + lane];
                agg_s[lane] = sacc;
            }
            __syncthreads();
        }
        if (any_maxcap && !(DCOMP_BIG_ABL & 1)) {
            // station.py:183-187: the UE with the highest FP64 rate is served; equal rates -> the oldest connection, then the lowest UE index
            // (dcomp_device.h shared_rates has the derivation: nearest UE, contenders within 1e-7, the collapsing FP64 key).  Owner threads.
            __syncthreads();
            for (int q = tid; q < GPB * B; q += BLK) {
                const int el = q / B, b = q - el * B, base = el * UPAD;
                uint32_t win = 0xFFFFFFFFu;
                if (env0 + el < p.E && mode_s[b] == DCOMP_MAX_CAP && agg_n[q] > 0.f) {
                    const double2 bp = bs_s[b];
                    auto has = [&](int v) { const uint4 sl = slot_s[base + v]; return (((b < 32 ? sl.x : sl.y) >> (b & 31)) & 1u) != 0u; };
                    double dmin = 1e300;
                    for (int v = 0; v < U; v++) if (has(v)) {
                        const double dx = bp.x - pos_s[base + v].x, dy = bp.y - pos_s[base + v].y;
                        dmin = fmin(dmin, __builtin_fma(dy, dy, dx * dx));
                    }
                    int ncand = 0, only = 0;
                    for (int v = 0; v < U; v++) if (has(v)) {
                        const double dx = bp.x - pos_s[base + v].x, dy = bp.y - pos_s[base + v].y;
                        if (__builtin_fma(dy, dy, dx * dx) <= dmin * (1.0 + 1e-7)) { ncand++; only = v; }
                    }
                    if (ncand == 1) win = (uint32_t)only;
                    else {
                        unsigned long long best = 0ull;
                        uint32_t bestw = 0xFFFFFFFFu;
                        for (int v = 0; v < U; v++) if (has(v)) {
                            const double dx = bp.x - pos_s[base + v].x, dy = bp.y - pos_s[base + v].y;
                            if (__builtin_fma(dy, dy, dx * dx) > dmin * (1.0 + 1e-7)) continue;
                            const unsigned long long key = maxcap_rate_key(p.pl_c1, p.pl_c2, pos_s[base + v].x, pos_s[base + v].y, bp.x, bp.y);
                            const uint32_t w = ((uint32_t)p.conn_since[((size_t)(env0 + el) * U + v) * B + b] << 10) | (uint32_t)v;     // (16-bit step, UE index < 1 024)
                            if (key > best || (key == best && w < bestw)) { best = key; bestw = w; }
                        }
                        win = bestw & 0x3FFu;
                    }
                }
                mc_win[q] = win;
            }
        }
        __syncthreads();
    };
    auto publish = [&](float third) {                              // this lane's row for the station-role loops
        pos_s[tid] = make_double2(px, py);
        slot_s[tid] = make_uint4(alive ? (uint32_t)conn : 0u, alive ? (uint32_t)(conn >> 32) : 0u, __float_as_uint(third), 0u);
    };

    float reward_before = 0.f, curr = 0.f;
    if (!RESET) {
        // 2. rates before the move (base.py:446) -> reward_before (base.py:158-167)
        publish(ewma);
        __syncthreads();
        sharing_aggregates(px, py, ewma);
        // 3. move (base.py:447 -> user.py:159-173); the old position stays for the pre-move rates below
        const double ox = px, oy = py;
        if (alive && !(DCOMP_BIG_ABL & 32)) {
            move_ue<false>(p, env, uidw, p.episode, px, py, mv, vrange);
            if (px < 0.0 || py < 0.0 || px > (double)p.map_w || py > (double)p.map_h) atomicOr(p.flags, DCOMP_FLAG_OUTSIDE_MAP);
        }
        // 4. this UE's pre-move shares (station.py:152-202), and for each: does the connection survive at the new position (user.py:175-188)?
        //    EWMA from the STALE rates of what stays (user.py:148-157).
        float curr_pre = 0.f, stale = 0.f;
        if (!(DCOMP_BIG_ABL & 64)) {
            const float inv_ewma = fast_rcp(ewma + EPS);
            for (unsigned long long m = conn; m; m &= m - 1ull) {
                const int b = __ffsll((long long)m) - 1, q = env_local * B + b;
                const double2 bp = bs_s[b];
                bool ir;
                float l;
                big_pair(ox, oy, bp, p, ir, l);
                const float out = big_share(mode_s[b], big_rate(l), agg_n[q], agg_s[q], inv_ewma, mc_win[q] == (uint32_t)u);
                curr_pre += out;
                const double dx = bp.x - px, dy = bp.y - py;
                const double dsq = __builtin_fma(dy, dy, dx * dx);
                bool stays = dsq < p.dt2;
                if ((float)dsq == p.dt2f || p.dsq_exact) stays = in_range_exact(px, py, bp.x, bp.y, p.dt2);
                if (stays) stale += out;
                else conn &= ~(1ull << b);
            }
        }
        reward_before = clamp_med3(ue_utility(curr_pre, step_util, dr_req), MIN_UTIL, MAX_UTIL) * (1.0f / MAX_UTIL);
        ewma = __builtin_fmaf(0.9f, stale, 0.1f * ewma);           // one explicit contraction: every kernel variant rounds alike
        __syncthreads();                                           // every reader of the pre-move slots / aggregates is done
        // 5. rates after the move (base.py:451)
        publish(ewma);
        __syncthreads();
        sharing_aggregates(px, py, ewma);
        if (!(DCOMP_BIG_ABL & 64)) {
            const float inv_ewma = fast_rcp(ewma + EPS);
            for (unsigned long long m = conn; m; m &= m - 1ull) {
                const int b = __ffsll((long long)m) - 1, q = env_local * B + b;
                bool ir;
                float l;
                big_pair(px, py, bs_s[b], p, ir, l);
                curr += big_share(mode_s[b], big_rate(l), agg_n[q], agg_s[q], inv_ewma, mc_win[q] == (uint32_t)u);
            }
        }
        if (active) {                                              // state write-back: every slot (unlisted slots are cleared)
            p.pos[idx] = alive ? make_double2(px, py) : make_double2(0.0, 0.0);
            p.mv[idx] = alive ? mv : 0ull;
            p.conn[idx] = alive ? (uint32_t)conn : 0u;
            x.conn_hi[idx] = alive ? (uint32_t)(conn >> 32) : 0u;
            p.ewma[idx] = alive ? ewma : 0.f;
            if (DYN) p.uid[idx] = alive ? (uint16_t)uidw : (uint16_t)0;
        }
        __syncthreads();                                           // the sparse loops are done with the aggregates / slots
    }
    const float util = ue_utility(curr, step_util, dr_req);
    // 6. publish the row as the outputs need it: the new position, the post-drop connection set, utility, reward_before
    pos_s[tid] = make_double2(px, py);
    slot_s[tid] = make_uint4(alive ? (uint32_t)conn : 0u, alive ? (uint32_t)(conn >> 32) : 0u, __float_as_uint(alive ? util : 0.f),
                             __float_as_uint(alive ? reward_before : 0.f));
    __syncthreads();
    const int kind = p.kind, n_eff = cur;
    const bool want_min = p.reward_agg == DCOMP_REWARD_MIN;
    // 7. per-station utility aggregates (station.py:63-83) -- multi-agent envs only: central observations and rewards carry none
    if (kind == DCOMP_MULTI && !(DCOMP_BIG_ABL & 2)) {
        float nn[EPW], su[EPW], mn[EPW];
#pragma unroll
        for (int el = 0; el < EPW; el++) {
            int r0, envl, uu0;
            const int nrows = piece_rows(el, r0, envl, uu0);
            int n = 0;
            float s = 0.f, m_ = MAX_UTIL;
            for (int uu = sub; uu < nrows; uu += SUB) {
                const uint4 sl = slot_s[r0 + uu];
                const int c = __builtin_amdgcn_sbfe((int)(hi_lane ? sl.y : sl.x), sh, 1u);         // 0 / -1
                n -= c;
                s += __uint_as_float(sl.z & (uint32_t)c);
                if (want_min) m_ = c ? fminf(m_, __uint_as_float(sl.z)) : m_;
            }
            nn[el] = (float)sub_sum_i(n); su[el] = sub_sum_f(s); mn[el] = want_min ? sub_min_f(m_) : m_;
        }
        if (NWAVE > 1) {
            if (st_writer) { part_s[wave * B + sb] = nn[0]; part_s[(NWAVE + wave) * B + sb] = su[0]; part_s[(2 * NWAVE + wave) * B + sb] = mn[0]; }
            __syncthreads();
            if (wave == 0 && st_writer) {
                float n = 0.f, s = 0.f, m_ = MAX_UTIL;
                for (int w = 0; w < NWAVE; w++) { n += part_s[w * B + sb]; s += part_s[(NWAVE + w) * B + sb]; m_ = fminf(m_, part_s[(2 * NWAVE + w) * B + sb]); }
                agg_n[sb] = n; agg_u[sb] = s; agg_m[sb] = m_;
            }
        } else if (st_writer) {
#pragma unroll
            for (int el = 0; el < EPW; el++) {
                const int envl = (wave * 64 + el * PL) / UPAD;
                agg_n[envl * B + sb] = nn[el]; agg_u[envl * B + sb] = su[el]; agg_m[envl * B + sb] = mn[el];
            }
        }
        __syncthreads();
    }
    // 8. observation rows (variants.py:271-305 / central.py:36-55), lane = station: the pair (row, station), the row's in-range set (the
    //    compare's lane mask), the relative snr against the row maximum, `connected` from the row's set, the per-env columns -- four
    //    coalesced stores per row, straight from registers.  Unlisted slots write zero rows (central.py:46-55).
    const int ROW = 4 * B + 1, UB = U * B;
    constexpr bool compact = COMPACT;                              // (its own instantiation: the branch in the row loop cost the row format 3-8 %; multi-agent envs only: the host checks)
    const int CWC = B + 1 + (B > 32 ? 2 : 1), REC = U * CWC + 2 * B;     // words per UE / per env-step of the compact record (dcomp_frag::ue_words / env_words)
    const float inv_u = 1.0f / (float)n_eff;
    if (!(DCOMP_BIG_ABL & 4)) {
#pragma unroll
        for (int el = 0; el < EPW; el++) {
            int r0, envl, uu0;
            const int nrows = piece_rows(el, r0, envl, uu0);
            float n_col = 0.f, u_col = 0.f;
            if (kind == DCOMP_MULTI && st_ok && nrows > 0) {
                const float n = agg_n[envl * B + sb];
                n_col = n * inv_u;
                u_col = agg_u[envl * B + sb] * fast_rcp(fmaxf(n, 1.f)) * (1.0f / MAX_UTIL);
            }
            // where this env's rows go: the row format, or (multi-agent envs, dcomp_out.obs_compact) the compact record of dcomp_fragment.h --
            // U x {dr[B], utility, connection word(s)} + ues_at_bs[B] | util_at_bs[B]: the station lanes store the dr blocks and, once per env,
            // the two per-env columns; utility and the set words are the UE lanes' (below)
            float *const dst_env = !o_obs ? nullptr : compact ? o_obs + (size_t)(env0 + envl) * REC : o_obs + (size_t)(env0 + envl) * U * (kind == DCOMP_MULTI ? ROW : 2 * B + 1);
            if (compact && dst_env && st_writer && nrows > 0 && uu0 == 0) {
                big_store(dst_env + U * CWC + sb, n_col);
                big_store(dst_env + U * CWC + B + sb, u_col);
            }
            const unsigned long long ok_mask = __builtin_amdgcn_ballot_w64(st_ok);
            const BigRsrc rs = big_rsrc(dst_env);
            // Rows as 16-byte stores (BigParams::row_x4; multi-agent rows, more than 32 stations): the row's 4B words as they lie in memory, four per
            // lane.  The lane-per-station values are transposed through ONE row of LDS per wavefront (two ds_write_b32 + one ds_read_b128 per lane and
            // row; the two per-env blocks are written once per env; LDS operations of a wavefront complete in order: no barrier).  One store
            // instruction then writes 1 KiB of consecutive bytes instead of four writing 256 B each at a stride of B floats: beyond the Infinity
            // Cache the four-block form reached 3.1 TB/s of writes (65 536 x 32 x 64: rows alone 566 of 692 us).
            const bool x4 = !COMPACT && LBP == 6 && x.row_x4 != 0 && kind == DCOMP_MULTI && dst_env != nullptr;
            float *const stage = reinterpret_cast<float *>(big_smem + cv.stage) + wave * 4 * B;
            if (x4 && st_ok) { stage[2 * B + lane] = n_col; stage[3 * B + lane] = u_col; }      // the two per-env blocks: once per env
            if (LBP == 6) {
                // more than 32 stations: ONE row per trip, everything about the row uniform (position / set: broadcast reads; destination: scalar).
                // The loop exists once per output form (OUT: 0 nothing, 1 compact record, 2 rows as 16-byte stores, 3 rows in four blocks, 4 central):
                // the form is picked per env piece, not per row (the chain of uniform branches in front of the central stores cost 65 536 x 10 x 40
                // central 9 % when the 16-byte form was added to it).
                auto rows6 = [&](auto out_tag) __attribute__((always_inline)) {
                constexpr int OUT = decltype(out_tag)::value;
                for (int uu = 0; uu < nrows; uu++) {
                    const int r = r0 + uu, ue = uu0 + uu;
                    const bool live = ue < cur;
                    const double2 q = pos_s[r];
                    const uint4 sl = slot_s[r];
                    float l = -3.0e38f;
                    unsigned long long bal = 0ull;
                    if (!(DCOMP_BIG_ABL & 8)) {
                        // the lane masks straight from the compares (a ballot of a derived bool costs a v_cndmask + v_cmp pair each)
                        bool ir;
                        float qf;
                        pair_eval_q(q.x, q.y, mybs.x, mybs.y, p, ir, l, qf);
                        const double dx = mybs.x - q.x, dy = mybs.y - q.y;
                        const double dsq = __builtin_fma(dy, dy, dx * dx);             // (the same expression as inside pair_eval_q: one evaluation)
                        bal = __builtin_amdgcn_ballot_w64(dsq < p.dt2) & ok_mask;
                        const unsigned long long rare = (__builtin_amdgcn_ballot_w64(qf == p.dt2f) | __builtin_amdgcn_ballot_w64(qf < NEAR_D2)) & ok_mask;
                        if (rare != 0ull || p.dsq_exact) {                               // wave-uniform: the fused d^2 cannot decide / a UE within 1.26 m of a station
                            bool fix = st_ok && ((bal >> lane) & 1ull);
                            if (st_ok && (qf == p.dt2f || p.dsq_exact)) fix = in_range_exact(q.x, q.y, mybs.x, mybs.y, p.dt2);
                            if (st_ok && qf < NEAR_D2 && (float)dsq < 1e-20f) l = pair_eval_tiny(q.x, q.y, mybs.x, mybs.y, p);
                            bal = __builtin_amdgcn_ballot_w64(fix);
                        }
                        l = st_ok ? l : -3.0e38f;
                    }
                    if (kind == DCOMP_MULTI) { if (lane == 0) inr_s[r] = bal; }      // (only the multi-agent rewards read the in-range sets)
                    float lmax_p = 0.f;
                    if (POL) {                                                       // dcomp_set_policy: the rules on the row's dr entries (the instantiation of its own)
                        lmax_p = wave_max_f32(l);
                        if (p.next_act) {
                            const int a = big_policy_wave(p, (unsigned long long)sl.x | ((unsigned long long)sl.y << 32), live ? fast_exp2(l - lmax_p) : 0.f, st_ok, ok_mask, lane, B);
                            if (lane == 0) p.next_act[(size_t)(env0 + envl) * U + ue] = (uint8_t)(live ? a : 0);
                        }
                    }
                    if (OUT != 0) {
                        // (buffer stores: the row's offset is a scalar, the lane's a loop-invariant register -- no address arithmetic per store;
                        //  an unlisted slot -- `live` is uniform -- takes the zero-row branch instead of a select per value)
                        const uint32_t lo = (uint32_t)lane * 4u, B4 = (uint32_t)B * 4u;
                        if (OUT == 1) {
                            const float lmax = POL ? lmax_p : wave_max_f32(l);
                            if (st_ok) big_bstore(rs, lo, (uint32_t)(ue * CWC) * 4u, live ? fast_exp2(l - lmax) : 0.f);
                        } else if (OUT == 2) {
                            const uint32_t ro = (uint32_t)(ue * ROW) * 4u;
                            float cf = 0.f, dr = 0.f, ut = 0.f;
                            if (live) {
                                const float lmax = POL ? lmax_p : wave_max_f32(l);
                                dr = fast_exp2(l - lmax);                                    // variants.py:276-284
                                cf = (float)__builtin_amdgcn_ubfe(hi_lane ? sl.y : sl.x, sh, 1u);
                                ut = __uint_as_float(sl.z) * (1.0f / MAX_UTIL);
                            }
                            float v[4] = {0.f, 0.f, 0.f, 0.f};                                // (an unlisted slot: a zero row, the per-env blocks too)
                            if (live) {
                                if (st_ok) { stage[lane] = cf; stage[B + lane] = dr; }
                                if (lane < B) {                                              // (4B words = B lanes of four; B <= 64)
                                    const float4 q4 = *reinterpret_cast<const float4 *>(stage + 4 * lane);
                                    v[0] = q4.x; v[1] = q4.y; v[2] = q4.z; v[3] = q4.w;
                                }
                            }
                            if (lane < B) big_bstore4(rs, lo * 4u, ro, v);
                            if (lane == 0) big_bstore(rs, 0u, ro + 4u * B4, ut);              // the row's own utility entry, right behind it
                        } else if (OUT == 3) {
                            const uint32_t ro = (uint32_t)(ue * ROW) * 4u;
                            if (live) {
                                const float lmax = POL ? lmax_p : wave_max_f32(l);
                                const float dr = fast_exp2(l - lmax);                        // variants.py:276-284
                                const float cf = (float)__builtin_amdgcn_ubfe(hi_lane ? sl.y : sl.x, sh, 1u);
                                if (st_ok) {
                                    big_bstore(rs, lo, ro, cf);
                                    big_bstore(rs, lo, ro + B4, dr);
                                    big_bstore(rs, lo, ro + 2u * B4, n_col);
                                    big_bstore(rs, lo, ro + 3u * B4, u_col);
                                }
                            } else if (st_ok) {
                                big_bstore(rs, lo, ro, 0.f); big_bstore(rs, lo, ro + B4, 0.f);
                                big_bstore(rs, lo, ro + 2u * B4, 0.f); big_bstore(rs, lo, ro + 3u * B4, 0.f);
                            }
                        } else {
                            const uint32_t ro = (uint32_t)(ue * B) * 4u, UB4 = (uint32_t)UB * 4u;
                            if (live) {
                                const float lmax = POL ? lmax_p : wave_max_f32(l);
                                const float dr = fast_exp2(l - lmax);
                                const float cf = (float)__builtin_amdgcn_ubfe(hi_lane ? sl.y : sl.x, sh, 1u);
                                if (st_ok) { big_bstore(rs, lo, ro, cf); big_bstore(rs, lo, ro + UB4, dr); }
                            } else if (st_ok) { big_bstore(rs, lo, ro, 0.f); big_bstore(rs, lo, ro + UB4, 0.f); }
                        }
                    }
                }
                };
                using std::integral_constant;
                if (!dst_env) rows6(integral_constant<int, 0>{});
                else if (compact) rows6(integral_constant<int, 1>{});
                else if (x4) rows6(integral_constant<int, 2>{});
                else if (kind == DCOMP_MULTI) rows6(integral_constant<int, 3>{});
                else rows6(integral_constant<int, 4>{});
            } else {
                // up to 32 stations: SUB rows per trip, lane = (row sub, station sb)
                const int trips = (nrows + SUB - 1) >> (6 - LBP);
                for (int it = 0; it < trips; it++) {
                    const int uu = it * SUB + sub;
                    const bool rv = uu < nrows;
                    const int r = r0 + (rv ? uu : 0), ue = uu0 + uu;
                    const bool live = ue < cur;
                    const double2 q = pos_s[r];
                    const uint4 sl = slot_s[r];
                    float l = -3.0e38f;
                    bool ir = false;
                    if (!(DCOMP_BIG_ABL & 8)) big_row_pair(q.x, q.y, mybs, p, st_ok && rv, ir, l);
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(ir);
                    if (sb == 0 && rv && kind == DCOMP_MULTI) inr_s[r] = (bal >> (sub * BP)) & ((1ull << BP) - 1ull);
                    if (dst_env || (POL && p.next_act)) {
                        const float lmax = BP == 32 ? group_reduce<32, OpMax>(l) : BP == 16 ? group_reduce<16, OpMax>(l) : group_reduce<8, OpMax>(l);
                        const float dr = live ? fast_exp2(l - lmax) : 0.f;                // variants.py:276-284
                        if (POL && p.next_act) {                                          // dcomp_set_policy: the rules on the rows of this trip
                            const bool ok = st_ok && rv;
                            const int a = BP == 32 ? big_policy_group<32>(p, sl.x, dr, ok, sub, sb, B) : BP == 16 ? big_policy_group<16>(p, sl.x, dr, ok, sub, sb, B)
                                                                                                              : big_policy_group<8>(p, sl.x, dr, ok, sub, sb, B);
                            if (sb == 0 && rv) p.next_act[(size_t)(env0 + envl) * U + ue] = (uint8_t)(live ? a : 0);
                        }
                        if (!dst_env) continue;
                        const float cf = (float)__builtin_amdgcn_ubfe(sl.x, sh, 1u);
                        const uint32_t B4 = (uint32_t)B * 4u;                              // (buffer stores: one 32-bit offset per lane, the blocks of the row as scalar offsets)
                        if (compact) {
                            if (st_ok && rv) big_bstore(rs, (uint32_t)(ue * CWC + sb) * 4u, 0u, dr);
                        } else if (kind == DCOMP_MULTI) {
                            const uint32_t off = (uint32_t)(ue * ROW + sb) * 4u;
                            if (st_ok && rv) {
                                big_bstore(rs, off, 0u, cf);
                                big_bstore(rs, off, B4, dr);
                                big_bstore(rs, off, 2u * B4, live ? n_col : 0.f);
                                big_bstore(rs, off, 3u * B4, live ? u_col : 0.f);
                            }
                        } else {
                            const uint32_t off = (uint32_t)(ue * B + sb) * 4u;
                            if (st_ok && rv) {
                                big_bstore(rs, off, 0u, cf);
                                big_bstore(rs, off, (uint32_t)UB * 4u, dr);
                            }
                        }
                    }
                }
            }
        }
    }
    // the rows' own utility entry: by the UE lanes (one store instruction per wavefront instead of a lane-0 store in every trip of the row loop)
    if (active && o_obs && !(DCOMP_BIG_ABL & 4)) {
        const float ut = alive ? util * (1.0f / MAX_UTIL) : 0.f;
        if (compact) {
            float *const rec = o_obs + (size_t)env * REC + (size_t)u * CWC;
            big_store(rec + B, ut);
            big_store(rec + B + 1, __uint_as_float(alive ? (uint32_t)conn : 0u));
            if (B > 32) big_store(rec + B + 2, __uint_as_float(alive ? (uint32_t)(conn >> 32) : 0u));
        } else if (kind == DCOMP_MULTI) { if (!(B > 32 && x.row_x4)) big_store(o_obs + (size_t)idx * ROW + 4 * B, ut); }      // (row_x4: the row loop has written it)
        else big_store(o_obs + (size_t)env * U * (2 * B + 1) + 2 * UB + u, ut);
    }
    __syncthreads();
    // 9. reward, info
    const unsigned long long in_range = kind == DCOMP_MULTI ? inr_s[tid] : 0ull;
    if (kind == DCOMP_CENTRAL) {                                   // central.py:65-73: over the UEs' rewards_before
        if (active && u == 0) {
            const int base = env_local * UPAD;
            float r = want_min ? 1.f : 0.f, su = 0.f;
            for (int v = 0; v < n_eff; v++) {
                const uint4 sl = slot_s[base + v];
                r = want_min ? fminf(r, __uint_as_float(sl.w)) : r + __uint_as_float(sl.w);
                su += __uint_as_float(sl.z);
            }
            if (p.reward_agg == DCOMP_REWARD_AVG) r = r * fast_rcp((float)n_eff);         // (one reciprocal: the form of write_outputs)
            if (o_reward) o_reward[env] = RESET ? 0.f : r;
            if (o_sum_util) o_sum_util[env] = su;
        }
    } else {
        float reward = 0.f;
        if (!RESET) {
            reward = util;                                         // multi_agent.py:52 (own utility, NOT normalised)
            if (p.reward_agg == DCOMP_REWARD_SUM) {                // multi_agent.py:73-79
                if (in_range != 0ull) {
                    const int base = env_local * UPAD;
                    const uint32_t clo = (uint32_t)conn, chi = (uint32_t)(conn >> 32);
                    float s = 0.f;
                    for (int v = 0; v < U; v++) { const uint4 sl = slot_s[base + v]; if ((sl.x & clo) | (sl.y & chi)) s += __uint_as_float(sl.w); }
                    reward = s;
                }
            } else if (p.reward_agg == DCOMP_REWARD_AVG) {         // multi_agent.py:60-71
                float n = 0.f, t = 0.f;
                for (unsigned long long m = in_range; m; m &= m - 1ull) { const int q = env_local * B + __ffsll((long long)m) - 1; n += agg_n[q]; t += agg_u[q]; }
                if (n > 0.f) reward = conn == 0ull ? (t + util) * fast_rcp(n + 1.f) : t * fast_rcp(n);
            } else {                                               // multi_agent.py:81-85, station.py:78-83
                float m_ = util;
                for (unsigned long long m = in_range; m; m &= m - 1ull) { const int q = env_local * B + __ffsll((long long)m) - 1; m_ = fminf(m_, agg_n[q] > 0.f ? agg_m[q] : MAX_UTIL); }
                reward = in_range != 0ull ? m_ : util;
            }
        }
        if (active && o_reward) big_store(&o_reward[idx], alive ? reward : 0.f);
        if (active && u == 0 && o_sum_util) {
            const int base = env_local * UPAD;
            float su = 0.f;
            for (int v = 0; v < n_eff; v++) su += __uint_as_float(slot_s[base + v].z);
            o_sum_util[env] = su;
        }
    }
    if (active) {                                                  // base.py:383-411
        if (o_ue_dr) big_store(&o_ue_dr[idx], alive ? curr : 0.f);
        if (o_ue_util) big_store(&o_ue_util[idx], alive ? util : 0.f);
        if (o_rb) big_store(&o_rb[idx], alive ? reward_before : 0.f);
    }
    };
    if constexpr (ROLL) {
        for (int t = 0; t < nsteps; t++) {
            one_step(t);
            __syncthreads();                                       // the next step republishes the slots / aggregates this one's rewards have just read
        }
    } else one_step(0);
}

using BigKernelFn = void (*)(const KParams, const BigParams);
// fn[pol][compact][which]: which = 0 step, 1 reset, 2 step with UE arrival / departure, 3 fused rollout (fixed UE list); compact = 1: the instantiations that write the compact record;
// pol = 1: the ones that carry the in-step heuristic policy (dcomp_set_policy)
struct BigKernels { BigKernelFn fn[2][2][4]; int gpb, block; };
BigKernels big_kernels_for_upad(int upad);

}  // namespace dcomp
