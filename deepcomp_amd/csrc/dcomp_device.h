// dcomp_device.h -- CDNA4 (gfx950) device code of the DeepCoMP env step.
//
// Mapping: one lane = one (env, UE).  An env occupies UPAD consecutive lanes (UPAD = next power of two
// >= U, <= 256); a 256-lane workgroup holds 256/UPAD envs.  The per-BS reductions over an env's UEs
// (connected-UE count, sum 1/rate, sum priority, sum/min utility) are DPP / ds_swizzle all-reduces inside
// a wavefront and one LDS exchange across wavefronts when U > 64.  The BS table (x, y, sharing mode) is
// wave-uniform: it lives in the kernel-argument segment and is read with scalar loads into SGPRs, the B
// loop is fully unrolled (B is a template parameter).  No [U,B] intermediate ever reaches HBM: one read
// and one write of the UE state, one write of the observation row (staged through LDS so that every store
// instruction covers 1 KiB contiguous, non-temporal).  Variants: dcomp_wide.h (B > 20 with >= 64 lanes per env),
// dcomp_dyn.h (UE arrival / departure).
//
// Numerics: position / movement / connect-drop decisions in FP64 with the reference's operation order
// (bit-exact masks); SNR, rates, utility, observation in FP32 in the log2 domain (no overflow at d -> 0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dcomp_types.h"

// Build switches.  The product build uses the defaults; tools/ablate.py builds timing-only variants.
// DCOMP_ABLATE    bit mask of pipeline stages to leave out (results are wrong by construction; timing only)
// DCOMP_NT_OBS    1: write-once output streams (observation rows, rewards, info) use non-temporal stores -- measured
//                 0.1008 -> 0.0875 ms per step at config 3 (they stop competing with the state lines for L2)
// DCOMP_BS_IN_LDS 1: read the BS table from an LDS copy instead of SGPRs (north_star wording) -- measured 10 % slower
// DCOMP_BLOCK     workgroup size; 128 / 64 measured 3-6 % slower than 256
#ifndef DCOMP_ABLATE
#define DCOMP_ABLATE 0
#endif
#ifndef DCOMP_NT_OBS
#define DCOMP_NT_OBS 1
#endif
#ifndef DCOMP_BS_IN_LDS
#define DCOMP_BS_IN_LDS 0
#endif
#ifndef DCOMP_BLOCK
#define DCOMP_BLOCK 256
#endif
#ifndef DCOMP_WIDE_MIN_B
#define DCOMP_WIDE_MIN_B 20  // envs of >= 64 lanes use the wide kernel (dcomp_wide.h) above this many base stations
                             // (4 096 x 128 UE: B = 16: narrow 0.046 / wide 0.055 ms; 20: 0.062 / 0.064; 24: 0.102 / 0.070)
#endif
#ifndef DCOMP_COMPILER_DIV
#define DCOMP_COMPILER_DIV 0 // 1 = the compiler's generic FP64 sqrt / division in move_ue (A/B of norm_and_unit)
#endif
#ifndef DCOMP_LOG2_MODE
#define DCOMP_LOG2_MODE 2    // log2(d^2): 0 = plain v_log_f32, 1 = frexp range reduction, 2 = 2^-12 prescale (default)
#endif
#ifndef DCOMP_SPARSE_PRE
#define DCOMP_SPARSE_PRE 1      // 1: sparse pre-move pass in step_kernel where the LDS row fits (B <= 11); 0: dense (A/B)
#endif
#ifndef DCOMP_SPARSE_MIN_B
#define DCOMP_SPARSE_MIN_B 7      // the sparse pre-move pass from this many stations up (B = 5: 0.5 % slower than the dense pass)
#endif
#ifndef DCOMP_CENTRAL_STAGED
#define DCOMP_CENTRAL_STAGED 1  // 0: central observation rows are stored straight from registers (round 1; A/B only)
#endif
#ifndef DCOMP_BS_VOLATILE
#define DCOMP_BS_VOLATILE 0    // experiment: re-read the BS table from the kernel-argument segment at every use (no hoisting out of the step loop)
#endif
#ifndef DCOMP_FORCE_KIND
#define DCOMP_FORCE_KIND -1     // experiment: compile write_outputs for one env kind only (register-pressure bisection)
#endif
#ifndef DCOMP_SCALAR_DRAW
#define DCOMP_SCALAR_DRAW 2    // Philox draws of the few redrawing lanes on the scalar unit (draw_triple_wave): 0 never, 1 every kernel, 2 fused rollout only
#endif
#ifndef DCOMP_MOVE_SELECTS
#define DCOMP_MOVE_SELECTS 2   // move_ue as straight-line code with selects instead of branches: 0 never, 1 every kernel, 2 fused rollout only
#endif
#ifndef DCOMP_EXP_NO_RESET
#define DCOMP_EXP_NO_RESET 0
#endif
#ifndef DCOMP_EXP_NO_TAPE
#define DCOMP_EXP_NO_TAPE 0
#endif
#ifndef DCOMP_XCD_REMAP
#define DCOMP_XCD_REMAP 1      // step kernels: every XCD owns a contiguous eighth of the env slots (xcd_contiguous_block)
#endif
#ifndef DCOMP_NT_STATE
#define DCOMP_NT_STATE 0     // experiment: bit 0 non-temporal state loads, bit 1 non-temporal state stores
#endif

// A rarely taken branch that LOADS from global memory (draw tape, velocity table, cluster table) consumes the value inside the
// branch: the compiler then waits for it (s_waitcnt vmcnt) inside the branch too.  Left to itself it puts the wait at the join,
// where every wave executes it -- and vmcnt counts loads AND stores in order, so a persistent kernel (step_kernel_wide) would wait
// there for the previous slot's observation rows to drain, in the middle of its compute phase.
#define VM_ARRIVED1(a) asm volatile("" : "+v"(a))
#define VM_ARRIVED2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define VM_ARRIVED3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))

namespace dcomp {

// ---- channel constants (station.py:26-30, 110-127): snr = K * (d + 1e-16)^(-GAMMA) -------------------------
// K and GAMMA are computed on the host from the reference's formula and passed in KParams.
constexpr float BW = 9e6f;                 // station.py:26
constexpr float LOG2E = 1.4426950408889634f;
constexpr float TEN_LOG10_2 = 3.0102999566398120f;   // 10*log10(x) = TEN_LOG10_2 * log2(x)   (utility.py:54)
constexpr float EPS = 1e-16f;              // constants.py:28
constexpr float MIN_UTIL = -20.f, MAX_UTIL = 20.f;   // constants.py:40-41

struct UeCfg {          // per-UE immutable config (same in every env), 16 B
    int16_t init_x, init_y;   // -1 = random
    uint8_t vel_lo, vel_hi;   // inclusive draw range
    uint8_t util;             // DCOMP_UTIL_*
    uint8_t pause;            // RandomWaypoint.pause_duration (movement.py:87), 0..127
    float dr_req;
    uint8_t border;           // RandomWaypoint.border_buffer, 1..255
    uint8_t pad[3];
};
// Movement parameters of a UE packed into one non-negative int: velocity range lo:8 | hi:8, pause_duration:7, border_buffer:8.
__device__ __forceinline__ int mv_cfg_pack(uint32_t lo, uint32_t hi, uint32_t pause, uint32_t border)
{
    return (int)(lo | (hi << 8) | (pause << 16) | (border << 23));
}
constexpr int MV_CFG_ARRIVED = 1 | (3 << 8) | (2 << 16) | (10 << 23);   // RandomWaypoint(map, velocity='slow') of an arriving UE (base.py:597-599)

struct KParams {
    // state (device)
    double2 *pos;
    unsigned long long *mv;
    uint32_t *conn;
    float *ewma;
    uint32_t *flags;
    uint16_t *conn_since;      // [E*U][B] step of connection (max-cap tie rule), NULL when no BS is max-cap
    uint16_t *uid;             // [E*U] UE id per slot (bit 15: arrived during the episode), NULL unless UEs arrive / depart
    uint16_t *orig_consumed;   // [E*U0] movement triples an initial UE had used when it left (0xFFFF: never left), optional
    // io (device)
    const uint8_t *action;
    float *obs, *reward, *sum_util, *ue_dr, *ue_util, *rb_out;
    int32_t obs_compact;       // 1: `obs` is dcomp_out.obs_compact -- the step writes the compact record (dcomp_fragment.h) instead of the rows
    uint8_t *next_act;         // optional [E][U]: the heuristic policy's action on the observation this launch writes (dcomp_set_policy)
    const uint32_t *policy_cluster;   // DCOMP_POLICY_CLUSTER: [B] cluster masks
    int32_t policy;            // DCOMP_POLICY_*
    int32_t policy_loop;       // fused rollout: steps 1..T-1 act on the policy's decision, only actions[0] is read
    float policy_eps;          // DCOMP_POLICY_DYNAMIC
    // tape (device)
    const int32_t *tape_pos0;
    const ushort4 *tape_triples;
    const UeCfg *ue_cfg;
    const double2 *ue_velq;    // optional [U0]: fixed velocities that are not an integer in 0..255 (movement.py:116-117 takes any number):
                               //   {velocity, largest q with sqrt(q) <= velocity}; x < 0: this UE uses the integer in its movement word
    int32_t tape_depth;
    int32_t tape_ids;          // tapes per env: initial UEs by position, then one per id of an arriving UE
    int32_t E, U;              // U = slots per env (max_ues, base.py:79-84)
    int32_t U0, cur_ue;        // UEs after reset / currently in the list (same in every env: the schedule is config)
    int32_t n_remove, n_add;   // arrival / departure of THIS step (base.py:433-443)
    const int32_t *ev_remove;  // tape mode: [E][n_remove] list positions (global random.randint, base.py:611)
    const int32_t *ev_add_xy;  // tape mode: [E][n_add][2] border points (map.py:52-65)
    uint32_t ev_rem_base, ev_add_base;   // Philox: departures / arrivals so far this episode
    int32_t map_w, map_h;
    int32_t kind, reward_agg, rng_mode;
    uint32_t all_log_util;     // 1: every UE uses the log utility (skip the per-UE config load)
    uint32_t any_maxcap;
    uint32_t maxcap_mask;      // bit b: BS b is max-cap
    double pl_c1, pl_c2;       // Okumura-Hata constants of station.py:110-116 (FP64, for the max-cap rate key)
    uint32_t time;             // env.time before this step (base.py:39)
    // fused rollout (step_kernel only): T steps per launch with the UE state in registers in between
    int32_t num_steps;         // T >= 1; actions[T][E][U]
    int32_t out_every_step;    // 1: the outputs of step t go to out + t * (one step's size), i.e. [T][...] buffers; 0: last step only
    int32_t horizon;           // > 0: reset inside the kernel when env.time reaches it (RLlib's horizon = episode_length,
                               //      env_setup.py:281) -- the same sequence as `if time == L: reset()` before every step
    int32_t tight_g;           // > 0: step_kernel packs envs tightly, G = U lanes each (see Seg); 0: padded power-of-two groups
    int32_t tight_gpw;         //      envs per wavefront = 64 / G
    int32_t tight_magic;       //      lane / G == (lane * magic) >> 16 for lane < 64
    uint32_t episode_inc;      // Philox episode-word increment per in-kernel reset (0: rand_episodes = False, base.py:171-173)
    uint32_t any_sum_mode;     // some BS is rate-fair or proportional-fair (needs a sum over its UEs)
    uint32_t seed_lo, seed_hi, episode;
    uint32_t env_base;         // global id of env 0
    float half_gamma;          // path-loss exponent c2/10, halved (applied to log2 d^2)
    float log2k;               // log2(K) + L2_OFF
    float log2k_s;             // log2(K) + L2_OFF - 12 * half_gamma  (pair_eval scales d^2 by 2^-12)
    double dt2;                // X = smallest double q with sqrt_rn(q) >= d_T: in range <=> dist_sq_ref < X (dcomp_connect_boundary_sq)
    float dt2f;                // (float)X: a FUSED d^2 whose float image differs from this cannot sit within 4 doubles of X (host-checked),
                               //   so there the fused compare IS the reference decision; on equality the pair is redone in the reference form
    uint32_t dsq_exact;        // 1: that host check failed (X too close to a float rounding boundary): every pair takes the reference form
    double bs_x[DCOMP_MASK32_MAX_BS], bs_y[DCOMP_MASK32_MAX_BS];   // (stations of the specialised kernels; 33 ... 64 stations: dcomp_big.h reads a device table)
    int32_t bs_mode[DCOMP_MASK32_MAX_BS];
};

// ---------------------------------------------------------------------------------------------- cross-lane
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float swz_xor16(float v)
{   // ds_swizzle bit mode: and=0x1f, or=0, xor=0x10 -> lane i <-> i^16 inside each 32-lane half
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
#define DCOMP_DPP_QUAD_X1 0xB1      // quad_perm [1,0,3,2]
#define DCOMP_DPP_QUAD_X2 0x4E      // quad_perm [2,3,0,1]
#define DCOMP_DPP_HALF_MIRROR 0x141 // i <-> 7-i in each 8
#define DCOMP_DPP_ROW_MIRROR 0x140  // i <-> 15-i in each 16

struct OpSum { __device__ __forceinline__ static float f(float a, float b) { return a + b; } };
// min / max of finite values as ONE v_med3_f32: fminf / fmaxf of lane-exchanged values cost two canonicalising v_max_f32
// x, x, x on top under IEEE mode (the compiler cannot prove a DPP / swizzle result quiet)
struct OpMin { __device__ __forceinline__ static float f(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -3.0e38f); } };
struct OpMax { __device__ __forceinline__ static float f(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.0e38f); } };

// All-reduce over aligned groups of W lanes (W = 1,2,4,...,64).  Every lane of a group ends with the
// bit-identical result (each butterfly stage combines the same two partial results in both partners).
template <int W, class Op>
__device__ __forceinline__ float group_reduce(float v)
{
#ifdef DCOMP_NO_DPP
#pragma unroll
    for (int m = 1; m < W; m <<= 1) v = Op::f(v, __shfl_xor(v, m, 64));
#else
    if (W >= 2) v = Op::f(v, dpp_f32<DCOMP_DPP_QUAD_X1>(v));
    if (W >= 4) v = Op::f(v, dpp_f32<DCOMP_DPP_QUAD_X2>(v));
    if (W >= 8) v = Op::f(v, dpp_f32<DCOMP_DPP_HALF_MIRROR>(v));
    if (W >= 16) v = Op::f(v, dpp_f32<DCOMP_DPP_ROW_MIRROR>(v));
    if (W >= 32) v = Op::f(v, swz_xor16(v));
    if (W >= 64) v = Op::f(v, __shfl_xor(v, 32, 64));
#endif
    return v;
}
// Set bits of a ballot mask inside my W-lane group.  All in 32-bit pieces: `(float)__popcll(m >> gbase & mask)` makes the
// compiler carry the count as a 64-bit integer and expand a u64 -> f32 conversion (7 VALU per count instead of 3).
template <int W>
__device__ __forceinline__ int group_popcount(unsigned long long m, int gbase)
{
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    if (W == 64) return __builtin_popcount(lo) + __builtin_popcount(hi);              // scalar
    if (W == 32) { const int cl = __builtin_popcount(lo), ch = __builtin_popcount(hi); return gbase ? ch : cl; }   // 2 x s_bcnt1 + v_cndmask
    const uint32_t half = (gbase & 32) ? hi : lo;
    return __builtin_popcount((half >> (gbase & 31)) & ((1u << W) - 1u));
}
// Number of lanes of my W-group whose predicate holds.
template <int W>
__device__ __forceinline__ int group_count(bool pred, int gbase)
{
    return group_popcount<W>(__ballot(pred), gbase);
}

// N independent all-reduces, stage by stage: the N butterflies interleave, so no DPP read-after-write stalls.
template <int W, class Op, int N>
__device__ __forceinline__ void group_reduce_vec(float (&v)[N])
{
#ifdef DCOMP_NO_DPP
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = group_reduce<W, Op>(v[i]);
#else
    // The empty asm after every combine keeps the SLP vectoriser from pairing the N independent adds of a stage into v_pk_add_f32: a packed
    // add cannot take a DPP operand, so every pair then cost v_mov_b32_dpp x 2 + v_pk_add_f32 instead of v_add_f32_dpp x 2 (round 5: the
    // launch is VALU-issue-bound at the clock the part holds under it, profiles/r05_c3_clock_trace.txt)
#define DCOMP_STAGE(EXPR)                          \
    _Pragma("unroll") for (int i = 0; i < N; i++) { v[i] = Op::f(v[i], EXPR); asm volatile("" : "+v"(v[i])); }
    if (W >= 2) { DCOMP_STAGE(dpp_f32<DCOMP_DPP_QUAD_X1>(v[i])) }
    if (W >= 4) { DCOMP_STAGE(dpp_f32<DCOMP_DPP_QUAD_X2>(v[i])) }
    if (W >= 8) { DCOMP_STAGE(dpp_f32<DCOMP_DPP_HALF_MIRROR>(v[i])) }
    if (W >= 16) { DCOMP_STAGE(dpp_f32<DCOMP_DPP_ROW_MIRROR>(v[i])) }
    if (W >= 32) { DCOMP_STAGE(swz_xor16(v[i])) }
    if (W >= 64) { DCOMP_STAGE(__shfl_xor(v[i], 32, 64)) }
#undef DCOMP_STAGE
#endif
}
// ---- tight packing of UE lists whose length is not a power of two (step_kernel_tight only)
// Padded groups waste lanes: U = 10 sits in groups of 16 (37.5 % idle), U = 5 in 8, U = 20 in 32.  In tight mode an env takes
// exactly G = U lanes and a wavefront holds floor(64 / G) envs (U = 10: six instead of four).  The per-env reductions then run
// over lane segments that are neither aligned nor a power of two wide, which DPP butterflies cannot do: an inclusive
// Hillis-Steele scan with ds_bpermute (ceil(log2 G) steps, restricted to the segment) and a broadcast of the segment's last
// lane -- every lane of an env still ends with the bit-identical total.  ~13 instead of 4 instructions per reduction, paid
// back by a third fewer wavefronts wherever the kernel is throughput-bound (dcomp_create decides).
template <bool TIGHT>
struct SegT {
    static constexpr bool tight = TIGHT;   // compile-time: the padded kernels carry none of the segmented code
    int g;                  // lanes per env
    int u;                  // my position inside the segment
    int last_addr;          // ds_bpermute address (4 * lane) of the segment's last lane
    uint32_t mlo, mhi;      // ballot mask of my segment's lanes
};
using SegPadded = SegT<false>;
template <int W, class S>
__device__ __forceinline__ int seg_popcount(unsigned long long m, int gbase, const S &sg)
{
    if constexpr (S::tight) return __builtin_popcount((uint32_t)m & sg.mlo) + __builtin_popcount((uint32_t)(m >> 32) & sg.mhi);
    else return group_popcount<W>(m, gbase);
}
template <int W, class Op, int N, class S>
__device__ __forceinline__ void seg_reduce_vec(float (&v)[N], const S &sg, int lane)
{
    if constexpr (S::tight) {
        for (int d = 1; d < sg.g; d <<= 1) {                    // uniform trip count
            const int addr = (lane - d) * 4;
            const bool take = sg.u >= d;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v[i])));
                v[i] = take ? Op::f(v[i], t) : v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(sg.last_addr, __float_as_int(v[i])));
    } else group_reduce_vec<W, Op, N>(v);
}
template <int W, class Op, class S>
__device__ __forceinline__ float seg_reduce(float x, const S &sg, int lane)
{
    float v[1] = {x};
    seg_reduce_vec<W, Op, 1>(v, sg, lane);
    return v[0];
}

template <class T>
__device__ __forceinline__ void stream_store(T *ptr, T v)
{
#if DCOMP_NT_OBS
    __builtin_nontemporal_store(v, ptr);
#else
    *ptr = v;
#endif
}
// N consecutive floats of one lane as 16-byte pieces + a remainder: global stores need dword alignment only, so a lane's run of
// B or 4B+1 floats leaves as ceil(N / 4) store instructions instead of N -- what counts in the latency-bound fused rollout,
// where every store instruction of 64 scattered 4-byte pieces occupies the memory pipeline as long as one of 64 x 16 bytes.
template <int N>
__device__ __forceinline__ void store_run(float *dst, const float (&v)[N])
{
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (int i = 0; i + 4 <= N; i += 4) { f4u q; q.x = v[i]; q.y = v[i + 1]; q.z = v[i + 2]; q.w = v[i + 3]; *reinterpret_cast<f4u *>(dst + i) = q; }
    constexpr int R = N & ~3;
    if constexpr (N - R >= 2) { f2u q; q.x = v[R]; q.y = v[R + 1]; *reinterpret_cast<f2u *>(dst + R) = q; }
    if constexpr ((N - R) & 1) dst[N - 1] = v[N - 1];
}
// LDS ops of one wave execute in order; this only stops the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------- channel
constexpr float NEAR_D2 = 1.6f;   // squared distance under which the rare fix-ups run (see eval_pairs)
// Every `l2` / `l2snr` value in the kernels is log2(snr) + L2_OFF.  log2(snr) of an in-range pair lies in [-25.6, -4]; most
// connections sit at the far end, where an f32 has an ulp of 1.9e-6 -- the rounding of the final FMA alone would cost
// 6.6e-7 relative in snr.  Shifted by 24 the same pairs lie in [-1.6, 2] (ulp 2.4e-7).  The shift is free: the host adds it
// to log2 K, the rate series absorbs 2^-24 in its coefficients, observations only use differences.
constexpr float L2_OFF = 24.0f;
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32 (1 ulp)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (1 ulp)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32 (1 ulp)
// min / clamp as ONE v_med3_f32: fminf / fmaxf on a value the compiler cannot prove quiet (a select, a DPP move) cost an
// extra canonicalising v_max_f32 x, x, x each under IEEE mode.  No NaN reaches these.
__device__ __forceinline__ float min_med3(float x, float c) { return __builtin_amdgcn_fmed3f(x, -3.0e38f, c); }   // (-inf would be folded back into fminf)
__device__ __forceinline__ float clamp_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// log2(snr) and in-range test of one (UE, BS) pair.  station.py:110-127, 222-226.
//   snr = K * (d + 1e-16)^(-gamma)  ->  log2 snr = log2 K - (gamma/2) * log2(d^2)   for d >> 1e-16.
// `tiny` flags d^2 < 1e-20 (UE sitting on a BS: waypoints and BS positions share the integer grid), where the
// +1e-16 of station.py:116 matters; those pairs are redone by pair_eval_tiny under a wave-uniform rare branch.
// (pair_eval_q: the same with the float d^2 handed back instead of the `near` test -- eval_pairs keeps the MINIMUM over the stations and
// tests once: one v_med3 per station instead of a compare and an OR)
// The reference's squared distance: two rounded squares and a rounded sum (shapely Point.distance -> GEOS sqrt(dx*dx + dy*dy),
// station.py:124), and its range decision: dist_sq_ref < X (X = KParams::dt2, dcomp_connect_boundary_sq).  fma(dy, dy, dx*dx) differs from the
// two-rounding sum in the last bit for ~18 % of pairs -- by at most 2 ulp -- so a decision taken on the FUSED value can differ only where the
// fused value lies within 2 ulp of X.  Round 6: the kernels keep the fused value for the common case (one FP64 instruction fewer per pair, and
// the unfused form in every pair cost the headline kernel 35 VGPRs = two of its six waves per SIMD: 74.1 -> 79.0 us, profiles/r06_ab_dsq.txt)
// and redo the decision in the reference's literal form exactly where the two CAN differ: when (float)fused == (float)X.  The host checks
// that every double within 4 ulp of X has that float image (dsq_exact = 1 otherwise: always the reference form).  Hit rate ~1e-7 per pair;
// tests/test_threshold_gpu.py puts 1 000+ placements per kernel family there.
#ifndef DCOMP_EDGE_MODE
// How a lane notes "(float)fused d^2 == (float)X" over its stations: 0 = v_cmp_eq_f32 + s_or per pair (one vector + one scalar instruction, and a
// VALU -> SALU hand-over), 1 = min |q - (float)X| (two vector instructions, nothing scalar), 2 = by kernel: form 0 where many waves per SIMD hide the
// hand-over and the vector ALU is the co-limit (step_kernel: config 3 +0.1-0.3 % against +0.0-1.2 %), form 1 where one or two waves per SIMD run
// (the fused rollout, the wide kernel beyond the Infinity Cache: config 2 +2.9 % against +4.2 %, config 5 whole +2.9 % against +4.8 %;
// profiles/r06_ab_edge_mode*.txt, all against the round-5 predicate).
#define DCOMP_EDGE_MODE 2
#endif
#ifndef DCOMP_DSQ_FUSED
#define DCOMP_DSQ_FUSED 0          // 1: the round-5 predicate, fused value only (A/B; fails the threshold tests)
#endif
__device__ __forceinline__ double dist_sq_ref(double dx, double dy)
{
#pragma clang fp contract(off)
    const double xx = dx * dx, yy = dy * dy;
    return xx + yy;
}
// the rare side of the decision, kept out of line: (float)fused d^2 == (float)X
#ifndef DCOMP_EXACT_INLINE
#define DCOMP_EXACT_INLINE 0         // 1: the rare reference-form re-check inlined at every site instead of one out-of-line function (A/B)
#endif
#if DCOMP_EXACT_INLINE
__device__ __forceinline__
#else
__device__ __noinline__
#endif
bool in_range_exact(double px, double py, double bx, double by, double dt2)
{
    return dist_sq_ref(bx - px, by - py) < dt2;
}
__device__ __forceinline__ void pair_eval_q(double px, double py, double bx, double by, const KParams &p, bool &in_range, float &l2snr, float &q);
__device__ __forceinline__ void pair_eval(double px, double py, double bx, double by, const KParams &p, bool &in_range,
                                          float &l2snr, bool &tiny)
{
    float q;
    pair_eval_q(px, py, bx, by, p, in_range, l2snr, q);
    tiny = q < NEAR_D2;                          // "near": superset of the d^2 < 1e-20 pairs the fix-up replaces
#if !DCOMP_DSQ_FUSED
    if (q == p.dt2f || p.dsq_exact) in_range = in_range_exact(px, py, bx, by, p.dt2);       // rare, per lane: the reference's literal form
#endif
}
__device__ __forceinline__ void pair_eval_q(double px, double py, double bx, double by, const KParams &p, bool &in_range, float &l2snr, float &q)
{
    double dx = bx - px, dy = by - py;
    double dsq = __builtin_fma(dy, dy, dx * dx);
    in_range = dsq < p.dt2;                      // PROVISIONAL where (float)dsq == p.dt2f: the caller redoes those pairs with in_range_exact
    q = (float)dsq;
    // v_log_f32's absolute error scales with |result| (log2 d^2 ~ 12 near the connect range -> ~1e-6, i.e. ~1.1e-6
    // relative in snr = 2^l2snr).  Scaling d^2 by 2^-12 first puts every in-range pair at |log2| < 4 for the price
    // of one multiply (the exact alternative, frexp + two FMAs, costs 3 % of the step; tools/numerics_report.py has
    // the measured errors of all three).  Tiny pairs give -inf/NaN here and are replaced by eval_pairs' fix-up.
#if DCOMP_LOG2_MODE == 0
    l2snr = __builtin_fmaf(-p.half_gamma, fast_log2(fmaxf(q, 1e-20f)), p.log2k);
#elif DCOMP_LOG2_MODE == 1
    const float qc = fmaxf(q, 1e-20f);
    const float lm = fast_log2(__builtin_amdgcn_frexp_mantf(qc));
    const float ef = (float)__builtin_amdgcn_frexp_expf(qc);
    l2snr = __builtin_fmaf(-p.half_gamma, ef, __builtin_fmaf(-p.half_gamma, lm, p.log2k));
#else
    l2snr = __builtin_fmaf(-p.half_gamma, fast_log2(q * 0x1p-12f), p.log2k_s);
#endif
}
__device__ __forceinline__ float pair_eval_tiny(double px, double py, double bx, double by, const KParams &p)
{
    double dx = bx - px, dy = by - py;
    float q = (float)__builtin_fma(dy, dy, dx * dx);
    float d = __builtin_amdgcn_sqrtf(q) + EPS;                  // d + 1e-16 (station.py:116)
    return __builtin_fmaf(-2.0f * p.half_gamma, fast_log2(d), p.log2k);
}
// All B pairs of one UE; returns the in-range mask.
#if DCOMP_BS_IN_LDS
__shared__ double g_bs_lds[2 * DCOMP_MASK32_MAX_BS];
#define DCOMP_BSX(b) (*(volatile double *)&g_bs_lds[b])
#define DCOMP_BSY(b) (*(volatile double *)&g_bs_lds[DCOMP_MASK32_MAX_BS + (b)])
#else
#if DCOMP_BS_VOLATILE
#define DCOMP_BSX(b) (*(const volatile double *)&p.bs_x[b])
#define DCOMP_BSY(b) (*(const volatile double *)&p.bs_y[b])
#else
#define DCOMP_BSX(b) p.bs_x[b]
#define DCOMP_BSY(b) p.bs_y[b]
#endif
#endif
// `near_wave` (optional, wave-uniform): some lane of this wave is within NEAR_D2^(1/2) = 1.26 m of a BS.  That one test
// triggers both rare fix-ups: the exact d + 1e-16 of a UE sitting ON a BS (here) and, in shared_rates, the rate of a pair
// with snr > 1/64 (d < 1.24 m), which the short log1p series does not cover.
// bsx / bsy (optional): the BS table in registers of the caller's choosing -- the fused rollout keeps it in VGPRs, see
// step_kernel_body; nullptr: the kernel-argument segment (SGPRs).
template <int B>
__device__ __forceinline__ uint32_t eval_pairs(double px, double py, const KParams &p, float (&l2)[B], bool *near_wave = nullptr,
                                               const double *bsx = nullptr, const double *bsy = nullptr)
{
    uint32_t in_range = 0;
    float qmin = 3.0e38f;
    // some station of this lane sits where the fused d^2 cannot decide (pair_eval_q): noted by compares (form 0) or as min |q - (float)X| (form 1)
    const bool vec_form = DCOMP_EDGE_MODE == 1 || (DCOMP_EDGE_MODE == 2 && bsx != nullptr);      // (bsx: the latency-bound fused rollout)
    bool edge = false;
    float emin = 3.0e38f;
#pragma unroll
    for (int b = 0; b < B; b++) {
        bool ir;
        float q;
        pair_eval_q(px, py, bsx ? bsx[b] : DCOMP_BSX(b), bsy ? bsy[b] : DCOMP_BSY(b), p, ir, l2[b], q);
        in_range |= (uint32_t)ir << b;
        qmin = min_med3(qmin, q);
        if (vec_form) emin = min_med3(emin, __builtin_fabsf(q - p.dt2f));
        else edge |= q == p.dt2f;
    }
    if (vec_form) edge = emin == 0.f;
#if !DCOMP_DSQ_FUSED
    if (__ballot(edge) != 0ull || p.dsq_exact) {  // rare (~1e-7 per pair), wave-uniform: every station of the wave again, in the reference's form
        in_range = 0;
#pragma unroll
        for (int b = 0; b < B; b++)
            in_range |= (uint32_t)in_range_exact(px, py, bsx ? bsx[b] : p.bs_x[b], bsy ? bsy[b] : p.bs_y[b], p.dt2) << b;
    }
#endif
    const bool nw = __ballot(qmin < NEAR_D2) != 0ull;
    if (near_wave) *near_wave = nw;
    if (nw) {                                    // rare (~3 % of the wavefronts): a lane within 1.26 m of a BS
#pragma unroll
        for (int b = 0; b < B; b++) {
            const double bx = bsx ? bsx[b] : p.bs_x[b], by = bsy ? bsy[b] : p.bs_y[b];
            float t = pair_eval_tiny(px, py, bx, by, p);
            double dx = bx - px, dy = by - py;
            if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l2[b] = t;
        }
    }
    return in_range;
}
// bw * log2(1 + snr) from log2(snr).  station.py:129-138.  Never forms 1+snr for small snr (1.0f + snr is
// exactly 1.0f below 6e-8 while the connect threshold is 2e-8): log1p(s)/s = 1 - s/2 + s^2/3 - s^3/4, truncation error
// s^4/5 < 1.2e-8 for snr < 1/64 (d > 1.24 m).  `needfix` flags the rare larger snr, redone by rate_unshared_any.
constexpr float RATE_SMALL_L2 = -6.0f + L2_OFF;   // (shifted) log2 of the largest snr the short series takes
__device__ __forceinline__ float rate_unshared_small(float l2snr, bool &needfix)
{
    needfix = l2snr > RATE_SMALL_L2;
    const float z = fast_exp2(min_med3(l2snr, RATE_SMALL_L2));        // z = snr * 2^24
    // bw/ln2 * s * (1 - s/2 + s^2/3 - s^3/4) with s = z * 2^-24: the powers of 2^-24 live in the coefficients
    constexpr float K = BW * LOG2E, S1 = 0x1p-24f, S2 = 0x1p-48f, S3 = 0x1p-72f, S4 = 0x1p-96f;
    float t = __builtin_fmaf(z, -0.25f * K * S4, 0.33333334f * K * S3);
    t = __builtin_fmaf(z, t, -0.5f * K * S2);
    t = __builtin_fmaf(z, t, K * S1);
    return z * t;
}
__device__ __forceinline__ float rate_unshared_any(float l2snr)
{
    // snr >= 1/64.  u = fl(1 + s) loses the low bits of s; log(1+s) = log(u) * s / (u - 1) puts them back (u - 1 is exact),
    // relative error ~2e-7 for any s.  d -> 0 (snr up to 3.5e52): log2(1+s) = log2(s) to f32 precision.
    const float l = l2snr - L2_OFF;                                    // exact enough here: |l2snr| < 2^8, l2snr >= 18
    const float s = fast_exp2(fminf(l, 100.f));
    const float u = 1.0f + s;
    const float lg = fast_log2(u) * (s * fast_rcp(u - 1.0f));
    return BW * (l > 100.f ? l : lg);
}
// user.py:76-92 -> utility.py:23-54
__device__ __forceinline__ float ue_utility(float dr, bool step_util, float dr_req)
{
    if (step_util) return dr >= dr_req ? MAX_UTIL : MIN_UTIL;
    if (dr == 0.f) return MIN_UTIL;
    float u = TEN_LOG10_2 * fast_log2(dr);
    return clamp_med3(u, MIN_UTIL, MAX_UTIL);
}

// ---------------------------------------------------------------------------------------------- RNG
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // (the 64-bit form makes the compiler emit ONE v_mad_u64_u32 per product instead of v_mul_hi_u32 + v_mul_lo_u32 -- all of
        // them quarter-rate instructions: config 4's share -1.9 %, config 3 -0.5 %; same integers)
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// packed movement word: wx:16 | wy:16 | vel:8 | pause:8 (bit 7 pausing, bits 0-6 curr_pause) | cursor:16
__device__ __forceinline__ unsigned long long mv_pack(uint32_t wx, uint32_t wy, uint32_t vel, uint32_t pausing, uint32_t cp,
                                                      uint32_t cursor)
{
    return (unsigned long long)(wx & 0xFFFF) | ((unsigned long long)(wy & 0xFFFF) << 16) | ((unsigned long long)(vel & 0xFF) << 32) |
           ((unsigned long long)((pausing << 7) | (cp & 0x7Fu)) << 40) | ((unsigned long long)(cursor & 0xFFFF) << 48);
}

// movement.py:110-130 (RandomWaypoint.reset): k-th movement triple of this UE in this episode.
// uidw = UE id (1-based) | UID_BORN for UEs that arrived during the episode (base.py:592-606: always 'slow', freshly
// seeded).  Draws are keyed by the id, not by the slot: slots shift when a UE leaves.
constexpr uint32_t UID_BORN = 0x8000u;
// mcfg: the UE's movement parameters (mv_cfg_pack) -- velocity draw range, border buffer.  Fixed UE lists load them once per
// kernel, next to the state: a load here would sit in the middle of the move and, in the fused rollout, wait behind the
// previous step's stores.
__device__ __forceinline__ void draw_triple(const KParams &p, int env, uint32_t uidw, uint32_t k, uint32_t episode, uint32_t &vel,
                                            uint32_t &wx, uint32_t &wy, int mcfg)
{
    const uint32_t id0 = (uidw & 0x7FFFu) - 1u;
    const bool born = (uidw & UID_BORN) != 0u;
    if (p.rng_mode == DCOMP_RNG_TAPE) {
        if (k >= (uint32_t)p.tape_depth) { atomicOr(p.flags, DCOMP_FLAG_TAPE_EMPTY); k = p.tape_depth - 1; }
        const uint32_t slot = born ? (uint32_t)p.U0 + id0 : id0;
        ushort4 t = p.tape_triples[((size_t)env * p.tape_ids + slot) * p.tape_depth + k];
        vel = t.x; wx = t.y; wy = t.z;
        VM_ARRIVED3(vel, wx, wy);
    } else {
        uint32_t r[4] = {0x12345678u + k * 977u, 0x9abcdef0u ^ uidw * 2654435761u, 0x0fedcba9u + (uint32_t)env * 40503u, 0u};
        if (!(DCOMP_ABLATE & 128)) philox4x32_10(p.env_base + (uint32_t)env, id0 | (born ? UID_BORN : 0u), episode, k + 1, p.seed_lo, p.seed_hi, r);
        const uint32_t vlo = (uint32_t)mcfg & 0xFFu, vhi = ((uint32_t)mcfg >> 8) & 0xFFu, bb = ((uint32_t)mcfg >> 23) & 0xFFu;
        vel = vlo + __umulhi(r[0], vhi - vlo + 1u);                                   // movement.py:112-117
        wx = bb + __umulhi(r[1], (uint32_t)(p.map_w - 2 * (int)bb + 1));              // movement.py:126-127
        wy = bb + __umulhi(r[2], (uint32_t)(p.map_h - 2 * (int)bb + 1));
    }
}
// The same draw for the FEW lanes of a wavefront that redraw in a given step (a UE redraws once per leg of its walk: 1-2 of 64
// lanes per step), on the SCALAR unit: the wave walks its redrawing lanes, reads the lane's counter words with v_readlane and
// runs Philox on uniform values -- s_mul_hi_u32 / s_mul_i32 / s_xor_b32, which do not occupy the vector ALU -- then hands
// the three results to that lane with v_cndmask.  The vector form (draw_triple under `if (redraw)`) issues the ~95 VALU
// instructions of the ten rounds for the whole wave whenever ANY lane redraws, i.e. in most steps: 6-12 % of the step's VALU
// work.  Same integers either way.  `redraw` may be false in every lane; lanes outside the caller's branch do not take part.
template <bool SCALAR>
__device__ __forceinline__ void draw_triple_wave(const KParams &p, int env, uint32_t uidw, uint32_t k, uint32_t episode, bool redraw,
                                                 uint32_t &vel, uint32_t &wx, uint32_t &wy, int mcfg)
{
    if (!SCALAR || (!DCOMP_EXP_NO_TAPE && p.rng_mode == DCOMP_RNG_TAPE)) {           // host-drawn tape: a per-lane load
        if (redraw) draw_triple(p, env, uidw, k, episode, vel, wx, wy, mcfg);
        return;
    }
    const int lane = (int)(threadIdx.x & 63u);
    unsigned long long need = __ballot(redraw);
    while (need != 0ull) {                                   // uniform: one trip per redrawing lane
        const int l = __ffsll((long long)need) - 1;
        need &= need - 1ull;
        // The key is uniform and loop-invariant, and so are the ten round keys derived from it: left alone, the compiler hoists the
        // whole key schedule out of the step loop and parks it in 20 scalar registers the loop does not have.  Opaque copies keep
        // the schedule where it is used (a few s_add per draw).
        uint32_t k0 = p.seed_lo, k1 = p.seed_hi, eb = p.env_base;
        asm volatile("" : "+s"(k0), "+s"(k1), "+s"(eb));
        const uint32_t e_s = (uint32_t)__builtin_amdgcn_readlane(env, l), u_s = (uint32_t)__builtin_amdgcn_readlane((int)uidw, l);
        const uint32_t k_s = (uint32_t)__builtin_amdgcn_readlane((int)k, l), m_s = (uint32_t)__builtin_amdgcn_readlane(mcfg, l);
        uint32_t r[4] = {0x12345678u + k_s * 977u, 0x9abcdef0u ^ u_s * 2654435761u, 0x0fedcba9u + e_s * 40503u, 0u};
        if (!(DCOMP_ABLATE & 128)) philox4x32_10(eb + e_s, ((u_s & 0x7FFFu) - 1u) | (u_s & UID_BORN), episode, k_s + 1u, k0, k1, r);
        const uint32_t vlo = m_s & 0xFFu, vhi = (m_s >> 8) & 0xFFu, bb = (m_s >> 23) & 0xFFu;
        const uint32_t v_s = vlo + __umulhi(r[0], vhi - vlo + 1u);                              // movement.py:112-117
        const uint32_t x_s = bb + __umulhi(r[1], (uint32_t)(p.map_w - 2 * (int)bb + 1));        // movement.py:126-127
        const uint32_t y_s = bb + __umulhi(r[2], (uint32_t)(p.map_h - 2 * (int)bb + 1));
        const bool mine = lane == l;
        vel = mine ? v_s : vel; wx = mine ? x_s : wx; wy = mine ? y_s : wy;
    }
}
// Movement parameters of the UE with id word uidw when the caller does not hold them (UE lists that change: dcomp_dyn.h).
__device__ __forceinline__ int load_mv_cfg(const KParams &p, uint32_t uidw)
{
    if (uidw & UID_BORN) return MV_CFG_ARRIVED;
    const UeCfg c = p.ue_cfg[(uidw & 0x7FFFu) - 1u];
    return mv_cfg_pack(c.vel_lo, c.vel_hi, c.pause, c.border);
}

// nrm = sqrt(vy*vy + vx*vx), nx = vx / nrm, ny = vy / nrm, correctly rounded: the compiler's own FP64 sqrt and division
// sequences (v_rsq_f64 / v_rcp_f64 + Newton steps + residual correction) without their range scaling and special-value
// fix-ups (v_div_scale / v_div_fmas / v_div_fixup / v_ldexp / v_cmp_class) -- the argument is a squared distance on the
// map, 0 < q < 2^41, the numerators are below 2^17 in magnitude -- and with ONE reciprocal shared by both divisions.
// 21 instead of 48 instructions; tests/test_parity_gpu.py holds the positions bit-exact against the CPU oracle.
__device__ __forceinline__ void norm_and_unit(double vx, double vy, double &nrm, double &nx, double &ny)
{
#pragma clang fp contract(off)
    const double x = __builtin_fma(vy, vy, vx * vx);
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    nrm = g;
    double rc = __builtin_amdgcn_rcp(g);
    double e = __builtin_fma(-g, rc, 1.0);
    rc = __builtin_fma(rc, e, rc);
    e = __builtin_fma(-g, rc, 1.0);
    rc = __builtin_fma(rc, e, rc);
    const double qx = vx * rc, qy = vy * rc;
    nx = __builtin_fma(__builtin_fma(-g, qx, vx), rc, qx);
    ny = __builtin_fma(__builtin_fma(-g, qy, vy), rc, qy);
}

// The position update of one movement step (movement.py:132-156) once pause / redraw are settled: snap onto the waypoint when it
// is within one step, else walk `velocity` along the unit vector.  SELECTS: straight-line form of the fused rollout (move_ue).
// TABLE: the UE's velocity may come from p.ue_velq (any non-negative number) instead of the integer in its movement word.
template <bool SELECTS, bool TABLE>
__device__ __forceinline__ void advance_ue(const KParams &p, uint32_t uidw, uint32_t vel, bool stay, double wx, double wy, double &px, double &py)
{
#pragma clang fp contract(off)
    double velf, qmax;
    {
        const uint32_t v2 = vel * vel;
        qmax = (double)v2;
        const int e2 = 2 * (31 - __clz((int)(vel | 1u)));
        qmax = (vel > 0u && v2 < (2u << e2)) ? qmax + __builtin_ldexp(1.0, e2 - 52) : qmax;
        velf = (double)vel;
    }
    if (TABLE) {
        double2 vq = (uidw & UID_BORN) ? make_double2(-1.0, 0.0) : p.ue_velq[(uidw & 0x7FFFu) - 1u];
        VM_ARRIVED2(vq.x, vq.y);
        velf = vq.x < 0.0 ? velf : vq.x;
        qmax = vq.x < 0.0 ? qmax : vq.y;
    }
    if (SELECTS) {
        double dx = px - wx, dy = py - wy;
        double q = dx * dx + dy * dy;
        const bool snap = q <= qmax;                                // snap onto the waypoint
        double vx = wx - px, vy = wy - py;
        double nrm, nx, ny;
        norm_and_unit(vx, vy, nrm, nx, ny);                         // np.linalg.norm, then two divisions (movement.py:151)
        const double mx = px + velf * nx, my = py + velf * ny;
        px = stay ? px : snap ? wx : mx;
        py = stay ? py : snap ? wy : my;
    } else if (!stay) {
        double dx = px - wx, dy = py - wy;
        double q = dx * dx + dy * dy;
        if (q <= qmax) { px = wx; py = wy; }                        // snap onto the waypoint
        else {
            double vx = wx - px, vy = wy - py;
            double nrm, nx, ny;
#if DCOMP_COMPILER_DIV
            nrm = __builtin_sqrt(__builtin_fma(vy, vy, vx * vx));
            nx = vx / nrm; ny = vy / nrm;
#else
            norm_and_unit(vx, vy, nrm, nx, ny);                     // np.linalg.norm, then two divisions (movement.py:151)
#endif
            px = px + velf * nx;
            py = py + velf * ny;
        }
    }
}

// One RandomWaypoint step in FP64, in the reference's operation order.  movement.py:132-181.
// Contraction is off: the only fused op is the explicit fma of the 2-element dot product (numpy).
// LAT (the fused rollout: latency-bound, one wave per SIMD, short of scalar registers): the redrawing lanes' Philox on the scalar
// unit and straight-line code with selects -- in a wave of 64 UEs some lane takes every side of every branch in nearly every
// step, so the branches only add exec-mask bookkeeping; lanes that stay put run the normalisation on a zero vector and
// discard the NaNs.  The throughput-bound step kernels keep the branchy form with the vector Philox: measured on one box,
// the LAT form costs them 1.5 % (config 3) to 5.7 % (65 536 x 10 x 5 central) -- their SIMDs have other waves to run while one
// sits in a branch, and the scalar unit is shared by the four SIMDs of a CU.  Same FP64 operations, same results.
template <bool LAT = false>
__device__ __forceinline__ void move_ue(const KParams &p, int env, uint32_t uidw, uint32_t episode, double &px, double &py,
                                        unsigned long long &mv, int mcfg)
{
#pragma clang fp contract(off)
    constexpr bool SCALAR = DCOMP_SCALAR_DRAW == 1 || (DCOMP_SCALAR_DRAW == 2 && LAT);
    constexpr bool SELECTS = DCOMP_MOVE_SELECTS == 1 || (DCOMP_MOVE_SELECTS == 2 && LAT);
    uint32_t wxi = (uint32_t)(mv & 0xFFFF), wyi = (uint32_t)((mv >> 16) & 0xFFFF), vel = (uint32_t)((mv >> 32) & 0xFF);
    uint32_t pz = (uint32_t)((mv >> 40) & 0xFF), cursor = (uint32_t)(mv >> 48);
    uint32_t pausing = (pz >> 7) & 1, cp = pz & 0x7Fu;
    const uint32_t pause_dur = ((uint32_t)mcfg >> 16) & 0x7Fu;
    double wx = (double)wxi, wy = (double)wyi;
    if (px == wx && py == wy) pausing = 1;                          // movement.py:169-170
    const bool stay = pausing && cp < pause_dur;                    // movement.py:172-175
    const bool redraw = pausing && !stay;                           // movement.py:176 -> reset(): new velocity + waypoint, then move
    cp = stay ? cp + 1u : cp;
    draw_triple_wave<SCALAR>(p, env, uidw, cursor, episode, redraw, vel, wxi, wyi, mcfg);
    cursor = redraw ? cursor + 1u : cursor;
    pausing = redraw ? 0u : pausing;
    cp = redraw ? 0u : cp;
    wx = (double)wxi; wy = (double)wyi;                             // (unchanged unless redrawn)
    // movement.py:142: `curr_pos.distance(waypoint) <= velocity` with distance = sqrt(dx*dx + dy*dy).
    // sqrt is monotone and correctly rounded, so the test is `q <= qmax(vel)`, qmax = largest double whose
    // rounded sqrt is <= vel: vel^2, plus one ulp when the mantissa m of vel has m < sqrt(2)
    // (dcomp_create verifies this closed form against a brute-force sqrt table for vel = 0..255).
    // Velocities that are not an integer in 0..255 (a caller's RandomWaypoint(map, velocity=2.5)): {velocity, qmax} from a per-UE
    // table the host built with the same definition -- a wave-uniform branch on a kernel argument, taken by no env of the
    // reference's own scenarios.
    if (p.ue_velq != nullptr) advance_ue<SELECTS, true>(p, uidw, vel, stay, wx, wy, px, py);
    else advance_ue<SELECTS, false>(p, uidw, vel, stay, wx, wy, px, py);
    mv = mv_pack(wxi, wyi, vel, pausing, cp, cursor);
}

// ---------------------------------------------------------------------------------------------- kernels
// LDS scratch shared by the cross-wave exchange, max-cap arg-min and the 'sum' neighbourhood reward.
// Sharing-model pattern baked into the kernel (station.py:152-202, env_setup.py:40-49):
//   MP_GENERIC   per-BS model read from the argument block at run time
//   MP_RES_FAIR  every BS resource-fair (no float sums over UEs at all)
//   MP_MIXED     the CLI default 'mixed': BS b -> [resource-fair, rate-fair, proportional-fair][b % 3]
enum { MP_GENERIC = 0, MP_RES_FAIR = 1, MP_MIXED = 2 };
template <int MP>
__device__ __forceinline__ int bs_mode_of(const KParams &p, int b)
{
    if (MP == MP_RES_FAIR) return DCOMP_RES_FAIR;
    if (MP == MP_MIXED) return (b % 3 == 0) ? DCOMP_RES_FAIR : (b % 3 == 1) ? DCOMP_RATE_FAIR : DCOMP_PROP_FAIR;
    return p.bs_mode[b];
}

template <int B, int UPAD>
struct Geo {
    static constexpr int WG = UPAD < 64 ? UPAD : 64;        // in-wave group width
    static constexpr int NW = UPAD > 64 ? UPAD / 64 : 1;    // waves per env
    static constexpr int GPB = DCOMP_BLOCK >= UPAD ? DCOMP_BLOCK / UPAD : 1;   // envs per block
    static constexpr int ROW_MULTI = 4 * B + 1;
};

// Observation staging: rows are written lane-per-row into LDS and copied out linearly, so every global store
// instruction covers 64 x 16 contiguous bytes (1 KiB) instead of 64 scattered 16-byte pieces.
template <int B>
struct StageGeo {
    static constexpr int ROW = 4 * B + 1;
    static constexpr int NPASS = (64 * ROW * 4 <= 6144) ? 1 : (32 * ROW * 4 <= 6144) ? 2 : (16 * ROW * 4 <= 6144) ? 4 : 8;
    static constexpr int RPP = 64 / NPASS;                  // rows per pass
    // Central observations (connected[U*B] | dr[U*B] | utility[U] per env, central.py:147-151) of the <= 64 rows of a wave are
    // 64 * (2B+1) contiguous floats: staged in ONE window when that fits ~6 KB (B <= 11), else in windows of the multi size.
    static constexpr int CENTRAL_ONE = 64 * (2 * B + 1) + 4;
    static constexpr int MULTI_WORDS = RPP * ROW + 4;       // + alignment phase
    static constexpr int STAGE_WORDS = (CENTRAL_ONE <= 1540 && CENTRAL_ONE > MULTI_WORDS) ? CENTRAL_ONE : MULTI_WORDS;
    // The sparse pre-move pass parks one row of B + 1 unshared rates per lane in the same bytes; B = 12..23 need 44 words more
    // than the staging windows -- taken; B >= 24 would double the buffer (8 passes of 8 rows) -- not taken, dense pass there.
    static constexpr int SPARSE_WORDS = 64 * (B + 1);
    static constexpr int WORDS = (SPARSE_WORDS > STAGE_WORDS && SPARSE_WORDS <= STAGE_WORDS + 64) ? SPARSE_WORDS : STAGE_WORDS;
};

template <int B, int UPAD>
struct alignas(16) BlockSharedT {
    // The max-cap and 'sum'-reward scratch is dead (workgroup barrier behind its last read) before the first staging write,
    // so it shares the staging bytes: 24.3 -> 21.3 KB per workgroup at B = 10, i.e. 7 instead of 6 workgroups per CU.
    union {
        alignas(16) float stage[DCOMP_BLOCK / 64][(StageGeo<B>::WORDS + 3) / 4 * 4];   // per-wave observation staging (multi-agent layout)
        struct {
            unsigned long long mc_key[Geo<B, UPAD>::GPB * B];   // max-cap: min squared-distance bits per (env-in-block, bs)
            unsigned long long mc_u[Geo<B, UPAD>::GPB * B];     //          max FP64 rate key 1 + snr among the contenders
            uint32_t mc_win[Geo<B, UPAD>::GPB * B];             //          (step of connection << 8 | UE) of the winner
            uint32_t mc_cnt[Geo<B, UPAD>::GPB * B];             //          number of contenders
            uint32_t nb_conn[256];                              // 'sum' reward: conn' and reward_before of the block's UEs
            float nb_rb[256];
        };
    };
    float xw[4][B + 4];                                     // per-wave partials of the cross-wave exchange
    double2 bs[B];                                          // BS positions for the per-lane station index of the sparse pre-move pass
};

// Sum `N` per-wave values across the NW waves of an env (values are wave-uniform when WG == 64).
template <int N, int NW, class Op, class SH>
__device__ __forceinline__ void xwave_reduce_(float (&v)[N], SH &sh, int wave, int lane)
{
    if (NW == 1) return;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) sh.xw[wave][i] = v[i];
    }
    __syncthreads();
    int w0 = (wave / NW) * NW;
#pragma unroll
    for (int i = 0; i < N; i++) {
        float a = sh.xw[w0][i];
#pragma unroll
        for (int k = 1; k < NW; k++) a = Op::f(a, sh.xw[w0 + k][i]);
        v[i] = a;
    }
    __syncthreads();
}

// What the reference's max-cap argmax actually compares (station.py:129-138, 183-187): bw * log2(1 + snr) in FP64, and
// 1 + snr absorbs all but ~30 bits of an snr of 1e-6 -- UEs whose squared distances differ by up to ~1e-8 relative get the
// SAME rate and the first one in connection order wins.  This returns that collapsing quantity, fl(1 + snr), through the
// reference's own operation chain (distance, path loss via log10, 10^x, / noise).  Rare path, kept out of line.
__device__ __noinline__ unsigned long long maxcap_rate_key(double pl_c1, double pl_c2, double px, double py, double bx, double by)
{
#pragma clang fp contract(off)
    const double dx = bx - px, dy = by - py;
    const double d = __builtin_sqrt(dx * dx + dy * dy);                  // shapely Point.distance
    const double pl = pl_c1 + pl_c2 * log10(d + 1e-16);                  // station.py:110-116
    const double snr = pow(10.0, (30.0 - pl) / 10.0) / 1e-9;             // station.py:118-127
    return (unsigned long long)__double_as_longlong(1.0 + snr);          // positive doubles order like their bit patterns
}

// Shared data rates of this UE at every BS.  station.py:152-220 with S_b = {u : conn[u,b]}.
//   in : conn mask, l2snr[b], ewma, (px,py) for the max-cap arg-min
//   out: dr[b] (0 where not connected), cnt[b] = |S_b|
// PRE: the unshared rates of the connected pairs were computed by the sparse pre-move pass and wait in the lane's LDS row `prow`.
// CARRY (fused rollout): the UNSHARED rate of every station at the UE's current position is computed once per position -- by the
// post-move call of step t (exported through `dru_io`, for all B stations, connected or not) and reused by the pre-move call of
// step t + 1 (`dru_io` read: no exp2 / series at all): 1 = export, 2 = import.
template <int B, int UPAD, int MP, class S = SegPadded, bool PRE = false, int CARRY = 0>
__device__ __forceinline__ void shared_rates(const KParams &p, BlockSharedT<B, UPAD> &sh, uint32_t conn, const float (&l2)[B], float ewma,
                                             double px, double py, int u, int idx, int env_local, int wave, int lane, int gbase,
                                             float (&dr)[B], float (&cnt)[B], int near_hint = -1, const S &sg = S{}, const float *prow = nullptr,
                                             float *dru_io = nullptr)
{
    // near_hint (wave-uniform): eval_pairs' "a lane of this wave is within 1.26 m of a BS" for the position l2 belongs to
    // (1 / 0), or -1 = unknown, then the per-pair snr > 1/64 test is made here.
    using G = Geo<B, UPAD>;
    float agg[B];
    const float inv_ewma = fast_rcp(ewma + EPS);          // station.py:150 priority denominator (beta = 1)
    bool fix = false;
#pragma unroll
    for (int b = 0; b < B; b++) {
        const bool c = (conn >> b) & 1u;
        unsigned long long m = __ballot(c);
        float dru = 0.f;
        if (PRE) dru = c ? prow[b] : 0.f;                  // entries of unconnected stations are never read
        else if (CARRY == 2) dru = c ? dru_io[b] : 0.f;    // computed by the previous step's post-move call
        else if (CARRY == 1) {                             // every station: the next step may connect to any of them
            bool f;
            const float t = rate_unshared_small(l2[b], f);
            dru_io[b] = t;
            dru = c ? t : 0.f;
        } else if (m != 0ull) {                            // wave-uniform: skip BSs nobody in this wave is connected to
            bool f;
            const float t = rate_unshared_small(l2[b], f);
            dru = c ? t : 0.f;
            if (near_hint < 0) fix |= c && f;
        }
        dr[b] = dru;
        cnt[b] = (float)seg_popcount<G::WG>(m, gbase, sg);
    }
    if (!PRE && CARRY != 2 && (near_hint < 0 ? (__ballot(fix) != 0ull) : (near_hint != 0))) {   // rare: a connected UE closer than 1.24 m to its BS (snr > 1/64)
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (CARRY == 1) { if (l2[b] > RATE_SMALL_L2) { dru_io[b] = rate_unshared_any(l2[b]); if ((conn >> b) & 1u) dr[b] = dru_io[b]; } }
            else if (((conn >> b) & 1u) && l2[b] > RATE_SMALL_L2) dr[b] = rate_unshared_any(l2[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
        const bool c = (conn >> b) & 1u;
        const int mode = bs_mode_of<MP>(p, b);
        float a = 0.f;
        if (mode == DCOMP_RATE_FAIR) a = c ? fast_rcp(dr[b]) : 0.f;       // station.py:177-180
        else if (mode == DCOMP_PROP_FAIR) a = dr[b] * inv_ewma;           // station.py:192-195 (0 when not connected)
        agg[b] = a;
    }
    if (MP == MP_MIXED) {
        // only the rate-/proportional-fair BSs (b % 3 != 0) need a sum over UEs
        constexpr int NS = B - (B + 2) / 3;
        if (NS > 0) {
            float sv[NS > 0 ? NS : 1];
#pragma unroll
            for (int b = 0, k = 0; b < B; b++) if (b % 3 != 0) sv[k++] = agg[b];
            seg_reduce_vec<G::WG, OpSum, (NS > 0 ? NS : 1)>(sv, sg, lane);
#pragma unroll
            for (int b = 0, k = 0; b < B; b++) if (b % 3 != 0) agg[b] = sv[k++];
        }
    } else if (MP == MP_GENERIC && p.any_sum_mode) seg_reduce_vec<G::WG, OpSum, B>(agg, sg, lane);
    if (G::NW > 1) {
        xwave_reduce_<B, G::NW, OpSum>(cnt, sh, wave, lane);
        if (MP == MP_MIXED || (MP == MP_GENERIC && p.any_sum_mode)) xwave_reduce_<B, G::NW, OpSum>(agg, sh, wave, lane);
    }
    uint32_t mc_winner = 0;
    if (MP == MP_GENERIC && p.any_maxcap) {
        // max-cap (station.py:183-187): only the UE with the highest FP64 unshared rate is served; equal rates -> first in
        // bs.conn_ues = oldest connection, then (same step) lowest UE index, the order base.py:259-263 appends them in.
        // Stage 1: smallest FP64 squared distance per (env, BS).  Stage 2: contenders = UEs within 1e-7 (relative) of it --
        // anything farther has a strictly smaller rate (the 1 + snr rounding collapses at most 1.1e-8 at the cell edge).
        // Stage 3, only where a BS has more than one contender: the exact collapsing key (maxcap_rate_key).  Stage 4: among
        // the UEs holding the maximal key, the smallest (step of connection, UE index).
        const int tid = threadIdx.x;
        for (int i = tid; i < G::GPB * B; i += DCOMP_BLOCK) { sh.mc_key[i] = ~0ull; sh.mc_u[i] = 0ull; sh.mc_win[i] = ~0u; sh.mc_cnt[i] = 0u; }
        __syncthreads();
        unsigned long long key[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
            key[b] = ~0ull;
            if (bs_mode_of<MP>(p, b) == DCOMP_MAX_CAP && ((conn >> b) & 1u)) {
                double dx = p.bs_x[b] - px, dy = p.bs_y[b] - py;
                key[b] = (unsigned long long)__double_as_longlong(__builtin_fma(dy, dy, dx * dx));
                atomicMin(&sh.mc_key[env_local * B + b], key[b]);
            }
        }
        __syncthreads();
        uint32_t cand = 0;
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (key[b] != ~0ull) {
                const double mn = __longlong_as_double((long long)sh.mc_key[env_local * B + b]);
                if (__longlong_as_double((long long)key[b]) <= mn * (1.0 + 1e-7)) { cand |= 1u << b; atomicAdd(&sh.mc_cnt[env_local * B + b], 1u); }
            }
        }
        __syncthreads();
        uint32_t contested = 0;
#pragma unroll
        for (int b = 0; b < B; b++) if (((cand >> b) & 1u) && sh.mc_cnt[env_local * B + b] > 1u) contested |= 1u << b;
        const bool heavy = __ballot(contested != 0u) != 0ull;                   // wave-uniform, rare
#pragma unroll
        for (int b = 0; b < B; b++) {
            key[b] = 1ull;                                                     // sole contender: any key wins
            if (heavy && ((contested >> b) & 1u)) key[b] = maxcap_rate_key(p.pl_c1, p.pl_c2, px, py, p.bs_x[b], p.bs_y[b]);
            if ((cand >> b) & 1u) atomicMax(&sh.mc_u[env_local * B + b], key[b]);
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < B; b++)
            if (((cand >> b) & 1u) && key[b] == sh.mc_u[env_local * B + b])
                atomicMin(&sh.mc_win[env_local * B + b], ((uint32_t)p.conn_since[(size_t)idx * B + b] << 8) | (uint32_t)u);
        __syncthreads();
#pragma unroll
        for (int b = 0; b < B; b++)
            if (((cand >> b) & 1u) && (sh.mc_win[env_local * B + b] & 0xFFu) == (uint32_t)u) mc_winner |= 1u << b;
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
        // No contraction in here: without the select below the caller's `curr += dr[b]` would see a bare product and the compiler could fuse it
        // into an FMA in one kernel instantiation and not in another (step vs fused rollout: one ulp apart, tests/test_rollout_gpu.py)
#pragma clang fp contract(off)
        const bool c = (conn >> b) & 1u;
        const int mode = bs_mode_of<MP>(p, b);
        float dru = dr[b], out = 0.f;
        if (mode == DCOMP_RES_FAIR) out = dru * fast_rcp(fmaxf(cnt[b], 1.f));                        // station.py:171-173
        else if (mode == DCOMP_RATE_FAIR) out = fast_rcp(agg[b]);                                    // station.py:180
        else if (mode == DCOMP_PROP_FAIR) out = (dru * inv_ewma) * fast_rcp(agg[b] + EPS) * dru;     // station.py:194-195
        else out = ((mc_winner >> b) & 1u) ? dru : 0.f;
        // dru is 0 where the UE is not connected, so the resource-fair and proportional-fair shares are 0 there by themselves (finite
        // factors: rcp(max(cnt, 1)), rcp(agg + eps)); only 1 / sum of a rate-fair station is the same for every lane and needs the select
        dr[b] = (mode == DCOMP_RES_FAIR || mode == DCOMP_PROP_FAIR) ? out : (c ? out : 0.f);
    }
}

// Observation row + reward of one UE.  variants.py:271-305, central.py:31-73, multi_agent.py:32-95.
// l2 / cnt are consumed (overwritten with the observation entries).
// `active`: this lane owns a slot (row) of the env; `alive`: a UE currently sits in that slot (always the same unless
// UEs arrive / depart, then dead slots produce zero rows: central.py:46-55); n_eff = UEs currently in the env.
struct Outs { float *obs, *reward, *sum_util, *ue_dr, *ue_util, *rb_out; uint8_t *next_act; int compact; };   // where this step's outputs go

// The same rules with the dr entries behind a functor (the wide kernel keeps a UE's row in LDS, not in registers).
template <int B, class F>
__device__ __forceinline__ int policy_action_fn(const KParams &p, uint32_t conn, F dr)
{
    float mx = dr(0);
    int best = 0;
#pragma unroll 4
    for (int b = 1; b < B; b++) { const float d = dr(b); if (d > mx) { mx = d; best = b; } }
    if (p.policy == DCOMP_POLICY_3GPP) return ((conn >> best) & 1u) ? 0 : conn ? __builtin_ffs((int)conn) : best + 1;
    uint32_t sel = B == 32 ? ~0u : (1u << (B & 31)) - 1u;
    if (p.policy == DCOMP_POLICY_DYNAMIC) {
        const float thr = mx * p.policy_eps;
        sel = 0;
#pragma unroll 4
        for (int b = 0; b < B; b++) sel |= (dr(b) >= thr ? 1u : 0u) << b;
    } else if (p.policy == DCOMP_POLICY_CLUSTER) { sel = p.policy_cluster[best]; VM_ARRIVED1(sel); }
    const uint32_t drop = conn & ~sel;
    if (drop) return __builtin_ffs((int)drop);
    const uint32_t cand = sel & ~conn;
    float m2 = -__builtin_huge_valf();
    int a = 0;
#pragma unroll 4
    for (int b = 0; b < B; b++) { const float d = dr(b); if (((cand >> b) & 1u) && d > m2) { m2 = d; a = b + 1; } }
    return a;
}

// The reference's heuristic baselines (agent/heuristics.py:13-187) on one UE's dr[] entries and connection mask -- the rules
// and first-maximum ties of heuristic_kernel (dcomp_api.hip), evaluated on the registers write_outputs is about to store.
template <int B>
__device__ __forceinline__ int policy_action(const KParams &p, uint32_t conn, const float (&dr)[B])
{
    float mx = dr[0];
    int best = 0;
#pragma unroll
    for (int b = 1; b < B; b++) if (dr[b] > mx) { mx = dr[b]; best = b; }          // np.argmax: the first maximum
    if (p.policy == DCOMP_POLICY_3GPP)                                              // heuristics.py:30-38
        return ((conn >> best) & 1u) ? 0 : conn ? __builtin_ffs((int)conn) : best + 1;
    uint32_t sel = B == 32 ? ~0u : (1u << (B & 31)) - 1u;                           // FullCoMP: every cell
    if (p.policy == DCOMP_POLICY_DYNAMIC) {                                         // heuristics.py:87-90
        const float thr = mx * p.policy_eps;
        sel = 0;
#pragma unroll
        for (int b = 0; b < B; b++) sel |= (dr[b] >= thr ? 1u : 0u) << b;
    } else if (p.policy == DCOMP_POLICY_CLUSTER) { sel = p.policy_cluster[best]; VM_ARRIVED1(sel); }      // heuristics.py:172-176
    const uint32_t drop = conn & ~sel;
    if (drop) return __builtin_ffs((int)drop);                                      // cells outside the set first, index order
    const uint32_t cand = sel & ~conn;
    float m2 = -__builtin_huge_valf();
    int a = 0;
#pragma unroll
    for (int b = 0; b < B; b++) if (((cand >> b) & 1u) && dr[b] > m2) { m2 = dr[b]; a = b + 1; }   // strongest first
    return a;
}
// STAGED: observation rows go through LDS and leave as linear 16-byte stores (1 KiB contiguous per store instruction) -- what
// a bandwidth-bound launch needs.  false: straight from registers; the fused rollout kernel, used for small batches with
// one or two waves per SIMD, is latency-bound and the LDS round trips cost it 0.7 us per step (2.88 -> 2.21 us at 4 096 x 10 x 5).
// POL: the in-step heuristic policy (o.next_act / pol_next): -1 = decided at run time (one uniform branch), 0 = compiled out,
// 1 = compiled in.  The fused rollout is instantiated both ways: it is latency-bound with one wave per SIMD, and the extra live
// values of the run-time form cost the plain tape-driven rollout 8 % (round 2: 2.15 -> 2.33 us per step at 4 096 x 10 x 5).
// KIND: the env kind (central.py / multi_agent.py layouts and rewards): -1 = read from the argument block, DCOMP_CENTRAL =
// compiled for the central env only.  The latency-bound fused rollout and the tight-packing kernel -- the two that serve the
// small central shapes (BASELINE config 2; 65 536 x 10 x 5) -- have a central-only instantiation: 4 % each, same box A/B.
template <int B, int UPAD, bool RESET, bool DYN = false, bool STAGED = true, class S = SegPadded, int POL = -1, int KIND = -1>
__device__ __forceinline__ void write_outputs(const KParams &p, const Outs &o, BlockSharedT<B, UPAD> &sh, bool active, int env, int env_local, int u, int idx,
                                              int wave, int lane, int gbase, uint32_t conn, uint32_t in_range,
                                              float (&l2)[B], float (&cnt)[B], float util, float curr_dr,
                                              float reward_before, bool alive, int n_eff, const S &sg = S{}, uint32_t *pol_next = nullptr)
{
    using G = Geo<B, UPAD>;
    using SG = StageGeo<B>;
    const int U = p.U;
    const int kind = KIND >= 0 ? KIND : DCOMP_FORCE_KIND >= 0 ? DCOMP_FORCE_KIND : p.kind;
    // per-BS utility aggregates over connected UEs (station.py:63-83)
    float tsum[B];
#pragma unroll
    for (int b = 0; b < B; b++) tsum[b] = ((conn >> b) & 1u) ? util : 0.f;
    if (!RESET && kind == DCOMP_MULTI && !(DCOMP_ABLATE & 16)) {     // central observations carry no per-station utilities
        seg_reduce_vec<G::WG, OpSum, B>(tsum, sg, lane);
        if (G::NW > 1) xwave_reduce_<B, G::NW, OpSum>(tsum, sh, wave, lane);
    }
    float l2max = l2[0];
#pragma unroll
    for (int b = 1; b < B; b++) l2max = fmaxf(l2max, l2[b]);

    // ---- reward
    float reward = 0.f;
    if (kind == DCOMP_CENTRAL) {
        float r[1];
        if (p.reward_agg == DCOMP_REWARD_MIN) {
            r[0] = seg_reduce<G::WG, OpMin>(alive ? reward_before : 1.f, sg, lane);
            xwave_reduce_<1, G::NW, OpMin>(r, sh, wave, lane);
        } else {
            r[0] = seg_reduce<G::WG, OpSum>(alive ? reward_before : 0.f, sg, lane);
            xwave_reduce_<1, G::NW, OpSum>(r, sh, wave, lane);
            if (p.reward_agg == DCOMP_REWARD_AVG) r[0] = r[0] * fast_rcp((float)n_eff);      // (v_rcp_f32, 1 ulp: the IEEE division sequence is 11 instructions)
        }
        reward = r[0];
    } else if (!RESET) {
        reward = util;                                              // multi_agent.py:52 (own utility, NOT normalised)
        if (p.reward_agg == DCOMP_REWARD_SUM) {
            // multi_agent.py:73-79: sum of rewards_before over UEs sharing any BS with this UE
            sh.nb_conn[threadIdx.x] = alive ? conn : 0u;
            sh.nb_rb[threadIdx.x] = reward_before;
            __syncthreads();
            if (in_range != 0) {
                float s = 0.f;
                const int base = (int)threadIdx.x - u;               // first lane of my env (padded and tight groups alike)
                for (int v = 0; v < U; v++) if (sh.nb_conn[base + v] & conn) s += sh.nb_rb[base + v];
                reward = s;
            }
            __syncthreads();
        } else {
            if (p.reward_agg == DCOMP_REWARD_AVG) {                 // multi_agent.py:60-71
                float n = 0.f, t = 0.f;
#pragma unroll
                for (int b = 0; b < B; b++) {                        // (bit -> 0.f / 1.f, two FMAs: exactly the conditional adds, four instructions per station)
                    const float ind = (float)((in_range >> b) & 1u);
                    n = __builtin_fmaf(ind, cnt[b], n);
                    t = __builtin_fmaf(ind, tsum[b], t);
                }
                if (n > 0.f) {                                       // ONE reciprocal (1 ulp) instead of two IEEE divisions (2 x 12 instructions)
                    const bool lone = conn == 0u;
                    reward = (lone ? t + util : t) * fast_rcp(lone ? n + 1.f : n);
                }
            } else {                                                // multi_agent.py:81-85, station.py:78-83
                float tmin[B];
#pragma unroll
                for (int b = 0; b < B; b++) tmin[b] = ((conn >> b) & 1u) ? util : MAX_UTIL;
                seg_reduce_vec<G::WG, OpMin, B>(tmin, sg, lane);
                if (G::NW > 1) xwave_reduce_<B, G::NW, OpMin>(tmin, sh, wave, lane);
                float m = util;
#pragma unroll
                for (int b = 0; b < B; b++) if ((in_range >> b) & 1u) m = fminf(m, cnt[b] > 0.f ? tmin[b] : MAX_UTIL);
                reward = (in_range != 0) ? m : util;
            }
        }
    }

    // ---- info (base.py:383-411)
    if (o.sum_util) {
        float s[1];
        s[0] = seg_reduce<G::WG, OpSum>(alive ? util : 0.f, sg, lane);
        xwave_reduce_<1, G::NW, OpSum>(s, sh, wave, lane);
        if (active && u == 0) o.sum_util[env] = s[0];
    }
    if (active) {
        if (o.ue_dr) stream_store(&o.ue_dr[idx], alive ? curr_dr : 0.f);
        if (o.ue_util) stream_store(&o.ue_util[idx], alive ? util : 0.f);
        if (o.rb_out) stream_store(&o.rb_out[idx], alive ? reward_before : 0.f);
    }

    // ---- observation entries (in place): l2 -> snr_b / max snr, tsum -> avg utility at BS, cnt -> UEs at BS / U
    const float inv_u = fast_rcp((float)n_eff);
    // DYN (UE lists that change, and reset of such envs): dead slots produce zero rows.  Otherwise every row that is stored
    // belongs to a live UE (alive == active), so the entries need no select.
    const bool live = DYN ? alive : true;
    const float util_n = live ? util * (1.0f / MAX_UTIL) : 0.f;
    // DYN + compact record: the per-env columns are stored ONCE per env, by the lane of the last SLOT -- which may be unlisted, so
    // that lane needs the env's values, not its own row's zeros (every lane of an env holds the same sums)
    float env_cols[DYN ? 2 * B : 1];
#pragma unroll
    for (int b = 0; b < B; b++) {
        l2[b] = live ? fast_exp2(l2[b] - l2max) : 0.f;                                          // variants.py:276-284
        // avg utility of the UEs at b, 0 for an idle BS (its sum is 0): variants.py:299, station.py:71-76
        const float avg = tsum[b] * fast_rcp(fmaxf(cnt[b], 1.f)) * (1.0f / MAX_UTIL);
        if (DYN) { env_cols[b] = cnt[b] * inv_u; env_cols[B + b] = avg; }
        tsum[b] = live ? avg : 0.f;
        cnt[b] = live ? cnt[b] * inv_u : 0.f;                                                   // variants.py:296
    }
    if (POL != 0 && (POL == 1 || o.next_act)) {                     // uniform: heuristic policy on the entries just made
        const int a = policy_action<B>(p, conn, l2);
        if (active && o.next_act) o.next_act[idx] = (uint8_t)(live ? a : 0);
        if (pol_next) *pol_next = live ? (uint32_t)a : 0u;
    }
    if ((DCOMP_ABLATE & 8) && kind == DCOMP_MULTI) {
        float acc = util_n + reward;
        for (int b = 0; b < B; b++) acc += l2[b] + cnt[b] + tsum[b];
        if (active && acc == 123456.f) o.obs[idx] = acc;         // keeps the producers alive, writes nothing
    } else if (kind == DCOMP_MULTI && !STAGED) {
        if (active && o.compact) {                              // (uniform) the compact record instead of the row, straight from registers
            if (o.reward) o.reward[idx] = alive ? reward : 0.f;
            float *rec = o.obs + (size_t)idx * (B + 2) + (size_t)env * (2 * B);     // word U (B + 2) env + (B + 2) u + 2B env
            float recv[B + 2];
#pragma unroll
            for (int b = 0; b < B; b++) recv[b] = l2[b];
            recv[B] = util_n;
            recv[B + 1] = __uint_as_float(conn);
            store_run<B + 2>(rec, recv);
            if (u == U - 1) {                                   // the per-env columns once, behind the last UE's record
                float tl[2 * B];
#pragma unroll
                for (int b = 0; b < B; b++) { tl[b] = cnt[b]; tl[B + b] = tsum[b]; }
                store_run<2 * B>(rec + B + 2, tl);
            }
        } else if (active) {
            if (o.reward) o.reward[idx] = alive ? reward : 0.f;
            float rowv[SG::ROW];
#pragma unroll
            for (int b = 0; b < B; b++) {
                rowv[b] = (float)((conn >> b) & 1u);
                rowv[B + b] = l2[b];
                rowv[2 * B + b] = cnt[b];
                rowv[3 * B + b] = tsum[b];
            }
            rowv[4 * B] = util_n;
            store_run<SG::ROW>(o.obs + (size_t)idx * SG::ROW, rowv);       // the lane's 4B+1 consecutive floats
        }
    } else if (kind == DCOMP_MULTI && o.compact) {
        // The compact record (dcomp_fragment.h: per UE dr[B] | utility | connection mask, then ues_at_bs[B] | util_at_bs[B] once
        // per env) INSTEAD of the rows -- a third of the store traffic, and no pack pass for a learner hand-off.  The records of a
        // wave's rows are one contiguous span (the per-env columns sit between the envs' records): staged in LDS as they lie in
        // memory and copied out linearly, as the rows are; spans larger than the staging buffer go in windows.
        if (active && o.reward) stream_store(&o.reward[idx], alive ? reward : 0.f);
        const unsigned long long am = __ballot(active);
        if (am != 0ull) {
            constexpr int CW = B + 2;
            const int first = __ffsll((long long)am) - 1, last = 63 - __clzll((long long)am);
            const int idx0 = __builtin_amdgcn_readlane(idx, first), env0 = __builtin_amdgcn_readlane(env, first);
            const size_t g0 = (size_t)idx0 * CW + (size_t)env0 * (2 * B);           // first word of the span
            const int loc = (idx - idx0) * CW + (env - env0) * (2 * B);             // my record inside it
            const bool tail = u == U - 1;                                           // the last UE's lane adds the per-env columns
            const int n = __builtin_amdgcn_readlane(loc + CW + (tail ? 2 * B : 0), last);
            const int ph = (int)((((size_t)o.obs >> 2) + g0) & 3);                  // 16-byte phase of the span start
            uint32_t *st = reinterpret_cast<uint32_t *>(sh.stage[wave]);
            constexpr int CH = (SG::WORDS - 4) & ~3;                                // window size (multiple of 4: constant phase)
            const bool one = n <= CH;                                               // the usual case: everything in one window
#pragma unroll 1
            for (int w0 = 0; w0 < n; w0 += CH) {
                const int cw = min(CH, n - w0);
                if (active) {
                    uint32_t *rec = st + ph - w0 + loc;
                    const int at = loc - w0;
#pragma unroll
                    for (int b = 0; b < B; b++) if (one || (unsigned)(at + b) < (unsigned)cw) rec[b] = __float_as_uint(l2[b]);
                    if (one || (unsigned)(at + B) < (unsigned)cw) rec[B] = __float_as_uint(util_n);
                    if (one || (unsigned)(at + B + 1) < (unsigned)cw) rec[B + 1] = conn;
                    if (tail) {
#pragma unroll
                        for (int b = 0; b < B; b++) {
                            if (one || (unsigned)(at + CW + b) < (unsigned)cw) rec[CW + b] = __float_as_uint(DYN ? env_cols[b] : cnt[b]);
                            if (one || (unsigned)(at + CW + B + b) < (unsigned)cw) rec[CW + B + b] = __float_as_uint(DYN ? env_cols[B + b] : tsum[b]);
                        }
                    }
                }
                wave_lds_fence();
                const int n_end = ph + cw;
                uint32_t *gbase_ptr = reinterpret_cast<uint32_t *>(o.obs) + g0 + w0 - ph;      // 16-byte aligned
                for (int j = lane * 4; j < n_end; j += 256) {
                    if (j >= ph && j + 4 <= n_end) {
                        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#if DCOMP_NT_OBS
                        __builtin_nontemporal_store(*reinterpret_cast<const u4v *>(st + j), reinterpret_cast<u4v *>(gbase_ptr + j));
#else
                        *reinterpret_cast<u4v *>(gbase_ptr + j) = *reinterpret_cast<const u4v *>(st + j);
#endif
                    } else {
                        for (int k = max(j, ph); k < min(j + 4, n_end); k++) gbase_ptr[k] = st[k];
                    }
                }
                wave_lds_fence();
            }
        }
    } else if (kind == DCOMP_MULTI) {
        if (active && o.reward) stream_store(&o.reward[idx], alive ? reward : 0.f);
        // rows of this wave are contiguous in memory: [row0, row0 + nrows)
        const unsigned long long am = __ballot(active);
        const int nrows = group_popcount<64>(am, 0);
        const int row0 = __builtin_amdgcn_readlane(idx, am ? __ffsll((long long)am) - 1 : 0);   // uniform source lane: no LDS round trip
        const int r = idx - row0;
        float *st = sh.stage[wave];
#pragma unroll 1
        for (int pass = 0; pass < SG::NPASS; pass++) {
            const int rbase = pass * SG::RPP;
            if (rbase >= nrows) break;
            const int rows = min(SG::RPP, nrows - rbase);
            const size_t g0 = (size_t)(row0 + rbase) * SG::ROW;            // first float of this pass in obs
            const int ph = (int)((((size_t)o.obs >> 2) + g0) & 3);         // 16-byte phase of that address
            if (active && r >= rbase && r < rbase + rows) {
                float *row = st + ph + (r - rbase) * SG::ROW;              // word stride 4B+1 is odd: conflict-free
#pragma unroll
                for (int b = 0; b < B; b++) {
                    row[b] = (float)((conn >> b) & 1u);
                    row[B + b] = l2[b];
                    row[2 * B + b] = cnt[b];
                    row[3 * B + b] = tsum[b];
                }
                row[4 * B] = util_n;
            }
            wave_lds_fence();
            const int n_end = ph + rows * SG::ROW;                         // staged floats live in st[ph, n_end)
            float *gbase_ptr = o.obs + g0 - ph;                            // 16-byte aligned
            for (int j = lane * 4; j < n_end; j += 256) {
                if (j >= ph && j + 4 <= n_end) {
#if DCOMP_NT_OBS
                    // observation rows are write-once streams nobody in this kernel re-reads: non-temporal stores
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(*reinterpret_cast<const f4v *>(st + j), reinterpret_cast<f4v *>(gbase_ptr + j));
#else
                    *reinterpret_cast<float4 *>(gbase_ptr + j) = *reinterpret_cast<const float4 *>(st + j);
#endif
                } else {
                    for (int k = max(j, ph); k < min(j + 4, n_end); k++) gbase_ptr[k] = st[k];
                }
            }
            wave_lds_fence();
        }
    } else if (G::NW == 1 && STAGED && DCOMP_CENTRAL_STAGED) {
        // Central layout: the records of the envs of one wave are contiguous in memory (REC floats each), so the wave stages
        // them in LDS exactly as they lie in memory and copies them out linearly -- every store instruction covers 1 KiB
        // instead of 64 scattered 4-byte pieces (round 1: 41 % of the HBM peak at 65 536 x 10 x 5).  Lane u owns B
        // consecutive floats of the `connected` and of the `dr` block (LDS word stride B between lanes: conflict-free for
        // odd B, 2-way -- free for ds_write_b32 -- for B = 2 mod 4).  Regions larger than the staging buffer go in windows.
        if (active && o.reward && u == 0) o.reward[env] = reward;
        const unsigned long long am = __ballot(active);
        if (am != 0ull) {
            const int REC = U * (2 * B + 1);
            const int env0 = __builtin_amdgcn_readlane(env, __ffsll((long long)am) - 1);
            const int envl = __builtin_amdgcn_readlane(env, 63 - __clzll((long long)am));
            const int n = (envl - env0 + 1) * REC;                             // floats this wave produces
            const size_t g0 = (size_t)env0 * REC;
            const int ph = (int)((((size_t)o.obs >> 2) + g0) & 3);             // 16-byte phase of the region start
            float *st = sh.stage[wave];
            constexpr int CH = (SG::WORDS - 4) & ~3;                           // window size (multiple of 4: constant phase)
            const int o_conn = (env - env0) * REC + u * B, o_dr = o_conn + U * B, o_ut = (env - env0) * REC + 2 * U * B + u;
#pragma unroll 1
            for (int w0 = 0; w0 < n; w0 += CH) {
                const int cnt = min(CH, n - w0);
                if (active) {
                    float *row = st + ph - w0;
                    const bool one = n <= CH;                                  // the usual case: everything in one window
#pragma unroll
                    for (int b = 0; b < B; b++) {
                        if (one || (unsigned)(o_conn + b - w0) < (unsigned)cnt) row[o_conn + b] = (float)((conn >> b) & 1u);
                        if (one || (unsigned)(o_dr + b - w0) < (unsigned)cnt) row[o_dr + b] = l2[b];
                    }
                    if (one || (unsigned)(o_ut - w0) < (unsigned)cnt) row[o_ut] = util_n;
                }
                wave_lds_fence();
                const int n_end = ph + cnt;
                float *gbase_ptr = o.obs + g0 + w0 - ph;                       // 16-byte aligned
                for (int j = lane * 4; j < n_end; j += 256) {
                    if (j >= ph && j + 4 <= n_end) {
#if DCOMP_NT_OBS
                        typedef float f4v __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(*reinterpret_cast<const f4v *>(st + j), reinterpret_cast<f4v *>(gbase_ptr + j));
#else
                        *reinterpret_cast<float4 *>(gbase_ptr + j) = *reinterpret_cast<const float4 *>(st + j);
#endif
                    } else {
                        for (int k = max(j, ph); k < min(j + 4, n_end); k++) gbase_ptr[k] = st[k];
                    }
                }
                wave_lds_fence();
            }
        }
    } else if (active) {                                                       // envs wider than a wavefront: direct stores
        if (o.reward && u == 0) o.reward[env] = reward;
        float *base = o.obs + (size_t)env * U * (2 * B + 1);
        float cv[B];
#pragma unroll
        for (int b = 0; b < B; b++) cv[b] = (float)((conn >> b) & 1u);
        store_run<B>(base + u * B, cv);                                        // the lane's B consecutive floats of each block
        store_run<B>(base + U * B + u * B, l2);
        base[2 * U * B + u] = util_n;
    }
}

// Start position and first movement draw of the UE in slot u (user.py:98-116, movement.py:110-122): what MobileEnv.reset
// (base.py:169-189) does per UE.  Used by reset_kernel and by step_kernel's in-kernel reset at the horizon.
__device__ __forceinline__ void reset_ue(const KParams &p, int env, int u, uint32_t episode, double &px, double &py, unsigned long long &mv)
{
    const UeCfg c = p.ue_cfg[u];
    int x, y;
    if (p.rng_mode == DCOMP_RNG_TAPE) { const size_t t = (size_t)env * p.U0 + u; x = p.tape_pos0[2 * t]; y = p.tape_pos0[2 * t + 1]; }
    else {
        uint32_t r[4];
        philox4x32_10(p.env_base + (uint32_t)env, (uint32_t)u, episode, 0u, p.seed_lo, p.seed_hi, r);
        x = (int)__umulhi(r[0], (uint32_t)p.map_w + 1u);
        y = (int)__umulhi(r[1], (uint32_t)p.map_h + 1u);
    }
    if (c.init_x >= 0) x = c.init_x;
    if (c.init_y >= 0) y = c.init_y;
    px = (double)x; py = (double)y;
    uint32_t vel, wx, wy;
    draw_triple(p, env, (uint32_t)u + 1u, 0u, episode, vel, wx, wy, mv_cfg_pack(c.vel_lo, c.vel_hi, c.pause, c.border));
    mv = mv_pack(wx, wy, vel, 0u, 0u, 1u);
}

__device__ __forceinline__ void store_state(const KParams &p, int idx, double px, double py, unsigned long long mv, uint32_t conn, float ewma)
{
#if DCOMP_NT_STATE & 2
    typedef double d2v __attribute__((ext_vector_type(2)));
    d2v q; q.x = px; q.y = py;
    __builtin_nontemporal_store(q, reinterpret_cast<d2v *>(p.pos) + idx);
    __builtin_nontemporal_store(mv, p.mv + idx);
    __builtin_nontemporal_store(conn, p.conn + idx);
    __builtin_nontemporal_store(ewma, p.ewma + idx);
#else
    p.pos[idx] = make_double2(px, py);
    p.mv[idx] = mv;
    p.conn[idx] = conn;
    p.ewma[idx] = ewma;
#endif
}

// One MobileEnv.step (base.py:413-466) of the UE in this lane: toggle -> rates -> move -> drop -> EWMA -> rates -> outputs.
// The UE state is passed by reference and stays in registers; `emit` (uniform): write observation / reward / info to `o`.
// STORE: write the state back BEFORE the outputs (plain step) -- position, movement word and EWMA are then dead while the
// observation rows are staged, which is worth a wave per SIMD (73 vs 92 VGPRs at B = 10).
// Pairs of a UE at its current position, carried from one step of the fused rollout to the next: a UE stands where the
// previous step's move left it, so the post-move evaluation of step t (log2 snr of every station, in-range mask, "a lane of this
// wave is near a station") IS the pre-move evaluation of step t + 1 -- the rollout kernel keeps it in registers instead of
// redoing B pair evaluations (FP64 distance, log2) per UE and step.  A plain step cannot: the values would have to go through HBM.
template <int B>
struct PairCarry { float l2[B]; float dru[B]; uint32_t in_range; bool near; };   // dru: unshared rate per station (shared_rates CARRY)

template <int B, int UPAD, int MP, bool STORE, class S, int POL = -1, bool CARRY = false, int KIND = -1>
__device__ __forceinline__ void step_once(const KParams &p, BlockSharedT<B, UPAD> &sh, const Outs &o, bool emit, bool active, int env,
                                          int env_local, int u, int idx, int wave, int lane, int gbase, uint32_t act, uint32_t time,
                                          uint32_t episode, bool step_util, float dr_req, int vrange, double &px, double &py,
                                          unsigned long long &mv, uint32_t &conn, float &ewma, const S &sg, uint32_t *pol_next = nullptr,
                                          const double *bsx = nullptr, const double *bsy = nullptr, PairCarry<B> *carry = nullptr,
                                          int env_shift = 0, int idx_shift = 0)
{
    // env_shift / idx_shift (uniform): where this step's outputs go in [T][...] fragment buffers -- the fused rollout writes step t
    // at env + t * E / idx + t * E * U of the SAME base pointers (no per-step pointer arithmetic on seven 64-bit scalars)
    float l2[B], dr[B], cnt[B];
    uint32_t in_range = B >= 32 ? 0xFFFFFFFFu : ((1u << (B & 31)) - 1u);
    bool near_pre = false, near_post = false;
    if (act > (uint32_t)B) { atomicOr(p.flags, DCOMP_FLAG_BAD_ACTION); act = 0; }
    // The pre-move position only matters where the UE is connected (rate before the move, base.py:446) and at the one station it
    // acts on (in range -> may connect, user.py:203-222): each lane walks its OWN set bits -- per-lane station index, BS position
    // from the LDS table -- and parks the unshared rate in its row of the (still idle) observation staging buffer; shared_rates
    // picks it up where the UE is connected.  Trip count = the largest need-set in the wave (3-4) instead of B dense pair +
    // rate evaluations per lane.  Where the row does not fit the staging buffer (B > 11) or a max-cap BS needs the aliased
    // scratch: the dense form.
    // Only in the plain step (STORE): the fused rollout is latency-bound, one wave per SIMD, and the LDS look-ups of this pass
    // cost it 7 % (2.16 -> 2.32 us per step at 4 096 x 10 x 5); and only from 7 stations up (B = 5: 0.5 % slower).
    constexpr bool SPARSE_OK = DCOMP_SPARSE_PRE && STORE && B >= DCOMP_SPARSE_MIN_B && !(DCOMP_ABLATE & 33) && 64 * (B + 1) <= StageGeo<B>::WORDS;
    if (SPARSE_OK && !(MP == MP_GENERIC && p.any_maxcap)) {
        float *const prow = sh.stage[wave] + lane * (B + 1);
        const uint32_t act_bit = act ? 1u << (act - 1u) : 0u;
        uint32_t need = active ? (conn | act_bit) : 0u, inr_old = 0u;
        while (__ballot(need != 0u) != 0ull) {
            if (need != 0u) {
                const int b = __ffs((int)need) - 1;
                need &= need - 1u;
                const double2 bp = sh.bs[b];
                bool ir, near;
                float l;
                pair_eval(px, py, bp.x, bp.y, p, ir, l, near);
                if (near) {                                        // rare, per lane: within 1.26 m of the station
                    const double dx = bp.x - px, dy = bp.y - py;
                    if ((float)__builtin_fma(dy, dy, dx * dx) < 1e-20f) l = pair_eval_tiny(px, py, bp.x, bp.y, p);
                }
                inr_old |= (uint32_t)ir << b;
                bool big;
                float dru = rate_unshared_small(l, big);
                if (big) dru = rate_unshared_any(l);               // rare: snr > 1/64
                prow[b] = dru;
            }
        }
        conn ^= act_bit & (conn | inr_old);                        // toggle (base.py:247-263 -> user.py:190-222); generic pattern: no max-cap BS here
        shared_rates<B, UPAD, MP, S, true>(p, sh, conn, l2, ewma, px, py, u, idx, env_local, wave, lane, gbase, dr, cnt, 0, sg, prow);
    } else {
        // 1. pairs at the pre-move position
        if (CARRY) {                                  // = the previous step's post-move pairs (fused rollout)
#pragma unroll
            for (int b = 0; b < B; b++) l2[b] = carry->l2[b];
            in_range = carry->in_range; near_pre = carry->near;
        } else if (!(DCOMP_ABLATE & 32)) in_range = eval_pairs<B>(px, py, p, l2, &near_pre, bsx, bsy);
        else { for (int b = 0; b < B; b++) l2[b] = -20.f; }
        // 2. toggle (base.py:247-263 -> user.py:190-222)
        if (act > 0) {
            const uint32_t bit = 1u << (act - 1);
            if (conn & bit) conn &= ~bit;
            else if (in_range & bit) {
                conn |= bit;
                if (MP == MP_GENERIC && (p.maxcap_mask & bit)) p.conn_since[(size_t)idx * B + (act - 1)] = (uint16_t)time;
            }
        }
        // 3. rates before the move (base.py:446) -> reward_before (base.py:158-167)
        if (CARRY) shared_rates<B, UPAD, MP, S, false, 2>(p, sh, conn, l2, ewma, px, py, u, idx, env_local, wave, lane, gbase, dr, cnt, (int)near_pre, sg, nullptr, carry->dru);
        else if (!(DCOMP_ABLATE & 1)) shared_rates<B, UPAD, MP, S>(p, sh, conn, l2, ewma, px, py, u, idx, env_local, wave, lane, gbase, dr, cnt, (int)near_pre, sg);
        else { for (int b = 0; b < B; b++) { dr[b] = 1.f; cnt[b] = 1.f; } }
    }
    float curr = 0.f;
#pragma unroll
    for (int b = 0; b < B; b++) curr += dr[b];
    const float util_pre = ue_utility(curr, step_util, dr_req);
    const float reward_before = clamp_med3(util_pre, MIN_UTIL, MAX_UTIL) * (1.0f / MAX_UTIL);
    // 4. move (base.py:447 -> user.py:159-173)
    if (active && !(DCOMP_ABLATE & 2)) {
        move_ue<!STORE>(p, env, (uint32_t)u + 1u, episode, px, py, mv, vrange);       // !STORE = the fused rollout
        // movement.py:165-166: inside [0, W] x [0, H].  Non-negative doubles order like their bit patterns, and a negative one (or a NaN) has
        // the top bit set: two unsigned 64-bit compares instead of four FP64 compares (px / py cannot be -0.0: a UE lands on integer waypoints
        // as (double)int, and x + (-0.0) = x otherwise)
        if ((unsigned long long)__double_as_longlong(px) > (unsigned long long)__double_as_longlong((double)p.map_w) ||
            (unsigned long long)__double_as_longlong(py) > (unsigned long long)__double_as_longlong((double)p.map_h)) atomicOr(p.flags, DCOMP_FLAG_OUTSIDE_MAP);
    }
    // 5. pairs at the new position; drop lost connections (user.py:175-188); EWMA from the stale rates (user.py:148-157)
    if (!(DCOMP_ABLATE & 64)) in_range = eval_pairs<B>(px, py, p, l2, &near_post, bsx, bsy);
    if (CARRY) {
#pragma unroll
        for (int b = 0; b < B; b++) carry->l2[b] = l2[b];
        carry->in_range = in_range; carry->near = near_post;
    }
    conn &= in_range;
    float stale = 0.f;
#pragma unroll
    for (int b = 0; b < B; b++)                        // (bit -> 0 / -1 with one v_bfe_i32, AND, add: the select form costs two instructions more per station)
        stale += __int_as_float(__float_as_int(dr[b]) & __builtin_amdgcn_sbfe((int)conn, b, 1));
    ewma = __builtin_fmaf(0.9f, stale, 0.1f * ewma);   // one explicit contraction: every kernel variant rounds alike
    // 6. rates after the move (base.py:451)
    if (CARRY) shared_rates<B, UPAD, MP, S, false, 1>(p, sh, conn, l2, ewma, px, py, u, idx, env_local, wave, lane, gbase, dr, cnt, (int)near_post, sg, nullptr, carry->dru);
    else if (!(DCOMP_ABLATE & 4)) shared_rates<B, UPAD, MP, S>(p, sh, conn, l2, ewma, px, py, u, idx, env_local, wave, lane, gbase, dr, cnt, (int)near_post, sg);
    curr = 0.f;
#pragma unroll
    for (int b = 0; b < B; b++) curr += dr[b];
    const float util = ue_utility(curr, step_util, dr_req);
    if (STORE && active) store_state(p, idx, px, py, mv, conn, ewma);
    // 7. observation, reward, info
    if (emit) {
        Outs o2 = o;                                  // next_action is ONE [E][U] buffer, not a [T][...] fragment: undo the shift for it
        if (POL != 0 && o2.next_act) o2.next_act -= idx_shift;
        write_outputs<B, UPAD, false, false, STORE, S, POL, KIND>(p, o2, sh, active, env + env_shift, env_local, u, idx + idx_shift, wave, lane, gbase, conn,
                                                         in_range, l2, cnt, util, curr, reward_before, active, p.U, sg, pol_next);
    }
}

// MobileEnv.step for all envs: one launch = one step.  ROLLOUT = true is the fused rollout: p.num_steps consecutive steps
// in ONE launch, the UE state (position, movement word, connections, EWMA) staying in registers in between -- no kernel
// boundary, no state round trip through HBM, no host launch per step (the loop this replaces: simulation.py:512-541).
// Two instantiations on purpose: with the step loop around it the compiler hoists the kernel-argument loads (BS table) out
// of the loop and spills them (73 -> 163 VGPRs, 0.081 -> 0.099 ms at config 3), so the plain step keeps its loop-free code.
// A uniform value the compiler must keep in a VECTOR register: the asm hides that every lane holds the same bits.  The fused
// rollout's step loop wants ~110 scalar registers (kernel-argument pointers and constants, the BS table, loop counters, saved
// exec masks of the divergent branches) and has 102: the compiler spilled 93 of them into VGPR lanes and paid a v_readlane /
// v_writelane for every use inside the loop.  The kernel runs at <= 4 waves per SIMD, where VGPRs are free up to 128.
__device__ __forceinline__ double in_vgpr(double v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Workgroup b runs on XCD b mod 8 (round-robin dispatch).  With `slot = b` the eight XCDs write eight interleaved combs of the
// observation buffer; with this map every XCD owns ONE contiguous eighth of the slots.  Nothing is re-read, so no L2 hit rate
// changes -- but the stream of dirty lines each XCD's L2 sends to memory is contiguous, and beyond the 256 MB Infinity Cache that is
// worth 15-19 % of the sustained write rate (tools/micro/store_patterns.hip, pattern 7 vs 5: 448.8 -> 377.5 us for 2.2 GB).
__device__ __forceinline__ int xcd_contiguous_block()
{
#if DCOMP_XCD_REMAP
    const unsigned b = blockIdx.x, n = gridDim.x, q = n >> 3, r = n & 7u, x = b & 7u;
    return (int)(x * q + (x < r ? x : r) + (b >> 3));
#else
    return (int)blockIdx.x;
#endif
}

template <int B, int UPAD, int MP, bool ROLLOUT, bool TIGHT = false, int POL = -1, int KIND = -1>
__device__ __forceinline__ void step_kernel_body(const KParams &p, BlockSharedT<B, UPAD> &sh)
{
    using G = Geo<B, UPAD>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int env_local = tid / UPAD, u = tid % UPAD;
    const int blk = ROLLOUT ? (int)blockIdx.x : xcd_contiguous_block();
    int env = blk * G::GPB + env_local;
    bool active = (env < p.E) && (u < p.U);
    int gbase = lane & ~(G::WG - 1);
    using S = SegT<TIGHT>;
    S sg{};
    if constexpr (TIGHT) {                                         // tight packing: G = U lanes per env, 64 / G envs per wavefront
        const int g = p.tight_g, gi = (lane * p.tight_magic) >> 16; // gi = lane / g
        u = lane - gi * g;
        env_local = wave * p.tight_gpw + gi;
        env = (blk * (DCOMP_BLOCK / 64) + wave) * p.tight_gpw + gi;
        active = gi < p.tight_gpw && env < p.E;
        gbase = gi * g;
        const unsigned long long m = gi < p.tight_gpw ? ((1ull << g) - 1ull) << gbase : 0ull;
        sg = S{g, u, (gbase + g - 1) * 4, (uint32_t)m, (uint32_t)(m >> 32)};
    }
    const int idx = env * p.U + u;
    if (DCOMP_SPARSE_PRE && !ROLLOUT && B >= DCOMP_SPARSE_MIN_B && 64 * (B + 1) <= StageGeo<B>::WORDS) {   // BS table for the sparse pre-move pass
        if (tid < B) sh.bs[tid] = make_double2(p.bs_x[tid], p.bs_y[tid]);
        __syncthreads();
    }

#if DCOMP_BS_IN_LDS
#pragma unroll
    for (int b = 0; b < B; b++) if (tid == b) { g_bs_lds[b] = p.bs_x[b]; g_bs_lds[DCOMP_MASK32_MAX_BS + b] = p.bs_y[b]; }
    __syncthreads();
#endif
    double px = 0.0, py = 0.0;
    unsigned long long mv = 0;
    uint32_t conn = 0, act = 0;
    float ewma = 0.f;
    bool step_util = false;
    float dr_req = 1.f;
    int vrange = MV_CFG_ARRIVED;
    if (active) {
#if DCOMP_NT_STATE & 1
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v q = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p.pos) + idx);
        px = q.x; py = q.y;
        mv = __builtin_nontemporal_load(p.mv + idx);
        conn = __builtin_nontemporal_load(p.conn + idx);
        ewma = __builtin_nontemporal_load(p.ewma + idx);
#else
        double2 q = p.pos[idx];
        px = q.x; py = q.y;
        mv = p.mv[idx];
        conn = p.conn[idx];
        ewma = p.ewma[idx];
#endif
        if (!ROLLOUT) act = p.action[idx];
        {
            const UeCfg c = p.ue_cfg[u];                                           // loaded here, next to the state
            step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req;
            vrange = mv_cfg_pack(c.vel_lo, c.vel_hi, c.pause, c.border);
        }
    }
    Outs o{p.obs, p.reward, p.sum_util, p.ue_dr, p.ue_util, p.rb_out, POL == 0 ? nullptr : p.next_act, p.obs_compact};
    if (!ROLLOUT) {
        step_once<B, UPAD, MP, true, S, -1, false, KIND>(p, sh, o, true, active, env, env_local, u, idx, wave, lane, gbase, act, p.time, p.episode,
                                                         step_util, dr_req, vrange, px, py, mv, conn, ewma, sg);
    } else {
        const int T = p.num_steps;
        const size_t EU = (size_t)p.E * p.U;
        uint32_t time = p.time, episode = p.episode;
        double bsx[B], bsy[B];                                     // the BS table in VGPRs (see in_vgpr)
#pragma unroll
        for (int b = 0; b < B; b++) { bsx[b] = in_vgpr(p.bs_x[b]); bsy[b] = in_vgpr(p.bs_y[b]); }
        PairCarry<B> carry;
        auto refresh_carry = [&]() {                               // pairs + unshared rates at the position the state holds
            carry.in_range = eval_pairs<B>(px, py, p, carry.l2, &carry.near, bsx, bsy);
#pragma unroll
            for (int b = 0; b < B; b++) {
                bool f;
                carry.dru[b] = rate_unshared_small(carry.l2[b], f);
                if (f) carry.dru[b] = rate_unshared_any(carry.l2[b]);       // (per lane; once per launch / episode)
            }
        };
        refresh_carry();
        // Actions: 8 steps' bytes per lane are fetched at a time, one chunk ahead.  A load per step would make every step
        // wait for its own HBM round trip -- and, vmcnt being one in-order counter for loads and stores, for the previous
        // step's observation stores to drain.
        const uint8_t *actp = p.action + (active ? idx : 0);      // idle lanes read a valid byte, masked below: no exec-masked blocks
        const bool ploop = POL != 0 && p.policy_loop;               // closed loop: steps 1..T-1 act on the policy's own decision
        const int TT = ploop ? 1 : T;                              // ... the tape is actions[0] only
        uint32_t pol_next = 0u;                                    // the registered policy's action on the observation of step t-1
        uint32_t nb[8];                                            // the NEXT chunk's bytes, still in flight
        auto load_chunk = [&](int t0) {                            // 8 independent loads (steps past T re-read the last one)
#pragma unroll
            for (int j = 0; j < 8; j++) nb[j] = (uint32_t)actp[(size_t)min(t0 + j, TT - 1) * EU];
        };
        load_chunk(0);
        unsigned long long act_cur = 0ull;
        int env_shift = 0, idx_shift = 0;                          // (dcomp_rollout_ex keeps T * E * U below 2^31)
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            if (!DCOMP_EXP_NO_RESET && p.horizon > 0 && time == (uint32_t)p.horizon) {    // RLlib's horizon (env_setup.py:281): reset(), then step
                episode += p.episode_inc;
                time = 0;
                conn = 0u; ewma = 0.f;
                if (active) reset_ue(p, env, u, episode, px, py, mv);
                refresh_carry();
            }
            if ((t & 7) == 0) {                                    // first use of the bytes requested 8 steps ago; request the next 8
                const uint32_t lo = nb[0] | (nb[1] << 8) | (nb[2] << 16) | (nb[3] << 24), hi = nb[4] | (nb[5] << 8) | (nb[6] << 16) | (nb[7] << 24);
                act_cur = ((unsigned long long)hi << 32) | lo;
                load_chunk(t + 8);
            }
            act = active ? (uint32_t)(act_cur >> (8 * (t & 7))) & 0xFFu : 0u;
            if (ploop && t > 0) act = active ? pol_next : 0u;
            step_once<B, UPAD, MP, false, S, POL, true, KIND>(p, sh, o, p.out_every_step || t == T - 1 || ploop, active, env, env_local, u, idx, wave, lane,
                                                        gbase, act, time, episode, step_util, dr_req, vrange, px, py, mv, conn, ewma, sg,
                                                        POL != 0 ? &pol_next : nullptr, bsx, bsy, &carry, env_shift, idx_shift);
            time += 1;
            if (p.out_every_step) { env_shift += p.E; idx_shift += (int)EU; }   // outputs of step t + 1 -> the next slice of the [T][...] buffers
            // the max-cap and 'sum'-reward scratch of the next step aliases the observation staging other waves may still be copying out
            if ((MP == MP_GENERIC && p.any_maxcap) || ((KIND >= 0 ? KIND : p.kind) == DCOMP_MULTI && p.reward_agg == DCOMP_REWARD_SUM)) __syncthreads();
        }
    }
    if (ROLLOUT && active) store_state(p, idx, px, py, mv, conn, ewma);
}

#ifdef DCOMP_MINW
#define DCOMP_STEP_BOUNDS __launch_bounds__(DCOMP_BLOCK, DCOMP_MINW)
#else
#define DCOMP_STEP_BOUNDS __launch_bounds__(DCOMP_BLOCK)
#endif
template <int B, int UPAD, int MP>
__global__ DCOMP_STEP_BOUNDS void step_kernel(const KParams p)
{
    __shared__ BlockSharedT<B, UPAD> sh;
    step_kernel_body<B, UPAD, MP, false>(p, sh);
}

// step_kernel with envs packed tightly (see SegT): only where dcomp_create ever picks it -- UE lists of 5, 9, 10, 17-21.
template <int B, int UPAD, int MP, int KIND = -1>
__global__ __launch_bounds__(DCOMP_BLOCK) void step_kernel_tight(const KParams p)
{
    __shared__ BlockSharedT<B, UPAD> sh;
    step_kernel_body<B, UPAD, MP, false, true, -1, KIND>(p, sh);
}

// POL = 0: the action-tape rollout (no policy code in it); POL = 1: with a registered heuristic policy -- next_action of every
// emitted step and, with policy_loop, the closed loop act = policy(obs); step(act) inside the launch.
template <int B, int UPAD, int MP, int POL, int KIND = -1>
__global__ __launch_bounds__(DCOMP_BLOCK) void rollout_kernel(const KParams p)
{
    __shared__ BlockSharedT<B, UPAD> sh;
    step_kernel_body<B, UPAD, MP, true, false, POL, KIND>(p, sh);
}

// The tape-driven central rollout with envs packed tightly (SegT<true>): for the batches dcomp_create packs tightly -- many waves
// per SIMD, where the fused kernel is bound by its VALU work and 37.5 % of a padded 10-UE wave is idle lanes.
template <int B, int UPAD, int MP>
__global__ __launch_bounds__(DCOMP_BLOCK) void rollout_kernel_tight(const KParams p)
{
    __shared__ BlockSharedT<B, UPAD> sh;
    step_kernel_body<B, UPAD, MP, true, true, 0, DCOMP_CENTRAL>(p, sh);
}

// MobileEnv.reset (base.py:169-189): user.py:98-116 + movement.py:110-122 + first observation.
template <int B, int UPAD>
__global__ __launch_bounds__(DCOMP_BLOCK) void reset_kernel(const KParams p)
{
    using G = Geo<B, UPAD>;
    __shared__ BlockSharedT<B, UPAD> sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env_local = tid / UPAD, u = tid % UPAD;
    const int env = blockIdx.x * G::GPB + env_local;
    const bool active = (env < p.E) && (u < p.U);
    const int idx = env * p.U + u;
    const int gbase = lane & ~(G::WG - 1);

    const bool alive = active && u < p.U0;            // slots beyond the initial ue_list are empty after reset (base.py:177-182)
    double px = 0.0, py = 0.0;
    bool step_util = false;
    float dr_req = 1.f;
    if (alive) {
        UeCfg c = p.ue_cfg[u];
        step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req;
        unsigned long long mv;
        reset_ue(p, env, u, p.episode, px, py, mv);
        p.pos[idx] = make_double2(px, py);
        p.mv[idx] = mv;
        p.conn[idx] = 0u;
        p.ewma[idx] = 0.f;
        if (p.uid) p.uid[idx] = (uint16_t)(u + 1);
        if (p.orig_consumed) p.orig_consumed[(size_t)env * p.U0 + u] = 0xFFFFu;
    } else if (active) {
        p.pos[idx] = make_double2(0.0, 0.0);
        p.mv[idx] = 0ull;
        p.conn[idx] = 0u;
        p.ewma[idx] = 0.f;
        if (p.uid) p.uid[idx] = 0;
    }
    float l2[B], cnt[B];
    const uint32_t in_range = eval_pairs<B>(px, py, p, l2);
#pragma unroll
    for (int b = 0; b < B; b++) cnt[b] = 0.f;
    const float util = ue_utility(0.f, step_util, dr_req);
    const Outs o{p.obs, p.reward, p.sum_util, p.ue_dr, p.ue_util, p.rb_out, p.next_act, p.obs_compact};
    write_outputs<B, UPAD, true, true>(p, o, sh, active, env, env_local, u, idx, wave, lane, gbase, 0u, in_range, l2, cnt, util, 0.f, 0.f, alive,
                                 p.U0);
}

}  // namespace dcomp
#include "dcomp_wide.h"
#include "dcomp_dyn.h"
namespace dcomp {

using KernelFn = void (*)(const KParams);
// step_wide: organisation for envs of >= 64 lanes (dcomp_wide.h); nullptr for narrower envs.  It has no max-cap
// path, the host falls back to `step` when a BS is max-cap.
struct KernelPair { KernelFn step, reset, step_wide, step_dyn, rollout, step_tight, rollout_pol, rollout_central, tight_central, rollout_tight_central; };   // step_dyn: UEs arrive / depart (dcomp_dyn.h), UPAD <= 64

template <int B, int UPAD, int MP>
inline KernelFn wide_or_null()
{
    // measured on MI355X: the chunked organisation only pays once the unrolled per-BS register arrays of
    // step_kernel no longer fit (B > DCOMP_WIDE_MIN_B); below that step_kernel is faster
    if constexpr (UPAD >= 64 && B > DCOMP_WIDE_MIN_B) return step_kernel_wide<B, UPAD, MP>;
    else return nullptr;
}

template <int B, int UPAD, int MP, int POL = 0, int KIND = -1>
inline KernelFn rollout_or_null()
{
    // The fused rollout serves small batches (dcomp_create: <= 4 waves per SIMD).  Not instantiated where it is never or hardly
    // ever picked -- shapes the wide kernel takes over, envs of more than one wavefront -- those are also the costly ones to
    // build; dcomp_rollout then launches the step kernel once per step (same results).
    if constexpr (UPAD > 64 || (UPAD >= 64 && B > DCOMP_WIDE_MIN_B)) return nullptr;
    else return rollout_kernel<B, UPAD, MP, POL, KIND>;
}

template <int B, int UPAD, int MP, int KIND = -1>
inline KernelFn tight_or_null()
{
    if constexpr (UPAD == 8 || UPAD == 16 || UPAD == 32) return step_kernel_tight<B, UPAD, MP, KIND>;
    else return nullptr;
}

template <int B, int UPAD, int MP>
inline KernelFn rollout_tight_or_null()
{
    // only where dcomp_create takes long central rollouts to the fused kernel at any batch size (B <= 8) and packs tightly
    if constexpr ((UPAD == 8 || UPAD == 16 || UPAD == 32) && B <= 8) return rollout_kernel_tight<B, UPAD, MP>;
    else return nullptr;
}

template <int B, int UPAD>
inline KernelFn dyn_or_null()
{
    return step_kernel_dyn<B, UPAD>;
}

template <int B, int UPAD>
inline KernelPair make_pair_(int mp)
{
    // Envs of >= 64 lanes with more than DCOMP_WIDE_MIN_B stations run the wide kernel; the narrow step kernel only serves them
    // when a BS is max-cap (the wide kernel has no max-cap path) -- always the generic sharing pattern.  The two specialised
    // narrow variants would never be launched there and are the most expensive instantiations of the build: left out.
    if constexpr (UPAD >= 64 && B > DCOMP_WIDE_MIN_B) {
        const KernelFn w = mp == MP_RES_FAIR ? wide_or_null<B, UPAD, MP_RES_FAIR>() : mp == MP_MIXED ? wide_or_null<B, UPAD, MP_MIXED>() : wide_or_null<B, UPAD, MP_GENERIC>();
        return KernelPair{step_kernel<B, UPAD, MP_GENERIC>, reset_kernel<B, UPAD>, w, dyn_or_null<B, UPAD>(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    } else {
    if (mp == MP_RES_FAIR) return KernelPair{step_kernel<B, UPAD, MP_RES_FAIR>, reset_kernel<B, UPAD>, wide_or_null<B, UPAD, MP_RES_FAIR>(), dyn_or_null<B, UPAD>(), rollout_or_null<B, UPAD, MP_GENERIC>(), tight_or_null<B, UPAD, MP_GENERIC>(), rollout_or_null<B, UPAD, MP_GENERIC, 1>(), rollout_or_null<B, UPAD, MP_GENERIC, 0, DCOMP_CENTRAL>(), tight_or_null<B, UPAD, MP_GENERIC, DCOMP_CENTRAL>(), rollout_tight_or_null<B, UPAD, MP_GENERIC>()};   // (all-resource-fair: the generic variants, run-time modes, serve the fused rollout and the tight packing -- build time)
    if (mp == MP_MIXED) return KernelPair{step_kernel<B, UPAD, MP_MIXED>, reset_kernel<B, UPAD>, wide_or_null<B, UPAD, MP_MIXED>(), dyn_or_null<B, UPAD>(), rollout_or_null<B, UPAD, MP_MIXED>(), tight_or_null<B, UPAD, MP_MIXED>(), rollout_or_null<B, UPAD, MP_MIXED, 1>(), rollout_or_null<B, UPAD, MP_MIXED, 0, DCOMP_CENTRAL>(), tight_or_null<B, UPAD, MP_MIXED, DCOMP_CENTRAL>(), rollout_tight_or_null<B, UPAD, MP_MIXED>()};
    return KernelPair{step_kernel<B, UPAD, MP_GENERIC>, reset_kernel<B, UPAD>, wide_or_null<B, UPAD, MP_GENERIC>(), dyn_or_null<B, UPAD>(), rollout_or_null<B, UPAD, MP_GENERIC>(), tight_or_null<B, UPAD, MP_GENERIC>(), rollout_or_null<B, UPAD, MP_GENERIC, 1>(), rollout_or_null<B, UPAD, MP_GENERIC, 0, DCOMP_CENTRAL>(), tight_or_null<B, UPAD, MP_GENERIC, DCOMP_CENTRAL>(), rollout_tight_or_null<B, UPAD, MP_GENERIC>()};
    }
}

// One translation unit per B instantiates all UPAD widths (dcomp_inst_bXX.hip).
template <int B>
inline KernelPair kernels_for_upad(int upad, int mp)
{
#ifdef DCOMP_ONLY_UPAD
    return make_pair_<B, DCOMP_ONLY_UPAD>(mp);      // build-time probe
#else
    switch (upad) {
    case 1: case 2: case 4: return make_pair_<B, 4>(mp);
    case 8: return make_pair_<B, 8>(mp);
    case 16: return make_pair_<B, 16>(mp);
    case 32: return make_pair_<B, 32>(mp);
    case 64: return make_pair_<B, 64>(mp);
    case 128: return make_pair_<B, 128>(mp);
    case 256: return make_pair_<B, 256>(mp);
    default: return KernelPair{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    }
#endif
}

}  // namespace dcomp
