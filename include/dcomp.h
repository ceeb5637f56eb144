/*
 * dcomp.h -- C ABI of the MI355X-native DeepCoMP environment step (libdcomp_hip.so).
 *
 * The reference (CN-UPB/DeepCoMP) has no FFI: its boundary is the Python class protocol
 * gym.Env / ray.rllib MultiAgentEnv  --  __init__(env_config), reset(), step(action), seed()
 * (deepcomp/env/single_ue/base.py:27,132,169,413; multi_ue/central.py:9-73; multi_ue/multi_agent.py:6-107).
 * This header is the native layer *underneath* that protocol; deepcomp_amd/env.py keeps the Python
 * surface and calls these entry points through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Every array named "device" is a HIP device pointer allocated by the caller (PyTorch);
 *     the library borrows it for the duration of one call and never allocates per step.
 *   - `stream` is the caller's hipStream_t passed as void* (NULL = default stream). Calls only
 *     enqueue work; nothing synchronises except dcomp_check().
 *   - Return value: DCOMP_OK (0) or a negative DCOMP_E* code; dcomp_last_error() gives the text.
 *     Nothing throws across the ABI.
 *   - One handle <-> one stream at a time; handles are independent; no global mutable state.
 *
 * Lane/data layout (E envs, U UEs, B base stations; idx = env*U + ue):
 *   pos   double[E*U][2]   UE position x,y (FP64: connect/drop decisions are bit-exact vs the reference)
 *   mv    uint64[E*U]      waypoint x:16 | y:16 | velocity:8 | pausing:1 (bit 47) + curr_pause:7 (bits 40-46) | draw cursor:16
 *   conn  uint32[E*U]      bit b set <=> UE connected to BS b                  (user.py:34 bs_dr keys)
 *   conn_hi uint32[E*U]    the same for stations 32 ... 63; required for envs of the generic kernel (csrc/dcomp_big.h): more than 32 stations
 *                          or more than 256 UE slots (dcomp_needs_conn_hi) -- rollouts there are one launch per step (no fused rollout kernel)
 *   ewma  float [E*U]      exponentially weighted average rate                 (user.py:148-157)
 *   flags uint32[4]        sticky device-side error bits, read by dcomp_check()
 *   With UE arrival / departure (cfg.max_ues >= cfg.num_ue; 0 = fixed list) every per-UE array has max_ues slots per env; slot =
 *   position in the reference's env.ue_list; uid uint16[E*max_ues] holds the UE ids.
 *   conn_since uint16[E*U][B]   step at which a connection was made -- only when some BS is max-cap (oldest
 *                          connection wins rate ties, station.py:184-186); NULL otherwise
 *   An episode has at most 65536 steps (16-bit draw cursor and conn_since); dcomp_step returns DCOMP_EUNSUPPORTED beyond.
 */
#ifndef DCOMP_H
#define DCOMP_H

#include <stddef.h>
#include <stdint.h>

#include "dcomp_types.h"      /* limits, enums, dcomp_cfg / dcomp_state / dcomp_out / dcomp_tape / dcomp_events */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ABI guard.  The structs of dcomp_types.h grow at their END from version to version (version 2 added dcomp_out.obs_compact, version 3 dcomp_state.conn_hi):
 * a caller compiled against an older header would make the library read past its struct.  So the handle is created through
 * dcomp_create_v, which takes the caller's idea of the ABI -- DCOMP_ABI_VERSION and the sizes of the four structs the library
 * reads through caller pointers -- and refuses (DCOMP_EABI; dcomp_last_error() names both sides) unless all of them are the
 * library's own.  C / C++ callers keep writing dcomp_create(cfg, &env): the macro below passes the values of THIS header.  FFI
 * callers (ctypes, cgo, JNI) call dcomp_create_v with the sizes of THEIR struct declarations.  The plain
 * `dcomp_create` symbol stays exported for one reason: a binary built against a version-1 header calls it, and gets DCOMP_EABI
 * instead of undefined behaviour.  Every struct passed to the library must be zero-initialised before its fields are set (a field
 * the caller does not know about is then NULL = "not requested"). */
#define DCOMP_ABI_VERSION 3
#define DCOMP_EABI (-7)        /* caller and library disagree about the ABI version or a struct size */
int dcomp_abi_version(void);                                             /* the library's DCOMP_ABI_VERSION */
int dcomp_create_v(int32_t abi_version, size_t cfg_size, size_t state_size, size_t out_size, size_t rollout_opts_size,
                   const dcomp_cfg *cfg, dcomp_env **out);               /* MobileEnv.__init__  base.py:27-84 */
int dcomp_create(const dcomp_cfg *cfg, dcomp_env **out);                 /* version-1 entry point: always DCOMP_EABI (see above) */
int dcomp_destroy(dcomp_env *env);
int dcomp_state_sizes(const dcomp_env *env, size_t *pos_bytes, size_t *mv_bytes, size_t *conn_bytes,
                      size_t *ewma_bytes, size_t *flags_bytes, size_t *since_bytes);
int dcomp_obs_dim(const dcomp_env *env, int32_t *floats_per_env, int32_t *reward_per_env);

/* MobileEnv.reset (base.py:169-189): draw start positions + first waypoint, clear connections and
 * EWMA, time = 0, write the first observation.  `tape` may be NULL in Philox mode. */
int dcomp_reset(dcomp_env *env, const dcomp_state *st, const dcomp_tape *tape, const dcomp_out *out, void *stream);

/* MobileEnv.step (base.py:413-466) for all E envs: action[E][U] uint8 in [0, B] (0 = no-op,
 * k = toggle BS k-1; base.py:259-263).  One fused kernel: toggle -> rates -> move -> drop -> EWMA ->
 * rates -> obs/reward. */
int dcomp_step(dcomp_env *env, const dcomp_state *st, const uint8_t *action, const dcomp_out *out, void *stream);

/* dcomp_step with this step's UE departures / arrivals (MobileEnv.step incl. base.py:433-443, add_new_ue / remove_ue
 * base.py:592-618).  Needs cfg.max_ues >= cfg.num_ue (> 0) and state.uid.  ev may be NULL (no event). */
int dcomp_step_dyn(dcomp_env *env, const dcomp_state *st, const uint8_t *action, const dcomp_out *out,
                   const dcomp_events *ev, void *stream);
int dcomp_num_ue(const dcomp_env *env);               /* UEs currently in every env's list */

/* T consecutive steps from an action tape actions[T][E][U] with no host work in between; outputs of the LAST step only.
 * Replaces the per-step Python loop of simulation.py:512-541.  The narrow kernel (dcomp_rollout_is_fused() == 1) runs all T
 * steps in ONE launch with the UE state in registers in between; envs of >= 64 lanes with more than 20 BSs and envs with UE
 * arrival / departure are launched once per step.  Fused: batches of up to 3 waves per SIMD, and -- at any batch size when
 * num_steps >= 4 -- central envs of <= 8 stations and multi-agent envs of <= 3 stations (short observation rows).  Connection masks, positions and movement state are identical to T dcomp_step calls
 * either way.  A batch that dcomp_step packs tightly (dcomp_lanes_per_env: UE lists of 5, 9, 10, 17-21 in batches of >= 4 096
 * waves; per-env sums in scan order instead of butterfly order) is packed the same way by the fused kernel of its tape-driven
 * central rollouts: same summation order as its dcomp_step; with a registered policy (dcomp_set_policy) the fused kernel keeps
 * the padded groups and the floats may differ from dcomp_step's in the last bits (<= 2e-6 relative). */
int dcomp_rollout(dcomp_env *env, const dcomp_state *st, const uint8_t *actions, int32_t num_steps,
                  const dcomp_out *out, void *stream);

/* dcomp_rollout with options.
 *   every_step          1: out->obs, out->reward and the info tensors are [T][...] buffers that receive the outputs of
 *                       every step (a rollout fragment for the learner); 0: outputs of the last step only
 *   horizon             > 0: whenever env.time has reached it the envs are reset before the next step, inside the rollout --
 *                       what RLlib does at its `horizon` = episode_length (env_setup.py:281); the first observation of
 *                       the new episode is not emitted (policy_loop: it is written to out->obs of a last-step-only rollout and
 *                       overwritten by the next step).  0: never (then an episode has at most 65536 steps)
 *   new_episode_draws   Philox mode: 1 = every such reset starts the next episode's draws (rand_episodes = True),
 *                       0 = the same episode again (the reference re-seeds at reset, base.py:171-173).  Tape mode replays the
 *                       tape handed to the last dcomp_reset, so it must be 0 there.
 *   policy_loop         1: closed loop with the policy registered through dcomp_set_policy -- step 0 takes actions[0][E][U]
 *                       (the next_action the previous launch wrote), every later step the policy's decision on the
 *                       observation of the step before, taken from registers: a heuristic agent's whole evaluation run
 *                       (simulation.py:512-541) in one call; `actions` holds ONE step.  With a horizon the run may cross episode
 *                       boundaries: at each one the library launches the reset kernel -- which writes the first observation of
 *                       the new episode AND the policy's action on it -- and continues the loop from that action: one launch
 *                       per stretch of an episode plus one reset launch per episode, no host work in between.  Where rollouts
 *                       are not fused (dcomp_rollout_fused_for: wide and generic kernels, UE arrival / departure) the same loop is
 *                       one launch per step, each step reading the next_action buffer the launch before it wrote (round 6).
 *   ev_n_remove/ev_n_add  UE departures / arrivals of an env with a changing UE list (cfg.max_ues > 0), per step of THIS
 *                       rollout: host arrays [num_steps]; entry t = the UEs that leave / arrive in step t (base.py:433-443; the
 *                       schedule is configuration, identical in every env).  NULL: no events.  Such envs are launched once
 *                       per step (dcomp_step_dyn's kernel); with a horizon the schedule entries are indexed by rollout step,
 *                       so the caller lays the episode's schedule out again after every reset.
 *   ev_remove_idx/ev_add_xy  tape mode only (NULL with Philox draws): the host-drawn list positions / border points of those
 *                       events (see dcomp_events), device arrays, the blocks [E][n_remove] resp. [E][n_add][2] of the steps
 *                       that have events concatenated in step order. */
typedef struct dcomp_rollout_opts {
    int32_t every_step;
    int32_t horizon;
    int32_t new_episode_draws;
    int32_t policy_loop;
    const int32_t *ev_n_remove;
    const int32_t *ev_n_add;
    const int32_t *ev_remove_idx;
    const int32_t *ev_add_xy;
} dcomp_rollout_opts;
int dcomp_rollout_ex(dcomp_env *env, const dcomp_state *st, const uint8_t *actions, int32_t num_steps,
                     const dcomp_out *out, const dcomp_rollout_opts *opts, void *stream);
int dcomp_rollout_is_fused(const dcomp_env *env);      /* 1: T steps = one launch */
/* Whether a rollout of num_steps steps with these options is ONE launch: fusion of the short-row shapes depends on the number of
 * steps (>= 4, or a policy loop), and an every-step fragment of >= 2^31 rows (num_steps * num_envs * num_ue) takes the
 * one-launch-per-step path (same results).  1 / 0; -1: bad arguments. */
int dcomp_rollout_fused_for(const dcomp_env *env, int32_t num_steps, int32_t every_step, int32_t policy_loop);
int dcomp_needs_conn_hi(const dcomp_env *env);         /* 1: the env runs on the generic kernel (more than 32 stations, more than 256 UE slots, or
                                                         * DCOMP_FORCE_BIG) and dcomp_state.conn_hi (stations 32-63 of every connection set) must be
                                                         * given; 0: it is ignored.  Ask after dcomp_create_v instead of re-deriving the rule. */
int dcomp_lanes_per_env(const dcomp_env *env);         /* lanes an env occupies in dcomp_step: next power of two >= num_ue, or
                                                         * num_ue itself when envs are packed tightly (throughput-bound batches of
                                                         * UE lists that are not a power of two long; DCOMP_TIGHT=0/1 overrides) */

/* Name of the kernel instantiation dcomp_step launches for this env, as rocprofv3 prints it (e.g. "step_kernel<10, 32, 2>" =
 * <num_bs, lanes per env, sharing pattern>): measurement tooling ties a tracked profile to what really runs. */
int dcomp_step_kernel_name(const dcomp_env *env, char *buf, int32_t len);

/* Synchronises `stream`, reads the sticky flags and maps them to DCOMP_EACTION / DCOMP_ETAPE /
 * DCOMP_EPOS (the reference raises AssertionError in these cases).  Clears the flags. */
int dcomp_check(dcomp_env *env, const dcomp_state *st, void *stream);

int dcomp_time(const dcomp_env *env);                 /* env.time (base.py:39) -- lock-step over the batch */
int64_t dcomp_episode(const dcomp_env *env);          /* resets so far - 1 (Philox counter word) */
int dcomp_set_episode(dcomp_env *env, int64_t episode);

/* MobileEnv.seed (base.py:132-143) for counter-based draws: a new Philox key.  Takes effect with the next draw; callers
 * normally follow it with dcomp_set_episode(env, 0) + dcomp_reset (the gym convention env.seed(s); env.reset()). */
int dcomp_set_seed(dcomp_env *env, uint64_t seed);

/* Tape mode: replace the borrowed draw tape of the episode in progress by a LONGER one holding the same draws (first
 * `depth_old` triples of every stream identical) -- for episodes that outlive the tape they were started with: the
 * reference's done() is always None (base.py:371-381) and --cont-train never resets (main.py:48-51).  The caller
 * synchronises the stream first (steps in flight read the old tape). */
int dcomp_set_tape(dcomp_env *env, const dcomp_tape *tape, int32_t depth);

/* Checkpoint / resume of an env batch (the reference never checkpoints env state, simulation.py:143-147 restores the
 * learner only; with counter-based draws the state tensors plus these five host-side counters are the whole env):
 * {time, episode, UEs currently listed, departures so far this episode, arrivals so far this episode}. */
int dcomp_get_counters(const dcomp_env *env, int64_t out[5]);
int dcomp_set_counters(dcomp_env *env, const int64_t in[5]);

/* Host-side helper for tape mode: CPython-compatible Mersenne Twister draws
 * (random.Random(seed).randint, user.py:94-109 / movement.py:110-130) for UEs
 * [0,U) of `num_envs` envs whose base seeds are seeds[e]; UE i uses seeds[e] + 100*(i+1)
 * (base.py:138-143).  Writes host arrays pos0[E*U][2], triples[E*U][depth][4]. */
int dcomp_mt_draw_tape(const dcomp_cfg *cfg, const int64_t *seeds, int32_t num_envs, int32_t depth,
                       int32_t *pos0, uint16_t *triples);

double dcomp_connect_threshold(void);                 /* smallest double d with snr(d) <= 2e-8 (station.py:10,222-226) */
/* X = the smallest double q with sqrt_rn(q) >= dcomp_connect_threshold(): the reference decides can_connect as
 * snr(sqrt(dx*dx + dy*dy)) > 2e-8 (station.py:122-127, 222-226), i.e. q < X with q = fl(fl(dx*dx) + fl(dy*dy)).  The kernels compare
 * exactly that (round 6; X is one ulp BELOW fl(d_T * d_T) for the reference's constants).  -1: the host's libm makes snr non-monotone
 * around d_T (dcomp_create_v then fails with DCOMP_EUNSUPPORTED). */
double dcomp_connect_boundary_sq(void);
const char *dcomp_last_error(void);
const char *dcomp_version(void);

/* Device self-test of the cross-lane reductions and the FP64 sqrt/div/fma used by the movement
 * step: fills out[n] (device) from x[n], y[n] (device doubles).  op: 0 sqrt(x) 1 x/y 2 fma(y,y,x*x)
 * 3 segmented all-reduce sums of (float)x over groups of `width` lanes; 4 / 5 / 6: norm, x/norm, y/norm of the
 * vector (x, y) as the movement step computes them (one shared reciprocal; must equal sqrt and two divisions). */
/* The reference's heuristic baselines (deepcomp/agent/heuristics.py) for every (env, UE) of a batch in one launch, reading
 * the packed observation tensor dcomp_reset / dcomp_step wrote and writing the action tensor dcomp_step takes: a
 * heuristic-driven rollout never leaves the device.  One decision per UE and step, as in the reference:
 *   DCOMP_POLICY_3GPP     heuristics.py:13-38   at most one connection, to the strongest cell: stay | drop the other | connect
 *   DCOMP_POLICY_FULLCOMP heuristics.py:41-65   connect to every cell in range, strongest unconnected cell first
 *   DCOMP_POLICY_DYNAMIC  heuristics.py:68-106  the set {b: dr_b >= epsilon * max dr}: first drop connected cells outside the
 *                                               set (index order), then connect inside it, strongest first
 *   DCOMP_POLICY_CLUSTER  heuristics.py:109-187 the same with the set = static cluster of the strongest cell;
 *                                               cluster_mask[b] (device, uint32 x num_bs): bit o set = cell o in b's cluster
 * Ties as numpy / sorted() resolve them: the first (lowest-index) maximum.  obs_kind = layout of `obs` (DCOMP_MULTI:
 * [E][num_ue][4B+1], DCOMP_CENTRAL: [E][num_ue(2B+1)]); UE slots >= num_active (zero-padded observations of a dynamic
 * env, central.py:46-55) get action 0.  action: uint8 [E][num_ue]. */
typedef struct dcomp_policy {
    int32_t policy;              /* DCOMP_POLICY_* */
    int32_t obs_kind;            /* DCOMP_CENTRAL | DCOMP_MULTI */
    int32_t num_envs, num_ue, num_bs;
    int32_t num_active;          /* UEs listed (<= num_ue) */
    float epsilon;               /* DCOMP_POLICY_DYNAMIC (heuristics.py:75) */
    const uint32_t *cluster_mask;/* DCOMP_POLICY_CLUSTER, device [num_bs]: bit o of word b = station o is in b's cluster; with more than 32
                                  * stations [num_bs][2] = {stations 0-31, stations 32-63} per station */
} dcomp_policy;
int dcomp_heuristic_actions(const dcomp_policy *p, const float *obs, uint8_t *action, void *stream);

/* The same rules INSIDE the step: after dcomp_set_policy every dcomp_reset / dcomp_step / dcomp_step_dyn / dcomp_rollout launch
 * also writes next_action[E][num_ue] = the policy's action on the observation that launch writes (from the registers the
 * observation is stored from: no second pass over the tensor), i.e. what dcomp_heuristic_actions would return on out->obs.
 * A heuristic-driven loop is then `dcomp_step(env, st, next_action, out, s)` over and over (next_action may be the buffer
 * the step reads its actions from: a lane reads its slot before it writes it -- except with UE arrival / departure, where
 * slots shift: use two buffers there).  p: policy / epsilon / cluster_mask are read (cluster_mask must stay valid), the
 * shape fields must be 0 or match the env; p == NULL or next_action == NULL switches it off.
 * Every step kernel has it (narrow, tight, wide, dynamic, and since round 6 the generic kernel of more than 32 stations / 256 UE slots: the rules
 * on the row's station lanes, argmax / sets / candidates as lane masks); DCOMP_EUNSUPPORTED is reserved for kernels that might not. */
int dcomp_set_policy(dcomp_env *env, const dcomp_policy *p, uint8_t *next_action);

/* Compact rollout fragments for the learner hand-off (the RCCL all-gather of rollouts; in the reference the sample
 * batches travel through Ray's object store, util/env_setup.py:266, util/simulation.py:143).  The multi-agent observation row
 * of RelNormEnv.get_ue_obs (single_ue/variants.py:271-305) is connected[B] | dr[B] | ues_at_bs[B] | util_at_bs[B] | utility;
 * ues_at_bs / util_at_bs are properties of the ENV replicated into every UE's row (variants.py:296-299) and `connected` is B
 * bits (variants.py:273).  Compact record of one env-step, dcomp_fragment_words(U, B) = U (B + 2) + 2B 32-bit words:
 *     U x { dr[B] f32 | utility f32 | connected bit mask u32 }  then  ues_at_bs[B] f32 | util_at_bs[B] f32
 * (1 616 B instead of 5 248 B at 32 x 10; 17 664 B instead of 66 048 B at 128 x 32).
 *   dcomp_pack_fragment    obs: device rows [num_env_steps][num_ue][4B+1] (any number of steps x envs, e.g. a [T][E][U][4B+1]
 *                          fragment) -> packed: device words [num_env_steps][dcomp_fragment_words].  flags: device int32[1],
 *                          OR-ed with 1 if a listed row's per-env columns differ from row 0's, 2 if a `connected` entry is
 *                          neither 0 nor 1 -- i.e. if the input is not an observation tensor and the record would not be
 *                          lossless.  Zero it before, read it whenever convenient.
 *   dcomp_unpack_fragment  the inverse.  unpack(pack(obs)) is BIT-IDENTICAL to obs when flags stayed 0: floats are copied, not
 *                          recomputed; rows of unlisted UE slots (all zeros, UE arrival / departure) are recognised by their
 *                          all-zero dr block (a listed UE's best station has dr == 1, variants.py:279-284).
 * Both only enqueue one streaming kernel on `stream`.
 * Multi-agent envs can skip the rows altogether: with dcomp_out.obs_compact set (and obs NULL) dcomp_reset / dcomp_step /
 * dcomp_step_dyn / dcomp_rollout_ex write this record themselves, word for word what dcomp_pack_fragment makes of the rows the same call
 * would have written (a third of the step's store traffic and no pack pass; [T][E][words] with every_step). */
int dcomp_fragment_words(int32_t num_ue, int32_t num_bs);       /* words per env-step; -1: bad arguments */
int dcomp_pack_fragment(const float *obs, int64_t num_env_steps, int32_t num_ue, int32_t num_bs, uint32_t *packed, int32_t *flags,
                        void *stream);
int dcomp_unpack_fragment(const uint32_t *packed, int64_t num_env_steps, int32_t num_ue, int32_t num_bs, float *obs, void *stream);

int dcomp_selftest(int op, int width, const double *x, const double *y, double *out, int64_t n, void *stream);

#ifndef DCOMP_BUILDING_LIBRARY
/* callers compiled against this header create their handles through the guarded entry point (see "ABI guard" above) */
#define dcomp_create(cfg, out) \
    dcomp_create_v(DCOMP_ABI_VERSION, sizeof(dcomp_cfg), sizeof(dcomp_state), sizeof(dcomp_out), sizeof(dcomp_rollout_opts), (cfg), (out))
#endif

#ifdef __cplusplus
}
#endif
#endif /* DCOMP_H */
