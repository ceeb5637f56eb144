/*
 * dcomp_types.h -- limits, enums and plain structs of the C ABI (include/dcomp.h declares the entry points and documents them).
 * Split off so that the device translation units, which need only these definitions, are not recompiled (10 minutes for the 32
 * station counts) whenever an entry point is added to dcomp.h.
 */
#ifndef DCOMP_TYPES_H
#define DCOMP_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCOMP_MAX_BS 64          /* the connection set of a UE is two 32-bit words: state.conn (stations 0-31), state.conn_hi (32-63) */
#define DCOMP_MASK32_MAX_BS 32   /* up to here `conn` alone holds it and the specialised kernels run (one instantiation per station count);
                                  * 33 ... 64 stations take the generic kernel of csrc/dcomp_big.h and need state.conn_hi */
#define DCOMP_MAX_UE 1024        /* an env is ONE workgroup (its per-station sums meet in one LDS): 1 024 lanes at most */
#define DCOMP_SPECIAL_MAX_UE 256 /* up to here the specialised kernels (256-lane workgroups); 257 ... 1 024 UEs per env take the generic kernel of
                                  * csrc/dcomp_big.h with a 512- / 1 024-lane workgroup (round 6: 88 bytes of LDS per lane + tables, whatever the
                                  * station count -- every num_bs <= 64 goes with every num_ue <= 1 024) */

enum { DCOMP_OK = 0, DCOMP_EINVAL = -1, DCOMP_EHIP = -2, DCOMP_EACTION = -3, DCOMP_ETAPE = -4, DCOMP_EPOS = -5,
       DCOMP_EUNSUPPORTED = -6 };

enum { DCOMP_CENTRAL = 0, DCOMP_MULTI = 1 };                    /* central.py:143-152 | multi_agent.py:6 */
enum { DCOMP_REWARD_AVG = 0, DCOMP_REWARD_SUM = 1, DCOMP_REWARD_MIN = 2 };   /* constants.py:24 */
enum { DCOMP_RES_FAIR = 0, DCOMP_RATE_FAIR = 1, DCOMP_MAX_CAP = 2, DCOMP_PROP_FAIR = 3 };  /* station.py:152-202 */
enum { DCOMP_UTIL_LOG = 0, DCOMP_UTIL_STEP = 1 };                /* utility.py:23-54 */
enum { DCOMP_RNG_TAPE = 0, DCOMP_RNG_PHILOX = 1 };

/* device-side sticky flag bits (flags[0]) */
#define DCOMP_FLAG_BAD_ACTION   1u   /* action outside [0, B]        (base.py:238, central.py:61 assert) */
#define DCOMP_FLAG_TAPE_EMPTY   2u   /* waypoint tape exhausted                                           */
#define DCOMP_FLAG_OUTSIDE_MAP  4u   /* UE left the map              (movement.py:165-166 assert)         */

typedef struct dcomp_env dcomp_env;

/* Immutable per-handle configuration.  Replaces the objects inside the reference's env_config dict
 * (env_setup.py:247-256): Map -> map_w/map_h (int()-truncated, map.py:20-21); bs_list -> bs_x/bs_y/
 * bs_sharing; ue_list -> ue_*; 'reward' -> reward_agg; 'seed' -> seed; 'episode_length'. */
typedef struct dcomp_cfg {
    int32_t num_envs;            /* E: envs owned by this handle (one GPU's shard) */
    int32_t num_ue;              /* U <= DCOMP_MAX_UE: UEs in the configured ue_list (= after every reset); the reference has no limit (base.py:79-84) */
    int32_t num_bs;              /* B <= DCOMP_MAX_BS (the reference has no limit: station.py:16-30) */
    int32_t map_w, map_h;
    int32_t env_kind;            /* DCOMP_CENTRAL | DCOMP_MULTI */
    int32_t reward_agg;          /* DCOMP_REWARD_* */
    int32_t rng_mode;            /* DCOMP_RNG_TAPE (reference-exact draws supplied by the host) | DCOMP_RNG_PHILOX */
    int32_t tape_depth;          /* movement triples per UE per episode in tape mode */
    int32_t device;              /* HIP device ordinal */
    int32_t max_ues;             /* slots per env when UEs arrive / depart (base.py:79-84), >= num_ue; 0 = fixed list */
    uint64_t seed;               /* Philox key */
    int64_t env_id_base;         /* global id of this shard's env 0 (results do not depend on the GPU count) */
    const double *bs_x, *bs_y;   /* host [B] */
    const int32_t *bs_sharing;   /* host [B] DCOMP_*_FAIR / MAX_CAP */
    const int32_t *ue_util;      /* host [U] DCOMP_UTIL_* or NULL (= log) */
    const float *ue_dr_req;      /* host [U] or NULL (= 1) -- step utility only (user.py:33) */
    const int32_t *ue_vel_lo, *ue_vel_hi;  /* host [U] inclusive velocity draw range; lo==hi: fixed (movement.py:112-117) */
    const int32_t *ue_init_x, *ue_init_y;  /* host [U] fixed start coordinate or -1 = 'random' (user.py:98-109); NULL = random */
    const int32_t *ue_pause_duration;      /* host [U] RandomWaypoint.pause_duration, 0..127 (movement.py:87,172-176); NULL = 2 */
    const int32_t *ue_border_buffer;       /* host [U] RandomWaypoint.border_buffer, 1..255 (movement.py:87,126-127); NULL = 10.
                                            * UEs that arrive during an episode always get the defaults (base.py:597-599). */
    const double *ue_velocity;             /* host [U] or NULL: fixed velocity of UE u as a number when it is not an integer in
                                            * 0..255 (movement.py:116-117 takes whatever the caller passed, e.g. 2.5), >= 0;
                                            * negative / NaN = use the integer range above.  Such a UE never draws a velocity. */
} dcomp_cfg;

typedef struct dcomp_state {     /* device, caller-allocated; sizes via dcomp_state_sizes() */
    double *pos;
    uint64_t *mv;
    uint32_t *conn;
    float *ewma;
    uint32_t *flags;
    uint16_t *conn_since;        /* NULL unless dcomp_state_sizes() reports since_bytes > 0 */
    uint16_t *uid;               /* [E*max_ues] UE id per slot, bit 15 = arrived during the episode; only with max_ues > 0 */
    uint16_t *orig_consumed;     /* [E*num_ue] optional: movement triples an initial UE had consumed when it left the
                                  * list (0xFFFF = never left) -- lets a tape-mode host continue that UE's stream */
    uint32_t *conn_hi;           /* [E*U] like conn: bit b set <=> connected to station 32 + b.  Required when num_bs > DCOMP_MASK32_MAX_BS or
                                  * num_ue > DCOMP_SPECIAL_MAX_UE (generic kernel), ignored (may be NULL) otherwise.  ABI version 3. */
} dcomp_state;

/* Outputs of reset()/step().  obs layout = RLlib's flatten order of the reference's Dict spaces
 * (sorted keys; variants.py:255-269, central.py:147-151):
 *   MULTI   obs[E][U][4B+1] = connected[B] | dr[B] | ues_at_bs[B] | util_at_bs[B] | utility[1]
 *   CENTRAL obs[E][U*(2B+1)] = connected[U*B] | dr[U*B] | utility[U]
 * reward: MULTI [E][U] (multi_agent.py:39-95), CENTRAL [E] (central.py:65-73).
 * Optional info tensors (base.py:383-411): sum_utility[E], ue_dr[E][U], ue_utility[E][U]; NULL to skip. */
typedef struct dcomp_out {
    float *obs;
    float *reward;
    float *sum_utility;
    float *ue_dr;
    float *ue_utility;
    float *reward_before;        /* optional [E][U]: clip(utility at the pre-move rates)/20 per UE (base.py:158-167, 446) --
                                  * the reward the single-agent env hands out (base.py:358-369); NULL to skip */
    uint32_t *obs_compact;       /* optional, MULTI only: [E][U (B + 2) + 2B] (U = max_ues slots when UEs arrive / depart) -- the lossless compact record of the
                                  * observation rows (dcomp_fragment_words / dcomp_unpack_fragment in dcomp.h: per UE dr[B] | utility |
                                  * connection mask, then ues_at_bs[B] | util_at_bs[B] once per env) written by the step itself INSTEAD of
                                  * the rows: set it and leave obs NULL.  dcomp_unpack_fragment() of it is bit-identical to the rows the
                                  * same call would have written to obs. */
} dcomp_out;

/* Tape-mode draws for one episode (device): pos0[E*U][2] int32 start positions and
 * triples[E*U][depth] of {velocity, wx, wy, 0} uint16 -- the values the reference's per-UE
 * random.Random streams hand out. */
typedef struct dcomp_tape {
    const int32_t *pos0;         /* [E*num_ue][2] */
    const uint16_t *triples;     /* [E*num_ids][depth][4]; per env: the initial UEs by position, then (UE arrival) one
                                  * 'slow' tape per id an arriving UE can get (seed + 100*id, base.py:602-604) */
    int32_t num_ids;             /* tapes per env; 0 = num_ue */
} dcomp_tape;

/* UE departure / arrival applied by one step, after the actions and before the rates (base.py:433-443).  The counts
 * are the same in every env (the schedule is configuration).  Tape mode: remove_idx[E][n_remove] = the reference's
 * random.randint(0, num_ue-1) list positions (base.py:611), add_xy[E][n_add][2] = map.rand_border_point()
 * (map.py:52-65), both device arrays; Philox mode: NULL (keyed draws in the kernel). */
typedef struct dcomp_events {
    int32_t n_remove, n_add;
    const int32_t *remove_idx;
    const int32_t *add_xy;
} dcomp_events;

/* heuristic policies (deepcomp/agent/heuristics.py; see dcomp_heuristic_actions in dcomp.h) */
enum { DCOMP_POLICY_3GPP = 0, DCOMP_POLICY_FULLCOMP = 1, DCOMP_POLICY_DYNAMIC = 2, DCOMP_POLICY_CLUSTER = 3 };

#ifdef __cplusplus
}
#endif
#endif /* DCOMP_TYPES_H */
