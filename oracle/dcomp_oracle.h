/*
 * dcomp_oracle.h -- CPU ORACLE for the DeepCoMP env-step hot path.   *** TEST INFRASTRUCTURE ***
 *
 * A plain-C, FP64, one-env-at-a-time restatement of the reference's reset()/step() semantics
 * (SURVEY.md Appendix A), including the parts that hide in Python object order (connection age for
 * max-cap ties, dict insertion order for rate sums).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path (deepcomp_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against the golden
 * fixtures under tests/golden/ that were produced by running the reference's own unmodified
 * deepcomp.env.* modules in the build container (tests/golden/gen_golden.py).
 */
#ifndef DCOMP_ORACLE_H
#define DCOMP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_CENTRAL = 0, ORC_MULTI = 1 };
enum { ORC_AVG = 0, ORC_SUM = 1, ORC_MIN = 2 };
enum { ORC_RES_FAIR = 0, ORC_RATE_FAIR = 1, ORC_MAX_CAP = 2, ORC_PROP_FAIR = 3 };
enum { ORC_UTIL_LOG = 0, ORC_UTIL_STEP = 1 };
enum { ORC_RNG_TAPE = 0, ORC_RNG_PHILOX = 1 };

typedef struct orc_env orc_env;

/* One env.  vel_lo/vel_hi: inclusive integer range a movement reset draws the velocity from
 * (lo==hi: fixed velocity, no draw; movement.py:112-117).  init_x/init_y: fixed start coordinate or
 * -1 for 'random' (user.py:98-109). */
orc_env *orc_create(int num_ue, int num_bs, int map_w, int map_h, int kind, int reward_agg,
                    const double *bs_x, const double *bs_y, const int32_t *bs_sharing,
                    const int32_t *ue_util, const double *ue_dr_req,
                    const int32_t *vel_lo, const int32_t *vel_hi,
                    const int32_t *init_x, const int32_t *init_y);
void orc_destroy(orc_env *e);

/* RNG.  Tape mode: the caller pre-draws, per UE, `depth` movement triples (vel, wx, wy) in the
 * reference's draw order -- triple 0 is consumed by reset(), the rest by waypoint
 * redraws -- plus the start position.  Philox mode: counter-based Philox4x32-10 keyed by
 * (seed, global env id, ue, episode, draw#); the device kernels use the identical mapping. */
void orc_set_tape(orc_env *e, int depth, const int32_t *pos0 /*[U][2]*/, const int32_t *triples /*[U][depth][3]*/);
void orc_set_philox(orc_env *e, uint64_t seed, int64_t global_env_id);

/* UE arrival / departure (base.py:433-443, 592-618).  orc_create's num_ue is the CAPACITY (max_ues, base.py:79-84);
 * orc_set_initial_ues says how many of the configured UEs exist after reset().  orc_set_events arms the events of
 * the NEXT step (applied after the actions, before the rates): departures by list position, then arrivals at the
 * given border points -- tape mode hands both in (reference draws: global random.randint, map.rand_border_point);
 * with NULL arrays the Philox mapping shared with the device kernels is used.  Tapes may cover more ids than the
 * capacity (orc_set_tape_ids): a UE's draws are keyed by its id. */
void orc_set_initial_ues(orc_env *e, int num_initial);
void orc_set_events(orc_env *e, int n_remove, const int32_t *remove_idx, int n_add, const int32_t *add_xy);
void orc_set_tape_ids(orc_env *e, int depth, int num_ids, const int32_t *pos0, const int32_t *triples);
int  orc_num_ue(const orc_env *e);
void orc_get_uids(const orc_env *e, int32_t *uids /*[capacity], 0 = empty slot*/);
int  orc_slot_born(const orc_env *e, int slot);          /* the UE in this slot arrived during the episode */
void orc_get_orig_consumed(const orc_env *e, int32_t *out /*[initial UEs]*/);

void orc_reset(orc_env *e);                      /* base.py:169-189 (state part; obs via orc_get_obs) */
int  orc_step(orc_env *e, const int32_t *action /*[U]*/);   /* base.py:413-466; returns 0 or <0 on bad action */

/* Outputs of the last reset()/step(). */
void orc_get_obs(const orc_env *e, double *connected /*[U*B]*/, double *dr /*[U*B]*/, double *utility /*[U]*/,
                 double *ues_at_bs /*[U*B] or NULL*/, double *util_at_bs /*[U*B] or NULL*/);
void orc_get_reward(const orc_env *e, double *reward /*[1] central, [U] multi*/);
void orc_get_reward_before(const orc_env *e, double *out /*[U]*/);   /* base.py:446 rewards, single-agent env base.py:360-369 */
void orc_get_state(const orc_env *e, double *pos /*[U*2]*/, double *wp /*[U*2]*/, double *vel /*[U]*/,
                   int32_t *pausing, int32_t *curr_pause, uint8_t *conn /*[U*B]*/, double *dr /*[U*B]*/,
                   double *curr_dr, double *ewma, double *utility, int32_t *conn_order /*[B*U], -1 padded*/);
double orc_sum_utility(const orc_env *e);
int    orc_time(const orc_env *e);
int    orc_tape_cursor(const orc_env *e, int ue);   /* movement triples consumed so far by this UE */
void   orc_set_episode(orc_env *e, int64_t episode); /* philox: episode index used by the next reset */

/* Stand-alone pieces (known-answer tables G1-G3). */
double orc_snr(double dist);                       /* station.py:110-127 */
int    orc_can_connect(double dist);               /* station.py:222-226 */
double orc_dr_unshared(double dist);               /* station.py:129-138 */
double orc_log_utility(double dr);                 /* utility.py:36-54 */
double orc_step_utility(double dr, double req);    /* utility.py:23-33 */
double orc_connect_threshold_distance(void);       /* smallest double d with can_connect(d) false */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Batched driver (OpenMP over envs) -- used for large-E parity checks and the CPU baseline.
 * action: [E][U]; out arrays may be NULL.  obs is packed per UE in the device layout
 * (see include/dcomp.h): multi [E][U][4B+1] = connected|dr|ues_at_bs|util_at_bs|utility,
 * central [E][U][2B+1] = connected|dr|utility. */
void orc_batch_reset(orc_env **envs, int num_envs, float *obs, int num_threads);
void orc_batch_step(orc_env **envs, int num_envs, const uint8_t *action, float *obs, float *reward,
                    uint32_t *conn_bits /*[E][U]*/, double *pos /*[E][U][2]*/, int num_threads);
void orc_batch_conn_hi(orc_env **envs, int num_envs, uint32_t *conn_hi /*[E][U]: stations 32-63 of the connection sets*/);
void orc_batch_rates(orc_env **envs, int num_envs, double *curr_dr, double *ewma, double *utility, double *dr_rel, int num_threads);
int  orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
