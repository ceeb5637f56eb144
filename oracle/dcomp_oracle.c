/*
 * dcomp_oracle.c -- CPU ORACLE (test infrastructure; see dcomp_oracle.h).
 *
 * Restates, in scalar FP64 and in the reference's own evaluation order, what
 * MobileEnv.reset()/step() and their callees compute.  Every function cites the reference
 * file:line (relative to /root/reference/deepcomp/) it follows.  Build with -ffp-contract=off:
 * the only fused multiply-add on the path is the explicit fma() in move_towards(), which is what
 * numpy's 2-element dot product does in the container the golden fixtures were recorded in.
 */
#include "dcomp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: util/constants.py:28-41, env/entities/station.py:10,26-30 ---- */
#define EPSILON 1e-16
#define MIN_UTILITY (-20.0)
#define MAX_UTILITY 20.0
#define SNR_THRESHOLD 2e-8
#define BS_BW 9e6
#define BS_FREQ 2500.0
#define BS_NOISE 1e-9
#define BS_TX_POWER 30.0
#define BS_HEIGHT 50.0
#define UE_HEIGHT 1.5
#define PAUSE_DURATION 2    /* movement.py:87 defaults; per-UE values via orc_set_movement */
#define BORDER_BUFFER 10

struct orc_env {
    int U;                    /* capacity = max_ues (base.py:79-84); strides of the per-UE arrays */
    int nU, U0;               /* UEs currently in ue_list / at reset (base.py:177-182) */
    int B, map_w, map_h, kind, reward_agg;
    int32_t *uid;             /* [U] int(ue.id) per slot (slot = position in ue_list) */
    int ev_nrem, ev_nadd;     /* pending arrival / departure for the next step (base.py:433-443) */
    int32_t ev_rem[64], ev_add_xy[128];
    int ev_given;
    int tape_ids;             /* draw tapes are keyed by uid-1; number of ids they cover */
    uint32_t n_arrivals, n_removals;   /* per-episode counters (Philox draw words) */
    int32_t *orig_consumed;   /* [U] movement triples an initial UE had consumed when it was removed (-1: still listed) */
    double *bs_x, *bs_y;
    int32_t *bs_sharing, *ue_util, *vel_lo, *vel_hi, *init_x, *init_y;
    int32_t *pause_dur, *border;   /* per configured UE: RandomWaypoint(pause_duration, border_buffer), movement.py:87-104 */
    double *vel_fixed;        /* per configured UE: a fixed velocity given as a number (movement.py:116-117), < 0: drawn / integer */
    double *ue_dr_req;
    /* UE state: user.py:27-46, movement.py:96-104 */
    double *px, *py, *wx, *wy, *vel, *ewma;
    int32_t *pausing, *curr_pause;
    /* ordered containers: ue.bs_dr (insertion-ordered dict) and bs.conn_ues (list) */
    int32_t *ue_bs;   /* [U][B] BS indices in dict order */
    double *ue_dr;    /* [U][B] cached rate, same order */
    int32_t *ue_nbs;  /* [U] */
    int32_t *bs_ues;  /* [B][U] UE indices in connection order */
    int32_t *bs_nues; /* [B] */
    int time;
    double total_utility;
    double *reward_before, *reward; /* [U] */
    /* rng */
    int rng_mode, tape_depth;
    int32_t *tape_pos0, *tape_triples, *cursor;
    uint64_t seed;
    int64_t global_env, episode;
};

/* per-UE config is keyed by the UE id; ids beyond the initial list are UEs that arrived during the episode:
 * base.py:592-599 creates them with velocity 'slow', log utility, dr_req 1 and a fixed start position */
#define UID_BORN 0x8000                 /* set in uid[]: the UE arrived during the episode (add_new_ue) */
static int ue_born(const orc_env *e, int u) { return (e->uid[u] & UID_BORN) != 0; }
static int ue_idnum(const orc_env *e, int u) { return e->uid[u] & 0x7FFF; }
/* index into the draw tapes: initial UEs [0, U0) by list position, arrived UEs U0 + id - 1 (an id can be re-used,
 * and a re-created UE is always 'slow' and re-seeded, whatever the initial UE with that id was) */
static int cfg_id(const orc_env *e, int u) { return ue_born(e, u) ? e->U0 + ue_idnum(e, u) - 1 : ue_idnum(e, u) - 1; }
static int ue_util_kind(const orc_env *e, int u) { return ue_born(e, u) ? ORC_UTIL_LOG : e->ue_util[ue_idnum(e, u) - 1]; }
static double ue_req(const orc_env *e, int u) { return ue_born(e, u) ? 1.0 : e->ue_dr_req[ue_idnum(e, u) - 1]; }
static int ue_vlo(const orc_env *e, int u) { return ue_born(e, u) ? 1 : e->vel_lo[ue_idnum(e, u) - 1]; }
static int ue_vhi(const orc_env *e, int u) { return ue_born(e, u) ? 3 : e->vel_hi[ue_idnum(e, u) - 1]; }
/* arriving UEs get RandomWaypoint(map, velocity='slow') with the default pause / border (base.py:597-599) */
static int ue_pause(const orc_env *e, int u) { return ue_born(e, u) ? PAUSE_DURATION : e->pause_dur[ue_idnum(e, u) - 1]; }
static int ue_border(const orc_env *e, int u) { return ue_born(e, u) ? BORDER_BUFFER : e->border[ue_idnum(e, u) - 1]; }

/* ------------------------------------------------------------------ channel: station.py:110-138,222-226 */
static double path_loss(double distance)
{   /* station.py:110-116 (Okumura-Hata, suburban) */
    double ch = 0.8 + (1.1 * log10(BS_FREQ) - 0.7) * UE_HEIGHT - 1.56 * log10(BS_FREQ);
    double const1 = 69.55 + 26.16 * log10(BS_FREQ) - 13.82 * log10(BS_HEIGHT) - ch;
    double const2 = 44.9 - 6.55 * log10(BS_HEIGHT);
    return const1 + const2 * log10(distance + EPSILON);
}
static double received_power(double distance)
{   /* station.py:118-120 */
    return pow(10.0, (BS_TX_POWER - path_loss(distance)) / 10.0);
}
double orc_snr(double distance) { return received_power(distance) / BS_NOISE; }          /* station.py:122-127 */
int orc_can_connect(double distance) { return orc_snr(distance) > SNR_THRESHOLD; }       /* station.py:222-226 */
double orc_dr_unshared(double distance) { return BS_BW * log2(1.0 + orc_snr(distance)); } /* station.py:129-138 */

double orc_connect_threshold_distance(void)
{
    double lo = 60.0, hi = 80.0; /* can_connect(lo) true, can_connect(hi) false */
    for (int i = 0; i < 200; i++) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        if (orc_can_connect(mid)) lo = mid; else hi = mid;
    }
    return hi;
}

static double point_distance(double ax, double ay, double bx, double by)
{   /* shapely Point.distance -> GEOS: sqrt(dx*dx + dy*dy) (station.py:124, movement.py:142) */
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}
static double bs_dist(const orc_env *e, int b, int u) { return point_distance(e->bs_x[b], e->bs_y[b], e->px[u], e->py[u]); }

/* ------------------------------------------------------------------ utility: utility.py:23-54, user.py:76-92 */
static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
double orc_log_utility(double dr)
{   /* utility.py:52-54 */
    if (dr == 0) return MIN_UTILITY;
    return clipd(10.0 * log10(dr), MIN_UTILITY, MAX_UTILITY);
}
double orc_step_utility(double dr, double req) { return dr >= req ? MAX_UTILITY : MIN_UTILITY; } /* utility.py:31-33 */

static double ue_curr_dr(const orc_env *e, int u)
{   /* user.py:64-69: sum(list(bs_dr.values())) -- Python sum, left to right from int 0 */
    double s = 0.0;
    for (int k = 0; k < e->ue_nbs[u]; k++) s += e->ue_dr[u * e->B + k];
    return s;
}
static double ue_utility(const orc_env *e, int u)
{   /* user.py:76-92 */
    double dr = ue_curr_dr(e, u);
    return ue_util_kind(e, u) == ORC_UTIL_STEP ? orc_step_utility(dr, ue_req(e, u)) : orc_log_utility(dr);
}

/* ------------------------------------------------------------------ sharing: station.py:140-220 */
static int ue_find_bs(const orc_env *e, int u, int b)
{
    for (int k = 0; k < e->ue_nbs[u]; k++) if (e->ue_bs[u * e->B + k] == b) return k;
    return -1;
}
static double bs_priority(const orc_env *e, int b, int u)
{   /* station.py:140-150, alpha = beta = 1 */
    return pow(orc_dr_unshared(bs_dist(e, b, u)), 1) / (pow(e->ewma[u], 1) + EPSILON);
}
static double bs_data_rate_shared(orc_env *e, int b, int u, double dr_unshared)
{   /* station.py:152-202.  The asker is temporarily appended when not connected (:164-168). */
    int *lst = &e->bs_ues[b * e->U];
    int n = e->bs_nues[b], already = 0;
    for (int k = 0; k < n; k++) if (lst[k] == u) already = 1;
    if (!already) lst[n++] = u;
    double shared = 0.0;
    switch (e->bs_sharing[b]) {
    case ORC_RES_FAIR: shared = dr_unshared / n; break;                          /* :171-173 */
    case ORC_RATE_FAIR: {                                                        /* :177-180 */
        double tot = 0.0;
        for (int k = 0; k < n; k++) tot += 1.0 / orc_dr_unshared(bs_dist(e, b, lst[k]));
        shared = 1.0 / tot;
        break; }
    case ORC_MAX_CAP: {                                                          /* :183-187, first max wins */
        int best = 0; double bestv = -1.0;
        for (int k = 0; k < n; k++) { double v = orc_dr_unshared(bs_dist(e, b, lst[k])); if (v > bestv) { bestv = v; best = k; } }
        int idx = 0; for (int k = 0; k < n; k++) if (lst[k] == u) { idx = k; break; }
        shared = (idx == best) ? orc_dr_unshared(bs_dist(e, b, u)) : 0.0;
        break; }
    case ORC_PROP_FAIR: {                                                        /* :192-195 */
        double tot = 0.0;
        for (int k = 0; k < n; k++) tot += bs_priority(e, b, lst[k]);
        shared = bs_priority(e, b, u) / (tot + EPSILON) * dr_unshared;
        break; }
    }
    return shared;
}
static double bs_data_rate(orc_env *e, int b, int u)
{   /* station.py:204-220 */
    double d = bs_dist(e, b, u);
    if (!orc_can_connect(d)) return 0.0;
    return bs_data_rate_shared(e, b, u, orc_dr_unshared(d));
}

/* ------------------------------------------------------------------ connections: user.py:175-229 */
static void ue_disconnect(orc_env *e, int u, int b)
{   /* user.py:224-229: del ue.bs_dr[bs]; bs.conn_ues.remove(ue) */
    int k = ue_find_bs(e, u, b), n = e->ue_nbs[u];
    for (int j = k; j + 1 < n; j++) { e->ue_bs[u * e->B + j] = e->ue_bs[u * e->B + j + 1]; e->ue_dr[u * e->B + j] = e->ue_dr[u * e->B + j + 1]; }
    e->ue_nbs[u] = n - 1;
    int *lst = &e->bs_ues[b * e->U]; int m = e->bs_nues[b], pos = 0;
    for (int j = 0; j < m; j++) if (lst[j] == u) { pos = j; break; }
    for (int j = pos; j + 1 < m; j++) lst[j] = lst[j + 1];
    e->bs_nues[b] = m - 1;
}
static void ue_connect_toggle(orc_env *e, int u, int b)
{   /* user.py:190-222 with disconnect=True */
    if (ue_find_bs(e, u, b) >= 0) { ue_disconnect(e, u, b); return; }
    if (orc_can_connect(bs_dist(e, b, u))) {
        int k = e->ue_nbs[u]++;
        e->ue_bs[u * e->B + k] = b;
        e->ue_dr[u * e->B + k] = bs_data_rate(e, b, u);    /* :216, before joining conn_ues */
        e->bs_ues[b * e->U + e->bs_nues[b]++] = u;         /* :217 */
    }
}
static void ue_check_bs_connection(orc_env *e, int u)
{   /* user.py:175-188 */
    int rem[64], nr = 0, B = e->B;
    int *tmp = (B > 64) ? (int *)malloc(sizeof(int) * B) : rem;
    for (int k = 0; k < e->ue_nbs[u]; k++) { int b = e->ue_bs[u * B + k]; if (!orc_can_connect(bs_dist(e, b, u))) tmp[nr++] = b; }
    for (int k = 0; k < nr; k++) ue_disconnect(e, u, tmp[k]);
    if (tmp != rem) free(tmp);
}

/* ------------------------------------------------------------------ RNG */
static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{   /* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), Philox-4x32-10 */
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static void philox_draw(const orc_env *e, int u, uint32_t draw, uint32_t r[4])
{
    /* counter = (global env id, ue, episode, draw#); key = 64-bit seed.  draw 0 = start position,
     * draw k>=1 = k-th movement triple of the episode. */
    uint32_t ctr[4] = {(uint32_t)e->global_env, (uint32_t)((ue_idnum(e, u) - 1) | (ue_born(e, u) ? UID_BORN : 0)), (uint32_t)e->episode, draw};
    uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)};
    orc_philox4x32_10(ctr, key, r);
}
static void draw_start_pos(orc_env *e, int u)
{   /* user.py:98-109: randint(0, W), randint(0, H) unless a fixed coordinate was given */
    int x, y;
    if (e->rng_mode == ORC_RNG_TAPE) { x = e->tape_pos0[cfg_id(e, u) * 2]; y = e->tape_pos0[cfg_id(e, u) * 2 + 1]; }
    else { uint32_t r[4]; philox_draw(e, u, 0, r); x = (int)mulhi32(r[0], (uint32_t)e->map_w + 1); y = (int)mulhi32(r[1], (uint32_t)e->map_h + 1); }
    e->px[u] = e->init_x[u] >= 0 ? e->init_x[u] : x;       /* reset only: slot == list position of an initial UE */
    e->py[u] = e->init_y[u] >= 0 ? e->init_y[u] : y;
}
static void movement_reset(orc_env *e, int u)
{   /* movement.py:110-130: velocity (iff 'slow'/'fast'), then waypoint x, y in [10, W-10] x [10, H-10] */
    int k = e->cursor[u]++;
    if (e->rng_mode == ORC_RNG_TAPE) {
        if (k >= e->tape_depth) k = e->tape_depth - 1; /* caller checks orc_tape_cursor() against depth */
        const int32_t *t = &e->tape_triples[(cfg_id(e, u) * e->tape_depth + k) * 3];
        e->vel[u] = t[0]; e->wx[u] = t[1]; e->wy[u] = t[2];
    } else {
        uint32_t r[4]; philox_draw(e, u, (uint32_t)k + 1, r);
        e->vel[u] = ue_vlo(e, u) + (int)mulhi32(r[0], (uint32_t)(ue_vhi(e, u) - ue_vlo(e, u) + 1));
        const int bb = ue_border(e, u);
        e->wx[u] = bb + (int)mulhi32(r[1], (uint32_t)(e->map_w - 2 * bb + 1));
        e->wy[u] = bb + (int)mulhi32(r[2], (uint32_t)(e->map_h - 2 * bb + 1));
    }
    /* movement.py:116-117: `self.velocity = self.init_velocity` -- any number the caller configured, never drawn */
    if (!ue_born(e, u) && e->vel_fixed[ue_idnum(e, u) - 1] >= 0.0) e->vel[u] = e->vel_fixed[ue_idnum(e, u) - 1];
    e->pausing[u] = 0;
    e->curr_pause[u] = 0;
}

/* ------------------------------------------------------------------ movement: movement.py:132-181 */
static void move_towards(orc_env *e, int u)
{   /* movement.py:132-156 */
    if (point_distance(e->px[u], e->py[u], e->wx[u], e->wy[u]) <= e->vel[u]) { e->px[u] = e->wx[u]; e->py[u] = e->wy[u]; return; }
    double vx = e->wx[u] - e->px[u], vy = e->wy[u] - e->py[u];
    /* np.linalg.norm(v) = sqrt(v.dot(v)); the 2-element dot evaluates as fma(vy, vy, vx*vx) here */
    double nrm = sqrt(fma(vy, vy, vx * vx));
    double nx = vx / nrm, ny = vy / nrm;
    e->px[u] = e->px[u] + e->vel[u] * nx;
    e->py[u] = e->py[u] + e->vel[u] * ny;
}
static void movement_step(orc_env *e, int u)
{   /* movement.py:158-181 */
    if (e->px[u] == e->wx[u] && e->py[u] == e->wy[u]) e->pausing[u] = 1;
    if (e->pausing[u]) {
        if (e->curr_pause[u] < ue_pause(e, u)) { e->curr_pause[u] += 1; return; }
        movement_reset(e, u);
    }
    move_towards(e, u);
}

/* ------------------------------------------------------------------ env: single_ue/base.py */
static void update_ue_drs_rewards(orc_env *e, int update_only)
{   /* base.py:315-335 -> user.py:143-146 ; calc_reward base.py:158-167 */
    for (int u = 0; u < e->nU; u++) {
        for (int k = 0; k < e->ue_nbs[u]; k++) e->ue_dr[u * e->B + k] = bs_data_rate(e, e->ue_bs[u * e->B + k], u);
        if (!update_only) {
            double cu = clipd(ue_utility(e, u), MIN_UTILITY, MAX_UTILITY);
            e->reward_before[u] = clipd(cu + 0.0, MIN_UTILITY, MAX_UTILITY) / MAX_UTILITY;
        }
    }
}
static void step_reward(orc_env *e)
{
    int U = e->nU, B = e->B, CAP = e->U;
    if (e->kind == ORC_CENTRAL) {   /* multi_ue/central.py:65-73 */
        double r;
        if (e->reward_agg == ORC_AVG) { double s = 0.0; for (int u = 0; u < U; u++) s += e->reward_before[u]; r = s / U; }
        else if (e->reward_agg == ORC_SUM) { r = 0.0; for (int u = 0; u < U; u++) r += e->reward_before[u]; }
        else { r = e->reward_before[0]; for (int u = 1; u < U; u++) if (e->reward_before[u] < r) r = e->reward_before[u]; }
        e->reward[0] = r;
        return;
    }
    /* multi_ue/multi_agent.py:39-95 */
    for (int u = 0; u < U; u++) {
        double agg = ue_utility(e, u);
        int nrange = 0;
        for (int b = 0; b < B; b++) if (orc_can_connect(bs_dist(e, b, u))) nrange++;
        if (nrange > 0) {
            if (e->reward_agg == ORC_AVG) {
                int nn = 0;
                for (int b = 0; b < B; b++) if (orc_can_connect(bs_dist(e, b, u))) nn += e->bs_nues[b];
                if (nn > 0) {
                    double tot = 0.0;
                    for (int b = 0; b < B; b++) if (orc_can_connect(bs_dist(e, b, u))) {
                        double tb = 0.0;   /* station.py:63-69 */
                        for (int k = 0; k < e->bs_nues[b]; k++) tb += ue_utility(e, e->bs_ues[b * CAP + k]);
                        tot += tb;
                    }
                    agg = e->ue_nbs[u] == 0 ? (tot + ue_utility(e, u)) / (nn + 1) : tot / nn;
                }
            } else if (e->reward_agg == ORC_SUM) {
                /* user.py:238-244: set of UEs at any BS this UE is connected to; sum of their rewards_before */
                agg = 0.0;
                for (int v = 0; v < U; v++) {
                    int shares = 0;
                    for (int k = 0; k < e->ue_nbs[u] && !shares; k++) if (ue_find_bs(e, v, e->ue_bs[u * B + k]) >= 0) shares = 1;
                    if (shares) agg += e->reward_before[v];
                }
            } else {
                /* min over in-range BS min_utility (station.py:78-83: MAX_UTILITY if idle) and own utility */
                agg = ue_utility(e, u);
                for (int b = 0; b < B; b++) if (orc_can_connect(bs_dist(e, b, u))) {
                    double mb = MAX_UTILITY;
                    if (e->bs_nues[b] > 0) { mb = ue_utility(e, e->bs_ues[b * CAP]); for (int k = 1; k < e->bs_nues[b]; k++) { double v = ue_utility(e, e->bs_ues[b * CAP + k]); if (v < mb) mb = v; } }
                    if (mb < agg) agg = mb;
                }
            }
        }
        e->reward[u] = agg;
    }
}

/* base.py:608-618: ue_list.pop(idx) + disconnect_from_all; later list positions move up by one */
static void remove_ue(orc_env *e, int idx)
{
    int B = e->B, CAP = e->U;
    if (!ue_born(e, idx)) e->orig_consumed[ue_idnum(e, idx) - 1] = e->cursor[idx];
    while (e->ue_nbs[idx] > 0) ue_disconnect(e, idx, e->ue_bs[idx * B]);
    for (int u = idx; u + 1 < e->nU; u++) {
        e->px[u] = e->px[u + 1]; e->py[u] = e->py[u + 1]; e->wx[u] = e->wx[u + 1]; e->wy[u] = e->wy[u + 1];
        e->vel[u] = e->vel[u + 1]; e->ewma[u] = e->ewma[u + 1]; e->pausing[u] = e->pausing[u + 1];
        e->curr_pause[u] = e->curr_pause[u + 1]; e->cursor[u] = e->cursor[u + 1]; e->uid[u] = e->uid[u + 1];
        e->reward_before[u] = e->reward_before[u + 1];
        e->ue_nbs[u] = e->ue_nbs[u + 1];
        for (int k = 0; k < B; k++) { e->ue_bs[u * B + k] = e->ue_bs[(u + 1) * B + k]; e->ue_dr[u * B + k] = e->ue_dr[(u + 1) * B + k]; }
    }
    e->nU -= 1;
    e->uid[e->nU] = 0; e->ue_nbs[e->nU] = 0;
    for (int b = 0; b < B; b++) for (int k = 0; k < e->bs_nues[b]; k++) if (e->bs_ues[b * CAP + k] > idx) e->bs_ues[b * CAP + k] -= 1;
}
/* base.py:592-606: new UE at a border point, id = last id + 1, seeded env_seed + 100*id, velocity 'slow' */
static void add_ue(orc_env *e, int x, int y)
{
    int u = e->nU;
    e->uid[u] = ((e->uid[u - 1] & 0x7FFF) + 1) | UID_BORN;
    e->nU += 1;
    e->px[u] = x; e->py[u] = y;
    e->cursor[u] = 0;
    movement_reset(e, u);
    e->ue_nbs[u] = 0; e->ewma[u] = 0.0; e->reward_before[u] = 0.0; e->reward[u] = 0.0;
}
static void apply_events(orc_env *e)
{
    int nrem = e->ev_nrem, nadd = e->ev_nadd;
    for (int k = 0; k < nrem; k++) {
        int idx;
        if (e->ev_given) idx = e->ev_rem[k];
        else {   /* Philox: random.randint(0, num_ue - 1) replaced by a keyed draw */
            uint32_t ctr[4] = {(uint32_t)e->global_env, 0xFFFE0000u + e->n_removals, (uint32_t)e->episode, 0u}, r[4];
            uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)};
            orc_philox4x32_10(ctr, key, r);
            idx = (int)mulhi32(r[0], (uint32_t)e->nU);
        }
        e->n_removals++;
        remove_ue(e, idx);
    }
    for (int k = 0; k < nadd; k++) {
        int x, y;
        if (e->ev_given) { x = e->ev_add_xy[2 * k]; y = e->ev_add_xy[2 * k + 1]; }
        else {   /* Philox version of map.rand_border_point (map.py:52-65) */
            uint32_t ctr[4] = {(uint32_t)e->global_env, 0xFFFF0000u + e->n_arrivals, (uint32_t)e->episode, 0u}, r[4];
            uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)};
            orc_philox4x32_10(ctr, key, r);
            int rx = (int)mulhi32(r[0], (uint32_t)e->map_w + 1), ry = (int)mulhi32(r[1], (uint32_t)e->map_h + 1);
            int border = (int)mulhi32(r[2], 4u);    /* left, right, top, bottom */
            x = border == 0 ? 0 : border == 1 ? e->map_w - 1 : rx;
            y = border == 2 ? e->map_h - 1 : border == 3 ? 0 : ry;
            if (border <= 1) y = ry;
        }
        e->n_arrivals++;
        add_ue(e, x, y);
    }
    e->ev_nrem = e->ev_nadd = e->ev_given = 0;
}
void orc_set_events(orc_env *e, int n_remove, const int32_t *remove_idx, int n_add, const int32_t *add_xy)
{
    e->ev_nrem = n_remove; e->ev_nadd = n_add;
    e->ev_given = (remove_idx != NULL) || (add_xy != NULL);
    for (int k = 0; k < n_remove && remove_idx; k++) e->ev_rem[k] = remove_idx[k];
    for (int k = 0; k < 2 * n_add && add_xy; k++) e->ev_add_xy[k] = add_xy[k];
}
void orc_set_initial_ues(orc_env *e, int num_initial) { e->U0 = num_initial; }
/* Test hook: Basestation.data_rate(ue) (station.py:204-220) in the CURRENT state, optionally with the UEs' EWMA rates
 * replaced first -- lets the sharing-model table of the reference (tests/golden/sharing.npz: connected and not-yet-connected
 * askers, with and without EWMA history) be checked row by row, proportional-fair included. */
double orc_probe_data_rate(orc_env *e, int b, int u, const double *ewma)
{
    if (ewma) for (int k = 0; k < e->nU; k++) e->ewma[k] = ewma[k];
    return bs_data_rate(e, b, u);
}
/* RandomWaypoint(pause_duration, border_buffer) of the first n configured UEs (movement.py:87-104) */
void orc_set_movement(orc_env *e, int n, const int32_t *pause_duration, const int32_t *border_buffer)
{
    for (int u = 0; u < n && u < e->U; u++) { e->pause_dur[u] = pause_duration[u]; e->border[u] = border_buffer[u]; }
}
/* fixed velocities of the first n configured UEs as numbers (RandomWaypoint(map, velocity=2.5)); negative: leave as configured */
void orc_set_velocity(orc_env *e, int n, const double *velocity)
{
    for (int u = 0; u < n && u < e->U; u++) e->vel_fixed[u] = velocity[u];
}
int orc_num_ue(const orc_env *e) { return e->nU; }
/* per initial UE: movement triples consumed this episode (at removal, or so far if still listed) */
void orc_get_orig_consumed(const orc_env *e, int32_t *out)
{
    for (int i = 0; i < e->U0; i++) out[i] = e->orig_consumed[i];
    for (int u = 0; u < e->nU; u++) if (!ue_born(e, u)) out[ue_idnum(e, u) - 1] = e->cursor[u];
}
int orc_slot_born(const orc_env *e, int slot) { return slot < e->nU ? ue_born(e, slot) : 0; }
void orc_get_uids(const orc_env *e, int32_t *uids) { for (int u = 0; u < e->U; u++) uids[u] = e->uid[u] & 0x7FFF; }

void orc_reset(orc_env *e)
{   /* base.py:169-189 -> user.py:111-116, station.py:106-108 (seeding is the caller's tape / the philox key) */
    e->time = 0;
    e->nU = e->U0;                                  /* base.py:177-182: restore the original ue_list */
    e->n_arrivals = e->n_removals = 0;
    e->ev_nrem = e->ev_nadd = e->ev_given = 0;
    for (int u = 0; u < e->U; u++) { e->uid[u] = u < e->U0 ? u + 1 : 0; e->ue_nbs[u] = 0; e->orig_consumed[u] = -1; }
    for (int u = 0; u < e->nU; u++) {
        e->cursor[u] = 0;
        draw_start_pos(e, u);
        movement_reset(e, u);
        e->ue_nbs[u] = 0;
        e->ewma[u] = 0.0;
        e->reward_before[u] = 0.0;
        e->reward[u] = 0.0;
    }
    for (int b = 0; b < e->B; b++) e->bs_nues[b] = 0;
}

int orc_step(orc_env *e, const int32_t *action)
{   /* base.py:413-466 */
    for (int u = 0; u < e->nU; u++) if (action[u] < 0 || action[u] > e->B) return -1;   /* central.py:61 */
    for (int u = 0; u < e->nU; u++) if (action[u] > 0) ue_connect_toggle(e, u, action[u] - 1);   /* base.py:247-263 */
    apply_events(e);                                                                      /* base.py:433-443 */
    update_ue_drs_rewards(e, 0);                                                          /* base.py:446 */
    for (int u = 0; u < e->nU; u++) {                                                     /* base.py:447 -> user.py:159-173 */
        movement_step(e, u);
        ue_check_bs_connection(e, u);
        e->ewma[u] = 0.9 * ue_curr_dr(e, u) + (1 - 0.9) * e->ewma[u];                     /* user.py:148-157 (stale rates) */
    }
    update_ue_drs_rewards(e, 1);                                                          /* base.py:451 */
    e->time += 1;
    e->total_utility += orc_sum_utility(e);                                               /* base.py:454-455 */
    step_reward(e);
    return 0;
}

double orc_sum_utility(const orc_env *e)
{   /* base.py:105-107 */
    double s = 0.0;
    for (int u = 0; u < e->nU; u++) s += ue_utility(e, u);
    return s;
}
int orc_time(const orc_env *e) { return e->time; }
int orc_tape_cursor(const orc_env *e, int ue) { return e->cursor[ue]; }
void orc_set_episode(orc_env *e, int64_t episode) { e->episode = episode; }

void orc_get_obs(const orc_env *e, double *connected, double *dr, double *utility, double *ues_at_bs, double *util_at_bs)
{   /* single_ue/variants.py:271-305 per UE; central.py:31-57 / multi_agent.py:32-37 only re-arrange */
    int U = e->nU, B = e->B, CAP = e->U;
    for (int u = U; u < CAP; u++) {          /* slots without a UE: zero padding (central.py:46-55) */
        for (int b = 0; b < B; b++) { connected[u * B + b] = 0.0; dr[u * B + b] = 0.0; if (ues_at_bs) ues_at_bs[u * B + b] = 0.0; if (util_at_bs) util_at_bs[u * B + b] = 0.0; }
        utility[u] = 0.0;
    }
    for (int u = 0; u < U; u++) {
        double mx = 0.0;
        for (int b = 0; b < B; b++) {
            connected[u * B + b] = ue_find_bs(e, u, b) >= 0 ? 1.0 : 0.0;
            double s = orc_snr(bs_dist(e, b, u));
            dr[u * B + b] = s;
            if (b == 0 || s > mx) mx = s;
        }
        for (int b = 0; b < B; b++) dr[u * B + b] = (mx == 0) ? 0.0 : dr[u * B + b] / mx;
        utility[u] = ue_utility(e, u) / MAX_UTILITY;
        if (ues_at_bs) for (int b = 0; b < B; b++) ues_at_bs[u * B + b] = (double)e->bs_nues[b] / U;
        if (util_at_bs) for (int b = 0; b < B; b++) {
            double avg = 0.0;   /* station.py:71-76 */
            if (e->bs_nues[b] > 0) { double s = 0.0; for (int k = 0; k < e->bs_nues[b]; k++) s += ue_utility(e, e->bs_ues[b * CAP + k]); avg = s / e->bs_nues[b]; }
            util_at_bs[u * B + b] = avg / MAX_UTILITY;
        }
    }
}
/* per-UE reward of base.py:446 (clip(utility)/20 at the pre-move rates): what the single-agent env returns for the
 * UE that acted (base.py:360-369) */
void orc_get_reward_before(const orc_env *e, double *out) { for (int u = 0; u < e->U; u++) out[u] = u < e->nU ? e->reward_before[u] : 0.0; }
void orc_get_reward(const orc_env *e, double *reward)
{
    int n = e->kind == ORC_CENTRAL ? 1 : e->U;
    for (int i = 0; i < n; i++) reward[i] = (e->kind == ORC_CENTRAL || i < e->nU) ? e->reward[i] : 0.0;
}
void orc_get_state(const orc_env *e, double *pos, double *wp, double *vel, int32_t *pausing, int32_t *curr_pause,
                   uint8_t *conn, double *dr, double *curr_dr, double *ewma, double *utility, int32_t *conn_order)
{
    int U = e->U, B = e->B;
    for (int u = 0; u < U; u++) {
        const int live = u < e->nU;
        if (pos) { pos[2 * u] = live ? e->px[u] : 0.0; pos[2 * u + 1] = live ? e->py[u] : 0.0; }
        if (wp) { wp[2 * u] = live ? e->wx[u] : 0.0; wp[2 * u + 1] = live ? e->wy[u] : 0.0; }
        if (vel) vel[u] = live ? e->vel[u] : 0.0;
        if (pausing) pausing[u] = live ? e->pausing[u] : 0;
        if (curr_pause) curr_pause[u] = live ? e->curr_pause[u] : 0;
        if (curr_dr) curr_dr[u] = live ? ue_curr_dr(e, u) : 0.0;
        if (ewma) ewma[u] = live ? e->ewma[u] : 0.0;
        if (utility) utility[u] = live ? ue_utility(e, u) : 0.0;
        for (int b = 0; b < B; b++) {
            int k = live ? ue_find_bs(e, u, b) : -1;
            if (conn) conn[u * B + b] = k >= 0;
            if (dr) dr[u * B + b] = k >= 0 ? e->ue_dr[u * B + k] : 0.0;
        }
    }
    if (conn_order) for (int b = 0; b < B; b++) for (int k = 0; k < U; k++) conn_order[b * U + k] = k < e->bs_nues[b] ? e->bs_ues[b * U + k] : -1;
}

/* ------------------------------------------------------------------ lifecycle */
static void *dup_mem(const void *src, size_t n) { void *p = malloc(n ? n : 1); if (src) memcpy(p, src, n); else memset(p, 0, n); return p; }
orc_env *orc_create(int U, int B, int map_w, int map_h, int kind, int reward_agg, const double *bs_x, const double *bs_y,
                    const int32_t *bs_sharing, const int32_t *ue_util, const double *ue_dr_req, const int32_t *vel_lo,
                    const int32_t *vel_hi, const int32_t *init_x, const int32_t *init_y)
{
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->U = U; e->B = B; e->map_w = map_w; e->map_h = map_h; e->kind = kind; e->reward_agg = reward_agg;
    e->bs_x = dup_mem(bs_x, sizeof(double) * B); e->bs_y = dup_mem(bs_y, sizeof(double) * B);
    e->bs_sharing = dup_mem(bs_sharing, sizeof(int32_t) * B);
    e->ue_util = dup_mem(ue_util, sizeof(int32_t) * U);
    e->ue_dr_req = dup_mem(ue_dr_req, sizeof(double) * U);
    e->vel_lo = dup_mem(vel_lo, sizeof(int32_t) * U); e->vel_hi = dup_mem(vel_hi, sizeof(int32_t) * U);
    e->pause_dur = dup_mem(NULL, sizeof(int32_t) * U); e->border = dup_mem(NULL, sizeof(int32_t) * U);
    for (int u = 0; u < U; u++) { e->pause_dur[u] = PAUSE_DURATION; e->border[u] = BORDER_BUFFER; }
    e->vel_fixed = dup_mem(NULL, sizeof(double) * U);
    for (int u = 0; u < U; u++) e->vel_fixed[u] = -1.0;
    e->init_x = dup_mem(init_x, sizeof(int32_t) * U); e->init_y = dup_mem(init_y, sizeof(int32_t) * U);
    if (!ue_dr_req) for (int u = 0; u < U; u++) e->ue_dr_req[u] = 1.0;
    if (!init_x) for (int u = 0; u < U; u++) e->init_x[u] = -1;
    if (!init_y) for (int u = 0; u < U; u++) e->init_y[u] = -1;
    size_t du = sizeof(double) * U;
    e->px = dup_mem(NULL, du); e->py = dup_mem(NULL, du); e->wx = dup_mem(NULL, du); e->wy = dup_mem(NULL, du);
    e->vel = dup_mem(NULL, du); e->ewma = dup_mem(NULL, du); e->reward_before = dup_mem(NULL, du); e->reward = dup_mem(NULL, du);
    e->pausing = dup_mem(NULL, sizeof(int32_t) * U); e->curr_pause = dup_mem(NULL, sizeof(int32_t) * U);
    e->cursor = dup_mem(NULL, sizeof(int32_t) * U); e->ue_nbs = dup_mem(NULL, sizeof(int32_t) * U);
    e->ue_bs = dup_mem(NULL, sizeof(int32_t) * U * B); e->ue_dr = dup_mem(NULL, sizeof(double) * U * B);
    e->bs_ues = dup_mem(NULL, sizeof(int32_t) * U * B); e->bs_nues = dup_mem(NULL, sizeof(int32_t) * B);
    e->uid = dup_mem(NULL, sizeof(int32_t) * U);
    e->orig_consumed = dup_mem(NULL, sizeof(int32_t) * U);
    e->U0 = e->nU = U; e->tape_ids = U;
    for (int u = 0; u < U; u++) e->uid[u] = u + 1;
    e->rng_mode = ORC_RNG_PHILOX; e->seed = 42;
    return e;
}
void orc_destroy(orc_env *e)
{
    if (!e) return;
    free(e->bs_x); free(e->bs_y); free(e->bs_sharing); free(e->ue_util); free(e->ue_dr_req); free(e->vel_lo); free(e->vel_hi);
    free(e->pause_dur); free(e->border); free(e->vel_fixed);
    free(e->init_x); free(e->init_y); free(e->px); free(e->py); free(e->wx); free(e->wy); free(e->vel); free(e->ewma);
    free(e->reward_before); free(e->reward); free(e->pausing); free(e->curr_pause); free(e->cursor); free(e->ue_nbs);
    free(e->ue_bs); free(e->ue_dr); free(e->bs_ues); free(e->bs_nues); free(e->tape_pos0); free(e->tape_triples); free(e->uid); free(e->orig_consumed);
    free(e);
}
void orc_set_tape_ids(orc_env *e, int depth, int num_ids, const int32_t *pos0, const int32_t *triples)
{   /* tapes are keyed by UE id - 1 (ids beyond the initial list: UEs that arrive during the episode) */
    free(e->tape_pos0); free(e->tape_triples);
    e->rng_mode = ORC_RNG_TAPE; e->tape_depth = depth; e->tape_ids = num_ids;
    e->tape_pos0 = dup_mem(pos0, sizeof(int32_t) * num_ids * 2);
    e->tape_triples = dup_mem(triples, sizeof(int32_t) * num_ids * depth * 3);
}
void orc_set_tape(orc_env *e, int depth, const int32_t *pos0, const int32_t *triples)
{
    orc_set_tape_ids(e, depth, e->U, pos0, triples);
}
void orc_set_philox(orc_env *e, uint64_t seed, int64_t global_env_id)
{
    e->rng_mode = ORC_RNG_PHILOX; e->seed = seed; e->global_env = global_env_id; e->episode = 0;
}

/* ------------------------------------------------------------------ batched driver */
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
static void pack_obs(const orc_env *e, float *o)
{
    int U = e->U, B = e->B;
    double *c = malloc(sizeof(double) * U * B * 4 + sizeof(double) * U);
    double *d = c + U * B, *n = d + U * B, *a = n + U * B, *ut = a + U * B;
    orc_get_obs(e, c, d, ut, n, a);
    int stride = e->kind == ORC_MULTI ? 4 * B + 1 : 2 * B + 1;
    for (int u = 0; u < U; u++) {
        float *r = o + (size_t)u * stride;
        for (int b = 0; b < B; b++) { r[b] = (float)c[u * B + b]; r[B + b] = (float)d[u * B + b]; }
        if (e->kind == ORC_MULTI) { for (int b = 0; b < B; b++) { r[2 * B + b] = (float)n[u * B + b]; r[3 * B + b] = (float)a[u * B + b]; } r[4 * B] = (float)ut[u]; }
        else r[2 * B] = (float)ut[u];
    }
    free(c);
}
void orc_batch_reset(orc_env **envs, int E, float *obs, int num_threads)
{
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for num_threads(num_threads) schedule(static)
    for (int i = 0; i < E; i++) {
        orc_env *e = envs[i];
        orc_reset(e);
        int stride = e->kind == ORC_MULTI ? 4 * e->B + 1 : 2 * e->B + 1;
        if (obs) pack_obs(e, obs + (size_t)i * e->U * stride);
    }
}
void orc_batch_step(orc_env **envs, int E, const uint8_t *action, float *obs, float *reward, uint32_t *conn_bits,
                    double *pos, int num_threads)
{
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for num_threads(num_threads) schedule(static)
    for (int i = 0; i < E; i++) {
        orc_env *e = envs[i];
        int U = e->U, B = e->B;
        int32_t act[1024];
        for (int u = 0; u < U; u++) act[u] = action[(size_t)i * U + u];
        orc_step(e, act);
        int stride = e->kind == ORC_MULTI ? 4 * B + 1 : 2 * B + 1;
        if (obs) pack_obs(e, obs + (size_t)i * U * stride);
        if (reward) { if (e->kind == ORC_MULTI) for (int u = 0; u < U; u++) reward[(size_t)i * U + u] = u < e->nU ? (float)e->reward[u] : 0.f; else reward[i] = (float)e->reward[0]; }
        if (conn_bits) for (int u = 0; u < U; u++) { uint32_t m = 0; if (u < e->nU) for (int k = 0; k < e->ue_nbs[u]; k++) if (e->ue_bs[u * B + k] < 32) m |= 1u << e->ue_bs[u * B + k]; conn_bits[(size_t)i * U + u] = m; }   /* stations 0-31; 32-63: orc_batch_conn_hi */
        if (pos) for (int u = 0; u < U; u++) { pos[((size_t)i * U + u) * 2] = u < e->nU ? e->px[u] : 0.0; pos[((size_t)i * U + u) * 2 + 1] = u < e->nU ? e->py[u] : 0.0; }
    }
}
/* Stations 32 ... 63 of every UE's connection set (user.py:34 bs_dr keys), as the device keeps them in state.conn_hi. */
void orc_batch_conn_hi(orc_env **envs, int E, uint32_t *conn_hi)
{
    for (int i = 0; i < E; i++) {
        const orc_env *e = envs[i];
        const int U = e->U, B = e->B;
        for (int u = 0; u < U; u++) {
            uint32_t m = 0;
            if (u < e->nU) for (int k = 0; k < e->ue_nbs[u]; k++) if (e->ue_bs[u * B + k] >= 32) m |= 1u << (e->ue_bs[u * B + k] - 32);
            conn_hi[(size_t)i * U + u] = m;
        }
    }
}
/* FP64 per-UE rates of every env of a batch, for the at-scale parity tests (north_star: 1e-5 RELATIVE on data-rate floats):
 * curr_dr = sum of the cached per-connection rates (user.py:64-69), ewma (user.py:148-157), utility (user.py:76-92),
 * dr_rel = snr_b / max_b snr_b in FP64 (variants.py:276-284) -- the observation entry before its float32 cast. */
void orc_batch_rates(orc_env **envs, int E, double *curr_dr, double *ewma, double *utility, double *dr_rel, int num_threads)
{
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for num_threads(num_threads) schedule(static)
    for (int i = 0; i < E; i++) {
        const orc_env *e = envs[i];
        const int U = e->U, B = e->B;
        for (int u = 0; u < U; u++) {
            const int live = u < e->nU;
            if (curr_dr) curr_dr[(size_t)i * U + u] = live ? ue_curr_dr(e, u) : 0.0;
            if (ewma) ewma[(size_t)i * U + u] = live ? e->ewma[u] : 0.0;
            if (utility) utility[(size_t)i * U + u] = live ? ue_utility(e, u) : 0.0;
        }
        if (dr_rel) {
            double *c = malloc(sizeof(double) * U * B * 4 + sizeof(double) * U);
            double *d = c + U * B, *n = d + U * B, *a = n + U * B, *ut = a + U * B;
            orc_get_obs(e, c, d, ut, n, a);
            memcpy(dr_rel + (size_t)i * U * B, d, sizeof(double) * U * B);
            free(c);
        }
    }
}
