// dcomp_api.hip -- C ABI (include/dcomp.h) of libdcomp_hip.so: handle management, kernel dispatch,
// the CPython-compatible Mersenne-Twister tape generator, device self-tests.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#define DCOMP_BUILDING_LIBRARY 1     // (no dcomp_create -> dcomp_create_v macro in here)
#include "../../include/dcomp.h"
#include "dcomp_blist.h"
#include "dcomp_device.h"
#include "dcomp_fragment.h"
#include "dcomp_big.h"

namespace dcomp {
#define DCOMP_DECL(n) KernelPair kernels_b##n(int upad, int mp);
DCOMP_B_LIST(DCOMP_DECL)
#undef DCOMP_DECL

static KernelPair lookup_kernels(int B, int upad, int mp)
{
    switch (B) {
#define DCOMP_CASE(n) case n: return kernels_b##n(upad, mp);
        DCOMP_B_LIST(DCOMP_CASE)
#undef DCOMP_CASE
    default: return KernelPair{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    }
}
}  // namespace dcomp

using dcomp::KParams;
using dcomp::UeCfg;

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(DCOMP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct dcomp_env {
    dcomp_cfg cfg;
    KParams kp;                 // constant part pre-filled
    dcomp::KernelPair kern;
    UeCfg *d_ue_cfg;
    double2 *d_ue_velq = nullptr;   // {velocity, qmax} of UEs whose fixed velocity is no integer in 0..255 (cfg.ue_velocity)
    bool dyn;                  // UE list changes during an episode (cfg.max_ues > 0)
    int mp_pattern;            // sharing-pattern specialisation the kernels were looked up with (dcomp::MP_*)
    bool fused;                // kern.step is step_kernel: T steps in one launch (the wide / dynamic kernels step once per launch)
    bool fused_big = false;    // the generic kernel's fused rollout (fixed UE list): T steps = one launch per stretch of an episode
    bool fused_long;           // ... for rollouts of >= 4 steps at ANY batch size (small central rows: see dcomp_create)
    int upad, grid;
    int wide_pad_lds = 0;      // step_kernel_wide: extra dynamic LDS per workgroup = fewer resident workgroups per CU (see dcomp_create)
    int wide_grid = 0;         // step_kernel_wide is persistent: workgroups launched (<= what the GPU holds at once), each walks grid / wide_grid slots
    int tight_g, tight_gpw, tight_magic, tight_grid;   // step_kernel's tight packing of non-power-of-two UE lists (0 = off)
    int cap, cur_ue;            // slots per env; UEs currently listed
    uint32_t n_removed, n_arrived;   // this episode (Philox draw words)
    int time;
    int64_t episode;            // index of the current episode (-1 before the first reset)
    // 33 ... 64 stations (or DCOMP_FORCE_BIG=1): the generic kernel of dcomp_big.h instead of `kern`
    bool big = false;
    dcomp::BigKernels bigk{};
    int big_step = 0;                                      // BigKernels::fn's `which` of a step: 0, or 2 with UE arrival / departure
    dcomp::BigParams bigp{};
    double2 *d_bs = nullptr;
    int32_t *d_mode = nullptr;
    size_t big_lds = 0;
};

// ---- channel constants from the reference's own formula (station.py:26-30,110-127), FP64 on the host ----
static double ref_snr(double d)
{
    const double f = 2500.0, hb = 50.0, hu = 1.5, tx = 30.0, noise = 1e-9;
    double ch = 0.8 + (1.1 * std::log10(f) - 0.7) * hu - 1.56 * std::log10(f);
    double c1 = 69.55 + 26.16 * std::log10(f) - 13.82 * std::log10(hb) - ch;
    double c2 = 44.9 - 6.55 * std::log10(hb);
    double pl = c1 + c2 * std::log10(d + 1e-16);
    return std::pow(10.0, (tx - pl) / 10.0) / noise;
}
// d_T = the smallest double d whose snr, computed the reference's way, is NOT above the threshold; -1 when that computed snr is not
// monotone within 4 096 doubles either side of d_T (then `d < d_T` would not be the reference's decision; dcomp_create refuses).
static double connect_threshold_checked()
{
    static const double cached = [] {
        double lo = 60.0, hi = 80.0;    // snr(lo) > 2e-8 >= snr(hi)
        for (int i = 0; i < 200; i++) {
            double mid = 0.5 * (lo + hi);
            if (mid == lo || mid == hi) break;
            if (ref_snr(mid) > 2e-8) lo = mid; else hi = mid;
        }
        double below = hi, above = hi;
        for (int k = 0; k < 4096; k++) {
            below = std::nextafter(below, 0.0);
            if (!(ref_snr(below) > 2e-8) || ref_snr(above) > 2e-8) return -1.0;
            above = std::nextafter(above, INFINITY);
        }
        return hi;
    }();
    return cached;
}
extern "C" double dcomp_connect_threshold(void) { return connect_threshold_checked(); }
// The reference decides `snr(sqrt(dx*dx + dy*dy)) > 2e-8` (station.py:122-127, 222-226): with q = fl(fl(dx*dx) + fl(dy*dy)) -- two rounded
// squares, one rounded sum, contraction off -- that is sqrt_rn(q) < d_T, and since the correctly rounded root is monotone,
//     in range  <=>  q < X,   X = the smallest double q with sqrt_rn(q) >= d_T.
// X is NOT fl(d_T * d_T) (one ulp above X for the reference's constants): the kernels compare against X (KParams::dt2), found here by
// walking the doubles around d_T * d_T with the host's IEEE sqrt (checked against brute force in dcomp_create, qmax(v)).
extern "C" double dcomp_connect_boundary_sq(void)
{
    const double dt = connect_threshold_checked();
    if (!(dt > 0.0)) return -1.0;
    double q = dt * dt;
    while (std::sqrt(q) >= dt) q = std::nextafter(q, 0.0);
    while (std::sqrt(q) < dt) q = std::nextafter(q, INFINITY);
    return q;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the FUNCTION on a DEVICE, shared by every env that launches it: it only ever
// goes UP -- a running maximum per (function, device) -- so that a later env with a smaller footprint does not lower the limit under an
// earlier one.  (Asking for the CU's whole 160 KB once and for all is refused by the runtime: "invalid argument".)
static hipError_t raise_lds_limit(const void *fn, int bytes)
{
    struct Raised { const void *fn; int dev, bytes; };
    static std::mutex mu;
    static std::vector<Raised> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &d : done) {
        if (d.fn != fn || d.dev != dev) continue;
        if (d.bytes >= bytes) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) d.bytes = bytes;
        return e;
    }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.push_back(Raised{fn, dev, bytes});
    return e;
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Rows (num_steps * num_envs * num_ue) from which an every-step fragment no longer fits the fused rollout kernel's 32-bit row indices.
// DCOMP_FUSED_ROW_LIMIT_LOG2 lowers it (tests: the one-launch-per-step fallback of dcomp_rollout_ex without a 2^31-row fragment).
static uint64_t fused_row_limit()
{
    const char *e = getenv("DCOMP_FUSED_ROW_LIMIT_LOG2");
    const int l2 = e ? atoi(e) : 31;
    return (uint64_t)1 << (l2 >= 1 && l2 <= 31 ? l2 : 31);
}

extern "C" const char *dcomp_last_error(void) { return g_err; }
#define DCOMP_STR2(x) #x
#define DCOMP_STR(x) DCOMP_STR2(x)
extern "C" const char *dcomp_version(void) { return "deepcomp_amd 0.2 (gfx950, ABI " DCOMP_STR(DCOMP_ABI_VERSION) ")"; }
extern "C" int dcomp_abi_version(void) { return DCOMP_ABI_VERSION; }

// The version-1 entry point.  A binary that still calls it was compiled against structs this library no longer reads the same way
// (dcomp_out had six pointers): refuse instead of reading past them.
extern "C" int dcomp_create(const dcomp_cfg *, dcomp_env **out)
{
    if (out) *out = nullptr;
    return fail(DCOMP_EABI, "dcomp_create is the ABI-1 entry point and this library is ABI %d: rebuild the caller against include/dcomp.h "
                            "(its dcomp_create macro calls dcomp_create_v) or call dcomp_create_v with the caller's struct sizes", DCOMP_ABI_VERSION);
}

extern "C" int dcomp_create_v(int32_t abi_version, size_t cfg_size, size_t state_size, size_t out_size, size_t rollout_opts_size,
                              const dcomp_cfg *cfg, dcomp_env **out)
{
    if (out) *out = nullptr;
    if (abi_version != DCOMP_ABI_VERSION || cfg_size != sizeof(dcomp_cfg) || state_size != sizeof(dcomp_state) || out_size != sizeof(dcomp_out) ||
        rollout_opts_size != sizeof(dcomp_rollout_opts))
        return fail(DCOMP_EABI, "ABI mismatch: caller has version %d, sizeof dcomp_cfg / dcomp_state / dcomp_out / dcomp_rollout_opts = %zu / %zu / %zu / %zu; "
                                "this library has version %d and %zu / %zu / %zu / %zu", (int)abi_version, cfg_size, state_size, out_size, rollout_opts_size,
                    DCOMP_ABI_VERSION, sizeof(dcomp_cfg), sizeof(dcomp_state), sizeof(dcomp_out), sizeof(dcomp_rollout_opts));
    if (!cfg || !out) return fail(DCOMP_EINVAL, "null argument");
    *out = nullptr;
    const int E = cfg->num_envs, U = cfg->num_ue, B = cfg->num_bs;
    if (E < 1 || U < 1 || U > DCOMP_MAX_UE || B < 1 || B > DCOMP_MAX_BS)
        return fail(DCOMP_EINVAL, "need num_envs>=1, 1<=num_ue<=%d, 1<=num_bs<=%d (got %d, %d, %d)", DCOMP_MAX_UE, DCOMP_MAX_BS, E, U, B);
    const int CAP = cfg->max_ues > 0 ? cfg->max_ues : U;           // slots per env (base.py:79-84)
    if (CAP < U) return fail(DCOMP_EINVAL, "max_ues (%d) < num_ue (%d)", CAP, U);                 // base.py:84
    const bool DYN = cfg->max_ues > 0;                             // departures alone need no extra slots: max_ues == num_ue
    if ((int64_t)E * CAP > (int64_t)1 << 30) return fail(DCOMP_EINVAL, "num_envs*max_ues too large");
    if (cfg->map_w < 3 || cfg->map_h < 3 || cfg->map_w > 65535 || cfg->map_h > 65535)
        return fail(DCOMP_EINVAL, "map must be 3..65535 in both dimensions");
    for (int u = 0; u < U; u++) {                                  // RandomWaypoint(pause_duration, border_buffer), movement.py:87-104
        const int pd = cfg->ue_pause_duration ? cfg->ue_pause_duration[u] : 2, bb = cfg->ue_border_buffer ? cfg->ue_border_buffer[u] : 10;
        if (pd < 0 || pd > 127) return fail(DCOMP_EINVAL, "ue %d: pause_duration %d outside 0..127 (7 bits of the movement word)", u, pd);
        if (bb < 1 || bb > 255) return fail(DCOMP_EINVAL, "ue %d: border_buffer %d outside 1..255 (movement.py:103 asserts > 0)", u, bb);
        if (cfg->map_w < 2 * bb + 1 || cfg->map_h < 2 * bb + 1)
            return fail(DCOMP_EINVAL, "ue %d: map %dx%d leaves no waypoint inside a border buffer of %d", u, cfg->map_w, cfg->map_h, bb);
    }
    if (cfg->max_ues > 0 && (cfg->map_w < 21 || cfg->map_h < 21))
        return fail(DCOMP_EINVAL, "UE arrival needs a map of at least 21x21 (arriving UEs use border_buffer 10, base.py:597-599)");
    if (cfg->env_kind != DCOMP_CENTRAL && cfg->env_kind != DCOMP_MULTI) return fail(DCOMP_EINVAL, "bad env_kind");
    if (cfg->reward_agg < 0 || cfg->reward_agg > 2) return fail(DCOMP_EINVAL, "bad reward_agg");
    if (cfg->rng_mode != DCOMP_RNG_TAPE && cfg->rng_mode != DCOMP_RNG_PHILOX) return fail(DCOMP_EINVAL, "bad rng_mode");
    if (cfg->rng_mode == DCOMP_RNG_TAPE && cfg->tape_depth < 1) return fail(DCOMP_EINVAL, "tape mode needs tape_depth >= 1");
    if (!cfg->bs_x || !cfg->bs_y || !cfg->bs_sharing || !cfg->ue_vel_lo || !cfg->ue_vel_hi)
        return fail(DCOMP_EINVAL, "bs_x, bs_y, bs_sharing, ue_vel_lo, ue_vel_hi are required");

    dcomp_env *env = new dcomp_env();
    env->cfg = *cfg;
    env->cfg.bs_x = env->cfg.bs_y = nullptr;   // host arrays are not retained
    env->cfg.ue_pause_duration = env->cfg.ue_border_buffer = nullptr;
    env->cfg.ue_velocity = nullptr;
    env->cap = CAP; env->dyn = DYN; env->cur_ue = U; env->n_removed = env->n_arrived = 0;
    env->upad = next_pow2(CAP) < 4 ? 4 : next_pow2(CAP);
    int mp = dcomp::MP_RES_FAIR;            // sharing pattern -> specialised kernel (dcomp_device.h bs_mode_of)
    for (int b = 0; b < B; b++) if (cfg->bs_sharing[b] != DCOMP_RES_FAIR) mp = dcomp::MP_MIXED;
    if (mp == dcomp::MP_MIXED) {
        static const int cyc[3] = {DCOMP_RES_FAIR, DCOMP_RATE_FAIR, DCOMP_PROP_FAIR};   // env_setup.py:40-49
        for (int b = 0; b < B; b++) if (cfg->bs_sharing[b] != cyc[b % 3]) mp = dcomp::MP_GENERIC;
    }
    // More than 32 stations: the generic kernel (dcomp_big.h; one instantiation per lane width, B a run-time value, connection set in two
    // state words).  DCOMP_FORCE_BIG=1 sends smaller station counts there too (tests: generic against specialised kernels).
    env->big = B > DCOMP_MASK32_MAX_BS || CAP > DCOMP_SPECIAL_MAX_UE || (getenv("DCOMP_FORCE_BIG") && atoi(getenv("DCOMP_FORCE_BIG")) != 0);
    env->mp_pattern = mp;
    if (env->big) {
        env->bigk = dcomp::big_kernels_for_upad(env->upad);
        if (!env->bigk.fn[0][0][0]) { delete env; return fail(DCOMP_EUNSUPPORTED, "no generic kernel for %d lanes per env", env->upad); }
        env->big_step = DYN ? 2 : 0;
        env->fused_big = !DYN && !getenv("DCOMP_NO_FUSED_BIG");                               // UE arrival / departure (round 6: the generic kernel has the event phase too)
        if ((size_t)dcomp::big_carve(B, env->bigk.gpb, env->bigk.block).total > 160 * 1024) {
            delete env;
            return fail(DCOMP_EINVAL, "%d UE slots x %d stations do not fit one workgroup's LDS (generic kernel)", CAP, B);
        }
        env->kern = dcomp::KernelPair{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        env->grid = (E + env->bigk.gpb - 1) / env->bigk.gpb;
    } else {
        env->kern = dcomp::lookup_kernels(B, env->upad, mp);
        if (!env->kern.step) { delete env; return fail(DCOMP_EUNSUPPORTED, "no kernel built for num_bs=%d (built: " DCOMP_B_LIST_STR ")", B); }
        // the wide kernel (UPAD >= 64) always uses 256-thread workgroups; the others DCOMP_BLOCK
        const int gpb = DCOMP_BLOCK >= env->upad ? DCOMP_BLOCK / env->upad : 1;
        env->grid = (E + gpb - 1) / gpb;
    }
    env->time = 0;
    env->episode = -1;

    KParams &kp = env->kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.E = E; kp.U = CAP; kp.U0 = U; kp.cur_ue = U; kp.tape_ids = U;
    kp.map_w = cfg->map_w; kp.map_h = cfg->map_h;
    kp.kind = cfg->env_kind; kp.reward_agg = cfg->reward_agg; kp.rng_mode = cfg->rng_mode;
    kp.tape_depth = cfg->tape_depth;
    kp.seed_lo = (uint32_t)cfg->seed; kp.seed_hi = (uint32_t)(cfg->seed >> 32);
    kp.env_base = (uint32_t)cfg->env_id_base;
    {   // snr = K * (d + eps)^(-gamma): gamma = c2/10, K = snr(1 m)   (station.py:110-127)
        double c2 = 44.9 - 6.55 * std::log10(50.0);
        {   // the same two constants as ref_snr, for the kernel's FP64 max-cap rate key
            const double f = 2500.0, hb = 50.0, hu = 1.5;
            const double ch = 0.8 + (1.1 * std::log10(f) - 0.7) * hu - 1.56 * std::log10(f);
            kp.pl_c1 = 69.55 + 26.16 * std::log10(f) - 13.82 * std::log10(hb) - ch;
            kp.pl_c2 = c2;
        }
        kp.half_gamma = (float)(c2 / 20.0);
        kp.log2k = (float)(std::log2(ref_snr(1.0)) + (double)dcomp::L2_OFF);
        kp.log2k_s = (float)(std::log2(ref_snr(1.0)) + (double)dcomp::L2_OFF - 12.0 * (c2 / 20.0));
        kp.dt2 = dcomp_connect_boundary_sq();       // X: q < X  <=>  snr(sqrt(q)) > 2e-8 in the reference's own arithmetic
        {   // the kernels decide on a FUSED d^2 and redo a pair in the reference's form when (float)fused == (float)X (dcomp_device.h, dist_sq_ref):
            // sound iff every double within 4 ulp of X has the float image of X -- else every pair takes the reference form
            kp.dt2f = (float)kp.dt2;
            double lo = kp.dt2, hi = kp.dt2;
            for (int k = 0; k < 4; k++) { lo = std::nextafter(lo, 0.0); hi = std::nextafter(hi, INFINITY); }
            kp.dsq_exact = ((float)lo == kp.dt2f && (float)hi == kp.dt2f) ? 0u : 1u;
            if (getenv("DCOMP_DSQ_EXACT") && atoi(getenv("DCOMP_DSQ_EXACT")) != 0) kp.dsq_exact = 1u;      // tests: the always-exact path
        }
        if (!(kp.dt2 > 0.0)) { delete env; return fail(DCOMP_EUNSUPPORTED, "the host's log10 / pow make snr(d) non-monotone around the connect threshold: d < d_T would not be the reference's decision"); }
    }
    for (unsigned v = 0; v < 256; v++) {
        // the kernel's closed form of qmax(v) = max{q : sqrt_rn(q) <= v} (move_ue) against brute force
        double q = (double)(v * v);
        while (std::sqrt(std::nextafter(q, INFINITY)) <= (double)v) q = std::nextafter(q, INFINITY);
        double cf = (double)(v * v);
        if (v > 0) {
            int e2 = 2 * (31 - __builtin_clz(v));
            if (v * v < (2u << e2)) cf += std::ldexp(1.0, e2 - 52);
        }
        if (cf != q) { delete env; return fail(DCOMP_EUNSUPPORTED, "host sqrt is not IEEE-correct: qmax(%u) mismatch", v); }
    }
    kp.any_maxcap = 0;
    for (int b = 0; b < B; b++) {
        int m = cfg->bs_sharing[b];
        if (m < 0 || m > 3) { delete env; return fail(DCOMP_EINVAL, "bs_sharing[%d]=%d not supported", b, m); }   // station.py:22
        if (b < DCOMP_MASK32_MAX_BS) { kp.bs_x[b] = cfg->bs_x[b]; kp.bs_y[b] = cfg->bs_y[b]; kp.bs_mode[b] = m; }
        if (m == DCOMP_MAX_CAP) { kp.any_maxcap = 1; if (b < 32) kp.maxcap_mask |= 1u << b; env->bigp.maxcap_mask |= 1ull << b; }
        if (m == DCOMP_RATE_FAIR || m == DCOMP_PROP_FAIR) { kp.any_sum_mode = 1; env->bigp.summode_mask |= 1ull << b; }
    }
    std::vector<UeCfg> uc(U);
    kp.all_log_util = 1;
    for (int u = 0; u < U; u++) {
        UeCfg c;
        std::memset(&c, 0, sizeof(c));
        int lo = cfg->ue_vel_lo[u], hi = cfg->ue_vel_hi[u];
        if (lo < 0 || hi < lo || hi > 255) { delete env; return fail(DCOMP_EINVAL, "ue %d: velocity range [%d,%d] invalid", u, lo, hi); }
        c.vel_lo = (uint8_t)lo; c.vel_hi = (uint8_t)hi;
        c.init_x = cfg->ue_init_x ? (int16_t)cfg->ue_init_x[u] : (int16_t)-1;
        c.init_y = cfg->ue_init_y ? (int16_t)cfg->ue_init_y[u] : (int16_t)-1;
        c.util = cfg->ue_util ? (uint8_t)cfg->ue_util[u] : (uint8_t)DCOMP_UTIL_LOG;
        if (c.util > DCOMP_UTIL_STEP) { delete env; return fail(DCOMP_EUNSUPPORTED, "ue %d: utility %d not implemented", u, (int)c.util); }   // user.py:92
        if (c.util != DCOMP_UTIL_LOG) kp.all_log_util = 0;
        c.dr_req = cfg->ue_dr_req ? cfg->ue_dr_req[u] : 1.0f;
        c.pause = (uint8_t)(cfg->ue_pause_duration ? cfg->ue_pause_duration[u] : 2);
        c.border = (uint8_t)(cfg->ue_border_buffer ? cfg->ue_border_buffer[u] : 10);
        uc[u] = c;
    }
    hipError_t e = hipSetDevice(cfg->device);
    if (e == hipSuccess) e = hipMalloc((void **)&env->d_ue_cfg, sizeof(UeCfg) * U);
    if (e == hipSuccess) e = hipMemcpy(env->d_ue_cfg, uc.data(), sizeof(UeCfg) * U, hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete env; return fail(DCOMP_EHIP, "device setup failed: %s", hipGetErrorString(e)); }
    kp.ue_cfg = env->d_ue_cfg;
    if (env->big) {
        std::vector<double> xy(2 * (size_t)B);
        std::vector<int32_t> md(B);
        for (int b = 0; b < B; b++) { xy[2 * b] = cfg->bs_x[b]; xy[2 * b + 1] = cfg->bs_y[b]; md[b] = cfg->bs_sharing[b]; }
        e = hipMalloc((void **)&env->d_bs, sizeof(double) * 2 * B);
        if (e == hipSuccess) e = hipMalloc((void **)&env->d_mode, sizeof(int32_t) * B);
        if (e == hipSuccess) e = hipMemcpy(env->d_bs, xy.data(), sizeof(double) * 2 * B, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(env->d_mode, md.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice);
        env->big_lds = (size_t)dcomp::big_carve(B, env->bigk.gpb, env->bigk.block).total;
        for (int pol = 0; pol < 2; pol++)
            for (int c = 0; c < 2; c++)
                for (int w = 0; w < 3; w++)                        // the step this env launches (plain or with events), the reset, the fused rollout
                    if (e == hipSuccess) e = raise_lds_limit(reinterpret_cast<const void *>(env->bigk.fn[pol][c][w == 0 ? env->big_step : w == 1 ? 1 : 3]), (int)env->big_lds);
        if (e != hipSuccess) { dcomp_destroy(env); return fail(DCOMP_EHIP, "device setup failed (generic kernel, %zu bytes of LDS per workgroup): %s", env->big_lds, hipGetErrorString(e)); }
        env->bigp.bs = env->d_bs; env->bigp.mode = env->d_mode; env->bigp.B = B;
        // multi-agent rows of more than 32 stations: one 16-byte store per lane and row (dcomp_big.h, row loop) once a step's rows no longer sit in
        // the Infinity Cache (measured at 32 x 64: 2 048 envs = 67 MB 41.4 -> 43.1 us, 8 192 = 270 MB 63.9 -> 62.5, 16 384 = 539 MB 196 -> 145,
        // 65 536 = 2.2 GB 693 -> 495); DCOMP_BIG_ROW_X4=0 / 1 forces the four-block form / the 16-byte form
        env->bigp.row_x4 = getenv("DCOMP_BIG_ROW_X4") ? atoi(getenv("DCOMP_BIG_ROW_X4")) : ((size_t)E * CAP * (4 * B + 1) * 4 >= ((size_t)128 << 20) ? 1 : 0);
    }
    if (cfg->ue_velocity) {
        // movement.py:116-117 / 142-156 with a velocity that is no integer in 0..255: the device takes {v, qmax(v)} from a table,
        // qmax(v) = largest double q with sqrt_rn(q) <= v -- `distance <= velocity` without a square root, as for the integers
        std::vector<double> vq(2 * (size_t)U, -1.0);
        bool any = false;
        for (int u = 0; u < U; u++) {
            const double v = cfg->ue_velocity[u];
            if (!(v >= 0.0)) continue;
            if (!std::isfinite(v) || v > 1e6) { dcomp_destroy(env); return fail(DCOMP_EINVAL, "ue %d: velocity %g out of range", u, v); }
            if (cfg->ue_vel_lo[u] != cfg->ue_vel_hi[u]) { dcomp_destroy(env); return fail(DCOMP_EINVAL, "ue %d: ue_velocity needs a fixed range (lo == hi)", u); }
            double q = v * v;
            while (q > 0.0 && std::sqrt(q) > v) q = std::nextafter(q, 0.0);
            while (std::sqrt(std::nextafter(q, INFINITY)) <= v) q = std::nextafter(q, INFINITY);
            vq[2 * u] = v; vq[2 * u + 1] = q;
            any = true;
        }
        if (any) {
            e = hipMalloc((void **)&env->d_ue_velq, sizeof(double) * 2 * U);
            if (e == hipSuccess) e = hipMemcpy(env->d_ue_velq, vq.data(), sizeof(double) * 2 * U, hipMemcpyHostToDevice);
            if (e != hipSuccess) { dcomp_destroy(env); return fail(DCOMP_EHIP, "device setup failed: %s", hipGetErrorString(e)); }
            kp.ue_velq = env->d_ue_velq;
        }
    }
    if (!env->big && env->kern.step_wide && !kp.any_maxcap && !getenv("DCOMP_NO_WIDE")) {
        env->kern.step = env->kern.step_wide;
        // persistent launch: as many workgroups as are resident at once (LDS-bound: 4 per CU at B = 32), never more than there are slots
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(env->kern.step_wide), DCOMP_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || cus < 1) cus = 256;
        long cap = DCOMP_WIDE_PERSIST ? (long)per_cu * cus : (long)env->grid;      // (one workgroup per slot unless the persistent experiment is built)
        if (const char *e = getenv("DCOMP_WIDE_GRID")) cap = (DCOMP_WIDE_PERSIST && atol(e) > 0) ? atol(e) : env->grid;
        env->wide_grid = (int)(cap < env->grid ? cap : env->grid);
        // Batches whose observation rows are >= 1 GB per step -- four times the Infinity Cache: every byte goes out at the sustained HBM
        // write rate -- run with TWO workgroups per CU instead of the four the kernel's 39.8 KB of LDS allow (24 000 B of unused dynamic
        // LDS per workgroup): 16 384 x 128 x 32 252-272 -> 251 us, 32 768 envs 517 -> 477 us (0.60 -> 0.65 of the HBM peak), repeatably;
        // at 8 192 envs (0.54 GB) the same cap costs 6 %, at 4 096 envs (inside the cache) 25 %.  Fewer waves writing at once stream better.
        const double row_bytes = (double)E * CAP * (cfg->env_kind == DCOMP_MULTI ? 4 * B + 1 : 2 * B + 1) * 4.0;
        // Between 0.4 and 1 GB three per CU (13 000 B) are the best of the three: 8 192 envs 141 -> 134 us, 6 144 envs 101.5 -> 98.5 us.
        env->wide_pad_lds = row_bytes >= 1.0e9 ? 24000 : row_bytes >= 4.0e8 ? 13000 : 0;
        if (const char *e = getenv("DCOMP_WIDE_PAD_LDS")) env->wide_pad_lds = atoi(e);                 // A/B
    }
    if (DYN && !env->big) {
        if (!env->kern.step_dyn) { dcomp_destroy(env); return fail(DCOMP_EUNSUPPORTED, "no dynamic-UE kernel for this shape"); }
        env->kern.step = env->kern.step_dyn;
    }
    {
        // The fused rollout kernel is the LATENCY-optimised variant (state in registers, no LDS staging, no kernel boundaries):
        // it wins while the grid leaves the SIMDs under-occupied (4 096 x 10 x 5: one wave per SIMD, 2.2 vs 5.3 us per step).  A
        // grid of many waves per SIMD is throughput-bound: there the plain step kernel (fewer registers, coalesced staged
        // stores) launched once per step is faster and launch latency hides behind the running kernel.
        long max_waves = 3 * 1024;                                   // 3 waves per SIMD of the 256 CUs: what the rollout kernel's ~136 VGPRs leave room for
                                                                     // (a fourth wave per SIMD would wait for a second round)
        if (const char *e = getenv("DCOMP_FUSE_MAX_WAVES")) max_waves = atol(e);
        const long waves = (long)env->grid * (DCOMP_BLOCK / 64);
        env->fused = !DYN && !env->big && env->kern.step != env->kern.step_wide && env->kern.rollout != nullptr && waves <= max_waves;
        // Central envs with short rows (2B + 1 <= 17 floats per UE) are faster through the fused kernel at EVERY batch size once a
        // rollout is a few steps long: no kernel boundary (5.4 us per launch, a quarter of a 65 536 x 10 x 5 step), no state round
        // trip, pairs carried from step to step -- 65 536 x 10 x 5: 16.3 vs 19-20 us per step, 16 384: 4.2 vs 11.2, 262 144: 58 vs
        // 63 (T = 50).  Rows of the multi-agent layout (4B + 1 floats per lane, stored straight from registers) stream too
        // badly for that: 65 536 x 32 x 10: 129 vs 77 us.
        // Multi-agent rows (4B + 1 floats per lane) are short enough up to three stations -- the reference's stock small and medium
        // maps: 65 536 x 5 x 3 multi 8.1 vs 12.1 us per step (T = 50); at 10 x 5 it is a draw (23.4 vs 23.3), at 16 x 8 a loss (51.8 vs 38.2).
        env->fused_long = !DYN && !env->big && env->kern.step != env->kern.step_wide && env->kern.rollout != nullptr &&
                          ((cfg->env_kind == DCOMP_CENTRAL && B <= 8) || (cfg->env_kind == DCOMP_MULTI && B <= 3)) && !getenv("DCOMP_NO_FUSED_LONG");
        // Tight packing of UE lists whose length is not a power of two (dcomp_device.h, struct Seg): G = U lanes per env,
        // 64 / G envs per wavefront, segmented ds_bpermute reductions (~13 instead of 4 instructions each).  It pays where the
        // launch is throughput-bound and the padding wastes many lanes: >= 4 padded waves per SIMD and >= 1.4x the lanes in use
        // (U = 5, 9, 10, 17-21; measured: 10 x 5 central +10 %, 20 x 10 multi +8 %, 5 x 3 central +14 %, but 12 x 7 multi -2 % at 1.25x).
        // Not with max-cap BSs (their LDS scratch is indexed by padded env slots), not for UE lists that change.
        env->tight_g = 0;
        const int U1 = CAP;
        if (!DYN && env->kern.step_tight && U1 >= 3 && (U1 & (U1 - 1)) != 0 && !kp.any_maxcap && env->kern.step != env->kern.step_wide) {
            const int gpw = 64 / U1;
            const double padded_use = (double)U1 / env->upad, tight_use = (double)(gpw * U1) / 64.0;
            int want = waves >= 4 * 1024 && tight_use >= 1.4 * padded_use;     // (never together with the fused rollout: <= 3 * 1024 waves)
            if (const char *e = getenv("DCOMP_TIGHT")) want = atoi(e) != 0;       // tests / A-B: force on or off
            if (want) {
                const int magic = 65536 / U1 + 1;
                bool ok = true;
                for (int l = 0; l < 64; l++) if (((l * magic) >> 16) != l / U1) ok = false;
                if (ok) {
                    env->tight_g = U1; env->tight_gpw = gpw; env->tight_magic = magic;
                    const int epb = gpw * (DCOMP_BLOCK / 64);
                    env->tight_grid = (E + epb - 1) / epb;
                }
            }
        }
    }
    *out = env;
    return DCOMP_OK;
}

extern "C" int dcomp_destroy(dcomp_env *env)
{
    if (!env) return DCOMP_OK;
    if (env->d_ue_cfg) (void)hipFree(env->d_ue_cfg);
    if (env->d_ue_velq) (void)hipFree(env->d_ue_velq);
    if (env->d_bs) (void)hipFree(env->d_bs);
    if (env->d_mode) (void)hipFree(env->d_mode);
    delete env;
    return DCOMP_OK;
}

extern "C" int dcomp_state_sizes(const dcomp_env *env, size_t *pos_bytes, size_t *mv_bytes, size_t *conn_bytes, size_t *ewma_bytes,
                                 size_t *flags_bytes, size_t *since_bytes)
{
    if (!env) return fail(DCOMP_EINVAL, "null env");
    size_t n = (size_t)env->cfg.num_envs * env->cap;
    if (pos_bytes) *pos_bytes = n * 16;
    if (mv_bytes) *mv_bytes = n * 8;
    if (conn_bytes) *conn_bytes = n * 4;
    if (ewma_bytes) *ewma_bytes = n * 4;
    if (flags_bytes) *flags_bytes = 16;
    if (since_bytes) *since_bytes = env->kp.any_maxcap ? n * env->cfg.num_bs * 2 : 0;
    return DCOMP_OK;
}

extern "C" int dcomp_obs_dim(const dcomp_env *env, int32_t *floats_per_env, int32_t *reward_per_env)
{
    if (!env) return fail(DCOMP_EINVAL, "null env");
    const int U = env->cap, B = env->cfg.num_bs;
    if (floats_per_env) *floats_per_env = env->cfg.env_kind == DCOMP_MULTI ? U * (4 * B + 1) : U * (2 * B + 1);
    if (reward_per_env) *reward_per_env = env->cfg.env_kind == DCOMP_MULTI ? U : 1;
    return DCOMP_OK;
}

// The generic kernel's instantiation for this launch: with / without the in-step policy, row format / compact record.
static dcomp::BigKernelFn big_fn(const dcomp_env *env, const KParams &kp, int which)
{
    return env->bigk.fn[kp.next_act ? 1 : 0][kp.obs_compact ? 1 : 0][which];
}

// One launch of the step kernel (plain step; also the per-step launches of a rollout that is not fused).
static void launch_step(dcomp_env *env, KParams &kp, void *stream)
{
    if (env->big) {
        hipLaunchKernelGGL(big_fn(env, kp, env->big_step), dim3(env->grid), dim3(env->bigk.block), env->big_lds, (hipStream_t)stream, kp, env->bigp);
        return;
    }
    if (env->tight_g) {
        kp.tight_g = env->tight_g; kp.tight_gpw = env->tight_gpw; kp.tight_magic = env->tight_magic;
        const dcomp::KernelFn k = (env->cfg.env_kind == DCOMP_CENTRAL && env->kern.tight_central) ? env->kern.tight_central : env->kern.step_tight;
        hipLaunchKernelGGL(k, dim3(env->tight_grid), dim3(DCOMP_BLOCK), 0, (hipStream_t)stream, kp);
        return;
    }
    const bool wide = env->wide_grid && env->kern.step == env->kern.step_wide;
    const int grid = wide ? env->wide_grid : env->grid;
    // (the occupancy cap is for launches that stream GBs of ROWS; with the compact record a launch is bound by its arithmetic and wants
    // every wave it can get: 32 768 x 128 x 32 236 us with four workgroups per CU, 369 us with two)
    hipLaunchKernelGGL(env->kern.step, dim3(grid), dim3(DCOMP_BLOCK), wide && !kp.obs_compact ? env->wide_pad_lds : 0, (hipStream_t)stream, kp);
}

static int check_horizon(const dcomp_env *env, int steps)
{
    if (env && (int64_t)env->time + steps > 65536)
        return fail(DCOMP_EUNSUPPORTED, "episodes longer than 65536 steps: conn_since and the draw cursor are 16-bit (reset() first)");
    return DCOMP_OK;
}

static int fill_params(dcomp_env *env, const dcomp_state *st, const dcomp_out *out, KParams &kp)
{
    if (!env || !st || !out) return fail(DCOMP_EINVAL, "null argument");
    if (!st->pos || !st->mv || !st->conn || !st->ewma || !st->flags) return fail(DCOMP_EINVAL, "state pointers must all be set");
    if (!out->obs && !out->obs_compact) return fail(DCOMP_EINVAL, "out->obs (or out->obs_compact) is required");
    if (out->obs_compact) {
        if (out->obs) return fail(DCOMP_EINVAL, "out->obs and out->obs_compact are alternatives: set one, leave the other NULL");
        if (env->cfg.env_kind != DCOMP_MULTI) return fail(DCOMP_EINVAL, "out->obs_compact: the compact record is defined for multi-agent observations (central observations carry no per-env columns)");
    }
    if (env->kp.any_maxcap && !st->conn_since) return fail(DCOMP_EINVAL, "a max-cap BS needs state.conn_since (see dcomp_state_sizes)");
    if (env->big) {
        if (!st->conn_hi) return fail(DCOMP_EINVAL, "more than %d stations: state.conn_hi (stations 32-63 of the connection set, sized like conn) is required", DCOMP_MASK32_MAX_BS);
        env->bigp.conn_hi = st->conn_hi;
    }
    if (env->dyn && !st->uid) return fail(DCOMP_EINVAL, "UE arrival/departure needs state.uid");
    kp = env->kp;
    kp.uid = st->uid;
    kp.orig_consumed = st->orig_consumed;
    kp.cur_ue = env->cur_ue;
    kp.conn_since = st->conn_since;
    kp.time = (uint32_t)env->time;
    kp.pos = (double2 *)st->pos; kp.mv = (unsigned long long *)st->mv; kp.conn = st->conn; kp.ewma = st->ewma; kp.flags = st->flags;
    kp.obs = out->obs_compact ? reinterpret_cast<float *>(out->obs_compact) : out->obs; kp.obs_compact = out->obs_compact ? 1 : 0;
    kp.reward = out->reward; kp.sum_util = out->sum_utility; kp.ue_dr = out->ue_dr; kp.ue_util = out->ue_utility; kp.rb_out = out->reward_before;
    kp.episode = (uint32_t)(env->episode < 0 ? 0 : env->episode);
    kp.num_steps = 1; kp.out_every_step = 0; kp.horizon = 0; kp.episode_inc = 0;
    kp.tight_g = 0; kp.tight_gpw = 0; kp.tight_magic = 0;
    return DCOMP_OK;
}

extern "C" int dcomp_reset(dcomp_env *env, const dcomp_state *st, const dcomp_tape *tape, const dcomp_out *out, void *stream)
{
    KParams kp;
    int rc = fill_params(env, st, out, kp);
    if (rc) return rc;
    if (env->cfg.rng_mode == DCOMP_RNG_TAPE) {
        if (!tape || !tape->pos0 || !tape->triples) return fail(DCOMP_EINVAL, "tape mode: reset needs the episode's draw tape");
        env->kp.tape_pos0 = tape->pos0;                     // borrowed until the next reset
        env->kp.tape_triples = (const ushort4 *)tape->triples;
        env->kp.tape_ids = tape->num_ids > 0 ? tape->num_ids : env->cfg.num_ue;
        kp.tape_pos0 = env->kp.tape_pos0; kp.tape_triples = env->kp.tape_triples; kp.tape_ids = env->kp.tape_ids;
    }
    env->episode += 1;
    env->time = 0;
    env->cur_ue = env->cfg.num_ue; env->n_removed = env->n_arrived = 0;      // base.py:177-182
    kp.cur_ue = env->cur_ue;
    kp.episode = (uint32_t)env->episode;
    if (env->big) hipLaunchKernelGGL(big_fn(env, kp, 1), dim3(env->grid), dim3(env->bigk.block), env->big_lds, (hipStream_t)stream, kp, env->bigp);
    else hipLaunchKernelGGL(env->kern.reset, dim3(env->grid), dim3(DCOMP_BLOCK), 0, (hipStream_t)stream, kp);
    HIP_TRY(hipGetLastError());
    return DCOMP_OK;
}

extern "C" int dcomp_step(dcomp_env *env, const dcomp_state *st, const uint8_t *action, const dcomp_out *out, void *stream)
{
    KParams kp;
    int rc = fill_params(env, st, out, kp);
    if (rc) return rc;
    if (!action) return fail(DCOMP_EINVAL, "null action");
    if (env->episode < 0) return fail(DCOMP_EINVAL, "step() before reset()");
    if ((rc = check_horizon(env, 1))) return rc;
    kp.action = action;
    kp.n_remove = kp.n_add = 0;
    launch_step(env, kp, stream);
    HIP_TRY(hipGetLastError());
    env->time += 1;
    return DCOMP_OK;
}

extern "C" int dcomp_step_dyn(dcomp_env *env, const dcomp_state *st, const uint8_t *action, const dcomp_out *out,
                              const dcomp_events *ev, void *stream)
{
    if (!env) return fail(DCOMP_EINVAL, "null env");
    if (!env->dyn) {
        if (ev && (ev->n_remove || ev->n_add)) return fail(DCOMP_EINVAL, "handle was created without max_ues (fixed UE list)");
        return dcomp_step(env, st, action, out, stream);
    }
    KParams kp;
    int rc = fill_params(env, st, out, kp);
    if (rc) return rc;
    if (!action) return fail(DCOMP_EINVAL, "null action");
    if (env->episode < 0) return fail(DCOMP_EINVAL, "step() before reset()");
    if ((rc = check_horizon(env, 1))) return rc;
    const int nrem = ev ? ev->n_remove : 0, nadd = ev ? ev->n_add : 0;
    if (nrem < 0 || nadd < 0 || (nrem > 0 && nadd > 0)) return fail(DCOMP_EINVAL, "one step either adds or removes UEs (base.py:436-443)");
    if (env->cur_ue - nrem < 1) return fail(DCOMP_EINVAL, "cannot remove %d of %d UEs", nrem, env->cur_ue);
    if (env->cur_ue + nadd > env->cap) return fail(DCOMP_EINVAL, "%d + %d UEs exceed max_ues = %d", env->cur_ue, nadd, env->cap);
    if (env->cfg.rng_mode == DCOMP_RNG_TAPE && ((nrem && !ev->remove_idx) || (nadd && !ev->add_xy)))
        return fail(DCOMP_EINVAL, "tape mode: events need the host-drawn indices / border points");
    kp.action = action;
    kp.n_remove = nrem; kp.n_add = nadd;
    kp.ev_remove = ev ? ev->remove_idx : nullptr; kp.ev_add_xy = ev ? ev->add_xy : nullptr;
    kp.ev_rem_base = env->n_removed; kp.ev_add_base = env->n_arrived;
    if (env->big) hipLaunchKernelGGL(big_fn(env, kp, env->big_step), dim3(env->grid), dim3(env->bigk.block), env->big_lds, (hipStream_t)stream, kp, env->bigp);
    else hipLaunchKernelGGL(env->kern.step, dim3(env->grid), dim3(DCOMP_BLOCK), 0, (hipStream_t)stream, kp);
    HIP_TRY(hipGetLastError());
    env->time += 1;
    env->cur_ue += nadd - nrem;
    env->n_removed += (uint32_t)nrem; env->n_arrived += (uint32_t)nadd;
    return DCOMP_OK;
}
extern "C" int dcomp_num_ue(const dcomp_env *env) { return env ? env->cur_ue : -1; }

// T consecutive steps.  step_kernel runs them in ONE launch (state in registers in between); the wide and the dynamic-UE
// kernels are launched once per step.  Same results either way, and the same as T dcomp_step / dcomp_step_dyn calls (+
// dcomp_reset calls at the horizon).
static int rollout_impl(dcomp_env *env, const dcomp_state *st, const uint8_t *actions, int32_t T, const dcomp_out *out,
                        const dcomp_rollout_opts *opts, void *stream)
{
    KParams kp;
    int rc = fill_params(env, st, out, kp);
    if (rc) return rc;
    if (!actions || T < 1) return fail(DCOMP_EINVAL, "bad action tape");
    if (env->episode < 0) return fail(DCOMP_EINVAL, "rollout() before reset()");
    const int L = opts ? opts->horizon : 0, every = opts ? (opts->every_step != 0) : 0;
    const uint32_t inc = (opts && opts->new_episode_draws) ? 1u : 0u;
    const int32_t *ev_rem = opts ? opts->ev_n_remove : nullptr, *ev_add = opts ? opts->ev_n_add : nullptr;
    const bool tape = env->cfg.rng_mode == DCOMP_RNG_TAPE;
    if (L < 0 || L > 65536) return fail(DCOMP_EINVAL, "horizon must be 0 (none) .. 65536");
    if (L > 0) {
        if (env->time > L) return fail(DCOMP_EINVAL, "env.time %d is already beyond the horizon %d", env->time, L);
        if (tape && inc) return fail(DCOMP_EINVAL, "tape mode replays the borrowed tape: only fixed episodes can reset inside a rollout");
        if (tape && env->dyn) return fail(DCOMP_EUNSUPPORTED, "tape mode with UE arrival / departure: every episode needs its own host-drawn tape "
                                                              "(the reference re-seeds UEs by list position at reset, base.py:171-173); dcomp_reset between rollouts");
    } else if ((rc = check_horizon(env, T))) return rc;
    if (!env->dyn && (ev_rem || ev_add)) return fail(DCOMP_EINVAL, "handle was created without max_ues (fixed UE list): no arrival / departure events");
    const bool loop = opts && opts->policy_loop != 0;
    // (round 6: the closed loop runs on every kernel -- where rollouts are not fused, one launch per step, each reading the actions the
    //  previous one wrote)
    if (loop && !env->kp.next_act) return fail(DCOMP_EINVAL, "policy_loop needs a policy (dcomp_set_policy)");
    const size_t EU = (size_t)env->cfg.num_envs * env->cap, E = (size_t)env->cfg.num_envs;
    const bool multi = env->cfg.env_kind == DCOMP_MULTI;
    const size_t obs_step = EU * (size_t)(multi ? 4 * env->cfg.num_bs + 1 : 2 * env->cfg.num_bs + 1);
    auto out_slice = [&](KParams &k, int t) {                  // where step t's outputs go
        if (!every) return;
        k.obs = out->obs_compact ? reinterpret_cast<float *>(out->obs_compact) + E * (size_t)dcomp_frag::env_words(env->cap, env->cfg.num_bs) * t
                                 : out->obs + obs_step * t;
        if (out->reward) k.reward = out->reward + (multi ? EU : E) * t;
        if (out->sum_utility) k.sum_util = out->sum_utility + E * t;
        if (out->ue_dr) k.ue_dr = out->ue_dr + EU * t;
        if (out->ue_utility) k.ue_util = out->ue_utility + EU * t;
        if (out->reward_before) k.rb_out = out->reward_before + EU * t;
    };
    auto launch_reset = [&](KParams &k) {                      // MobileEnv.reset at the horizon (base.py:169-189); its observation lands
        env->time = 0;                                         // where the next step's will (and is overwritten by it)
        env->episode += tape ? 1 : inc;                        // (tape mode: `episode` counts resets, as dcomp_reset does; the draws do not use it)
        env->cur_ue = env->cfg.num_ue; env->n_removed = env->n_arrived = 0;
        k.cur_ue = env->cur_ue;
        k.episode = (uint32_t)env->episode;
        k.time = 0u;
        if (env->big) hipLaunchKernelGGL(big_fn(env, k, 1), dim3(env->grid), dim3(env->bigk.block), env->big_lds, (hipStream_t)stream, k, env->bigp);
        else hipLaunchKernelGGL(env->kern.reset, dim3(env->grid), dim3(DCOMP_BLOCK), 0, (hipStream_t)stream, k);
    };
    if (env->dyn) {
        // The whole arrival / departure schedule of the T steps is validated BEFORE the first launch (it used to be checked step by
        // step inside the loop below: an invalid entry at step t then left the env t steps advanced and the caller's host-side
        // event streams consumed).  The call either enqueues all T steps or nothing.
        int cur = env->cur_ue, time = env->time;
        for (int t = 0; t < T; t++) {
            if (L > 0 && time == L) { time = 0; cur = env->cfg.num_ue; }
            const int nrem = ev_rem ? ev_rem[t] : 0, nadd = ev_add ? ev_add[t] : 0;
            if (nrem < 0 || nadd < 0 || (nrem > 0 && nadd > 0)) return fail(DCOMP_EINVAL, "step %d: one step either adds or removes UEs (base.py:436-443)", t);
            if (cur - nrem < 1) return fail(DCOMP_EINVAL, "step %d: cannot remove %d of %d UEs", t, nrem, cur);
            if (cur + nadd > env->cap) return fail(DCOMP_EINVAL, "step %d: %d + %d UEs exceed max_ues = %d", t, cur, nadd, env->cap);
            if (tape && ((nrem && !opts->ev_remove_idx) || (nadd && !opts->ev_add_xy)))
                return fail(DCOMP_EINVAL, "tape mode: events need the host-drawn indices / border points");
            cur += nadd - nrem;
            time += 1;
        }
    }
    // the fused kernel addresses step t's outputs as (idx + t * E * U) on the caller's base pointers with 32-bit row indices: a
    // fragment beyond that takes the one-launch-per-step path below (same results)
    const bool fits32 = !(every && (uint64_t)T * EU >= fused_row_limit());
    if (fits32 && (env->fused || (env->fused_long && (T >= 4 || loop)))) {
        // with a registered policy: the variant that carries the rules; tape-driven central envs: the central-only instantiation
        dcomp::KernelFn kern = kp.next_act ? env->kern.rollout_pol : (!multi && env->kern.rollout_central) ? env->kern.rollout_central : env->kern.rollout;
        int grid = env->grid;
        // A batch dcomp_create packs tightly for dcomp_step (throughput-bound: >= 4 padded waves per SIMD, >= 1.4x the lanes in use) runs
        // its tape-driven central rollouts tightly packed too: the fused kernel is bound by its VALU work there, and a third fewer
        // waves do the same steps (65 536 x 10 x 5: six envs per wavefront instead of four).  Same packing, same summation order as
        // dcomp_step uses for this env.
        if (env->tight_g && !multi && !kp.next_act && env->kern.rollout_tight_central && !getenv("DCOMP_NO_TIGHT_ROLLOUT")) {
            kern = env->kern.rollout_tight_central;
            grid = env->tight_grid;
            kp.tight_g = env->tight_g; kp.tight_gpw = env->tight_gpw; kp.tight_magic = env->tight_magic;
        }
        if (!loop || L == 0 || env->time + T <= L) {
            // ONE launch; resets at the horizon of a tape-driven rollout happen inside the kernel
            kp.action = actions; kp.num_steps = T; kp.out_every_step = every; kp.horizon = L; kp.episode_inc = inc; kp.policy_loop = loop;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(DCOMP_BLOCK), 0, (hipStream_t)stream, kp);
            int time = env->time;
            int64_t episode = env->episode;
            for (int t = 0; t < T; t++) {
                if (L > 0 && time == L) { time = 0; episode += tape ? 1 : inc; }
                time += 1;
            }
            env->time = time;
            env->episode = episode;
        } else {
            // Closed loop across episode boundaries: one launch per stretch of an episode; at the horizon the reset kernel computes
            // the first observation of the new episode and the policy's action on it (next_action), which the next stretch starts from.
            const uint8_t *act_src = actions;
            for (int t = 0; t < T;) {
                if (env->time == L) {
                    out_slice(kp, t);
                    launch_reset(kp);
                    act_src = kp.next_act;
                }
                const int n = T - t < L - env->time ? T - t : L - env->time;
                out_slice(kp, t);
                kp.action = act_src; kp.num_steps = n; kp.out_every_step = every; kp.horizon = 0; kp.episode_inc = 0; kp.policy_loop = 1;
                kp.time = (uint32_t)env->time; kp.episode = (uint32_t)env->episode;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(DCOMP_BLOCK), 0, (hipStream_t)stream, kp);
                env->time += n;
                t += n;
                act_src = kp.next_act;                             // (a lane reads its slot before it writes it)
            }
        }
        HIP_TRY(hipGetLastError());
        return DCOMP_OK;
    }
    // one launch per step: wide envs, envs whose UE list changes (event feed), batches too large for the latency-optimised kernel
    if (env->fused_big) {
        // The generic kernel's fused rollout (big_kernel<..., ROLL>): one launch per stretch of an episode, the UE state in registers from step to
        // step, step t's outputs in slice t of the caller's buffers; at the horizon the reset kernel (which, with a policy, also decides the first
        // action of the new episode).  64-bit offsets throughout: no row limit.
        const uint8_t *act_src = actions;
        for (int t = 0; t < T;) {
            out_slice(kp, t);
            if (L > 0 && env->time == L) { launch_reset(kp); if (loop) act_src = kp.next_act; }
            const int n = (L > 0 && L - env->time < T - t) ? L - env->time : T - t;
            kp.action = loop ? act_src : actions + EU * t;
            kp.num_steps = n; kp.out_every_step = every; kp.policy_loop = loop; kp.horizon = 0; kp.episode_inc = 0;
            kp.time = (uint32_t)env->time; kp.episode = (uint32_t)env->episode;
            hipLaunchKernelGGL(big_fn(env, kp, 3), dim3(env->grid), dim3(env->bigk.block), env->big_lds, (hipStream_t)stream, kp, env->bigp);
            env->time += n;
            t += n;
            if (loop) act_src = kp.next_act;
        }
        HIP_TRY(hipGetLastError());
        return DCOMP_OK;
    }
    // The closed loop here: step 0 acts on actions[0] (the next_action the previous launch wrote), every later step -- and the first step of a
    // new episode -- on what the launch before it decided.  next_action is ONE buffer read and written in place: a lane reads the action of
    // its own slot when the kernel starts and writes the decision for that same slot at its end (with UE arrival / departure the UEs move
    // between slots through LDS, not through this buffer).
    size_t rem_off = 0, add_off = 0;
    const uint8_t *act_src = actions;
    for (int t = 0; t < T; t++) {
        out_slice(kp, t);
        if (L > 0 && env->time == L) { launch_reset(kp); if (loop) act_src = kp.next_act; }
        kp.episode = (uint32_t)env->episode;
        kp.action = loop ? act_src : actions + EU * t;
        kp.time = (uint32_t)env->time;
        kp.n_remove = kp.n_add = 0;
        if (env->dyn) {                                            // base.py:433-443: this step's departures / arrivals
            const int nrem = ev_rem ? ev_rem[t] : 0, nadd = ev_add ? ev_add[t] : 0;       // (validated above, all T steps)
            kp.cur_ue = env->cur_ue;
            kp.n_remove = nrem; kp.n_add = nadd;
            kp.ev_remove = (tape && nrem) ? opts->ev_remove_idx + rem_off : nullptr;
            kp.ev_add_xy = (tape && nadd) ? opts->ev_add_xy + add_off : nullptr;
            kp.ev_rem_base = env->n_removed; kp.ev_add_base = env->n_arrived;
            rem_off += E * (size_t)nrem; add_off += E * (size_t)nadd * 2;
            env->cur_ue += nadd - nrem;
            env->n_removed += (uint32_t)nrem; env->n_arrived += (uint32_t)nadd;
        }
        launch_step(env, kp, stream);
        env->time += 1;
        if (loop) act_src = kp.next_act;
    }
    HIP_TRY(hipGetLastError());
    return DCOMP_OK;
}

extern "C" int dcomp_rollout(dcomp_env *env, const dcomp_state *st, const uint8_t *actions, int32_t num_steps, const dcomp_out *out,
                             void *stream)
{
    return rollout_impl(env, st, actions, num_steps, out, nullptr, stream);
}

extern "C" int dcomp_rollout_ex(dcomp_env *env, const dcomp_state *st, const uint8_t *actions, int32_t num_steps, const dcomp_out *out,
                                const dcomp_rollout_opts *opts, void *stream)
{
    return rollout_impl(env, st, actions, num_steps, out, opts, stream);
}

extern "C" int dcomp_rollout_is_fused(const dcomp_env *env) { return env ? ((env->fused || env->fused_long || env->fused_big) ? 1 : 0) : -1; }
// Whether THIS rollout is one launch: fusion of the short-row shapes (fused_long) depends on the number of steps -- a rollout of
// fewer than 4 tape-driven steps goes out as one launch per step, as does an every-step fragment of >= 2^31 rows.
extern "C" int dcomp_rollout_fused_for(const dcomp_env *env, int32_t num_steps, int32_t every_step, int32_t policy_loop)
{
    if (!env || num_steps < 1) return -1;
    const uint64_t EU = (uint64_t)env->cfg.num_envs * env->cap;
    if (env->fused_big) return 1;                                  // (one launch per stretch of an episode; 64-bit offsets: no row limit)
    if (every_step && (uint64_t)num_steps * EU >= fused_row_limit()) return 0;
    return (env->fused || (env->fused_long && (num_steps >= 4 || policy_loop))) ? 1 : 0;
}
// 1: this env runs on the generic kernel, whose connection sets take two words per UE -- dcomp_state.conn_hi is REQUIRED (dcomp_reset / dcomp_step
// fail with DCOMP_EINVAL without it); 0: the specialised kernels, conn_hi is ignored.  The caller asks instead of re-deriving the rule.
extern "C" int dcomp_needs_conn_hi(const dcomp_env *env) { return env ? (env->big ? 1 : 0) : -1; }
extern "C" int dcomp_lanes_per_env(const dcomp_env *env) { return env ? (env->tight_g ? env->tight_g : env->upad) : -1; }

// The instantiation dcomp_step launches for this env, spelled as rocprofv3 prints it ("step_kernel<10, 32, 2>"): bench.py ties a
// tracked --pmc profile to the kernel the library really dispatches to.
extern "C" int dcomp_step_kernel_name(const dcomp_env *env, char *buf, int32_t len)
{
    if (!env || !buf || len < 1) return fail(DCOMP_EINVAL, "null argument");
    const int B = env->cfg.num_bs, W = env->upad, MP = env->mp_pattern;
    if (env->big) std::snprintf(buf, (size_t)len, "big_kernel<%d, false, %s, false, %s, false>", W < 4 ? 4 : W, env->dyn ? "true" : "false", env->kp.next_act ? "true" : "false");
    else if (env->tight_g) {
        const bool cen = env->cfg.env_kind == DCOMP_CENTRAL && env->kern.tight_central;
        std::snprintf(buf, (size_t)len, "step_kernel_tight<%d, %d, %d, %d>", B, W, MP, cen ? 0 : -1);
    } else if (env->dyn) std::snprintf(buf, (size_t)len, "step_kernel_dyn<%d, %d, %d>", B, W, MP);
    else if (env->kern.step == env->kern.step_wide && env->kern.step_wide) std::snprintf(buf, (size_t)len, "step_kernel_wide<%d, %d, %d>", B, W, MP);
    else std::snprintf(buf, (size_t)len, "step_kernel<%d, %d, %d>", B, W, MP);
    return DCOMP_OK;
}

extern "C" int dcomp_check(dcomp_env *env, const dcomp_state *st, void *stream)
{
    if (!env || !st || !st->flags) return fail(DCOMP_EINVAL, "null argument");
    uint32_t h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, st->flags, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (h[0]) HIP_TRY(hipMemsetAsync(st->flags, 0, sizeof(h), (hipStream_t)stream));
    if (h[0] & DCOMP_FLAG_BAD_ACTION) return fail(DCOMP_EACTION, "an action does not fit the action space [0, %d]", env->cfg.num_bs);
    if (h[0] & DCOMP_FLAG_TAPE_EMPTY) return fail(DCOMP_ETAPE, "waypoint draw tape exhausted (depth %d)", env->cfg.tape_depth);
    if (h[0] & DCOMP_FLAG_OUTSIDE_MAP) return fail(DCOMP_EPOS, "a UE position is outside the map");
    return DCOMP_OK;
}

extern "C" int dcomp_time(const dcomp_env *env) { return env ? env->time : -1; }
extern "C" int64_t dcomp_episode(const dcomp_env *env) { return env ? env->episode : -1; }
extern "C" int dcomp_get_counters(const dcomp_env *env, int64_t out[5])
{
    if (!env || !out) return fail(DCOMP_EINVAL, "null argument");
    out[0] = env->time; out[1] = env->episode; out[2] = env->cur_ue; out[3] = env->n_removed; out[4] = env->n_arrived;
    return DCOMP_OK;
}
extern "C" int dcomp_set_counters(dcomp_env *env, const int64_t in[5])
{
    if (!env || !in) return fail(DCOMP_EINVAL, "null argument");
    if (in[0] < 0 || in[1] < -1 || in[2] < 1 || in[2] > env->cap || in[3] < 0 || in[4] < 0) return fail(DCOMP_EINVAL, "counters out of range");
    if (!env->dyn && (in[2] != env->cfg.num_ue || in[3] || in[4])) return fail(DCOMP_EINVAL, "fixed UE list: num_ue / event counters cannot change");
    env->time = (int)in[0]; env->episode = in[1]; env->cur_ue = (int)in[2];
    env->n_removed = (uint32_t)in[3]; env->n_arrived = (uint32_t)in[4];
    return DCOMP_OK;
}
extern "C" int dcomp_set_seed(dcomp_env *env, uint64_t seed)
{
    if (!env) return fail(DCOMP_EINVAL, "null env");
    env->cfg.seed = seed;
    env->kp.seed_lo = (uint32_t)seed; env->kp.seed_hi = (uint32_t)(seed >> 32);
    return DCOMP_OK;
}
extern "C" int dcomp_set_tape(dcomp_env *env, const dcomp_tape *tape, int32_t depth)
{
    if (!env || !tape || !tape->pos0 || !tape->triples) return fail(DCOMP_EINVAL, "null argument");
    if (env->cfg.rng_mode != DCOMP_RNG_TAPE) return fail(DCOMP_EINVAL, "handle draws with Philox: it has no tape");
    if (depth < env->cfg.tape_depth) return fail(DCOMP_EINVAL, "the replacement tape (depth %d) must contain the current one (depth %d)", depth, env->cfg.tape_depth);
    const int ids = tape->num_ids > 0 ? tape->num_ids : env->cfg.num_ue;
    if (env->episode >= 0 && ids != env->kp.tape_ids) return fail(DCOMP_EINVAL, "tape has %d streams per env, the episode was started with %d", ids, env->kp.tape_ids);
    env->cfg.tape_depth = depth;
    env->kp.tape_depth = depth;
    env->kp.tape_pos0 = tape->pos0;
    env->kp.tape_triples = (const ushort4 *)tape->triples;
    env->kp.tape_ids = ids;
    return DCOMP_OK;
}
extern "C" int dcomp_set_episode(dcomp_env *env, int64_t episode)
{
    if (!env) return fail(DCOMP_EINVAL, "null env");
    env->episode = episode - 1;     // the next reset() starts `episode`
    return DCOMP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// CPython-compatible MT19937: random.Random(int).randint(a, b)   (stdlib: Modules/_randommodule.c, Lib/random.py)
namespace {
struct PyMT {
    uint32_t mt[624];
    int idx;
    void init_genrand(uint32_t s)
    {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void init_by_array(const uint32_t *key, int len)
    {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (624 > len ? 624 : len); k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (int k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    void seed(int64_t s)
    {   // random.seed(int): key = 32-bit little-endian digits of abs(s)
        uint64_t a = (uint64_t)(s < 0 ? -s : s);
        uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
        init_by_array(key, key[1] ? 2 : 1);
    }
    uint32_t next()
    {
        if (idx >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            for (; kk < 623; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    }
    uint32_t randbelow(uint32_t n)
    {   // Random._randbelow_with_getrandbits: k = n.bit_length(); r = getrandbits(k) until r < n
        int k = 32 - __builtin_clz(n);
        uint32_t r = next() >> (32 - k);
        while (r >= n) r = next() >> (32 - k);
        return r;
    }
    int randint(int a, int b) { return a + (int)randbelow((uint32_t)(b - a + 1)); }
};
}  // namespace

extern "C" int dcomp_mt_draw_tape(const dcomp_cfg *cfg, const int64_t *seeds, int32_t num_envs, int32_t depth, int32_t *pos0,
                                  uint16_t *triples)
{
    if (!cfg || !seeds || !pos0 || !triples || depth < 1 || num_envs < 1) return fail(DCOMP_EINVAL, "bad argument");
    const int U = cfg->num_ue, W = cfg->map_w, H = cfg->map_h;
    if (!cfg->ue_vel_lo || !cfg->ue_vel_hi) return fail(DCOMP_EINVAL, "velocity ranges required");
    auto work = [&](int e0, int e1) {
    for (int e = e0; e < e1; e++) {
        PyMT pos_rng, mov_rng;
        for (int u = 0; u < U; u++) {
            const int64_t s = seeds[e] + 100 * (int64_t)(u + 1);     // base.py:138-143
            pos_rng.seed(s); mov_rng.seed(s);                        // user.py:94-96 (same seed for both streams)
            const size_t idx = (size_t)e * U + u;
            const int ix = cfg->ue_init_x ? cfg->ue_init_x[u] : -1, iy = cfg->ue_init_y ? cfg->ue_init_y[u] : -1;
            pos0[2 * idx] = ix < 0 ? pos_rng.randint(0, W) : ix;                 // user.py:102-103
            pos0[2 * idx + 1] = iy < 0 ? pos_rng.randint(0, H) : iy;             // user.py:106-107
            for (int k = 0; k < depth; k++) {
                const int lo = cfg->ue_vel_lo[u], hi = cfg->ue_vel_hi[u];
                uint16_t *t = triples + (idx * depth + k) * 4;
                t[0] = (uint16_t)(lo != hi ? mov_rng.randint(lo, hi) : lo);      // movement.py:112-117
                const int bb = cfg->ue_border_buffer ? cfg->ue_border_buffer[u] : 10;
                t[1] = (uint16_t)mov_rng.randint(bb, W - bb);                    // movement.py:126
                t[2] = (uint16_t)mov_rng.randint(bb, H - bb);                    // movement.py:127
                t[3] = 0;
            }
        }
    }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    if ((int64_t)num_envs * U < 4096) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) {
        int e0 = (int)((int64_t)num_envs * t / nt), e1 = (int)((int64_t)num_envs * (t + 1) / nt);
        if (e1 > e0) th.emplace_back(work, e0, e1);
    }
    for (auto &t : th) t.join();
    return DCOMP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// heuristic baselines (deepcomp/agent/heuristics.py) on the packed observation tensor
namespace {
struct PolicyParams {
    int policy, multi, U, B, active;
    int aligned;                  // obs is 16-byte aligned (multi-agent layout: float4 loads)
    int64_t rows;                 // E * U
    float eps;
    const uint32_t *cluster;
};

// One lane per (env, UE).  A wave first copies its 64 rows into LDS with consecutive lanes on consecutive addresses (per-lane
// row reads would touch 64 cache lines per load instruction).  Multi-agent layout: the 64 rows of 4B+1 floats are ONE
// contiguous, 256-byte aligned span -> 16-byte loads of the whole span (the ues_at_bs | util_at_bs half the rules do not read
// shares its cache lines with connected | dr, so skipping it would not save HBM traffic); central layout (small tensors):
// the connected and dr runs of each row, float by float.  Row stride in LDS 4B+1 / 2B+1 words: odd = conflict-free per-lane walks.
__global__ void __launch_bounds__(256) heuristic_kernel(PolicyParams p, const float *__restrict__ obs, uint8_t *__restrict__ act)
{
    extern __shared__ float4 lds4[];
    float *const lds = reinterpret_cast<float *>(lds4);
    const int B = p.B, W = 2 * B, S = p.multi ? 4 * B + 1 : W + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float *buf = lds + wave * 64 * S;
    const int64_t r0 = ((int64_t)blockIdx.x * wpb + wave) * 64;
    const int64_t left = p.rows - r0;
    const int nrows = left >= 64 ? 64 : left > 0 ? (int)left : 0;
    if (p.multi) {                                                     // connected[B] | dr[B] lead the row (variants.py:255-269)
        const float *src = obs + r0 * S;
        const int nfl = nrows * S;
        const int nq = p.aligned ? nfl >> 2 : 0;
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *src4 = reinterpret_cast<const f4 *>(src);
        for (int q0 = lane; q0 < nq; q0 += 64 * 16) {                  // 16 loads in flight per lane, then the LDS stores
            f4 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = q0 + 64 * k < nq ? __builtin_nontemporal_load(src4 + q0 + 64 * k) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; k++) if (q0 + 64 * k < nq) reinterpret_cast<f4 *>(buf)[q0 + 64 * k] = v[k];
        }
        for (int i0 = 4 * nq + lane; i0 < nfl; i0 += 64 * 8) {         // tail / unaligned tensor: float by float
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = i0 + 64 * k < nfl ? src[i0 + 64 * k] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) if (i0 + 64 * k < nfl) buf[i0 + 64 * k] = v[k];
        }
    } else {                                                           // connected[U*B] | dr[U*B] | utility[U] (central.py:147-151)
        const int64_t env_floats = (int64_t)p.U * (W + 1);
        for (int i0 = lane; i0 < nrows * W; i0 += 64 * 8) {
            float v[8];
            int at[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + 64 * k;
                const int row = i / W, col = i - row * W;
                const int64_t r = r0 + row;
                const int64_t e = r / p.U;
                const int u = (int)(r - e * p.U);
                at[k] = row * S + col;
                v[k] = i < nrows * W ? obs[e * env_floats + (col < B ? u * B + col : p.U * B + u * B + col - B)] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) if (i0 + 64 * k < nrows * W) buf[at[k]] = v[k];
        }
    }
    __syncthreads();
    if (lane >= nrows) return;
    const float *row = buf + lane * S;
    const int64_t r = r0 + lane;
    const int u = (int)(r % p.U);
    unsigned long long conn = 0;                                       // (64-bit sets: up to 64 stations, round 5)
    float mx = -__builtin_huge_valf();
    int best = 0;
#pragma unroll 8
    for (int b = 0; b < B; b++) {
        conn |= (row[b] > 0.5f ? 1ull : 0ull) << b;
        const float d = row[B + b];
        if (d > mx) { mx = d; best = b; }                              // strict: the first maximum (np.argmax, heuristics.py:27)
    }
    int a = 0;
    if (u >= p.active) a = 0;
    else if (p.policy == DCOMP_POLICY_3GPP) {
        if ((conn >> best) & 1ull) a = 0;                              // heuristics.py:30-31: already at the best cell
        else if (conn) a = __builtin_ffsll((long long)conn);           // :33-36: drop the (first) other connection
        else a = best + 1;                                             // :38
    } else {
        unsigned long long sel = B == 64 ? ~0ull : (1ull << B) - 1ull; // FullCoMP: every cell
        if (p.policy == DCOMP_POLICY_DYNAMIC) {
            const float thr = mx * p.eps;                              // heuristics.py:87-90
            sel = 0;
#pragma unroll 8
            for (int b = 0; b < B; b++) sel |= (row[B + b] >= thr ? 1ull : 0ull) << b;
        } else if (p.policy == DCOMP_POLICY_CLUSTER)                   // :172-176; one word per station up to 32 stations, two (lo, hi) beyond
            sel = B <= 32 ? (unsigned long long)p.cluster[best] : ((unsigned long long)p.cluster[2 * best] | ((unsigned long long)p.cluster[2 * best + 1] << 32));
        const unsigned long long drop = conn & ~sel;
        if (drop) a = __builtin_ffsll((long long)drop);                // :96-99 / :178-181: index order
        else {
            const unsigned long long cand = sel & ~conn;
            float m2 = -__builtin_huge_valf();
#pragma unroll 8
            for (int b = 0; b < B; b++) {
                const float d = row[B + b];
                if (((cand >> b) & 1ull) && d > m2) { m2 = d; a = b + 1; }   // strongest first, first of equals (:57-63, :101-106)
            }
        }
    }
    act[r] = (uint8_t)a;
}
}  // namespace

extern "C" int dcomp_set_policy(dcomp_env *env, const dcomp_policy *p, uint8_t *next_action)
{
    if (!env) return fail(DCOMP_EINVAL, "null argument");
    if (!p || !next_action) { env->kp.next_act = nullptr; return DCOMP_OK; }
    if (p->policy < DCOMP_POLICY_3GPP || p->policy > DCOMP_POLICY_CLUSTER) return fail(DCOMP_EINVAL, "unknown policy %d", p->policy);
    if ((p->num_envs && p->num_envs != env->cfg.num_envs) || (p->num_ue && p->num_ue != env->kp.U) || (p->num_bs && p->num_bs != env->cfg.num_bs))
        return fail(DCOMP_EINVAL, "policy shape (%d, %d, %d) is not the env's (%d, %d, %d)", p->num_envs, p->num_ue, p->num_bs,
                    env->cfg.num_envs, env->kp.U, env->cfg.num_bs);
    if (p->policy == DCOMP_POLICY_CLUSTER && !p->cluster_mask) return fail(DCOMP_EINVAL, "DCOMP_POLICY_CLUSTER needs cluster_mask");
    if (p->policy == DCOMP_POLICY_DYNAMIC && !(p->epsilon >= 0.f && p->epsilon <= 1.f)) return fail(DCOMP_EINVAL, "epsilon must be in [0, 1]");
    env->kp.policy = p->policy; env->kp.policy_eps = p->epsilon; env->kp.policy_cluster = p->cluster_mask;
    env->kp.next_act = next_action;
    return DCOMP_OK;
}

extern "C" int dcomp_heuristic_actions(const dcomp_policy *p, const float *obs, uint8_t *action, void *stream)
{
    if (!p || !obs || !action) return fail(DCOMP_EINVAL, "null argument");
    if (p->num_envs < 1 || p->num_ue < 1 || p->num_ue > DCOMP_MAX_UE || p->num_bs < 1 || p->num_bs > DCOMP_MAX_BS)
        return fail(DCOMP_EINVAL, "need num_envs>=1, 1<=num_ue<=%d, 1<=num_bs<=%d (got %d, %d, %d)", DCOMP_MAX_UE, DCOMP_MAX_BS, p->num_envs, p->num_ue, p->num_bs);
    if (p->policy < DCOMP_POLICY_3GPP || p->policy > DCOMP_POLICY_CLUSTER) return fail(DCOMP_EINVAL, "unknown policy %d", p->policy);
    if (p->obs_kind != DCOMP_CENTRAL && p->obs_kind != DCOMP_MULTI) return fail(DCOMP_EINVAL, "obs_kind must be DCOMP_CENTRAL or DCOMP_MULTI");
    if (p->num_active < 0 || p->num_active > p->num_ue) return fail(DCOMP_EINVAL, "num_active (%d) outside [0, num_ue]", p->num_active);
    if (p->policy == DCOMP_POLICY_CLUSTER && !p->cluster_mask) return fail(DCOMP_EINVAL, "DCOMP_POLICY_CLUSTER needs cluster_mask");
    if (p->policy == DCOMP_POLICY_DYNAMIC && !(p->epsilon >= 0.f && p->epsilon <= 1.f)) return fail(DCOMP_EINVAL, "epsilon must be in [0, 1]");
    PolicyParams k;
    k.policy = p->policy; k.multi = p->obs_kind == DCOMP_MULTI; k.U = p->num_ue; k.B = p->num_bs; k.active = p->num_active;
    k.rows = (int64_t)p->num_envs * p->num_ue; k.eps = p->epsilon; k.cluster = p->cluster_mask;
    k.aligned = (reinterpret_cast<uintptr_t>(obs) & 15u) == 0;
    const size_t per_wave = (size_t)64 * ((k.multi ? 4 : 2) * p->num_bs + 1) * sizeof(float);
    int wpb = (int)(60 * 1024 / per_wave);
    wpb = wpb > 4 ? 4 : wpb < 1 ? 1 : wpb;
    if (per_wave * wpb > 64 * 1024) {           // 64 rows of more than 60 stations: one wave per workgroup, above the default 64 KiB of dynamic LDS
        HIP_TRY(raise_lds_limit(reinterpret_cast<const void *>(heuristic_kernel), 72 * 1024));
    }
    const int64_t waves = (k.rows + 63) / 64;
    dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
    hipLaunchKernelGGL(heuristic_kernel, grid, block, per_wave * wpb, (hipStream_t)stream, k, obs, action);
    HIP_TRY(hipGetLastError());
    return DCOMP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// device self-tests
namespace {
__global__ void selftest_fp64(int op, const double *x, const double *y, double *out, int64_t n)
{
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i], b = y[i], r;
    if (op == 0) r = __builtin_sqrt(a);
    else if (op == 1) r = a / b;
    else if (op == 2) r = __builtin_fma(b, b, a * a);
    else {                                             // 4, 5, 6: move_ue's norm_and_unit(vx = x, vy = y) -> nrm, nx, ny
        double nrm, nx, ny;
        dcomp::norm_and_unit(a, b, nrm, nx, ny);
        r = op == 4 ? nrm : op == 5 ? nx : ny;
    }
    out[i] = r;
}
template <int W>
__global__ void selftest_reduce(const double *x, double *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? (float)x[i] : 0.f;
    float s = dcomp::group_reduce<W, dcomp::OpSum>(v);
    float m = dcomp::group_reduce<W, dcomp::OpMin>(v);
    int lane = threadIdx.x & 63;
    int c = dcomp::group_count<W>(v > 0.f, lane & ~(W - 1));
    if (i < n) out[i] = (double)s + 1024.0 * (double)c + 1048576.0 * (double)m;
}
}  // namespace

extern "C" int dcomp_selftest(int op, int width, const double *x, const double *y, double *out, int64_t n, void *stream)
{
    if (!x || !out || n < 1) return fail(DCOMP_EINVAL, "bad argument");
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if ((op >= 0 && op <= 2) || (op >= 4 && op <= 6)) {
        if (!y) return fail(DCOMP_EINVAL, "y required");
        hipLaunchKernelGGL(selftest_fp64, grid, block, 0, s, op, x, y, out, n);
    } else if (op == 3) {
        switch (width) {
        case 2: hipLaunchKernelGGL(selftest_reduce<2>, grid, block, 0, s, x, out, n); break;
        case 4: hipLaunchKernelGGL(selftest_reduce<4>, grid, block, 0, s, x, out, n); break;
        case 8: hipLaunchKernelGGL(selftest_reduce<8>, grid, block, 0, s, x, out, n); break;
        case 16: hipLaunchKernelGGL(selftest_reduce<16>, grid, block, 0, s, x, out, n); break;
        case 32: hipLaunchKernelGGL(selftest_reduce<32>, grid, block, 0, s, x, out, n); break;
        case 64: hipLaunchKernelGGL(selftest_reduce<64>, grid, block, 0, s, x, out, n); break;
        default: return fail(DCOMP_EINVAL, "width must be 2..64 (power of two)");
        }
    } else return fail(DCOMP_EINVAL, "unknown op");
    HIP_TRY(hipGetLastError());
    return DCOMP_OK;
}

// ---- compact rollout fragments for the learner hand-off (dcomp_fragment.h) ----
extern "C" int dcomp_fragment_words(int32_t num_ue, int32_t num_bs)
{
    if (num_ue < 1 || num_ue > DCOMP_MAX_UE || num_bs < 1 || num_bs > DCOMP_MAX_BS) return -1;    // (round 6: two mask words per UE with more than 32 stations)
    return dcomp_frag::env_words(num_ue, num_bs);
}

static int fragment_params(dcomp_frag::FragParams &p, int64_t n, int U, int B, int &grid, size_t &lds_pack, size_t &lds_unpack)
{
    if (n < 1 || U < 1 || U > DCOMP_MAX_UE || B < 1 || B > DCOMP_MAX_BS)
        return fail(DCOMP_EINVAL, "fragment: need num_env_steps >= 1, 1 <= num_ue <= %d, 1 <= num_bs <= %d", DCOMP_MAX_UE, DCOMP_MAX_BS);
    if (dcomp_frag::fill(p, n, U, B, grid, lds_pack, lds_unpack)) return fail(DCOMP_EINVAL, "fragment too long for one launch: split it (num_env_steps * chunks >= 2^31)");
    return DCOMP_OK;
}

extern "C" int dcomp_pack_fragment(const float *obs, int64_t num_env_steps, int32_t num_ue, int32_t num_bs, uint32_t *packed, int32_t *flags,
                                   void *stream)
{
    if (!obs || !packed || !flags) return fail(DCOMP_EINVAL, "null argument");
    dcomp_frag::FragParams p{};
    int grid;
    size_t lp, lu;
    const int rc = fragment_params(p, num_env_steps, num_ue, num_bs, grid, lp, lu);
    if (rc) return rc;
    p.obs_in = obs; p.packed_out = packed; p.flags = flags;
    hipLaunchKernelGGL(dcomp_frag::pack_kernel, dim3(grid), dim3(dcomp_frag::BLOCK), lp, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DCOMP_OK;
}

extern "C" int dcomp_unpack_fragment(const uint32_t *packed, int64_t num_env_steps, int32_t num_ue, int32_t num_bs, float *obs, void *stream)
{
    if (!obs || !packed) return fail(DCOMP_EINVAL, "null argument");
    dcomp_frag::FragParams p{};
    int grid;
    size_t lp, lu;
    const int rc = fragment_params(p, num_env_steps, num_ue, num_bs, grid, lp, lu);
    if (rc) return rc;
    p.packed_in = packed; p.obs_out = obs;
    hipLaunchKernelGGL(dcomp_frag::unpack_kernel, dim3(grid), dim3(dcomp_frag::BLOCK), lu, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return DCOMP_OK;
}
