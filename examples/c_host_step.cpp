// examples/c_host_step.cpp -- the C ABI of include/dcomp.h driven from a host that is NOT Python.
//
//   hipcc --offload-arch=gfx950 -O2 -I include -o /tmp/c_host_step examples/c_host_step.cpp -L deepcomp_amd/csrc -ldcomp_hip \
//         -Wl,-rpath,$PWD/deepcomp_amd/csrc && /tmp/c_host_step
//
// What a C / C++ / cgo / JNI host does to replace `env = CentralRelNormEnv(env_config); env.reset(); env.step(a)`
// (deepcomp/env/multi_ue/central.py:9-73, single_ue/base.py:27-84, 169-189, 413-466):
//   1. fill a ZERO-INITIALISED dcomp_cfg from the scenario (env_setup.py:164-176: the 4-station "custom" map)
//   2. dcomp_create(&cfg, &env)   -- the header's macro: dcomp_create_v with THIS translation unit's DCOMP_ABI_VERSION and struct sizes
//   3. allocate the state / output buffers on the device (sizes: dcomp_state_sizes, dcomp_obs_dim), zero them
//   4. dcomp_reset, then dcomp_step per action batch, on the caller's stream; dcomp_check when it wants the assertions
// It prints one FNV-1a checksum per output tensor after 12 steps of 64 envs; tests/test_c_host_gpu.py runs the same 12 steps through
// deepcomp_amd.env.BatchedMobileEnv and holds the two byte for byte.  Only plain pointers and sizes cross the boundary.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "dcomp.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define DC_OK(x) do { int rc_ = (x); if (rc_ != DCOMP_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, dcomp_last_error()); return 3; } } while (0)

static uint64_t fnv1a(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    const int E = 64, U = 6, B = 4, T = 12;
    const int kind = (argc > 1 && !strcmp(argv[1], "multi")) ? DCOMP_MULTI : DCOMP_CENTRAL;
    // env_setup.py:164-176 (custom map, 194 x 120) with the CLI's 'mixed' sharing rule (env_setup.py:40-49)
    const double bs_x[B] = {10, 97, 184, 97}, bs_y[B] = {60, 10, 60, 110};
    const int32_t sharing[B] = {DCOMP_RES_FAIR, DCOMP_RATE_FAIR, DCOMP_PROP_FAIR, DCOMP_RES_FAIR};
    const int32_t vel_lo[U] = {0, 1, 1, 1, 5, 5}, vel_hi[U] = {0, 3, 3, 3, 10, 10};          // one static, three slow, two fast UEs (env_setup.py:145-161)

    dcomp_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));                                   // zero first: fields this host does not set mean "default"
    cfg.num_envs = E; cfg.num_ue = U; cfg.num_bs = B;
    cfg.map_w = 194; cfg.map_h = 120;
    cfg.env_kind = kind; cfg.reward_agg = DCOMP_REWARD_AVG;
    cfg.rng_mode = DCOMP_RNG_PHILOX; cfg.seed = 42; cfg.env_id_base = 0; cfg.device = 0;
    cfg.bs_x = bs_x; cfg.bs_y = bs_y; cfg.bs_sharing = sharing;
    cfg.ue_vel_lo = vel_lo; cfg.ue_vel_hi = vel_hi;

    printf("%s, header ABI %d, library ABI %d\n", dcomp_version(), DCOMP_ABI_VERSION, dcomp_abi_version());
    dcomp_env *env = NULL;
    DC_OK(dcomp_create(&cfg, &env));                                // = dcomp_create_v(DCOMP_ABI_VERSION, sizeof(dcomp_cfg), ... , &cfg, &env)

    size_t pos_b, mv_b, conn_b, ewma_b, flags_b, since_b;
    DC_OK(dcomp_state_sizes(env, &pos_b, &mv_b, &conn_b, &ewma_b, &flags_b, &since_b));
    int32_t obs_floats = 0, rew_floats = 0;
    DC_OK(dcomp_obs_dim(env, &obs_floats, &rew_floats));

    dcomp_state st;
    dcomp_out out;
    memset(&st, 0, sizeof(st));
    memset(&out, 0, sizeof(out));
    HIP_OK(hipMalloc((void **)&st.pos, pos_b));     HIP_OK(hipMemset(st.pos, 0, pos_b));
    HIP_OK(hipMalloc((void **)&st.mv, mv_b));       HIP_OK(hipMemset(st.mv, 0, mv_b));
    HIP_OK(hipMalloc((void **)&st.conn, conn_b));   HIP_OK(hipMemset(st.conn, 0, conn_b));
    HIP_OK(hipMalloc((void **)&st.ewma, ewma_b));   HIP_OK(hipMemset(st.ewma, 0, ewma_b));
    HIP_OK(hipMalloc((void **)&st.flags, flags_b)); HIP_OK(hipMemset(st.flags, 0, flags_b));
    const size_t obs_b = (size_t)E * obs_floats * 4, rew_b = (size_t)E * rew_floats * 4, eu_b = (size_t)E * U * 4;
    HIP_OK(hipMalloc((void **)&out.obs, obs_b));
    HIP_OK(hipMalloc((void **)&out.reward, rew_b));
    HIP_OK(hipMalloc((void **)&out.sum_utility, (size_t)E * 4));
    HIP_OK(hipMalloc((void **)&out.ue_dr, eu_b));
    HIP_OK(hipMalloc((void **)&out.ue_utility, eu_b));
    uint8_t *d_act = NULL;
    HIP_OK(hipMalloc((void **)&d_act, (size_t)E * U));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    DC_OK(dcomp_reset(env, &st, NULL, &out, stream));              // MobileEnv.reset; Philox draws need no tape

    std::vector<uint8_t> act((size_t)E * U);
    uint32_t lcg = 12345u;
    for (int t = 0; t < T; t++) {                                   // actions in [0, B]: 0 = no-op, k = toggle station k - 1 (base.py:259-263)
        for (size_t i = 0; i < act.size(); i++) { lcg = lcg * 1664525u + 1013904223u; act[i] = (uint8_t)((lcg >> 24) % (B + 1)); }
        HIP_OK(hipMemcpyAsync(d_act, act.data(), act.size(), hipMemcpyHostToDevice, stream));
        DC_OK(dcomp_step(env, &st, d_act, &out, stream));          // MobileEnv.step for all 64 envs: one kernel launch
        HIP_OK(hipStreamSynchronize(stream));                      // (the host buffer `act` is reused next turn)
    }
    DC_OK(dcomp_check(env, &st, stream));                          // the reference's in-code assertions, lazily (base.py:238, movement.py:165)
    if (dcomp_time(env) != T) { fprintf(stderr, "env.time = %d\n", dcomp_time(env)); return 4; }

    std::vector<unsigned char> h(obs_b > pos_b ? obs_b : pos_b);
    HIP_OK(hipMemcpy(h.data(), out.obs, obs_b, hipMemcpyDeviceToHost));        printf("obs %016llx\n", (unsigned long long)fnv1a(h.data(), obs_b));
    HIP_OK(hipMemcpy(h.data(), out.reward, rew_b, hipMemcpyDeviceToHost));     printf("reward %016llx\n", (unsigned long long)fnv1a(h.data(), rew_b));
    HIP_OK(hipMemcpy(h.data(), out.ue_dr, eu_b, hipMemcpyDeviceToHost));       printf("ue_dr %016llx\n", (unsigned long long)fnv1a(h.data(), eu_b));
    HIP_OK(hipMemcpy(h.data(), st.pos, pos_b, hipMemcpyDeviceToHost));         printf("pos %016llx\n", (unsigned long long)fnv1a(h.data(), pos_b));
    HIP_OK(hipMemcpy(h.data(), st.conn, conn_b, hipMemcpyDeviceToHost));       printf("conn %016llx\n", (unsigned long long)fnv1a(h.data(), conn_b));

    // a caller built against OLDER structs is refused, not served: pretend this host's dcomp_out had no obs_compact member
    dcomp_env *stale = NULL;
    const int rc = dcomp_create_v(DCOMP_ABI_VERSION, sizeof(dcomp_cfg), sizeof(dcomp_state), sizeof(dcomp_out) - sizeof(void *), sizeof(dcomp_rollout_opts), &cfg, &stale);
    printf("stale caller: %d (%s)\n", rc, rc == DCOMP_EABI ? "DCOMP_EABI" : "?!");

    DC_OK(dcomp_destroy(env));
    return rc == DCOMP_EABI ? 0 : 5;
}
