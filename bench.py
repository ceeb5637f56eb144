#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused HIP env step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python bench.py --gpus N --steps K --warmup W            # no launcher: spawns its N ranks itself (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one MobileEnv.step() of every env of the batch (E envs per GPU).  Workload at N=1: BASELINE
config 3 -- 65 536 parallel envs, 32 UE x 10 BS, multi-agent (DD-CoMP) per-UE observations, mixed sharing,
log utility, reward 'avg', episode length 100 (reset kernel every 100 steps, inside the timed region), uniform
random actions pre-generated on the device.  N>1: the env axis is sharded, 65 536 envs per GPU (weak scaling),
global env ids keep the draws independent of the GPU count; no collective on the data path.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def bytes_per_env_step(U, B, kind, log_metrics=True):
    """Compulsory HBM bytes of one env-step in this implementation's layout:
    read pos 16 + mv 8 + conn 4 + ewma 4 + action 1; write pos 16 + mv 8 + conn 4 + ewma 4; obs; reward; info."""
    per_ue = 33 + 32 + (8 if log_metrics else 0)
    if kind == 'multi':
        return U * (per_ue + 4 * (4 * B + 1) + 4) + (4 if log_metrics else 0)
    return U * (per_ue + 4 * (2 * B + 1)) + 4 + (4 if log_metrics else 0)


def survey_bytes_per_env_step(U, B, kind):
    """SURVEY.md 8(d) figure incl. the FP64-position surcharge (+24 U)."""
    return U * (55 + 16 * B + 24) + 4 if kind == 'multi' else U * (51 + 8 * B + 24) + 8


# BASELINE.md section 2, mixed sharing, 1 core: env-steps/s of the reference's own step() by (env kind, U, B)
REFERENCE_STEP_PER_CORE = {('central', 3, 3): 2122.0, ('central', 10, 5): 398.0, ('multi', 32, 10): 69.9, ('multi', 128, 32): 9.4}


def cpu_baseline(scn, kind, U, B, budget_s=12.0, restore_affinity=None):
    """The CPU oracle (oracle/dcomp_oracle.c, OpenMP over envs) on a bounded sample of the same workload, on the box's HOST cores: the
    rank's NUMA binding (pin_to_gpu_numa_node) is lifted for it (restore_affinity = the mask the process started with).  How many
    threads: the GPU box is a slice of a shared host, and a team as large as the visible CPU count collapses there (256 threads: 4 x 10^3
    env-steps/s, 128: 1.2 x 10^5, one: 3.1 x 10^4 -- profiles/r05_cpu_baseline_threads.txt), so a 0.4 s probe per candidate team size picks
    the fastest and `cores` reports the team that was used."""
    from oracle import oracle as orc
    if restore_affinity:
        try:
            os.sched_setaffinity(0, restore_affinity)
        except OSError:
            pass
    avail = len(os.sched_getaffinity(0))
    max_threads = max(1, min(orc.lib().orc_max_threads(), avail))

    def make(E, nt):
        envs = []
        for e in range(E):
            o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing,
                              [s['velocity'] for s in scn.ue_specs], kind=orc.MULTI if kind == 'multi' else orc.CENTRAL)
            o.set_philox(42, e)
            envs.append(o)
        return orc.OracleBatch(envs, num_threads=nt)

    rng = np.random.default_rng(7)
    cands = sorted({n for n in (max_threads, avail // 2, avail // 4, avail // 8, 64, 32, 16, 8) if 1 <= n <= max_threads}, reverse=True)
    probe = make(max(cands) * 4, max(cands))
    probe.reset()
    a = rng.integers(0, B + 1, size=(probe.E, U)).astype(np.uint8)
    tried = {}
    for nt in cands:
        probe.num_threads = nt
        probe.step(a)                                # thread-pool start-up outside the probe
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.4:
            probe.step(a)
            n += 1
        tried[nt] = probe.E * n / (time.perf_counter() - t0)
    threads = max(tried, key=tried.get)
    rate = tried[threads]
    steps = 100
    E = int(max(threads, min(32768, rate * budget_s / steps)))
    E = (E // threads) * threads
    batch = make(E, threads)
    batch.reset()
    acts = rng.integers(0, B + 1, size=(steps, E, U)).astype(np.uint8)
    t0 = time.perf_counter()
    for t in range(steps):
        batch.step(acts[t])
    dt = time.perf_counter() - t0
    out = {'value': E * steps / dt, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
           'sample': f'{E} envs x {steps} steps ({U} UE x {B} BS, {kind}), oracle/dcomp_oracle.c with OpenMP over envs, {dt:.1f} s',
           'cpus_visible': avail, 'team_sizes_probed_env_steps_per_s': {str(k): round(v) for k, v in tried.items()}}
    # "single core and all cores".  The same port on ONE thread (~3 s), which is what links this box to the reference's
    # own step(): REFERENCE_STEP_PER_CORE below was timed in the build container with the reference's unmodified deepcomp.env.* (BASELINE.md
    # section 2; the Python reference cannot travel to this box), so port-on-one-core / that figure is how much faster the C restatement is
    # than the reference per core -- and all-cores / single-core is this box's parallel speed-up, which normalises `value` across boxes.
    del batch, probe
    one = make(64, 1)
    one.reset()
    a1 = rng.integers(0, B + 1, size=(8, 64, U)).astype(np.uint8)
    one.step(a1[0])
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 2.5:
        one.step(a1[n & 7])
        n += 1
    dt1 = time.perf_counter() - t0
    ref = REFERENCE_STEP_PER_CORE.get((kind, U, B))
    out['single_core'] = {'value': 64 * n / dt1, 'unit': 'env-steps/s', 'cores': 1,
                          'sample': f'64 envs x {n} steps on one thread, {dt1:.1f} s', 'all_cores_over_single_core': out['value'] / (64 * n / dt1)}
    if ref:
        out['reference_step'] = {'value': ref, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'reference',
                                 'where': "the reference's own MultiAgentMobileEnv / CentralRelNormEnv step() on one core of the BUILD container (8 x Xeon 2.1 GHz), "
                                          'BASELINE.md section 2 -- not re-timed here: no reference source travels to the GPU box',
                                 'port_single_core_over_reference_step': 64 * n / dt1 / ref}
    return out


def measure_rollout(torch, BatchedMobileEnv, scenarios, build_from_scenario, dev, E, U, B, kind, T, L=100, steps=2000, launches_too=False):
    """Fused rollout (dcomp_rollout_ex): T steps per launch with the UE state in registers, EVERY step's observations /
    rewards / info written into [T, ...] fragment buffers, reset at the horizon inside the kernel.  Secondary figures, not
    the headline (the headline is one step() per launch, policy in the loop)."""
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)
    env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=42, episode_length=L, rng='philox', rand_episodes=True, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    tape = torch.randint(0, B + 1, (2, T, E, U), generator=g, device=dev, dtype=torch.uint8)
    frag = {'obs': torch.empty((T,) + tuple(env.obs.shape), device=dev), 'reward': torch.empty((T,) + tuple(env.reward.shape), device=dev),
            'sum_utility': torch.empty((T, E), device=dev), 'ue_dr': torch.empty((T, E, U), device=dev),
            'ue_utility': torch.empty((T, E, U), device=dev)}
    n_calls = max(1, steps // T)
    env.reset()

    def run(n):
        for i in range(n):
            env.rollout(tape[i & 1], out=frag, horizon=L)
    run(max(2, n_calls // 10))
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    run(n_calls)
    b.record()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    env.check()
    n = n_calls * T
    bpe = survey_bytes_per_env_step(U, B, kind)
    # the fused kernel reads / writes the UE state once per LAUNCH, not per step: its own traffic per env-step
    state_b = U * (33 + 32 - 1)
    fused_bpe = bytes_per_env_step(U, B, kind) - state_b + state_b / T
    ksec = a.elapsed_time(b) * 1e-3
    fused = bool(env.rollout_is_fused(T))
    out = {'value': E * n / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / n * 1e3, 'steps': n, 'steps_per_launch': T,
           'fused_one_launch': fused, 'kernel_ms_per_launch': a.elapsed_time(b) / n_calls,
           # TWO figures, never one alone: SURVEY 8(d)'s per-step bytes / time is an ALGORITHMIC throughput (a fused kernel keeps
           # the UE state in registers and never moves those bytes); the HBM fraction proper uses the bytes the kernel moves.
           'algorithmic_bytes_per_env_step': bpe, 'achieved_GBps_algorithmic': bpe * E * n / ksec / 1e9,
           'frac_of_hbm_peak_algorithmic': bpe * E * n / ksec / 1e9 / HBM_PEAK_GBS,
           'kernel_bytes_per_env_step': fused_bpe if fused else bytes_per_env_step(U, B, kind),
           'kernel_bytes_how': 'layout bytes the launch must move: outputs of every step + actions, UE state once per launch',
           'outputs': 'every step ([T, ...] fragment buffers), reset at the horizon inside the kernel'}
    out['achieved_GBps_kernel_traffic'] = out['kernel_bytes_per_env_step'] * E * n / ksec / 1e9
    out['frac_of_hbm_peak_kernel_traffic'] = out['achieved_GBps_kernel_traffic'] / HBM_PEAK_GBS
    ent, src = traffic_from_profile(f'{E}x{U}x{B}_{kind}_mixed_rollout_T{T}', want_entry=True)
    if ent:                                         # rocprofv3 --pmc bytes of the same launch shape (tracked profile, current kernels)
        pmc = (2.0 * ent['fetch_kib'] + ent['write_kib']) * 1024.0
        out['pmc_bytes_per_launch'] = pmc
        out['achieved_GBps_pmc_traffic'] = pmc / (a.elapsed_time(b) / n_calls * 1e-3) / 1e9
        out['frac_of_hbm_peak_pmc_traffic'] = out['achieved_GBps_pmc_traffic'] / HBM_PEAK_GBS
    out['traffic_source'] = src
    if launches_too:                                # the same workload as one launch per step (what round 1 measured)
        acts = tape[0]
        k_total = max(T, steps)

        def loop(n, k=0):
            for i in range(n):
                if k % L == 0:
                    env.reset()
                env.step(acts[k % T])
                k += 1
        loop(200)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        loop(k_total)
        torch.cuda.synchronize(dev)
        out['one_launch_per_step_env_steps_per_s'] = E * k_total / (time.perf_counter() - t0)
    return out


def measure_steps(torch, BatchedMobileEnv, scenarios, build_from_scenario, dev, E, U, B, kind, steps=300, L=100, sharing='mixed', ring=0, compact=False):
    """One launch per step on another shape (secondary figures): HIP events around back-to-back launches after 300 untimed
    ones (steady state), SURVEY 8(d) bytes / launch duration.  ring = k > 0: the steps write their observations / rewards into k
    DIFFERENT buffers in turn (a rollout fragment, step_into) instead of rewriting env.obs -- a buffer that is rewritten every step
    partly never leaves the 256 MB Infinity Cache, a fragment does.  compact: the steps write the lossless compact record
    (dcomp_out.obs_compact, U (B + 2) + 2B words per env-step) INSTEAD of the rows; its figure counts the bytes of THAT layout --
    it is not a fraction of the SURVEY 8(d) roofline, whose bytes are the row format's."""
    scn = scenarios.grid_map(B, sharing).with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)
    env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=42, episode_length=L, rng='philox', rand_episodes=True, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    pool = torch.randint(0, B + 1, (4, E, U), generator=g, device=dev, dtype=torch.uint8)
    frag = [(torch.empty_like(env.obs), torch.empty_like(env.reward)) for _ in range(ring)]
    packed = torch.empty((E, env.compact_words), dtype=torch.int32, device=dev) if compact else None
    ms, n = 0.0, 0
    for phase, count in (('warm', 300), ('timed', steps)):
        t = 0
        while t < count:
            env.reset()
            k = min(L, count - t)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(k):
                if compact:
                    env.step_compact(pool[i & 3], packed, env.reward)
                elif ring:
                    env.step_into(pool[i & 3], *frag[i % ring])
                else:
                    env.step(pool[i & 3])
            b.record()
            torch.cuda.synchronize(dev)
            if phase == 'timed':
                ms += a.elapsed_time(b)
                n += k
            t += k
    env.check()
    kms = ms / n
    bpe = survey_bytes_per_env_step(U, B, kind)
    if compact:                                  # the row format's 4 (4B + 1) bytes of observation per UE become 4 (B + 2), + 8B per env
        cb = bpe - U * 4 * (4 * B + 1) + 4 * env.compact_words
        return {'kernel_ms': kms, 'env_steps_per_s_kernel_only': E / (kms * 1e-3), 'bytes_per_env_step_in_the_compact_layout': cb,
                'bytes_per_env_step_in_the_row_format': bpe, 'achieved_GBps_by_the_bytes_it_moves': cb * E / (kms * 1e-3) / 1e9,
                'frac_of_hbm_peak_by_the_bytes_it_moves': cb * E / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'observation': 'lossless compact record written by the step itself (dcomp_out.obs_compact); dcomp_unpack_fragment restores the rows bit for bit',
                'how': f'{n} back-to-back launches after 300 untimed ones, HIP events'}
    return {'kernel_ms': kms, 'env_steps_per_s_kernel_only': E / (kms * 1e-3), 'algorithmic_bytes_per_env_step': bpe,
            'achieved_GBps': bpe * E / (kms * 1e-3) / 1e9, 'frac_of_hbm_peak': bpe * E / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'lanes_per_env': env.lanes_per_env, 'how': f'{n} back-to-back launches after 300 untimed ones, HIP events'}


def measure_sharded(torch, dist, BatchedMobileEnv, scenarios, build_from_scenario, dev, rank, world, backend, total_envs, U, B, kind, steps=200, warm=100, L=100):
    """EVERY rank calls this, at every N (N = 1: the whole job on one GPU = the base of the strong-scaling curve): a BASELINE
    multi-GPU configuration of FIXED total size (config 4: 262 144 x 32 x 10, config 5: 32 768 x 128 x 32) split over the ranks --
    total_envs / N envs per GPU, global env ids, no collective on the data path, barrier + synchronize on both sides, MAX over
    ranks.  With world = 8 the job IS the BASELINE configuration (32 768 resp. 4 096 envs per GPU)."""
    E = total_envs // world
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)
    env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=42, episode_length=L, rng='philox', rand_episodes=True, device=dev,
                           env_id_base=rank * E)
    g = torch.Generator(device=dev).manual_seed(11 + rank)
    pool = torch.randint(0, B + 1, (4, E, U), generator=g, device=dev, dtype=torch.uint8)
    use_dist = dist is not None and dist.is_initialized()

    def run(n, t=0):
        for i in range(n):
            if t % L == 0:
                env.reset()
            env.step(pool[t & 3])
            t += 1
        return t

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
    t = run(warm)
    fence()
    t0 = time.perf_counter()
    run(steps, t)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    env.check()
    bpe = survey_bytes_per_env_step(U, B, kind)
    return {'value': world * E * steps / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'total_envs': world * E,
            'envs_per_gpu': E, 'num_ue': U, 'num_bs': B, 'n_gpus': world, 'scaling': 'strong', 'kernel': 'dcomp::' + (env.step_kernel_name or '?'),
            'achieved_GBps_per_gpu': bpe * E * steps / dt / 1e9, 'frac_of_hbm_peak_per_gpu': bpe * E * steps / dt / 1e9 / HBM_PEAK_GBS,
            'how': f'host clock over the step loop after {warm} untimed steps (resets included), barrier + synchronize on both sides, MAX over ranks'}


STRONG = (('config4_strong_262144x32x10', (262144, 32, 10, 'multi')), ('config5_strong_32768x128x32', (32768, 128, 32, 'multi')))


def measure_also(torch, BatchedMobileEnv, scenarios, build_from_scenario, dev):
    """The other BASELINE configurations that fit one GPU, next to the headline (secondary figures).  Runs on EVERY rank at every N
    before the headline's warm-up (the same ~3 s of launches: the same warm state for every point of a scaling curve)."""
    mk = (torch, BatchedMobileEnv, scenarios, build_from_scenario, dev)
    central = measure_steps(*mk, 65536, 10, 5, 'central')
    # the same batch through rollout(): short central rows take the fused kernel at every batch size once a rollout is >= 4 steps
    # long (no kernel boundary: the 5.4 us per-launch constant is a quarter of this shape's step)
    r = measure_rollout(*mk, 65536, 10, 5, 'central', T=50, steps=1500)
    central['through_rollout_T50'] = {k: r[k] for k in ('ms_per_step', 'fused_one_launch', 'achieved_GBps_algorithmic', 'frac_of_hbm_peak_algorithmic',
                                                        'achieved_GBps_kernel_traffic', 'frac_of_hbm_peak_kernel_traffic', 'kernel_bytes_per_env_step',
                                                        'algorithmic_bytes_per_env_step', 'traffic_source') if k in r}
    central['through_rollout_T50'].update({k: r[k] for k in ('achieved_GBps_pmc_traffic', 'frac_of_hbm_peak_pmc_traffic') if k in r})
    central['through_rollout_T50']['how'] = ('50 steps per launch, every step\'s outputs into [T, ...] buffers, resets at the horizon inside the kernel; '
                                             '`algorithmic` = SURVEY 8(d) per-step bytes / time (a throughput: the fused kernel never moves the per-step state '
                                             'bytes), `kernel_traffic` = the bytes the launch really moves / time (the HBM fraction proper)')
    out = {'config2_4096x10x5_central_fused_rollout': measure_rollout(*mk, 4096, 10, 5, 'central', T=100, steps=4000, launches_too=True),
           'central_65536x10x5': central,
           # "report both resource-fair everywhere and mixed" -- the headline workload with every BS resource-fair
           'config3_65536x32x10_multi_resource_fair': measure_steps(*mk, 65536, 32, 10, 'multi', sharing='resource-fair'),
           # the headline workload stepping into a ring of 8 fragment buffers (2.75 GB) instead of rewriting ONE 344 MB observation
           # tensor, of which 256 MB can stay in the Infinity Cache from step to step: what a sampler that keeps every step sees
           'config3_65536x32x10_multi_into_8_fragment_buffers': measure_steps(*mk, 65536, 32, 10, 'multi', ring=8),
           'config3_65536x32x10_multi_compact_record': measure_steps(*mk, 65536, 32, 10, 'multi', compact=True)}
    # one GPU's share of the two multi-GPU BASELINE configurations at N = 1 / 2 / 4 / 8 (strong scaling: total size fixed), kernel
    # time by HIP events -- the per-GPU roofline of every point of the curve a multi-GPU node will draw
    for name, total, U, B in (('config5_per_gpu_share_of_32768x128x32', 32768, 128, 32), ('config4_per_gpu_share_of_262144x32x10', 262144, 32, 10)):
        out[name] = {f'N{n}_{total // n}_envs': measure_steps(*mk, total // n, U, B, 'multi', steps=200 if total // n * U * B > 6e7 else 300)
                     for n in (1, 2, 4, 8)}
    # config 5 with the compact record: the whole job on one GPU and the N = 8 share (the launches the row format binds to the HBM write rate)
    out['config5_compact_record'] = {f'N{n}_{32768 // n}_envs': measure_steps(*mk, 32768 // n, 128, 32, 'multi', steps=200, compact=True) for n in (1, 8)}
    out['config5_share_4096x128x32_multi'] = out['config5_per_gpu_share_of_32768x128x32']['N8_4096_envs']
    out['config4_share_32768x32x10_multi'] = out['config4_per_gpu_share_of_262144x32x10']['N8_32768_envs']      # last: the headline's own kernel
    return out


# ------------------------------------------------------------------------------------------------- the predicted 1 -> 8 curve
# Measured inputs (one MI355X, profiles/ of rounds 4-5) and stated assumptions; `python bench.py --predict` prints
# the table without a GPU, DESIGN.md section 7 holds a copy, tests/test_bench_cpu.py pins the two together.
PREDICT = {
    'kernel_ms_config3': 0.0765,            # step_kernel<10,32,2>, 65 536 envs, HIP events (BENCH_r05: 0.0765; steady state 0.0757)
    'n1_ms_per_step_steps20': 0.0778,       # BENCH_r05 ms_per_step (plain process, no process group): kernel + the closing synchronize
    'dist_ms_per_step_steps20_one_rank': 0.0835,   # `--gpus 1 --spawn` (process group with ONE rank, the N > 1 code path): 0.082-0.085
    'rank_spread': 0.03,                    # box-to-box / GPU-to-GPU spread of the same kernel seen over five rounds: +-3 %; value uses the MAX over ranks
    'closing_allreduce_us_per_log2N': 8.0,  # ASSUMPTION: RCCL 4-byte all-reduce over xGMI ~ 20-45 us at 2-8 ranks = +8 us per doubling on top of the one-rank figure
    'config4_share_ms': {1: 0.305, 2: 0.146, 4: 0.076, 8: 0.041},     # 262 144 / N envs x 32 x 10, kernel time (HIP events)
    'config5_share_ms': {1: 0.486, 2: 0.248, 4: 0.124, 8: 0.054},     # 32 768 / N envs x 128 x 32
    'sharded_host_overhead': 0.02,          # measure_sharded times 200 steps by the host clock incl. two reset launches and the closing barrier: +2 %
    'link_GBps_per_direction': 76.5,        # ASSUMPTION: one xGMI link = 153 GB/s both directions together (the task statement's "7 links x ~153 GB/s per GPU") -> 76.5 GB/s one way
    'link_efficiency': 0.8,                 # ASSUMPTION: fraction of a link's peak a large RCCL transfer sustains
}


def predict(steps=20, fragment=4):
    """The curve the first 8-GPU run should be read against.  Weak scaling (the headline: 65 536 envs per GPU): every rank launches the same
    kernel, nothing is exchanged on the data path, so ms_per_step(N) = the one-rank figure of the N > 1 code path + the closing collective's
    growth with N, stretched by the slowest of N ranks (MAX over ranks).  Strong scaling (configs 4 and 5): the measured per-GPU shares.
    With the rollout hand-off north_star names (every observation to every rank): a DIRECT all-gather on the fully connected mesh moves each
    rank's fragment over N - 1 links at once, so a fragment takes bytes_per_rank / (link rate) whatever N is -- and that, not stepping,
    bounds the job.  A ring would take (N - 1) x as long."""
    P = PREDICT
    E, U, B = 65536, 32, 10
    link = P['link_GBps_per_direction'] * P['link_efficiency'] * 1e9
    rows = []
    for n in (1, 2, 4, 8):
        import math
        if n == 1:
            ms = P['n1_ms_per_step_steps20'] if steps <= 50 else P['kernel_ms_config3'] + (P['n1_ms_per_step_steps20'] - P['kernel_ms_config3']) * 20 / steps
        else:
            fence = (P['dist_ms_per_step_steps20_one_rank'] - P['kernel_ms_config3']) * 20 / steps       # fixed costs of the bracket, spread over the steps
            fence += P['closing_allreduce_us_per_log2N'] * 1e-3 * math.log2(n) / steps
            # MAX over n ranks of a +-spread population: expected maximum of n uniform samples = mean + spread (n - 1) / (n + 1)
            ms = (P['kernel_ms_config3'] + fence) * (1 + P['rank_spread'] * (n - 1) / (n + 1))
        value = n * E / (ms * 1e-3)
        frag_rows = fragment * (E * U * (4 * B + 1) + E * U) * 4              # bytes one rank sends per fragment (rows + rewards)
        frag_cmp = fragment * (E * (U * (B + 2) + 2 * B) + E * U) * 4          # ... as the compact record
        step_ms = fragment * P['kernel_ms_config3']

        def with_handoff(bytes_per_rank):
            if n == 1:
                return E * fragment / (step_ms * 1e-3), 0.0
            t = bytes_per_rank / link * 1e3                                   # all N - 1 peers at once, one link each
            per = max(step_ms, t)                                             # overlapped on the side stream: the longer of the two
            return n * E * fragment / (per * 1e-3), t
        r_rate, r_ms = with_handoff(frag_rows)
        c_rate, c_ms = with_handoff(frag_cmp)
        rows.append({'n_gpus': n, 'weak_ms_per_step': ms, 'weak_value_env_steps_per_s': value,
                     'config4_strong_ms_per_step': P['config4_share_ms'][n] * (1 + P['sharded_host_overhead']),
                     'config5_strong_ms_per_step': P['config5_share_ms'][n] * (1 + P['sharded_host_overhead']),
                     'with_rollout_handoff_rows_env_steps_per_s': r_rate, 'rows_fragment_transfer_ms': r_ms,
                     'with_rollout_handoff_compact_env_steps_per_s': c_rate, 'compact_fragment_transfer_ms': c_ms})
    base = rows[0]
    for r in rows:
        r['weak_speedup'] = r['weak_value_env_steps_per_s'] / base['weak_value_env_steps_per_s']
        r['config4_strong_speedup'] = base['config4_strong_ms_per_step'] / r['config4_strong_ms_per_step']
        r['config5_strong_speedup'] = base['config5_strong_ms_per_step'] / r['config5_strong_ms_per_step']
    return {'steps': steps, 'fragment_steps': fragment, 'inputs': P, 'rows': rows,
            'reading': ('weak: value(N) / value(1) -- north_star asks for >= 6 at N = 8; config 5 is super-linear because its 4 096-env share fits the '
                        '256 MB Infinity Cache while the whole job does not; the hand-off columns are LINK-bound at every N > 1 (a fragment of '
                        f'{fragment} steps is {fragment * P["kernel_ms_config3"]:.2f} ms of stepping against tens of ms on a link), which is why `value` '
                        'carries the per-env summary and the full-observation hand-off is reported next to it')}


def predict_table(pr):
    L = [f"| N | weak ms/step (--steps {pr['steps']}) | weak env-steps/s | speed-up | config 4 strong ms/step (speed-up) | config 5 strong ms/step (speed-up) | "
         f"+ rows hand-off env-steps/s (ms per {pr['fragment_steps']}-step fragment on a link) | + compact-record hand-off |", '|---|---|---|---|---|---|---|---|']
    for r in pr['rows']:
        L.append(f"| {r['n_gpus']} | {r['weak_ms_per_step']:.4f} | {r['weak_value_env_steps_per_s']:.3e} | {r['weak_speedup']:.2f} | "
                 f"{r['config4_strong_ms_per_step']:.3f} ({r['config4_strong_speedup']:.2f}) | {r['config5_strong_ms_per_step']:.3f} ({r['config5_strong_speedup']:.2f}) | "
                 f"{r['with_rollout_handoff_rows_env_steps_per_s']:.2e} ({r['rows_fragment_transfer_ms']:.1f}) | "
                 f"{r['with_rollout_handoff_compact_env_steps_per_s']:.2e} ({r['compact_fragment_transfer_ms']:.1f}) |")
    return '\n'.join(L)


def stream_ceiling(torch, dev, write_bytes, rw_bytes, iters=100):
    """"also report against a measured device-copy bandwidth on the box".  torch's own elementwise kernels
    (fill = write-only, out-of-place add = read + write) on buffers of the step kernel's traffic, HIP-event timed."""
    def timed(fn):
        for _ in range(10):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / iters * 1e-3
    dst = torch.empty(write_bytes // 4, dtype=torch.float32, device=dev)
    src = torch.ones(rw_bytes // 4, dtype=torch.float32, device=dev)
    t_fill = timed(lambda: dst.fill_(1.0))
    t_copy = timed(lambda: torch.add(src, 1.0, out=dst[:src.numel()]))
    return {'fill_GBps': write_bytes / t_fill / 1e9, 'copy_GBps': 2 * rw_bytes / t_copy / 1e9,
            'what': f'torch fill_ of {write_bytes / 1e6:.0f} MB (write-only) / out-of-place add of {rw_bytes / 1e6:.0f} MB (read+write)'}


def self_spawn(n):
    """`python bench.py --gpus N` with no launcher around it (the reference's scale-out needs none either: `num_workers`,
    env_setup.py:266): start the N ranks here -- N copies of this command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, exactly what torch.distributed.run would have set -- relay their output and print rank 0's JSON line
    LAST.  A rank that dies takes the others down (no hung rendezvous)."""
    import socket
    import subprocess
    # a port BELOW the kernel's ephemeral range (32768-60999): one found with bind(0) lies inside it, and in the seconds until rank 0 has
    # imported torch and bound its store, any outgoing connection on the box can be handed the same number (seen once: EADDRINUSE)
    import random
    port = None
    for _ in range(64):
        cand = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(('127.0.0.1', cand))
            port = cand
        except OSError:
            pass
        finally:
            s.close()
        if port is not None:
            break
    if port is None:
        sys.exit("bench.py: no free rendezvous port between 20000 and 32000")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   DCOMP_BENCH_SPAWNED='1')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))   # only rank 0 owns stdout
    import threading
    got = {'json': None}

    def relay():                                       # rank 0's stdout; the other ranks' goes to stderr (RCCL banners)
        for line in procs[0].stdout:
            if line.startswith('{"metric"'):
                got['json'] = line.rstrip('\n')
            else:
                sys.stdout.write(line)
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    failed = None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0) and failed is None:
                failed = (r, p.returncode)
        if failed is not None:                         # a dead rank would leave the others hanging in a collective / rendezvous
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    th.join(timeout=10)
    sys.stdout.flush()
    bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
    if got['json'] is not None and not bad:
        print(got['json'], flush=True)
        sys.exit(0)
    sys.stderr.write(f'bench.py: self-spawned ranks failed (rank, exit code): {bad or failed}\n')
    sys.exit(1)


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _fmt_cpulist(cpus):
    out, run = [], []
    for c in sorted(cpus) + [None]:
        if run and (c is None or c != run[-1] + 1):
            out.append(str(run[0]) if len(run) == 1 else f'{run[0]}-{run[-1]}')
            run = []
        if c is not None:
            run.append(c)
    return ','.join(out)


def pin_to_gpu_numa_node(torch, local_rank, enable=True, sysfs='/sys'):
    """Bind THIS rank (and the threads it starts later: RCCL proxies, the gloo store) to the CPUs of the NUMA node its GPU hangs off.
    Eight Python launch loops at 40-80 us per step on a two-socket host otherwise run wherever the scheduler puts them, half of them a
    socket away from their GPU's PCIe root.  Works under any launcher -- torch.distributed.run sets no affinity --
    because every rank does it for itself: PCI address of cuda:<local_rank> -> /sys/bus/pci/devices/<bdf>/numa_node ->
    /sys/devices/system/node/node<n>/cpulist, intersected with the affinity the process was given (cgroup / taskset limits are kept).
    Returns what was done, for config.rank_placement."""
    info = {'local_rank': local_rank, 'pinned': False}
    try:
        before = os.sched_getaffinity(0)
        info['cpus_before'] = len(before)
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f'{getattr(pr, "pci_domain_id", 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        info['pci'] = bdf
        node = int(open(f'{sysfs}/bus/pci/devices/{bdf}/numa_node').read())
        info['numa_node'] = node
        if node < 0:
            info['why_not'] = 'the platform reports no NUMA node for this GPU (numa_node = -1)'
            return info
        cpus = _parse_cpulist(open(f'{sysfs}/devices/system/node/node{node}/cpulist').read()) & before
        if not cpus:
            info['why_not'] = f'no CPU of node {node} is in this process\'s affinity mask'
            return info
        info['cpus'] = _fmt_cpulist(cpus)
        if not enable:
            info['why_not'] = '--no-pin'
            return info
        if cpus != before:
            os.sched_setaffinity(0, cpus)
        info['pinned'] = True
    except Exception as ex:      # noqa: BLE001 -- placement is an optimisation: never fail the run over it
        info['why_not'] = f'{type(ex).__name__}: {ex}'[:200]
    return info


def traffic_from_profile(workload_key, kernel_name=None, want_entry=False):
    """roofline.traffic comes from a TRACKED rocprofv3 --pmc summary (profiles/traffic.json, registered by
    tools/register_traffic.py from a summary tools/summarize_prof.py wrote on the GPU box), never from a literal in this file:
    HBM bytes per launch = FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in
    KiB.  An entry is valid while (a) the KERNEL sources the profiled library was built from are the ones this library was built
    from -- content hash of csrc/dcomp_inst.hip + the device headers + flags (build.kernel_fingerprint; no git needed on the GPU
    box; host-side ABI edits cannot change a kernel's traffic) -- and (b) the library dispatches this workload to the very
    instantiation that was profiled (dcomp_step_kernel_name).  Otherwise it is STALE: traffic is null and the source says why."""
    path = os.path.join(REPO, 'profiles', 'traffic.json')
    try:
        ent = json.load(open(path)).get(workload_key)
    except (OSError, ValueError):
        ent = None
    if not ent:
        return None, f'no entry for {workload_key!r} in profiles/traffic.json'
    from deepcomp_amd import build as hip_build
    src = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/{ent['tag']}_summary.txt, kernel {ent['kernel']}, commit {ent.get('commit', '?')}"
    if ent.get('kernel_fingerprint') != hip_build.kernel_fingerprint() and ent.get('source_fingerprint') != hip_build.source_fingerprint():
        return None, 'STALE (kernel sources changed since): ' + src
    if ent['kernel'].startswith('big_kernel') and ent.get('source_fingerprint') != hip_build.source_fingerprint() \
            and ent.get('generic_fingerprint') != hip_build.generic_fingerprint():
        return None, 'STALE (csrc/dcomp_big.h changed since): ' + src
    if kernel_name is not None and kernel_name != ent['kernel']:
        return None, f'STALE (this library dispatches to {kernel_name}): ' + src
    if want_entry:
        return ent, src
    return (2.0 * ent['fetch_kib'] + ent['write_kib']) * 1024.0, src


SIMDS, CLOCK_GHZ, CYCLES_PER_VALU = 1024, 2.4, 4      # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies a 16-lane SIMD for 4 cycles


class SmiSampler:
    """amdsmi gpu_metrics polled from a thread while a block of launches runs: the gfx clock and socket power the part really holds under
    this kernel (sustained stepping is power-limited: ~2.1 GHz at ~1.37 kW, not the 2.4 GHz peak; profiles/r05_c3_clock_trace.txt).
    The firmware refreshes its (filtered) table every ~20 ms, so the block has to last a few tens of ms.  Never fails the run."""

    def __init__(self, index=0, period=0.004):
        import threading
        self.rows, self._stop, self.err = [], threading.Event(), None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self._h = amdsmi.amdsmi_get_processor_handles()[index]
            self._get = amdsmi.amdsmi_get_gpu_metrics_info
            self._get(self._h)
        except Exception as ex:      # noqa: BLE001
            self.err = f'{type(ex).__name__}: {ex}'[:120]
            return
        self._period = period
        self._th = threading.Thread(target=self._loop, daemon=True)
        self._th.start()

    def _loop(self):
        while not self._stop.is_set():
            try:
                m = self._get(self._h)
                clk = [c for c in (m.get('current_gfxclks') or []) if isinstance(c, (int, float)) and 0 < c < 60000]
                if clk:
                    self.rows.append((sum(clk) / len(clk), m.get('current_socket_power'), m.get('ppt_residency_acc')))
            except Exception:        # noqa: BLE001
                pass
            self._stop.wait(self._period)

    def stop(self):
        if self.err:
            return {'error': self.err}
        self._stop.set()
        self._th.join(timeout=1.0)
        if not self.rows:
            return {'error': 'no sample'}
        last = self.rows[-3:]                     # the table is a filtered view: the last readings of the block are the closest to its steady state
        pw = [r[1] for r in last if isinstance(r[1], (int, float))]
        ppt = [r[2] for r in self.rows if isinstance(r[2], (int, float))]
        return {'gfxclk_mhz': sum(r[0] for r in last) / len(last), 'socket_power_w': sum(pw) / len(pw) if pw else None,
                'power_throttle_residency_advanced': bool(ppt and ppt[-1] > ppt[0]), 'samples': len(self.rows),
                'how': 'amdsmi gpu_metrics (firmware-filtered, ~20 ms refresh) polled while the steady-state launches and ~0.25 s of further stepping ran; mean of the last 3 readings'}


def valu_bound(ent, kernel_ms):
    """the path is transcendental / VALU heavy for a streaming kernel, so the vector-issue bound is reported
    next to the HBM one.  SQ_INSTS_VALU per launch (tracked --pmc profile) x 4 cycles, spread over 1 024 SIMDs at the 2.4 GHz peak
    clock = the shortest time the launch's vector instructions can issue in (FP64 and transcendental instructions take longer
    than 4 cycles, so the true floor is higher and `frac` a lower bound of the ALU occupancy)."""
    n = ent.get('valu_insts_per_launch')
    if not n:
        return None
    floor_ms = n * CYCLES_PER_VALU / (SIMDS * CLOCK_GHZ * 1e9) * 1e3
    return {'insts_per_launch': n, 'insts_per_wave': n / ent['waves_per_launch'] if ent.get('waves_per_launch') else None,
            'issue_floor_ms': floor_ms, 'frac': floor_ms / kernel_ms,
            'peak': f'{SIMDS} SIMDs x {CLOCK_GHZ} GHz / {CYCLES_PER_VALU} cycles per wave64 VALU instruction'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--ues', type=int, default=32)
    ap.add_argument('--bs', type=int, default=10)
    ap.add_argument('--kind', default='multi', choices=['multi', 'central'])
    ap.add_argument('--sharing', default='mixed')
    ap.add_argument('--eps-length', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rollout', type=int, default=0, help='issue steps in chunks of T through dcomp_rollout (one host call per chunk)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1 ('nccl' = RCCL; 'gloo' for single-GPU dry runs)")
    ap.add_argument('--force-dist', action='store_true', help='initialise torch.distributed even with one rank (exercises the RCCL path on one GPU)')
    ap.add_argument('--same-device', action='store_true', help='dry run: every rank uses cuda:0')
    ap.add_argument('--no-gather', action='store_true', help='N>1: skip the per-episode all-gather of the rollout summary')
    ap.add_argument('--gather', default='summary', choices=['summary', 'obs'],
                    help="learner hand-off inside the timed region (N>1 or --force-dist): 'summary' = end-of-episode rewards + utility once per "
                         "episode (default); 'obs' = every fragment of --fragment steps of observations + rewards, all-gathered on a side stream "
                         "while the next fragment is being stepped")
    ap.add_argument('--fragment', type=int, default=4, help='steps per rollout fragment for --gather obs and for the post-run obs hand-off probe')
    ap.add_argument('--compact-step', action='store_true',
                    help="NOT the headline: the steps write the lossless compact observation record themselves (dcomp_out.obs_compact) instead of "
                         "the rows; roofline.achieved then counts the bytes of THAT layout (profiling aid: tools/profile_gpu.sh ... --compact-step)")
    ap.add_argument('--compact', action='store_true',
                    help="--gather obs: hand the observations over as the lossless compact record (U (B + 2) + 2B words per env-step instead of "
                         "U (4B + 1), 3.2x fewer bytes at 32 x 10), written by the steps themselves (dcomp_out.obs_compact: no rows, no pack pass)")
    ap.add_argument('--compact-via-pack', action='store_true',
                    help="--gather obs --compact: the steps write rows and dcomp_pack_fragment packs them inside the timed region (the two-pass form)")
    ap.add_argument('--gather-every', type=int, default=0,
                    help="--gather summary: steps between two hand-offs (all-gather of the per-env reward + sum_utility since the last one). "
                         "0 = min(episode length, max(4, steps // 2)): at least one collective falls inside ANY timed region")
    ap.add_argument('--idle-before-timing-us', type=int, default=0,
                    help="diagnostic: the host sleeps this long (GPU idle) right before the timed region -- how much of an N > 1 line's slower "
                         "kernels is the idle time its barrier adds in front of a 1.6 ms region")
    ap.add_argument('--prewarm', type=int, default=300,
                    help='untimed launches of the headline kernel on every rank before the warm-up steps (clock / power management settles over '
                         'the first ~300 launches after idle; identical at every N so that the points of a scaling curve share one warm state)')
    ap.add_argument('--no-also', action='store_true', help='skip the secondary BASELINE config 2 measurement')
    ap.add_argument('--no-stream', action='store_true', help='skip the measured fill/copy bandwidth (roofline.measured_stream)')
    ap.add_argument('--no-check', action='store_true', help='skip the device error-flag check (ablation builds)')
    ap.add_argument('--traffic-bytes', type=float, default=None, help='HBM bytes per launch from a rocprofv3 --pmc pass')
    ap.add_argument('--also-after', action='store_true', help='measure the secondary configurations after the timed region (round-2 order; A/B)')
    ap.add_argument('--no-pin', action='store_true', help="do not bind the rank to the CPUs of its GPU's NUMA node (A/B; config.rank_placement says what was done)")
    ap.add_argument('--spawn', action='store_true', help='start the ranks from this process even for --gpus 1 (what --gpus N > 1 does by itself '
                                                         'when no launcher set WORLD_SIZE)')
    ap.add_argument('--predict', action='store_true',
                    help='no GPU needed: print the PREDICTED 1 / 2 / 4 / 8-GPU curve (weak scaling of the headline at --steps, strong scaling of configs 4 / 5, '
                         'throughput with the rollout hand-off) from the measured single-GPU figures and the stated link assumptions, then exit')
    ap.add_argument('--rccl-direct', action='store_true',
                    help="N > 1: before the communicator is created, set RCCL's hints for the DIRECT all-gather (each rank writes its shard to every peer over "
                         "that pair's own xGMI link) instead of the ring for the observation hand-off (deepcomp_amd.sharded.rccl_direct_hints); compare with the default")
    ap.add_argument('--gather-algo', default='collective', choices=['collective', 'p2p'],
                    help="how RolloutGather moves a fragment: 'collective' = all_gather_into_tensor (RCCL chooses), 'p2p' = one send to / one receive from every "
                         "peer in one batch (the direct all-gather spelled out)")
    ap.add_argument('--sustained-launches', type=int, default=40000,
                    help='N = 1, default workload: back-to-back headline launches of the sustained leg (also.sustained; blocks of 1 000, HIP events); 0 = skip')
    args = ap.parse_args()
    if args.predict:
        pr = predict(steps=args.steps if args.steps != 1000 else 20, fragment=args.fragment)
        print(predict_table(pr))
        print(json.dumps({'predicted': pr}), flush=True)
        return
    if args.rccl_direct:
        from deepcomp_amd.sharded import rccl_direct_hints
        os.environ.update(rccl_direct_hints())          # inherited by self-spawned ranks; read by RCCL when the communicator comes up
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.spawn):
        self_spawn(args.gpus)                  # does not return

    import torch
    import torch.distributed as dist
    from deepcomp_amd import build as hip_build
    if not hip_build.up_to_date():           # fresh checkout without binaries (they are git-ignored)
        if int(os.environ.get('RANK', '0')) == 0:
            hip_build.build()
        else:
            for _ in range(600):
                if hip_build.up_to_date():
                    break
                time.sleep(1.0)
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.same_device:
        local_rank = 0
    spawned = os.environ.get('DCOMP_BENCH_SPAWNED') == '1'      # a rank self_spawn() started: always a process group, also with one rank
    use_dist = world > 1 or args.force_dist or spawned
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; run `python bench.py --gpus {args.gpus}` without a "
                 f"launcher (it spawns its ranks itself) or torch.distributed.run with --nproc-per-node {args.gpus}")
    if not args.same_device and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible (--same-device --backend gloo: dry run on one)")
    t_wall0 = time.time()
    affinity_at_start = os.sched_getaffinity(0)
    placement = pin_to_gpu_numa_node(torch, local_rank, enable=not args.no_pin)
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local_rank)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    E, U, B, K, W, L = args.envs, args.ues, args.bs, args.steps, args.warmup, args.eps_length
    scn = scenarios.grid_map(B, args.sharing).with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)
    env = BatchedMobileEnv(m, bs, ues, args.kind, num_envs=E, seed=42, episode_length=L, rng='philox', rand_episodes=True,
                           device=dev, env_id_base=rank * E, log_metrics=True)
    # uniform random actions, Philox-seeded on device, outside the timed region; a pool of 16 tensors is cycled
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    pool = torch.randint(0, B + 1, (16, E, U), generator=g, device=dev, dtype=torch.uint8)

    T = args.rollout
    if T:
        assert L % T == 0 and K % T == 0 and W % T == 0, "--rollout T must divide the episode length, steps and warmup"
        tape = torch.randint(0, B + 1, (4, T, E, U), generator=g, device=dev, dtype=torch.uint8)

    # N>1: the learner hand-off.  Each GPU keeps its observations (data-parallel learner); what every rank
    # needs from the others is the per-env end-of-episode summary (rewards + utility): ONE RCCL all-gather per episode, issued
    # asynchronously on a side stream.  (All-gathering the observations themselves would be 8 x 344 MB per step.)
    gather = None
    pending = []
    if use_dist and not args.no_gather:
        from deepcomp_amd.sharded import RolloutGather
        gather = RolloutGather(use_side_stream=(args.backend == 'nccl'), reuse_buffers=3, algo=args.gather_algo)
        # the summary of one hand-off is ONE tensor [E, reward columns + 1] (per-env rewards | sum_utility): one staging kernel, one collective
        summary_stage = [torch.empty((E, env.reward.numel() // E + 1), device=dev) for _ in range(3)]

    # --gather obs: the rollout hand-off north_star describes.  Steps write into a fragment buffer [F, E, U, 4B+1] (two of
    # them, alternating); a finished fragment is all-gathered on the side stream while the next one is being stepped.
    F = args.fragment
    # summary hand-off period: once per episode (the cadence of a learner's per-episode logging), or once per timed region where that is
    # shorter -- with the half-period phase shift below exactly one collective then falls inside a region of K < L steps (round 4 first
    # used K // 2: two hand-offs in the driver's 20 steps, 10 % of a 1.6 ms region; one is what "self-proving" needs)
    G = args.gather_every or min(L, max(4, K))
    frag_bufs, frag_pending, gather_stats = None, [None, None], {'wait_s': 0.0, 'fragments': 0, 'collectives': 0, 'bytes_sent': 0, 'since': 0}

    def count_collectives(frag):
        """RolloutGather issues ONE all_gather_into_tensor per tensor of the fragment."""
        gather_stats['collectives'] += len(frag)
        gather_stats['bytes_sent'] += sum(v.numel() * v.element_size() for v in frag.values())
    if gather is not None and args.gather == 'obs':
        assert L % F == 0 and K % F == 0 and W % F == 0, "--fragment must divide the episode length, steps and warmup"
        direct_compact = args.compact and not args.compact_via_pack
        frag_bufs = [{'obs': None if direct_compact else torch.empty((F,) + tuple(env.obs.shape), device=dev),
                      'reward': torch.empty((F,) + tuple(env.reward.shape), device=dev)} for _ in range(2)]
    codec = None
    if args.compact or (use_dist and not args.no_gather and args.kind == 'multi'):
        from deepcomp_amd.fragment import FragmentCodec
        codec = FragmentCodec(U, B, device=dev) if args.kind == 'multi' else None
    if args.compact and codec is None:
        sys.exit("bench.py: --compact packs multi-agent observation rows (--kind multi)")
    if frag_bufs is not None and args.compact:
        for fb in frag_bufs:
            fb['packed'] = torch.empty((F, E, codec.words), dtype=torch.int32, device=dev)

    packed_main = None
    if args.compact_step:
        if args.kind != 'multi' or T or frag_bufs is not None:
            sys.exit("bench.py: --compact-step is for multi-agent envs stepped one launch per step, without --gather obs")
        packed_main = torch.empty((E, env.compact_words), dtype=torch.int32, device=dev)

    def timed_wait(h):
        """h.wait() makes the compute stream wait for the collective; the HIP events around it time that stall on the GPU
        (0 when the gather had finished long before), the host clock what the host itself blocked."""
        t0 = time.perf_counter()
        if dev.type == 'cuda' and args.backend == 'nccl':
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            h.wait()
            b.record()
            gather_stats.setdefault('stall_events', []).append((a, b))
        else:
            h.wait()
        gather_stats['wait_s'] += time.perf_counter() - t0

    def step_into_fragment(t):
        """One step whose outputs land in the current fragment buffer; hands the fragment over when it is full."""
        k, f = (t // F) & 1, t % F
        if f == 0 and frag_pending[k] is not None:      # about to overwrite a buffer: its gather must have read it
            timed_wait(frag_pending[k])
            frag_pending[k] = None
        if args.compact and not args.compact_via_pack:     # the step writes the compact record itself: no rows, no pack pass
            env.step_compact(pool[t & 15], frag_bufs[k]['packed'][f], frag_bufs[k]['reward'][f])
        else:
            env.step_into(pool[t & 15], frag_bufs[k]['obs'][f], frag_bufs[k]['reward'][f])
        if f == F - 1:
            send = frag_bufs[k]
            if args.compact:                          # the hand-off carries the compact record
                if args.compact_via_pack:             # ... packed here, on the compute stream
                    codec.pack(send['obs'], out=send['packed'])
                send = {'obs_compact': send['packed'], 'reward': send['reward']}
            frag = send if args.backend == 'nccl' else {n: v.cpu() for n, v in send.items()}
            frag_pending[k] = gather.all_gather_async(frag)
            count_collectives(frag)
            gather_stats['fragments'] += 1

    def hand_off_summary():
        """The default hand-off: every G steps each rank all-gathers its envs' current reward + sum_utility (the per-env summary a
        data-parallel learner's logging / early stopping needs from the other shards), asynchronously on the side stream."""
        if gather is None:
            return
        stage = summary_stage[gather_stats['collectives'] % 3]              # (at most 3 hand-offs are in flight: `pending` below)
        torch.cat((env.reward.view(E, -1), env.sum_utility.view(E, 1)), dim=1, out=stage)
        frag = {'reward_and_sum_utility': stage}
        if args.backend != 'nccl':
            frag = {k: v.cpu() for k, v in frag.items()}
        pending.append(gather.all_gather_async(frag))
        count_collectives(frag)
        if len(pending) > 2:
            timed_wait(pending.pop(0))

    def run(nsteps, t_start, spans=None):
        """spans: list that receives one HIP-event pair per run of back-to-back step launches between two resets -- the
        events sit on torch's current stream, the stream dcomp_step enqueues on, so a pair brackets pure step-kernel time."""
        t = t_start
        open_span = None

        def close():
            nonlocal open_span
            if open_span is not None:
                b = ev_pool.pop() if ev_pool else torch.cuda.Event(enable_timing=True)
                b.record()
                spans.append((open_span[0], b, t - open_span[1]))
                open_span = None

        def begin():
            nonlocal open_span
            if spans is not None and open_span is None:
                a = ev_pool.pop() if ev_pool else torch.cuda.Event(enable_timing=True)   # normally created before the timed region: hipEventCreate costs tens of microseconds,
                a.record()                      # and in front of the first launch that is idle GPU time inside a 1.7 ms region
                open_span = (a, t)
        if T:
            for i in range(nsteps // T):
                if t % L == 0:
                    close()
                    env.reset()
                begin()
                env.rollout(tape[i & 3])
                t += T
            close()
            return t
        for i in range(nsteps):
            if t % L == 0:
                close()
                env.reset()
            begin()
            if frag_bufs is not None:
                step_into_fragment(t)
            elif packed_main is not None:
                env.step_compact(pool[t & 15], packed_main, env.reward)
            else:
                env.step(pool[t & 15])
            t += 1
            if gather is not None and frag_bufs is None:
                gather_stats['since'] += 1
                if gather_stats['since'] >= G:         # every G steps, phase-shifted by G // 2 against the start of the timed region
                    gather_stats['since'] = 0
                    close()
                    hand_off_summary()
        close()
        return t

    fence_token = torch.zeros(1, device=dev) if use_dist and args.backend == 'nccl' else None

    def fence(closing=False):
        """barrier + synchronize.  Closing side over RCCL: the barrier's collective is enqueued BEHIND this rank's last launch (stream order)
        and completes once every rank has reached it, then ONE device synchronize waits for both -- the same bracket without two further
        host round trips inside a 1.6 ms region."""
        if closing and fence_token is not None:
            dist.all_reduce(fence_token)
            torch.cuda.synchronize(dev)
            return
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def drain():
        for h in pending + [h for h in frag_pending if h is not None]:
            timed_wait(h)
        pending.clear()
        frag_pending[0] = frag_pending[1] = None

    # The box's measured streaming rate is taken BEFORE the warm-up steps: it is part of the output anyway, and a
    # GPU that has just streamed for ~20 ms starts the step launches closer to its steady clocks than one that sat idle
    # through the set-up (the first launches after idle run 5-15 % slower, see roofline.steady_state).
    # EVERYTHING before the warm-up steps runs on EVERY rank at EVERY N (rank 0's figures are printed): the N = 1 point and the N = 8
    # point of a scaling curve enter their timed regions from the same warm state.
    stream_probe = None
    if not args.no_stream:
        bpe0 = bytes_per_env_step(U, B, args.kind)
        stream_probe = stream_ceiling(torch, dev, E * (bpe0 - U * 33) // 4 * 4, E * U * 33 // 4 * 4)
        stream_probe['when'] = 'before the warm-up steps, on every rank'
    # The secondary figures (the other BASELINE configurations on this GPU: ~2 s of back-to-back launches of the same kernels)
    # are measured BEFORE the headline: they are part of the output anyway, and the GPU then enters the W warm-up + K timed steps
    # at the clocks a running job has, instead of inside the 5-15 % slower transient of the first ~300 launches after idle
    # (tools/kprobe.py) that a 25-launch run would otherwise never leave.  `--also-after` restores the old order (A/B).
    default_workload = (E, U, B, args.kind, args.sharing) == (65536, 32, 10, 'multi', 'mixed')
    # The hand-off machinery is brought up HERE, before the secondary configurations and the pre-warm launches: the first use of the side
    # stream and of the communicator (stream + channel set-up, the three rotating sets of gather buffers) is followed by the same 5-15 %
    # slower transient of ~200 launches as a cold start (tools/diag_stream.py: 92 / 84 us per step right after the first hand-off, 76 us
    # with a hand-off in the middle of 100 steps in steady state).  Until the end of round 4 the first hand-off came right before the
    # timed region and the transient fell INTO it: `--force-dist --steps 100` read 0.097 ms per step against 0.081 plain.
    if gather is not None:
        env.reset()
        for _ in range(3):
            if frag_bufs is None:
                hand_off_summary()
            else:
                pending.append(gather.all_gather_async({'reward': env.reward} if args.backend == 'nccl' else {'reward': env.reward.cpu()}))
        drain()
        torch.cuda.synchronize(dev)
    also_first = None
    if not args.no_also and default_workload and not args.also_after:
        also_first = measure_also(torch, BatchedMobileEnv, scenarios, build_from_scenario, dev)
    if args.prewarm > 0:                        # >= 300 untimed launches of the headline kernel itself (no hand-off, no events)
        tp = 0
        while tp < args.prewarm:
            env.reset()
            for i in range(min(L, args.prewarm - tp)):
                if packed_main is not None:
                    env.step_compact(pool[i & 15], packed_main, env.reward)
                else:
                    env.step(pool[i & 15])
            tp += min(L, args.prewarm - tp)
        torch.cuda.synchronize(dev)
    t_env = run(W, 0)
    if gather is not None and frag_bufs is None:
        hand_off_summary()                      # one UNTIMED hand-off: the first collective of a communicator sets up its channels (milliseconds);
    drain()                                     # with --warmup 5 no hand-off of the warm-up steps would have done that before the timed region
    fence()
    # the hand-offs of the timed region fall on its steps G/2, 3G/2, ...: every one has G/2 steps of stepping to overlap with (a
    # hand-off issued at the region's last step could only be waited for), and K >= 4 steps always contain at least one
    gather_stats.update(wait_s=0.0, fragments=0, stall_events=[], collectives=0, bytes_sent=0, since=G // 2)
    spans = []
    ev_pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (K // L + 3) + 8)]
    for e in ev_pool[:2]:
        e.record()                              # first use of the event machinery outside the timed region
    torch.cuda.synchronize(dev)
    if args.idle_before_timing_us:
        time.sleep(args.idle_before_timing_us * 1e-6)
    t0 = time.perf_counter()
    t_env = run(K, t_env, spans)
    drain()
    fence(closing=True)
    elapsed = time.perf_counter() - t0
    elapsed_ranks, placements = [elapsed], [placement]
    if use_dist:
        # every rank's own clock over the same bracket: `value` uses the MAX (the contract), the line also shows min / max / all, so
        # that a straggler rank is visible instead of hidden behind the MAX
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        elapsed_ranks = [float(t.item()) for t in every]
        elapsed = max(elapsed_ranks)
        placements = [None] * world
        dist.all_gather_object(placements, placement)
    if not args.no_check:
        env.check()
    handoff = None
    if gather is not None:
        torch.cuda.synchronize(dev)
        stall_ms = sum(a.elapsed_time(b) for a, b in gather_stats.get('stall_events', []))
        per_frag = F * ((E * codec.words if args.compact else env.obs.numel()) + env.reward.numel()) * 4
        handoff = {'mode': args.gather, 'backend': 'rccl' if args.backend == 'nccl' else args.backend, 'rccl_ranks': dist.get_world_size(),
                   'overlapped_on_side_stream': args.backend == 'nccl', 'algo': args.gather_algo,
                   'rccl_hints': {k: os.environ[k] for k in ('RCCL_DIRECT_ALLGATHER_THRESHOLD', 'NCCL_PROTO', 'NCCL_ALGO') if k in os.environ} or None,
                   # counted where the collectives are ISSUED (count_collectives), reset after the warm-up: what really ran between t0
                   # and the closing fence of the timed region (drain() waits for all of them before the fence)
                   'collectives_in_timed_region': gather_stats['collectives'],
                   'bytes_in_timed_region': {'sent_per_rank': gather_stats['bytes_sent'], 'received_per_rank': gather_stats['bytes_sent'] * world},
                   'compute_stream_stall_ms_total': stall_ms, 'host_blocked_ms_total': gather_stats['wait_s'] * 1e3}
        if args.gather == 'obs':
            handoff.update(fragment_steps=F, fragments=gather_stats['fragments'], compact=bool(args.compact),
                           compact_written_by=(None if not args.compact else 'dcomp_pack_fragment after the steps' if args.compact_via_pack else 'the steps (dcomp_out.obs_compact)'), bytes_sent_per_rank_per_fragment=per_frag,
                           bytes_received_per_rank_per_fragment=per_frag * world,
                           compute_stream_stall_ms_per_fragment=stall_ms / max(1, gather_stats['fragments']))
        else:
            handoff.update(what=f'reward + sum_utility of every env, all-gathered every {G} steps as one tensor [E, reward columns + 1] (one collective per hand-off), asynchronous',
                           period_steps=G, bytes_sent_per_rank_per_handoff=4 * (env.reward.numel() + E))

    def probe_obs_handoff(nfrag=4, compact=False, direct=False):
        """The rollout hand-off north_star names, measured next to the headline (never part of `value`): fragments of F steps
        of observations + rewards all-gathered over RCCL on a side stream while the next fragment is stepped."""
        from deepcomp_amd.sharded import RolloutGather
        g2 = gather if gather is not None else RolloutGather(use_side_stream=(args.backend == 'nccl'), algo=args.gather_algo)
        bufs = [{'obs': None if direct else torch.empty((F,) + tuple(env.obs.shape), device=dev),
                 'reward': torch.empty((F,) + tuple(env.reward.shape), device=dev)} for _ in range(2)]
        if compact:
            for b_ in bufs:
                b_['packed'] = torch.empty((F, E, codec.words), dtype=torch.int32, device=dev)
        env.reset()

        def outgoing(k):
            if not compact:
                return bufs[k] if args.backend == 'nccl' else {n: v.cpu() for n, v in bufs[k].items()}
            if not direct:
                codec.pack(bufs[k]['obs'], out=bufs[k]['packed'])
            send = {'obs_compact': bufs[k]['packed'], 'reward': bufs[k]['reward']}
            return send if args.backend == 'nccl' else {n: v.cpu() for n, v in send.items()}

        def steps(k):
            for f in range(F):
                if direct:                                   # dcomp_out.obs_compact: the step writes the record itself, no rows, no pack pass
                    env.step_compact(pool[f & 15], bufs[k]['packed'][f], bufs[k]['reward'][f])
                else:
                    env.step_into(pool[f & 15], bufs[k]['obs'][f], bufs[k]['reward'][f])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        steps(0); steps(1)                                   # warm
        for i in range(4):                                   # ... the hand-off too: the first calls allocate the gathered tensors (GBs: a hipMalloc
            g2.all_gather_async(outgoing(i & 1)).wait()      # of 1.4 GB inside the timed loop read 30 ms per fragment on one box, 0.9 ms on others)
        fence()
        ev[0].record()
        for i in range(nfrag):
            steps(i & 1)
        ev[1].record()                                       # stepping alone
        fence()
        hs = [None, None]
        t0 = time.perf_counter()
        ev[2].record()
        for i in range(nfrag):
            k = i & 1
            if hs[k] is not None:
                hs[k].wait()
            steps(k)
            hs[k] = g2.all_gather_async(outgoing(k))
        for h in hs:
            if h is not None:
                h.wait()
        ev[3].record()
        fence()
        wall = time.perf_counter() - t0
        per = F * ((E * codec.words if compact else env.obs.numel()) + env.reward.numel()) * 4
        step_ms, both_ms = ev[0].elapsed_time(ev[1]) / nfrag, ev[2].elapsed_time(ev[3]) / nfrag
        return {'compact': compact, 'written_by': ('the step itself (dcomp_out.obs_compact)' if direct else 'dcomp_pack_fragment after the steps') if compact else 'rows',
                'fragment_steps': F, 'fragments': nfrag, 'rccl_ranks': dist.get_world_size(), 'backend': 'rccl' if args.backend == 'nccl' else args.backend,
                'bytes_sent_per_rank_per_fragment': per, 'bytes_received_per_rank_per_fragment': per * world,
                'ms_per_fragment_stepping_only': step_ms, 'ms_per_fragment_with_overlapped_all_gather': both_ms,
                'exposed_handoff_ms_per_fragment': max(0.0, both_ms - step_ms), 'wall_s': wall,
                'all_gather_GBps_per_rank_ingress': per * max(world - 1, 1) / max(both_ms, 1e-9) / 1e6,
                'env_steps_per_s_with_obs_handoff': world * E * F / (both_ms * 1e-3)}

    obs_probe = None
    if use_dist and args.gather != 'obs' and not args.no_gather:
        try:                                   # a side measurement: it must never cost the headline line
            obs_probe = probe_obs_handoff()
            if codec is not None:                  # the same hand-off with the lossless compact record (pack kernel included)
                obs_probe['compact_record'] = probe_obs_handoff(compact=True)
                obs_probe['compact_record_from_the_step'] = probe_obs_handoff(compact=True, direct=True)
        except Exception as ex:                # noqa: BLE001
            obs_probe = {'error': f'{type(ex).__name__}: {ex}'[:300]}
            torch.cuda.synchronize(dev)

    # The multi-GPU BASELINE configurations next to the weak-scaling headline, as STRONG scaling (total size fixed: config 4 =
    # 262 144 x 32 x 10, config 5 = 32 768 x 128 x 32; total / N envs per GPU).  Every rank takes part, at every N: the N = 1 run
    # records the base of the curve (the whole job on one GPU), the N = 8 run IS the BASELINE configuration.
    sharded = None
    if not args.no_also:
        sharded = {}
        mk = (torch, dist if use_dist else None, BatchedMobileEnv, scenarios, build_from_scenario, dev, rank, world, args.backend)
        for name, shape in STRONG:
            try:
                sharded[name] = measure_sharded(*mk, *shape)
            except Exception as ex:            # noqa: BLE001 -- a side measurement must never cost the headline line
                sharded[name] = {'error': f'{type(ex).__name__}: {ex}'[:300]}
                torch.cuda.synchronize(dev)

    # Duration of one step-kernel launch, over the TIMED region itself: the event pairs run() recorded around each run of
    # back-to-back launches between two resets (an event pair per launch would also time the launch latency of an empty
    # queue: round 1's kernel_ms > ms_per_step).
    torch.cuda.synchronize(dev)
    kern_ms = sum(a.elapsed_time(b) for a, b, _ in spans) / max(1, sum(n for _, _, n in spans))
    # what sits BETWEEN the runs of back-to-back step launches on the compute stream (reset launches, the copies of a hand-off, idle)
    gaps_ms = sum(spans[i][1].elapsed_time(spans[i + 1][0]) for i in range(len(spans) - 1))
    # The first few hundred launches after idle run 5-15 % slower (clock / power management settling: 91 -> 115 -> 82 us per
    # launch over 300 launches on a cold MI355X, tools/kprobe.py): with the driver's --steps 20 the timed region lies inside
    # that transient.  The steady state is reported NEXT to it, never instead of it: >= 300 further launches untimed, then 200 timed.
    steady_ms, steady_clk, sustained = None, None, None
    if world == 1 and not T and frag_bufs is None and gather is None:
        t = run(max(0, 300 - K), t_env)
        sp2 = []
        ev_pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(2 * (200 // L + 3)))
        smi = SmiSampler(local_rank)              # gfx clock / socket power WHILE the launches run (profiles/r05_c3_clock_trace.txt)
        t = run(200, t, sp2)
        if smi.err is None:                       # the firmware's table is a ~20 ms filtered view and the clock needs ~0.3 s of load to settle at its
            t = run(3000, t)                      # power-limited value: keep stepping (untimed) until the reading means something
        torch.cuda.synchronize(dev)
        steady_clk = smi.stop()
        steady_ms = sum(a.elapsed_time(b) for a, b, _ in sp2) / sum(n for _, _, n in sp2)
        # SUSTAINED behaviour inside the driver-run line (">= 1 000 steps after 100 warm-up"): >= 3 s of
        # back-to-back headline launches, timed with HIP events per run of launches between two resets (100 launches), reported in
        # blocks of 1 000 launches; the gfx clock and socket power are read while it runs.
        if default_workload and args.sustained_launches >= 2000 and packed_main is None:
            n_s = args.sustained_launches // 1000 * 1000
            sp3 = []
            ev_pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(2 * (n_s // L + 3)))
            smi3 = SmiSampler(local_rank, period=0.02)
            t_s0 = time.perf_counter()
            t = run(n_s, t, sp3)
            torch.cuda.synchronize(dev)
            sus_wall = time.perf_counter() - t_s0
            sus_clk = smi3.stop()
            blocks, acc_ms, acc_n = [], 0.0, 0
            for a_, b_, n_ in sp3:
                acc_ms += a_.elapsed_time(b_)
                acc_n += n_
                if acc_n >= 1000:
                    blocks.append(acc_ms / acc_n)
                    acc_ms, acc_n = 0.0, 0
            bs_ = sorted(blocks)
            pick = lambda q: bs_[min(len(bs_) - 1, int(q * len(bs_)))]      # noqa: E731
            sustained = {'launches': n_s, 'blocks_of': 1000, 'blocks': len(blocks), 'wall_s': sus_wall,
                         'kernel_ms_p10': pick(0.10), 'kernel_ms_median': pick(0.50), 'kernel_ms_p90': pick(0.90),
                         'kernel_ms_first_block': blocks[0], 'kernel_ms_last_block': blocks[-1], 'last_over_first': blocks[-1] / blocks[0],
                         'kernel_ms_mean': sum(blocks) / len(blocks), 'clock_and_power_at_end': sus_clk,
                         'how': f'{n_s} back-to-back launches of the headline kernel after the timed region and the steady-state leg (resets every {L} steps '
                                'in between, outside the event pairs), HIP events per 100-launch run, grouped into blocks of 1 000 launches'}
        else:
            sustained = None
    resets_timed = sum(1 for s in range(t_env - K, t_env) if s % L == 0)
    also_late = None
    if not args.no_also and default_workload and args.also_after:
        also_late = measure_also(torch, BatchedMobileEnv, scenarios, build_from_scenario, dev)

    if rank == 0:
        bpe = bytes_per_env_step(U, B, args.kind)
        sbpe = survey_bytes_per_env_step(U, B, args.kind)
        if packed_main is not None:            # the compact layout's own bytes: 4 (B + 2) per UE + 8B per env instead of 4 (4B + 1) per UE
            sbpe = sbpe - U * 4 * (4 * B + 1) + 4 * env.compact_words
            bpe = bpe - U * 4 * (4 * B + 1) + 4 * env.compact_words
        # algorithmic bytes per env-step x the env-steps one launch processes / launch duration
        achieved = sbpe * E / (kern_ms * 1e-3) / 1e9
        out = {
            'metric': 'env steps/sec', 'value': world * E * K / elapsed, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K,
            'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64 positions / f32 rates', 'data': 'synthetic',
            'config': {'workload': f'{E} envs/GPU x {U} UE x {B} BS, {args.kind}-agent obs, sharing={args.sharing}, '
                                   f'log utility, reward avg, episode {L} ({resets_timed} reset launch(es) inside the timed {K} steps), random actions' + (f', rollout chunks of {T}' if T else '') +
                                   (', observations written as the COMPACT record (dcomp_out.obs_compact; not the BASELINE output format -- roofline bytes are the compact layout\'s)' if packed_main is not None else ''),
                       'envs_per_gpu': E, 'num_ue': U, 'num_bs': B, 'pair_steps_per_s': world * E * K / elapsed * U * B,
                       'parallelism': f'env-shard x{world}' + ("; every rank bound to the CPUs of its GPU's NUMA node" if all(pl and pl.get('pinned') for pl in placements) else ''),
                       'rank_placement': placements,
                       'collective': ('none on the data path' if gather is None else
                                      f'all-gather of {F}-step observation + reward fragments ({F * (env.obs.numel() + env.reward.numel()) * 4 * world / 1e6:.0f} MB received per rank), side stream, overlapped'
                                      if args.gather == 'obs' else
                                      f'all-gather of per-env rewards + sum_utility ({4 * E * (env.reward.numel() // E + 1) * world / 1e6:.1f} MB received per rank) every {G} steps, async, '
                                      f'{gather_stats["collectives"]} collective(s) inside the timed region'),
                       'prewarm': (('secondary configurations (`also`), then ' if also_first is not None else '') +
                                   f'{args.prewarm} untimed launches of the headline kernel, then the {W} warm-up steps -- on every rank, identical at every N '
                                   '(methodology: rounds 1-2 entered the timed region cold, round 3 after `also` at N = 1 only; `--prewarm 0 --also-after` = the round-2 order)')},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'traffic': args.traffic_bytes, 'kernel': 'dcomp::' + (env.step_kernel_name or 'step_kernel'), 'kernel_ms': kern_ms,
                         'kernel_ms_how': f'HIP events over the timed region: {sum(n for _, _, n in spans)} launches in {len(spans)} back-to-back run(s) between resets',
                         'between_runs_ms_total': gaps_ms,
                         'launch_bound': kern_ms < 0.02,
                         'algorithmic_bytes_per_env_step': sbpe, 'layout_bytes_per_env_step': bpe},
        }
        out['elapsed_per_rank_ms'] = {'min': min(elapsed_ranks) * 1e3, 'max': max(elapsed_ranks) * 1e3, 'ranks': [e * 1e3 for e in elapsed_ranks],
                                      'what': 'each rank\'s own host clock over the timed region (fence to fence); value and ms_per_step use the MAX'}
        if handoff is not None:
            out['handoff'] = handoff
        if obs_probe is not None:
            out.setdefault('also', {})['obs_handoff_probe'] = obs_probe
            if handoff is not None and 'error' not in obs_probe:
                # north_star's hand-off is the ROLLOUT batch (observations), `value` carries the per-env summary only (SURVEY 8e's link
                # budget): the throughput with every observation delivered to every rank stands next to it, at top level, so that a
                # 1 -> 8 curve of `value` is never read as "rollouts delivered"
                def _wr(pr):
                    return {'env_steps_per_s': pr['env_steps_per_s_with_obs_handoff'], 'exposed_handoff_ms_per_fragment': pr['exposed_handoff_ms_per_fragment'],
                            'bytes_received_per_rank_per_fragment': pr['bytes_received_per_rank_per_fragment'],
                            'all_gather_GBps_per_rank_ingress': pr['all_gather_GBps_per_rank_ingress']}
                wr = {'what': (f'{F}-step fragments of observations + rewards of EVERY env all-gathered to EVERY rank on a side stream while the next fragment is '
                               'stepped (measured after the timed region, never part of `value`; whole-job env-steps/s, detail in also.obs_handoff_probe)'),
                      'fragment_steps': F, 'rows': _wr(obs_probe)}
                if isinstance(obs_probe.get('compact_record'), dict):
                    wr['compact_record_packed_after_the_steps'] = _wr(obs_probe['compact_record'])
                if isinstance(obs_probe.get('compact_record_from_the_step'), dict):
                    wr['compact_record_written_by_the_steps'] = _wr(obs_probe['compact_record_from_the_step'])
                wr['value_carries'] = 'the summary hand-off (handoff.what), not these'
                handoff['with_rollout_handoff'] = wr
        if sharded:
            out.setdefault('also', {}).update(sharded)
        if steady_ms is not None:
            out['roofline']['steady_state'] = {'kernel_ms': steady_ms, 'achieved': sbpe * E / (steady_ms * 1e-3) / 1e9,
                                               'frac': sbpe * E / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               'how': f'200 launches timed after {max(300, K)} untimed ones (same HIP-event method)',
                                               'clock_and_power': steady_clk}
        if sustained is not None:
            sustained['achieved_at_median'] = sbpe * E / (sustained['kernel_ms_median'] * 1e-3) / 1e9
            sustained['frac_at_median'] = sustained['achieved_at_median'] / HBM_PEAK_GBS
            out.setdefault('also', {})['sustained'] = sustained
        if args.traffic_bytes is None:
            ent, src = traffic_from_profile(f'{E}x{U}x{B}_{args.kind}_{args.sharing}' + ('_compact' if packed_main is not None else ''), env.step_kernel_name, want_entry=True)
            out['roofline']['traffic'] = (2.0 * ent['fetch_kib'] + ent['write_kib']) * 1024.0 if ent else None
            out['roofline']['traffic_source'] = src
            if ent and ent.get('avg_ns'):
                # the same 7 652 B x 65 536 over the TRACKED rocprofv3 --kernel-trace mean of this kernel (profiles/<tag>_summary.txt): what a
                # reader of profiles/ recomputes, next to the HIP-event figure of this run
                out['roofline']['frac_profile'] = sbpe * E / (ent['avg_ns'] * 1e-9) / 1e9 / HBM_PEAK_GBS
                out['roofline']['frac_profile_how'] = f"algorithmic bytes / the rocprofv3 --kernel-trace --stats mean ({ent['avg_ns'] / 1e3:.2f} us over {ent.get('calls', '?')} launches) in profiles/{ent['tag']}_summary.txt"
            if ent and valu_bound(ent, kern_ms):
                out['roofline']['valu'] = valu_bound(ent, kern_ms)
                if steady_ms is not None and steady_clk and steady_clk.get('gfxclk_mhz'):
                    # the same floor at the clock the part HOLDS under this kernel (power-limited), against the steady-state launch time
                    ghz = steady_clk['gfxclk_mhz'] / 1e3
                    fl = ent['valu_insts_per_launch'] * CYCLES_PER_VALU / (SIMDS * ghz * 1e9) * 1e3
                    out['roofline']['valu']['at_sustained_clock'] = {'gfxclk_ghz': ghz, 'issue_floor_ms': fl, 'frac': fl / steady_ms,
                                                                     'of': 'roofline.steady_state.kernel_ms'}
        else:
            out['roofline']['traffic_source'] = '--traffic-bytes'
        if not args.no_stream:
            # the launch writes (obs + reward + info + state) and reads (state + actions); ceiling for that mix
            sc = stream_probe
            sc['frac_of_fill'] = achieved / sc['fill_GBps']
            out['roofline']['measured_stream'] = sc
        if also_first is not None:
            out.setdefault('also', {}).update(also_first)
            out['also']['measured'] = 'before the warm-up steps of the headline, on every rank'
        if also_late is not None:
            out.setdefault('also', {}).update(also_late)
            out['also']['measured'] = 'after the timed region (round-2 order)'
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(scn, args.kind, U, B, restore_affinity=affinity_at_start)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out['wall_s'] = time.time() - t_wall0          # this rank's whole run after `import torch` (set-up, secondary figures, CPU baseline)
        # RCCL prints a version banner through C stdio; flush it first so that the JSON line is the LAST line of output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:      # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
