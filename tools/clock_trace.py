#!/usr/bin/env python3
"""Where does the launch-time spread of the headline kernel come from?  (rocprofv3 min 71.5 / mean 77-80 us.)

    python tools/clock_trace.py [--launches 3000] [--out gpurun_out/clock_trace]           # plain: HIP events per block of launches
    rocprofv3 --kernel-trace -d DIR -o kt --output-format csv -- python tools/clock_trace.py ...   # per-launch durations beside it
    python tools/clock_trace.py --report gpurun_out/clock_trace [--kernel-trace DIR]

Two things run side by side: (1) a SAMPLER process polling amdsmi's gpu_metrics (per-XCD gfx clock, memory clock, socket power, hotspot /
HBM temperature, the firmware's throttle-residency accumulators) as fast as the call returns, stamped with CLOCK_MONOTONIC / BOOTTIME /
REALTIME; (2) the bench workload (BASELINE config 3, the bench loop's reset-every-100 cadence), launched back to back with one HIP
event per BLOCK launches and a host stamp per block.  The report lines launch duration up against (a) the clock / power samples of the
same interval, (b) the launch's position inside its episode (launches since the last reset_kernel), (c) the time since the stream last
idled.  Nothing here is on a product path.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def sysfs_sampler(path, period, why):
    """Fallback without amdsmi: hwmon freq1_input (gfx clock, Hz), freq2_input (memory clock), power1_average / power1_input (uW), temp2_input."""
    import glob
    hw = [d for d in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*') if os.path.exists(os.path.join(d, 'freq1_input'))]

    def rd(name):
        try:
            return int(open(os.path.join(hw[0], name)).read())
        except (OSError, ValueError, IndexError):
            return None
    with open(path, 'w') as f:
        f.write(json.dumps({'t': (time.monotonic_ns(), 0, 0), 'error': 'amdsmi unavailable: ' + why + '; sysfs hwmon ' + (hw[0] if hw else 'not found')}) + '\n')
        while hw:
            t = (time.monotonic_ns(), time.clock_gettime_ns(time.CLOCK_BOOTTIME), time.time_ns())
            p = rd('power1_average') or rd('power1_input')
            rec = {'t': t, 'current_gfxclks': [rd('freq1_input') / 1e6] if rd('freq1_input') else [], 'current_uclk': (rd('freq2_input') or 0) / 1e6,
                   'current_socket_power': p / 1e6 if p else None, 'temperature_hotspot': (rd('temp2_input') or 0) / 1e3}
            f.write(json.dumps(rec) + '\n')
            f.flush()
            if period > 0:
                time.sleep(period)


def sampler(path, period):
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
        amdsmi.amdsmi_get_gpu_metrics_info(h)
    except Exception as ex:      # noqa: BLE001
        return sysfs_sampler(path, period, f'{type(ex).__name__}: {ex}'[:200])
    keys = ('current_gfxclks', 'current_uclk', 'current_socclk', 'current_socket_power', 'average_socket_power', 'temperature_hotspot', 'temperature_mem',
            'temperature_hbm', 'throttle_status', 'indep_throttle_status', 'ppt_residency_acc', 'prochot_residency_acc', 'socket_thm_residency_acc',
            'hbm_thm_residency_acc', 'vr_thm_residency_acc', 'accumulation_counter', 'energy_accumulator', 'firmware_timestamp', 'average_gfx_activity',
            'average_umc_activity', 'gfxclk_lock_status', 'voltage_gfx')
    with open(path, 'w') as f:
        while True:
            t = (time.monotonic_ns(), time.clock_gettime_ns(time.CLOCK_BOOTTIME), time.time_ns())
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            except Exception as ex:      # noqa: BLE001
                f.write(json.dumps({'t': t, 'error': str(ex)[:200]}) + '\n')
                f.flush()
                time.sleep(0.5)
                continue
            rec = {'t': t, 't_after': time.monotonic_ns()}
            for k in keys:
                v = m.get(k)
                if isinstance(v, (list, tuple)):
                    v = [x for x in v if isinstance(x, (int, float)) and x not in (65535, 0xFFFFFFFF)]
                rec[k] = v
            f.write(json.dumps(rec) + '\n')
            f.flush()
            if period > 0:
                time.sleep(period)


def workload(args):
    import torch
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    os.makedirs(args.out, exist_ok=True)
    samp = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--sampler', os.path.join(args.out, 'samples.jsonl'), '--period', str(args.period)])
    try:
        E, U, B, L = args.envs, args.ues, args.bs, 100
        dev = torch.device('cuda', 0)
        scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
        m, bs, ues = build_from_scenario(scn)
        env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, episode_length=L, rng='philox', rand_episodes=True, device=dev)
        g = torch.Generator(device=dev).manual_seed(7)
        pool = torch.randint(0, B + 1, (16, E, U), generator=g, device=dev, dtype=torch.uint8)
        env.reset()
        torch.cuda.synchronize()
        time.sleep(args.idle)                       # the sampler sees the idle clocks first
        blocks = []
        BL = args.block

        def phase(name, launches, reset_every=L, fixed_action=False):
            """launches back to back (resets every `reset_every`), one event per BL launches"""
            evs = [torch.cuda.Event(enable_timing=True)]
            torch.cuda.synchronize()
            host0 = time.monotonic_ns()
            evs[0].record()
            t, meta = 0, []
            while t < launches:
                if reset_every and t % reset_every == 0:
                    env.reset()
                for i in range(BL):
                    env.step(pool[0 if fixed_action else (t + i) & 15])
                t += BL
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
                meta.append((t - BL, time.monotonic_ns()))
            torch.cuda.synchronize()
            host1 = time.monotonic_ns()
            for k in range(len(meta)):
                blocks.append({'phase': name, 'first_launch': meta[k][0], 'pos_in_episode': (meta[k][0] % reset_every) if reset_every else meta[k][0],
                               'us_per_launch': evs[k].elapsed_time(evs[k + 1]) * 1e3 / BL, 'gpu_ms_since_phase_start': evs[0].elapsed_time(evs[k + 1]),
                               'host_issue_ns': meta[k][1], 'phase_host0_ns': host0, 'phase_host1_ns': host1,
                               'has_reset': bool(reset_every and meta[k][0] % reset_every == 0)})
        phase('A_cold_after_idle', args.launches)
        time.sleep(args.idle)
        phase('B_after_%gs_idle' % args.idle, args.launches)
        phase('C_back_to_back_with_B', args.launches)
        phase('D_no_resets_long_episode', args.launches, reset_every=0)
        env.reset()
        phase('E_same_action_tensor', 1000, fixed_action=True)
        env.check()
        json.dump({'blocks': blocks, 'block': BL, 'envs': E, 'ues': U, 'bs': B, 'kernel': env.step_kernel_name}, open(os.path.join(args.out, 'blocks.json'), 'w'))
        time.sleep(0.5)
    finally:
        samp.kill()
        samp.wait()
    report(args.out, None)


def load_samples(out):
    rows = []
    for line in open(os.path.join(out, 'samples.jsonl')):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        if 'error' not in r:
            rows.append(r)
    return rows


def mean(xs):
    xs = list(xs)
    return sum(xs) / len(xs) if xs else float('nan')


def clk(r):
    v = r.get('current_gfxclks') or []
    return mean(v) if v else float('nan')


def report(out, ktrace):
    import csv
    import glob
    d = json.load(open(os.path.join(out, 'blocks.json')))
    S = load_samples(out)
    BL = d['block']
    print(f"# kernel {d['kernel']}  {d['envs']} envs x {d['ues']} UE x {d['bs']} BS; {BL} launches per event block; {len(S)} amdsmi samples"
          + (f", median sampling period {sorted((S[i + 1]['t'][0] - S[i]['t'][0]) for i in range(len(S) - 1))[len(S) // 2] / 1e6:.1f} ms" if len(S) > 2 else ''))
    if S:
        resid = [k for k in ('ppt_residency_acc', 'prochot_residency_acc', 'socket_thm_residency_acc', 'hbm_thm_residency_acc', 'vr_thm_residency_acc') if S[0].get(k) is not None]
        print('# throttle-residency accumulators, last - first sample:', {k: (S[-1][k] - S[0][k]) for k in resid},
              ' accumulation_counter:', (S[-1].get('accumulation_counter') or 0) - (S[0].get('accumulation_counter') or 0))
    phases = []
    for b in d['blocks']:
        if b['phase'] not in phases:
            phases.append(b['phase'])
    print('\n== per phase: launch duration against the clock / power samples of the same host interval ==')
    print(f"{'phase':34s} {'launches':>8s} {'mean us':>8s} {'min blk':>8s} {'max blk':>8s} | {'gfxclk MHz mean (min..max)':>28s} {'uclk':>6s} {'W mean (max)':>14s} {'hotspot C':>9s} {'us x GHz':>9s}")
    for ph in phases:
        bl = [b for b in d['blocks'] if b['phase'] == ph]
        t0, t1 = bl[0]['phase_host0_ns'], bl[0]['phase_host1_ns']
        ss = [r for r in S if t0 <= r['t'][0] <= t1]
        us = mean(b['us_per_launch'] for b in bl)
        ck = [clk(r) for r in ss if clk(r) == clk(r)]
        pw = [r['current_socket_power'] for r in ss if isinstance(r.get('current_socket_power'), (int, float))]
        print(f"{ph:34s} {len(bl) * BL:8d} {us:8.2f} {min(b['us_per_launch'] for b in bl):8.2f} {max(b['us_per_launch'] for b in bl):8.2f} | "
              f"{mean(ck):10.0f} ({min(ck) if ck else 0:.0f}..{max(ck) if ck else 0:.0f}) n={len(ck):<4d} {mean(r['current_uclk'] for r in ss if r.get('current_uclk')):6.0f} "
              f"{mean(pw):7.0f} ({max(pw) if pw else 0:.0f}) {mean(r['temperature_hotspot'] for r in ss if r.get('temperature_hotspot')):9.0f} {us * mean(ck) / 1e3:9.1f}")
    print('\n== idle samples (before phase A / between A and B): what the part idles at ==')
    first = d['blocks'][0]['phase_host0_ns']
    idle = [r for r in S if r['t'][0] < first]
    if idle:
        print(f"   before A: gfxclk {mean(clk(r) for r in idle):.0f} MHz, power {mean(r['current_socket_power'] for r in idle if r.get('current_socket_power') is not None):.0f} W, n={len(idle)}")
    print('\n== time series: consecutive windows of 100 launches (phase, first launch, us/launch, nearest samples) ==')
    print(f"{'phase':34s} {'launch':>7s} {'us/launch':>9s} {'gfxclk':>7s} {'uclk':>6s} {'W':>5s} {'C':>4s}")
    per_win = max(1, 100 // BL)
    for ph in phases:
        bl = [b for b in d['blocks'] if b['phase'] == ph]
        for w in range(0, len(bl), per_win):
            grp = bl[w:w + per_win]
            lo = grp[0]['host_issue_ns'] - 2_000_000
            hi = grp[-1]['host_issue_ns'] + 2_000_000
            ss = [r for r in S if lo <= r['t'][0] <= hi] or sorted(S, key=lambda r: abs(r['t'][0] - hi))[:1]
            if w // per_win < 12 or (w // per_win) % 5 == 0:
                print(f"{ph:34s} {grp[0]['first_launch']:7d} {mean(b['us_per_launch'] for b in grp):9.2f} {mean(clk(r) for r in ss):7.0f} "
                      f"{mean(r.get('current_uclk') or 0 for r in ss):6.0f} {mean(r.get('current_socket_power') or 0 for r in ss):5.0f} {mean(r.get('temperature_hotspot') or 0 for r in ss):4.0f}")
    print('\n== position inside the episode (launches since the last reset_kernel), steady phases B + C, mean over episodes ==')
    pos = {}
    for b in d['blocks']:
        if b['phase'].startswith(('B_', 'C_')):
            pos.setdefault(b['pos_in_episode'], []).append(b['us_per_launch'])
    print('   ' + '  '.join(f"{p}:{mean(v):.1f}" for p, v in sorted(pos.items())))
    longp = [b for b in d['blocks'] if b['phase'].startswith('D_')]
    if longp:
        print('== no resets (phase D): us/launch by launches since the one reset ==')
        step = max(1, len(longp) // 30)
        print('   ' + '  '.join(f"{b['first_launch']}:{b['us_per_launch']:.1f}" for b in longp[::step]))
    # rocprofv3 --kernel-trace of the SAME process: per-launch durations
    kt = ktrace or out
    files = glob.glob(os.path.join(kt, '**', '*kernel_trace.csv'), recursive=True)
    if files:
        rows = []
        with open(files[0]) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
        rows.sort()
        steps = [(s, e) for s, e, n in rows if 'step_kernel' in n]
        print(f"\n== rocprofv3 --kernel-trace of the same process: {len(steps)} step launches ==")
        durs = [(e - s) / 1e3 for s, e in steps]
        sd = sorted(durs)
        print(f"   min {sd[0]:.2f}  p10 {sd[len(sd) // 10]:.2f}  median {sd[len(sd) // 2]:.2f}  mean {mean(durs):.2f}  p90 {sd[len(sd) * 9 // 10]:.2f}  max {sd[-1]:.2f} us")
        # position after a reset_kernel from the trace itself
        since, byp = None, {}
        for s, e, n in rows:
            if 'reset_kernel' in n:
                since = 0
            elif 'step_kernel' in n and since is not None:
                byp.setdefault(since, []).append((e - s) / 1e3)
                since += 1
        if byp:
            ks = sorted(byp)
            print('   duration by launches since the last reset_kernel (mean over all resets, us):')
            print('   ' + '  '.join(f"{k}:{mean(byp[k]):.1f}(n={len(byp[k])})" for k in ks if k < 100 and (k < 12 or k % 10 == 0 or k > 96)))
            print(f"   launches 0-4 after a reset: {mean(x for k in ks if k < 5 for x in byp[k]):.2f} us;  launches 20-99: {mean(x for k in ks if 20 <= k < 100 for x in byp[k]):.2f} us"
                  f";  the {sum(1 for d in durs if d < 72.0)} launches under 72 us: {sum(1 for k in ks if k < 5 for x in byp[k] if x < 72.0)} of them are launches 0-4 after a reset")
        # gaps between consecutive step launches (device idle between kernels)
        gaps = [(steps[i + 1][0] - steps[i][1]) / 1e3 for i in range(len(steps) - 1)]
        sg = sorted(gaps)
        print(f"   gap between consecutive step launches: median {sg[len(sg) // 2]:.2f} us, p90 {sg[len(sg) * 9 // 10]:.2f} us")
        # which host clock the trace uses, then duration vs the gfx clock sampled nearest to each launch
        for ci, cname in enumerate(('CLOCK_MONOTONIC', 'CLOCK_BOOTTIME', 'CLOCK_REALTIME')):
            if S and S[0]['t'][ci] - 5e9 <= steps[0][0] <= S[-1]['t'][ci] + 5e9:
                import bisect
                ts = [r['t'][ci] for r in S]
                buckets = {}
                for (s, e), du in zip(steps, durs):
                    j = min(len(S) - 1, bisect.bisect_left(ts, s))
                    c = clk(S[j])
                    if c == c:
                        buckets.setdefault(int(c // 50) * 50, []).append(du)
                print(f"   trace timestamps are {cname}; launch duration by the gfx clock sampled nearest to the launch (50 MHz bins):")
                for c in sorted(buckets):
                    v = buckets[c]
                    print(f"     {c:5d}-{c + 49} MHz: n={len(v):5d}  mean {mean(v):7.2f} us  min {min(v):7.2f}  -> mean x clock = {mean(v) * (c + 25) / 1e3:7.1f} kcycles")
                break


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--sampler')
    ap.add_argument('--period', type=float, default=0.002)
    ap.add_argument('--report')
    ap.add_argument('--kernel-trace')
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'clock_trace'))
    ap.add_argument('--launches', type=int, default=3000)
    ap.add_argument('--block', type=int, default=10)
    ap.add_argument('--idle', type=float, default=2.0)
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--ues', type=int, default=32)
    ap.add_argument('--bs', type=int, default=10)
    a = ap.parse_args()
    if a.sampler:
        sampler(a.sampler, a.period)
    elif a.report:
        report(a.report, a.kernel_trace)
    else:
        workload(a)
