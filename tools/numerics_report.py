#!/usr/bin/env python3
"""Measured FP32-vs-FP64 error of the device path against the oracle (GPU box): max relative error of per-UE data rates
and relative-SNR observations, max absolute error of utilities / rewards, over E envs x T steps of random actions."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv
from oracle import oracle as orc

E, U, B, T = 1024, 32, 10, 60
if len(sys.argv) > 2:                       # tools/numerics_report.py <envs> <steps>: a larger sample for the tail
    E, T = int(sys.argv[1]), int(sys.argv[2])
scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=2, num_slow=22, num_fast=8)
m, bs, ues = build_from_scenario(scn)
core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=7, rng='philox')
oenvs = []
for e in range(E):
    o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, [s['velocity'] for s in scn.ue_specs], kind=orc.MULTI)
    o.set_philox(7, e)
    oenvs.append(o)
ob = orc.OracleBatch(oenvs)
rng = np.random.default_rng(1)
core.reset()
ob.reset()
err = dict(rate_rel=0.0, obs_dr_rel=0.0, util_abs=0.0, reward_abs=0.0, ewma_rel=0.0, n_rates=0)
for t in range(T):
    a = rng.integers(0, B + 1, size=(E, U)).astype(np.uint8)
    a[rng.random((E, U)) < 0.5] = 0
    core.step(torch.from_numpy(a).cuda())
    o_obs, o_rew, o_conn, o_pos = ob.step(a)
    assert np.array_equal(core.state_host()['conn'], o_conn) and np.array_equal(core.state_host()['pos'], o_pos)
    st = [o.state() for o in oenvs]
    want_dr = np.stack([s['curr_dr'] for s in st])
    want_ut = np.stack([s['utility'] for s in st])
    want_ew = np.stack([s['ewma'] for s in st])
    got_dr = core.ue_dr.cpu().numpy().astype(np.float64)
    nz = want_dr > 0
    err['rate_rel'] = max(err['rate_rel'], float(np.max(np.abs(got_dr[nz] - want_dr[nz]) / want_dr[nz])))
    err['n_rates'] += int(nz.sum())
    ew = core.ewma.cpu().numpy().reshape(E, U).astype(np.float64)
    nz2 = want_ew > 1e-30          # below that the f32 state word is denormal (a rate that decayed x0.1 for 30+ steps)
    err['ewma_rel'] = max(err['ewma_rel'], float(np.max(np.abs(ew[nz2] - want_ew[nz2]) / want_ew[nz2])))
    err['util_abs'] = max(err['util_abs'], float(np.max(np.abs(core.ue_utility.cpu().numpy() - want_ut))))
    err['reward_abs'] = max(err['reward_abs'], float(np.max(np.abs(core.reward.cpu().numpy() - o_rew))))
    od = core.obs_views()['dr'].cpu().numpy().astype(np.float64)
    wd = o_obs[:, :, B:2 * B].astype(np.float64)
    big = wd > 1e-6
    err['obs_dr_rel'] = max(err['obs_dr_rel'], float(np.max(np.abs(od[big] - wd[big]) / wd[big])))
print(f'{E} envs x {U} UE x {B} BS, {T} steps, {err["n_rates"]} non-zero per-UE rates compared; masks and FP64 positions bit-exact')
print(f'max relative error  data rate {err["rate_rel"]:.2e} | EWMA rate (> 1e-30) {err["ewma_rel"]:.2e} | obs dr (relative SNR, vs f32-rounded oracle) {err["obs_dr_rel"]:.2e}')
print(f'max absolute error  utility [-20,20] {err["util_abs"]:.2e} | multi-agent reward [-20,20] {err["reward_abs"]:.2e}')
