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


# obs['dr'] (relative SNR, variants.py:276-284) by MAGNITUDE, against the oracle's FP64 value: the parity bar (tests/parity.py) is
# relative down to float32's smallest normal number, so the small entries are measured here too -- 32 stations (far stations at
# 1e-13) and UEs parked on / next to stations (the others' entries fall to 1e-10 ... 1e-50).
E2, U2, B2, T2 = 256, 32, 32, 40
OFFS = (0.0, 1e-9, 1e-6, 1e-4, 1e-2, 0.3)
scn = scenarios.grid_map(B2, 'mixed').with_ues(num_static=6, num_slow=20, num_fast=6)
for i, off in enumerate(OFFS):                                   # static UE i stands `off` metres from station i
    x, y = scn.bs_pos[i]
    scn.ue_specs[i]['pos_x'], scn.ue_specs[i]['pos_y'] = int(x), int(y)
    scn.bs_pos[i] = (x + off, y)
m, bs, ues = build_from_scenario(scn)
core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E2, seed=9, rng='philox')
init_xy = [(s['pos_x'], s['pos_y']) if s['pos_x'] != 'random' else (-1, -1) for s in scn.ue_specs]
oenvs = []
for e in range(E2):
    o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, [s['velocity'] for s in scn.ue_specs], kind=orc.MULTI, init_xy=init_xy)
    o.set_philox(9, e)
    oenvs.append(o)
ob = orc.OracleBatch(oenvs)
core.reset()
ob.reset()
tiny = float(np.finfo(np.float32).tiny)
dec = {}
below = [0, 0.0]
for t in range(T2):
    a = rng.integers(0, B2 + 1, size=(E2, U2)).astype(np.uint8)
    core.step(torch.from_numpy(a).cuda())
    o_obs, o_rew, o_conn, o_pos = ob.step(a)
    assert np.array_equal(core.state_host()['conn'], o_conn) and np.array_equal(core.state_host()['pos'], o_pos)
    want = ob.rates(want_dr_rel=True)['dr_rel']
    got = core.obs_views()['dr'].cpu().numpy().astype(np.float64).reshape(want.shape)
    nrm = want >= tiny
    rel = np.abs(got[nrm] - want[nrm]) / want[nrm]
    d = np.floor(np.log10(want[nrm])).astype(int)
    for k in np.unique(d):
        r = rel[d == k]
        c = dec.setdefault(int(k), [0, 0.0])
        c[0] += r.size
        c[1] = max(c[1], float(r.max()))
    below[0] += int((~nrm).sum())
    below[1] = max(below[1], float(np.abs(got[~nrm]).max()) if (~nrm).any() else 0.0)
print(f'\nobs dr by magnitude, {E2} envs x {U2} UE x {B2} BS, {T2} steps; static UEs {OFFS} m from a station; FP64 oracle values:')
for k in sorted(dec, reverse=True):
    print(f'   [1e{k:+03d}, 1e{k + 1:+03d}): n = {dec[k][0]:9d}   max relative error {dec[k][1]:.2e}')
print(f'   below 2^-126: n = {below[0]}, largest device value {below[1]:.3e} (must be flushed / denormal)')
