"""GPU tests of the RLlib-protocol adapters against the REFERENCE-RUN fixtures -- not against
other instances of the same HIP path: tests/golden/estack_* hold 8 reference runs (seeds 42 + 20000 e) of the 10 x 5 central
and the 32 x 10 multi-agent env under one action tape each.  Caller contract: deepcomp/util/env_setup.py:262-316 (observation /
action spaces, agent ids = ue.id, horizon = episode_length), deepcomp/util/simulation.py:143."""
import os

import numpy as np
import pytest

from tests.test_parity_gpu import ATOL_OBS, ATOL_UTIL, GOLDEN, RTOL_RATE, _entities_from_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _load(stack):
    return [np.load(os.path.join(GOLDEN, f'{stack}_e{e}.npz')) for e in range(8)]


def _env_config(g, num_envs):
    m, bs, ues = _entities_from_fixture(g)
    return {'map': m, 'bs_list': bs, 'ue_list': ues, 'seed': int(g['cfg_seed']), 'episode_length': int(g['cfg_eps_len']),
            'reward': {0: 'avg', 1: 'sum', 2: 'min'}[int(g['cfg_reward'])], 'rand_episodes': bool(g['cfg_rand_episodes']),
            'num_envs': num_envs, 'rng': 'reference'}


def _check_central(obs, g, prefix, i, U, B):
    assert obs['connected'].shape == (U * B,) and obs['dr'].shape == (U * B,) and obs['utility'].shape == (U,)
    assert np.array_equal(obs['connected'], g[f'{prefix}_obs_connected'][i].ravel())          # central.py:36-44: UE-major
    np.testing.assert_allclose(obs['dr'], g[f'{prefix}_obs_dr'][i].ravel(), rtol=RTOL_RATE, atol=1e-30)
    np.testing.assert_allclose(obs['utility'], g[f'{prefix}_obs_utility'][i], atol=ATOL_OBS, rtol=0)


def test_central_vector_env_against_reference_runs(torch_cuda):
    torch = torch_cuda
    from deepcomp_amd.rllib_adapter import CENTRAL_KEYS, CentralVectorEnv, flatten_obs
    gs = _load('estack_grid10x5_central')
    U, B = 10, 5
    vec = CentralVectorEnv(_env_config(gs[0], 8))
    assert vec.num_envs == 8 and tuple(sorted(vec.observation_space.spaces)) == CENTRAL_KEYS
    assert list(vec.observation_space.spaces) == list(CENTRAL_KEYS)                      # gym's sorted-key order = RLlib's flatten order
    assert list(vec.action_space.nvec) == [B + 1] * U                                    # central.py:28
    obs = vec.vector_reset()
    for e in range(8):
        _check_central(obs[e], gs[e], 'reset', 0, U, B)
        assert np.array_equal(flatten_obs(obs[e]), vec.poll_tensors()[0][e].cpu().numpy())   # packed row == RLlib's flattening
    T = gs[0]['actions'].shape[0]
    for t in range(T):
        obs, rew, dones, infos = vec.vector_step([g['actions'][t].tolist() for g in gs])
        assert dones == [False] * 8
        for e in range(8):
            _check_central(obs[e], gs[e], 'step', t, U, B)
            assert rew[e] == pytest.approx(float(gs[e]['step_reward'][t, 0]), abs=ATOL_OBS)
            assert infos[e]['time'] == int(gs[e]['step_time'][t])
            assert infos[e]['scalar_metrics']['sum_utility'] == pytest.approx(float(gs[e]['step_sum_utility'][t]), abs=ATOL_UTIL * U)
            vm = infos[e]['vector_metrics']                                             # base.py:383-411: per-UE dr / utility by 'UE <id>'
            assert list(vm['dr']) == [f'UE {u + 1}' for u in range(U)] == list(vm['utility'])
            np.testing.assert_allclose(list(vm['dr'].values()), gs[e]['step_curr_dr'][t], rtol=RTOL_RATE, atol=1e-30)
            np.testing.assert_allclose(list(vm['utility'].values()), gs[e]['step_utility'][t], rtol=0, atol=ATOL_UTIL)
            assert np.array_equal(flatten_obs(obs[e]), vec.poll_tensors()[0][e].cpu().numpy())
    # the horizon: RLlib resets the copies one by one, in its own order; every one must get ITS first observation
    order = [5, 2, 7, 0, 1, 3, 4, 6]
    time_before = vec.core.time
    for k, e in enumerate(order):
        _check_central(vec.reset_at(e), gs[e], 'reset', 0, U, B)
        assert vec.core.time == 0 and (k > 0 or time_before > 0)
    episodes = vec.core.episode
    _check_central(vec.reset_at(3), gs[3], 'reset', 0, U, B)              # a second request for a served index: a new reset
    assert vec.core.episode == episodes + 1
    obs, rew, _, _ = vec.vector_step([g['actions'][0].tolist() for g in gs])
    for e in range(8):
        _check_central(obs[e], gs[e], 'step', 0, U, B)                    # fixed episodes (base.py:171-173): the same episode again
    obs = [{k: v.copy() for k, v in o.items()} for o in obs]              # (the dicts are views of ONE host buffer the next step refills: keep a copy)
    # a request for the whole batch always resets (two in a row: two episodes), and so does a per-index request right after it
    ep = vec.core.episode
    vec.vector_reset(); vec.vector_reset()
    assert vec.core.episode == ep + 2
    _check_central(vec.reset_at(4), gs[4], 'reset', 0, U, B)
    assert vec.core.episode == ep + 3
    _check_central(vec.reset_at(1), gs[1], 'reset', 0, U, B)              # ... which then serves the other indices
    assert vec.core.episode == ep + 3
    # zero-copy path == protocol path
    vec.vector_step([g['actions'][0].tolist() for g in gs])
    vec.reset_at(0)
    a = torch.from_numpy(np.stack([g['actions'][0] for g in gs]).astype(np.uint8)).cuda()
    vec.send_action_tensor(a)
    o, r = vec.poll_tensors()
    for e in range(8):
        assert np.array_equal(o[e].cpu().numpy(), flatten_obs(obs[e])) and float(r[e]) == pytest.approx(rew[e], abs=1e-7)


def _check_agent(obs, g, prefix, i, u, B):
    assert set(obs) == {'connected', 'dr', 'utility', 'ues_at_bs', 'util_at_bs'}                # variants.py:302-303
    assert np.array_equal(obs['connected'], g[f'{prefix}_obs_connected'][i][u])
    np.testing.assert_allclose(obs['dr'], g[f'{prefix}_obs_dr'][i][u], rtol=RTOL_RATE, atol=1e-30)
    np.testing.assert_allclose(obs['utility'], [g[f'{prefix}_obs_utility'][i][u]], atol=ATOL_OBS, rtol=0)
    np.testing.assert_allclose(obs['ues_at_bs'], g[f'{prefix}_obs_ues_at_bs'][i][u], atol=1e-6, rtol=0)
    np.testing.assert_allclose(obs['util_at_bs'], g[f'{prefix}_obs_util_at_bs'][i][u], atol=ATOL_OBS, rtol=0)


def test_multi_agent_base_env_against_reference_runs(torch_cuda):
    torch = torch_cuda
    from deepcomp_amd.rllib_adapter import MULTI_KEYS, MultiAgentBaseEnv, flatten_obs
    gs = _load('estack_grid32x10_multi')
    U, B = 32, 10
    base = MultiAgentBaseEnv(_env_config(gs[0], 8))
    assert base.agent_ids == [str(i + 1) for i in range(U)]                             # ue.id strings (env_setup.py:304-309)
    assert list(base.observation_space.spaces) == list(MULTI_KEYS) and base.action_space.n == B + 1
    obs, rew, dones, infos, off = base.poll()
    assert sorted(obs) == list(range(8)) and off == {}
    for e in range(8):
        assert list(obs[e]) == base.agent_ids and dones[e] == {'__all__': False}
        for u in (0, 7, 31):
            _check_agent(obs[e][str(u + 1)], gs[e], 'reset', 0, u, B)
            assert np.array_equal(flatten_obs(obs[e][str(u + 1)]), base.poll_tensors()[0][e, u].cpu().numpy())
    T = min(12, gs[0]['actions'].shape[0])
    for t in range(T):
        acts = {e: {str(u + 1): int(gs[e]['actions'][t][u]) for u in range(U) if gs[e]['actions'][t][u] or u % 2}
                for e in range(8)}                                                      # some no-op agents left out (multi_agent.py:30)
        base.send_actions(acts)
        obs, rew, dones, infos, _ = base.poll()
        for e in range(8):
            for u in range(U):
                _check_agent(obs[e][str(u + 1)], gs[e], 'step', t, u, B)
                assert rew[e][str(u + 1)] == pytest.approx(float(gs[e]['step_reward'][t][u]), abs=ATOL_UTIL)
            assert infos[e]['1']['time'] == t + 1 and dones[e] == {'__all__': False}
    for e in (6, 0, 3, 7, 1, 2, 5, 4):                                                  # try_reset in any order
        o = base.try_reset(e)
        for u in (0, 13, 31):
            _check_agent(o[str(u + 1)], gs[e], 'reset', 0, u, B)
    assert base.core.time == 0
    # zero-copy path: one more step through tensors == the fixtures' first step
    a = torch.from_numpy(np.stack([g['actions'][0] for g in gs]).astype(np.uint8)).cuda()
    base.send_action_tensor(a)
    o, r = base.poll_tensors()
    base.core.check()
    oh = o.cpu().numpy()
    for e in range(8):
        assert np.array_equal(oh[e, :, :B], gs[e]['step_obs_connected'][0])
        np.testing.assert_allclose(r[e].cpu().numpy(), gs[e]['step_reward'][0], atol=ATOL_UTIL, rtol=0)


def test_protocol_adapters_build_their_views_once(torch_cuda):
    """the protocol methods rebuilt E (x U) observation dicts every step.  Now the dicts are views over ONE pinned
    host buffer that a step refills in place: the same objects come back every step (nothing allocated per env for observations), their
    CONTENT is the new step's; env_config['persistent_views'] = False hands out fresh arrays; the fixtures hold both."""
    from deepcomp_amd.rllib_adapter import CentralVectorEnv, MultiAgentBaseEnv
    gs = _load('estack_grid10x5_central')
    U, B = 10, 5
    vec = CentralVectorEnv(_env_config(gs[0], 8))
    fresh = CentralVectorEnv(dict(_env_config(gs[0], 8), persistent_views=False, info_level='scalar'))
    r0, f0 = vec.vector_reset(), fresh.vector_reset()
    keep = [o['dr'].copy() for o in f0]
    o0 = None
    for t in range(5):
        acts = [g['actions'][t].tolist() for g in gs]
        o1, rew, dones, infos = vec.vector_step(acts)
        f1, frew, _, finfos = fresh.vector_step(acts)
        o0 = o1 if o0 is None else o0
        assert o1 is o0 and all(a is b for a, b in zip(o1, o0)) and o1[3]['dr'].base is not None       # the same list of the same dicts of views
        assert o1 is not r0                                                   # (a reset's observations live in a buffer of their own; round 6)
        assert f1 is not f0 and rew == frew
        assert 'vector_metrics' in infos[0] and 'vector_metrics' not in finfos[0] and finfos[2]['scalar_metrics'] == infos[2]['scalar_metrics']
        for e in range(8):
            _check_central(o1[e], gs[e], 'step', t, U, B)
            _check_central(f1[e], gs[e], 'step', t, U, B)
    assert all(np.array_equal(k, o['dr']) for k, o in zip(keep, f0))                 # fresh arrays: the reset observation is still intact
    with pytest.raises(ValueError):
        CentralVectorEnv(dict(_env_config(gs[0], 2), info_level='everything'))
    gm = _load('estack_grid32x10_multi')
    U, B = 32, 10
    base = MultiAgentBaseEnv(_env_config(gm[0], 8))
    reset_obs = base.poll()[0]
    obs0 = None
    for t in range(4):
        base.send_actions({e: {str(u + 1): int(gm[e]['actions'][t][u]) for u in range(U)} for e in range(8)})
        obs, rew, dones, infos, _ = base.poll()
        obs0 = obs if obs0 is None else obs0
        assert obs is obs0 and obs[5]['7'] is obs0[5]['7'] and obs is not reset_obs
        assert infos[0]['1'] == {'time': t + 1} and infos[7]['32']['time'] == t + 1 and dones[4] == {'__all__': False}
        for e in (0, 3, 7):
            for u in (0, 9, 31):
                _check_agent(obs[e][str(u + 1)], gm[e], 'step', t, u, B)
                assert rew[e][str(u + 1)] == pytest.approx(float(gm[e]['step_reward'][t][u]), abs=ATOL_UTIL)


def test_step_views_survive_a_reset_requested_env_by_env(torch_cuda):
    """RLlib's sampler calls try_reset(env_id) / reset_at(i) inside its per-env loop at the horizon, BEFORE it has
    preprocessed the other envs' last-step observations; the first such request resets the whole batch.  With persistent views the reset
    used to refill the buffer those observations are views of.  Now resets have buffers of their own: after reset_at(0) / try_reset(0) the
    step observations of envs 1 .. 7 that the caller still holds are what the step wrote -- dict mode and flat_obs mode."""
    from deepcomp_amd.rllib_adapter import CentralVectorEnv, MultiAgentBaseEnv
    gs = _load('estack_grid10x5_central')
    for flat in (False, True):
        vec = CentralVectorEnv(dict(_env_config(gs[0], 8), flat_obs=flat))
        vec.vector_reset()
        for t in range(3):
            obs, _, _, _ = vec.vector_step([g['actions'][t].tolist() for g in gs])
        want = [o.copy() if flat else {k: v.copy() for k, v in o.items()} for o in obs]
        first = vec.reset_at(0)                          # resets the whole batch
        for e in range(1, 8):                            # the step views the sampler has not consumed yet
            if flat:
                assert np.array_equal(obs[e], want[e])
            else:
                assert all(np.array_equal(obs[e][k], want[e][k]) for k in want[e])
        r3 = vec.reset_at(3)
        assert not np.array_equal(np.asarray(r3 if flat else r3['dr']), np.asarray(want[3] if flat else want[3]['dr']))   # a reset observation, not the step's
        keep_first = first.copy() if flat else {k: v.copy() for k, v in first.items()}
        vec.vector_step([g['actions'][0].tolist() for g in gs])
        assert np.array_equal(first, keep_first) if flat else all(np.array_equal(first[k], keep_first[k]) for k in keep_first)   # ... and a step leaves the reset's alone
    gm = _load('estack_grid32x10_multi')
    U = 32
    for flat in (False, True):
        base = MultiAgentBaseEnv(dict(_env_config(gm[0], 8), flat_obs=flat))
        base.poll()
        for t in range(3):
            base.send_actions({e: {str(u + 1): int(gm[e]['actions'][t][u]) for u in range(U)} for e in range(8)})
            obs = base.poll()[0]
        snap = lambda o: o.copy() if flat else {k: v.copy() for k, v in o.items()}      # noqa: E731
        same = lambda a, b: np.array_equal(a, b) if flat else all(np.array_equal(a[k], b[k]) for k in b)      # noqa: E731
        want = {e: {a: snap(o) for a, o in obs[e].items()} for e in obs}
        base.try_reset(0)
        for e in range(1, 8):
            assert all(same(obs[e][a], want[e][a]) for a in want[e]), f'env {e}: the last step\'s observations changed under a reset'


def test_flat_obs_rows_are_the_flattening_of_the_reference_dicts(torch_cuda):
    """env_config['flat_obs'] (round 6): observation_space is the flattened Box, the protocol methods hand out rows of ONE
    pinned array.  The rows must be what RLlib's Dict-flattening preprocessor makes of the REFERENCE's observation dicts (sorted keys:
    connected, dr, [ues_at_bs, util_at_bs,] utility): held to the reference-run estack fixtures piece by piece, and bit-identical to
    flatten_obs() of the dict-mode adapter stepping the same batch."""
    from deepcomp_amd.rllib_adapter import CentralVectorEnv, MultiAgentBaseEnv, flatten_obs
    gs = _load('estack_grid10x5_central')
    U, B = 10, 5
    flat = CentralVectorEnv(dict(_env_config(gs[0], 8), flat_obs=True))
    dic = CentralVectorEnv(_env_config(gs[0], 8))
    assert flat.observation_space.shape == (U * (2 * B + 1),) and flat.observation_space.low == -1 and flat.observation_space.high == 1
    fo, do = flat.vector_reset(), dic.vector_reset()
    assert fo[0].base is not None and fo[0].shape == (U * (2 * B + 1),)
    for e in range(8):
        assert np.array_equal(fo[e], flatten_obs(do[e])) and flat.observation_space.contains(fo[e])
        _check_central({'connected': fo[e][:U * B], 'dr': fo[e][U * B:2 * U * B], 'utility': fo[e][2 * U * B:]}, gs[e], 'reset', 0, U, B)
    for t in range(gs[0]['actions'].shape[0]):
        acts = [g['actions'][t].tolist() for g in gs]
        fo, frew, _, _ = flat.vector_step(acts)
        do, drew, _, _ = dic.vector_step(acts)
        assert frew == drew
        for e in range(8):
            assert np.array_equal(fo[e], flatten_obs(do[e])) and flat.observation_space.contains(fo[e])
            _check_central({'connected': fo[e][:U * B], 'dr': fo[e][U * B:2 * U * B], 'utility': fo[e][2 * U * B:]}, gs[e], 'step', t, U, B)
    assert fo[2].base is flat.vector_step([g['actions'][0].tolist() for g in gs])[0][2].base        # rows of ONE persistent array
    gm = _load('estack_grid32x10_multi')
    U, B = 32, 10
    flat = MultiAgentBaseEnv(dict(_env_config(gm[0], 8), flat_obs=True))
    dic = MultiAgentBaseEnv(_env_config(gm[0], 8))
    assert flat.observation_space.shape == (4 * B + 1,)
    fo, do = flat.poll()[0], dic.poll()[0]

    def as_dict(row):
        return {'connected': row[:B], 'dr': row[B:2 * B], 'ues_at_bs': row[2 * B:3 * B], 'util_at_bs': row[3 * B:4 * B], 'utility': row[4 * B:]}
    for e in range(8):
        for u in range(U):
            a = str(u + 1)
            assert np.array_equal(fo[e][a], flatten_obs(do[e][a]))
            _check_agent(as_dict(fo[e][a]), gm[e], 'reset', 0, u, B)
    for t in range(10):
        acts = {e: {str(u + 1): int(gm[e]['actions'][t][u]) for u in range(U)} for e in range(8)}
        flat.send_actions(acts); dic.send_actions(acts)
        fo, frew = flat.poll()[:2]
        do, drew = dic.poll()[:2]
        assert frew == drew
        for e in range(8):
            assert list(fo[e]) == flat.agent_ids
            for u in range(U):
                a = str(u + 1)
                assert np.array_equal(fo[e][a], flatten_obs(do[e][a])) and fo[e][a].shape == (4 * B + 1,)
                _check_agent(as_dict(fo[e][a]), gm[e], 'step', t, u, B)


# ------------------------------------------------------------------------------------ heuristic policy kernel (f4)
def _policy_env(kind, U, B, E, **kw):
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=U // 8, num_slow=U - U // 8 - U // 4, num_fast=U // 4)
    m, bs, ues = build_from_scenario(scn)
    return BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=5, rng='philox', **kw), bs


def _spec_views(env, obs):
    """[E, U, B] views of either layout for the tensor-expression form of the rules (agents.py __call__)."""
    if env.kind == 1:
        return {'connected': obs[..., :env.B], 'dr': obs[..., env.B:2 * env.B]}
    E, U, B = env.E, env.U, env.B
    return {'connected': obs[:, :U * B].reshape(E, U, B), 'dr': obs[:, U * B:2 * U * B].reshape(E, U, B)}


@pytest.mark.parametrize('kind,U,B,E', [('multi', 32, 10, 300), ('central', 10, 5, 257), ('multi', 7, 3, 50), ('central', 33, 16, 21),
                                        ('multi', 128, 32, 5), ('central', 5, 32, 9), ('multi', 1, 1, 70)])
def test_policy_kernel_equals_the_rules(kind, U, B, E):
    """dcomp_heuristic_actions against the tensor-expression form of the reference's rules (itself held against the
    reference-recorded decisions in test_agents_cpu.py): identical actions on live observations AND on synthetic ones
    full of exact ties / empty connection sets / everything connected."""
    import torch
    from deepcomp_amd import agents
    env, bs = _policy_env(kind, U, B, E)
    env.reset()
    ags = [agents.Heuristic3GPP(), agents.FullCoMP(), agents.DynamicSelection(0.3), agents.DynamicSelection(1.0),
           agents.DynamicSelection(0.0)]
    if B >= 3:
        ags.append(agents.StaticClustering(3, bs, seed=5, device='cuda'))
    g = torch.Generator(device='cuda').manual_seed(1)
    for t in range(12):
        for ag in ags:
            assert torch.equal(ag.act(env), ag(_spec_views(env, env.obs))), (type(ag).__name__, t)
        env.step(ags[t % len(ags)].act(env))                     # sticky policies: many simultaneous connections
    env.check()
    syn = torch.zeros_like(env.obs)
    v = _spec_views(env, syn)
    v['dr'].copy_((torch.randint(0, 4, v['dr'].shape, generator=g, device='cuda') / 3.0))      # quantised: ties everywhere
    v['connected'].copy_((torch.rand(v['connected'].shape, generator=g, device='cuda') < 0.4).float())
    v['connected'][0] = 1.0
    v['connected'][-1] = 0.0
    v['dr'][E // 2] = 0.0
    shifted = torch.zeros(syn.numel() + 1, device='cuda')[1:].view_as(syn)       # 4- but not 16-byte aligned: scalar copy path
    shifted.copy_(syn)
    assert shifted.data_ptr() % 16 != 0
    for ag in ags:
        want = ag(_spec_views(env, syn))
        assert torch.equal(env.heuristic_actions(*_policy_args(ag), obs=syn), want), type(ag).__name__
        assert torch.equal(env.heuristic_actions(*_policy_args(ag), obs=shifted), want), type(ag).__name__


def _policy_args(ag):
    from deepcomp_amd import agents
    if isinstance(ag, agents.Heuristic3GPP):
        return ('3gpp',)
    if isinstance(ag, agents.FullCoMP):
        return ('fullcomp',)
    if isinstance(ag, agents.DynamicSelection):
        return ('dynamic', ag.epsilon)
    return ('cluster', 0.0, ag._bits)


def test_policy_kernel_on_reference_recorded_decisions():
    """The kernel on the observations of tests/golden/heuristics.npz (recorded from the reference's own agents,
    gen_golden.py::gen_heuristics): same decisions as the reference took."""
    import os
    import torch
    from deepcomp_amd import agents
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'heuristics.npz'))
    dr, conn = G['obs_dr'], G['obs_connected']                                   # [T, U, B]
    T, U, B = dr.shape
    env, bs = _policy_env('multi', U, B, T)
    obs = torch.zeros_like(env.obs)
    obs[..., :B] = torch.from_numpy(conn).float().cuda()
    obs[..., B:2 * B] = torch.from_numpy(dr).float().cuda()
    for name, args in [('3gpp', ('3gpp',)), ('fullcomp', ('fullcomp',)), ('dynamic05', ('dynamic', 0.5)), ('dynamic09', ('dynamic', 0.9))]:
        got = env.heuristic_actions(*args, obs=obs).cpu().numpy()
        assert np.array_equal(got, G['act_' + name]), name
    ag = agents.StaticClustering(3, bs, seed=1, device='cuda')
    ag.act(env)
    assert np.array_equal(env.heuristic_actions('cluster', 0.0, ag._bits, obs=obs).cpu().numpy(), G['act_static3'])


def test_policy_kernel_argument_validation():
    import torch
    env, _ = _policy_env('multi', 4, 3, 8)
    env.reset()
    with pytest.raises(ValueError):
        env.heuristic_actions('greedy')
    with pytest.raises(ValueError):
        env.heuristic_actions('cluster')
    with pytest.raises(ValueError):
        env.heuristic_actions('3gpp', obs=env.obs.double())
    with pytest.raises(ValueError):
        env.heuristic_actions('3gpp', out=torch.zeros((8, 3), dtype=torch.uint8, device='cuda'))
    with pytest.raises(ValueError):
        env.heuristic_actions('dynamic', epsilon=1.5)


@pytest.mark.parametrize('kind', ['central', 'multi'])
def test_policy_kernel_with_ue_arrival_and_departure(kind):
    """Dynamic UE lists (base.py:433-443): the listed UEs are the first num_ue slots, the rest of the observation is zero
    padding (central.py:46-55) -> action 0 there, the rules on the listed slots; the actions drive the env without an
    assertion (bad action / UE outside the map) while UEs come and go."""
    import torch
    from deepcomp_amd import agents
    E, U0, B, CAP = 37, 4, 5, 9
    env, bs = _policy_env(kind, U0, B, E, ue_arrival={2: 2, 4: -1, 6: 3, 9: -2}, max_ues=CAP, episode_length=12)
    env.reset()
    ags = [agents.Heuristic3GPP(), agents.FullCoMP(), agents.DynamicSelection(0.4), agents.StaticClustering(2, bs, seed=3, device='cuda')]
    seen = set()
    for t in range(11):
        n = env.num_ue
        seen.add(n)
        for ag in ags:
            got = ag.act(env)
            want = ag(_spec_views(env, env.obs)).clone()
            want[:, n:] = 0
            assert got.shape == (E, CAP) and torch.equal(got, want), (type(ag).__name__, t, n)
        env.step(ags[t % len(ags)].act(env))
    env.check()
    assert len(seen) >= 4 and max(seen) > U0


@pytest.mark.parametrize('kind,U,B,E,kw', [('multi', 32, 10, 300, {}), ('central', 10, 5, 257, {}), ('multi', 7, 3, 50, {}),
                                           ('central', 33, 16, 21, {}), ('multi', 5, 6, 33000, {}),       # 33 000 x 5: tight packing
                                           ('central', 5, 32, 9, {}), ('multi', 100, 12, 9, {}),
                                           ('multi', 128, 32, 5, {}), ('central', 70, 24, 7, {}), ('multi', 200, 21, 3, {}),   # wide kernel
                                           ('multi', 4, 5, 37, dict(ue_arrival={2: 2, 4: -1, 6: 3, 9: -2}, max_ues=9)),
                                           # the generic kernel (round 6: it carries the rules too): a row over the whole wavefront (> 32 stations),
                                           # several rows per trip (<= 32 stations, > 256 UE slots), envs wider than a wavefront, UE arrival / departure
                                           ('multi', 32, 64, 40, {}), ('central', 10, 40, 30, {}), ('multi', 70, 33, 4, {}), ('central', 12, 50, 9, {}),
                                           ('multi', 300, 12, 3, {}), ('central', 260, 7, 2, {}), ('multi', 600, 20, 2, {}), ('multi', 257, 32, 2, {}),
                                           ('multi', 6, 40, 17, dict(ue_arrival={2: 3, 4: -2, 6: 3, 9: -4}, max_ues=12)),
                                           ('central', 5, 36, 11, dict(ue_arrival={1: 2, 5: -1, 8: 2}, max_ues=9))])
def test_in_step_policy_equals_the_policy_kernel(kind, U, B, E, kw):
    """dcomp_set_policy: the step / reset / rollout launches write next_action = dcomp_heuristic_actions(obs they wrote),
    for every policy, over a closed loop driven by those very actions (incl. resets, a fused rollout fragment, UE arrival)."""
    import torch
    from deepcomp_amd import agents
    env, bs = _policy_env(kind, U, B, E, episode_length=12, **kw)
    assert env.lanes_per_env == 5 or E != 33000
    ags = [agents.Heuristic3GPP(), agents.FullCoMP(), agents.DynamicSelection(0.3)]
    if B >= 3:
        ags.append(agents.StaticClustering(2, bs, seed=5, device='cuda'))
    for ag in ags:
        env.reset()
        first = ag.act(env)                                     # stand-alone kernel; registers the policy with the env
        assert env._policy_key is not None and env.next_action is not None
        env.step(first)
        for t in range(1, 11):
            act = ag.act(env)
            assert act.data_ptr() == env.next_action.data_ptr()                       # no launch: the step wrote it
            want = env.heuristic_actions(*_policy_args(ag))
            assert torch.equal(act, want), (type(ag).__name__, t)
            env.step(act)
            assert torch.equal(act, want)                       # the tensor handed to step() is not the one it writes
        env.reset()
        assert torch.equal(ag.act(env), env.heuristic_actions(*_policy_args(ag)))     # the reset kernel writes it too
        if not kw:
            tape = torch.stack([env.next_action.clone()] * 3)
            env.rollout(tape)
            assert torch.equal(ag.act(env), env.heuristic_actions(*_policy_args(ag)))
        env.check()
    env.set_policy(None)
    assert env.next_action is None


@pytest.mark.parametrize('kind,U0,B,E,rng', [('multi', 4, 5, 37, 'philox'), ('central', 6, 12, 300, 'philox'), ('multi', 5, 40, 21, 'philox'),
                                             ('central', 4, 6, 5, 'reference')])
def test_closed_loop_rollout_with_ue_arrival_and_departure(kind, U0, B, E, rng):
    """rollout_policy() on envs whose UE list changes (one launch per step inside ONE dcomp_rollout_ex call, the event feed and the
    actions of the launch before read in place; rng='reference': cut at the episode boundaries, where the host draws the next tape)
    against `act = heuristic_actions(obs); step(act)` with reset() at the horizon: every step's observation and reward, the final
    state and the next decision bit-identical.  Narrow dynamic kernel and the generic one (40 stations)."""
    import torch
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    L, T = 11, 27
    arrival = {1: 2, 3: -1, 4: 3, 7: -3, 9: 1}
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=1, num_slow=U0 - 2, num_fast=1)
    m, bs, ues = build_from_scenario(scn)
    mk = lambda: BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=13, rng=rng, rand_episodes=True, episode_length=L, ue_arrival=arrival, max_ues=U0 + 5)
    for name, eps in (('3gpp', 0.0), ('dynamic', 0.5), ('fullcomp', 0.0)):
        ref, env = mk(), mk()
        ref.reset()
        want_obs, want_rew, want_n = [], [], []
        for t in range(T):
            if ref.time == L:
                ref.reset()
            ref.step(ref.heuristic_actions(name, eps))
            want_obs.append(ref.obs.clone()); want_rew.append(ref.reward.clone()); want_n.append(ref.num_ue)
        ref.check()
        assert len(set(want_n)) >= 3
        assert env.set_policy(name, eps)
        env.reset()
        out = {'obs': torch.full((T,) + tuple(env.obs.shape), float('nan'), device='cuda'), 'reward': torch.empty((T,) + tuple(env.reward.shape), device='cuda')}
        env.rollout_policy(T, out=out, horizon=L)
        env.check()
        assert torch.equal(out['obs'], torch.stack(want_obs)) and torch.equal(out['reward'], torch.stack(want_rew)), name
        assert env.time == ref.time and env.num_ue == ref.num_ue
        for k in ('pos', 'mv', 'conn', 'ewma', 'uid') + (('conn_hi',) if env.conn_hi is not None else ()):
            assert torch.equal(getattr(env, k), getattr(ref, k)), (name, k)
        assert torch.equal(env.next_action, ref.heuristic_actions(name, eps))


@pytest.mark.parametrize('kind,U,B,E,rng', [('central', 10, 5, 4096, 'philox'), ('multi', 32, 10, 200, 'philox'), ('multi', 7, 3, 50, 'philox'),
                                            ('central', 12, 16, 40, 'philox'), ('multi', 5, 4, 9, 'reference'),
                                            ('multi', 128, 32, 3, 'philox'),
                                            ('multi', 24, 48, 6, 'philox'), ('central', 300, 9, 2, 'philox')])      # generic kernel: one launch per step
def test_closed_loop_rollout_equals_step_by_step(kind, U, B, E, rng):
    """rollout_policy(T) -- the policy's decisions taken inside the fused rollout kernel, reset() at the horizon -- against the
    same loop issued as `act = heuristic_actions(obs); step(act)`: every step's observation / reward and the final state
    bit-identical (the step-by-step loop is what test_heuristic_driven_rollout_matches_oracle ties to the oracle)."""
    import torch
    from deepcomp_amd import agents, scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    L, T = 9, 25
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=U // 8, num_slow=U - U // 8 - U // 4, num_fast=U // 4)
    m, bs, ues = build_from_scenario(scn)
    mk = lambda: BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=11, rng=rng, rand_episodes=(rng == 'philox'), episode_length=L)
    cl = agents.StaticClustering(2, bs, seed=5, device='cuda') if B >= 3 else None
    for name, eps in [('3gpp', 0.0), ('fullcomp', 0.0), ('dynamic', 0.4)] + ([('cluster', 0.0)] if cl else []):
        ref, env = mk(), mk()
        if name == 'cluster':
            cl.act(ref)                                   # builds cl._bits
        cm = cl._bits if name == 'cluster' else None
        ref.reset()
        want_obs, want_rew = [], []
        for t in range(T):
            if ref.time == L:
                ref.reset()
            ref.step(ref.heuristic_actions(name, eps, cm))
            want_obs.append(ref.obs.clone()); want_rew.append(ref.reward.clone())
        ref.check()
        assert env.set_policy(name, eps, cm)              # wide kernel (128 x 32): in-step policy, one launch per step
        with pytest.raises(RuntimeError):
            env.rollout_policy(T)                         # no next_action before the first reset / step
        env.reset()
        out = {'obs': torch.empty((T,) + tuple(env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(env.reward.shape), device='cuda')}
        env.rollout_policy(T, out=out, horizon=L)
        env.check()
        assert torch.equal(out['obs'], torch.stack(want_obs)) and torch.equal(out['reward'], torch.stack(want_rew)), name
        assert env.time == ref.time and env.episode == ref.episode
        for k in ('pos', 'mv', 'conn', 'ewma') + (('conn_hi',) if env.conn_hi is not None else ()):
            assert torch.equal(getattr(env, k), getattr(ref, k)), (name, k)
        assert torch.equal(env.next_action, ref.heuristic_actions(name, eps, cm))
        env2 = mk()                                       # last-step outputs only (out=None), then on with single steps
        env2.set_policy(name, eps, cm)
        env2.reset()
        env2.rollout_policy(L - 2)
        assert torch.equal(env2.obs, want_obs[L - 3]) and torch.equal(env2.reward, want_rew[L - 3])
        env2.step(env2.next_action)
        assert torch.equal(env2.obs, want_obs[L - 2])
