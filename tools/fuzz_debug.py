import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import fuzz_parity as F
from deepcomp_amd.env import BatchedMobileEnv
from oracle import oracle as orc
idx = int(sys.argv[1])
rng = np.random.default_rng(0)
for i in range(idx + 1):
    c = F.random_case(rng)
print(F.describe(c))
E, U, B, kind, reward = c['E'], c['U'], c['B'], c['kind'], c['reward']
core = BatchedMobileEnv(c['m'], c['bs'], c['ues'], kind, num_envs=E, seed=c['seed'], reward=reward, rng='philox',
                        rand_episodes=True, env_id_base=c['base'], episode_length=1000)
envs = []
for e in range(E):
    o = orc.OracleEnv(c['w'], c['h'], c['bs_xy'], c['sh'], c['vel'], kind=orc.CENTRAL if kind == 'central' else orc.MULTI,
                      reward_agg={'avg': 0, 'sum': 1, 'min': 2}[reward], ue_util=c['util'], ue_dr_req=c['req'], init_xy=c['init'])
    o.set_philox(c['seed'], c['base'] + e); envs.append(o)
ob = orc.OracleBatch(envs)
arng = np.random.default_rng(c['seed'] ^ 0x5bd1e995)
core.reset(); ob.reset()
for t in range(c['steps']):
    if t == c['steps'] // 2:
        for o in ob.envs: o.set_episode(1)
        core.reset(); ob.reset()
    a = arng.integers(0, B + 1, size=(E, U)).astype(np.uint8)
    a[arng.random((E, U)) < c['p_noop']] = 0
    core.step(torch.from_numpy(a).cuda())
    obs_o, rew_o, conn_o, pos_o = ob.step(a)
    st = core.state_host()
    assert np.array_equal(st['pos'], pos_o) and np.array_equal(st['conn'], conn_o)
    got = core.ue_utility.cpu().numpy()
    want = np.stack([o.state()['utility'] for o in ob.envs])
    bad = np.argwhere(np.abs(got - want) > 1e-3)
    if len(bad):
        print('step', t, 'utility mismatches (env, ue):', bad.tolist())
        for e, u in bad[:4]:
            cm = int(conn_o[e][u]) if np.ndim(conn_o[e]) else conn_o[e]
            print(' env', e, 'ue', u, 'conn mask', bin(int(st['conn'][e][u])), 'dev util', got[e, u], 'oracle', want[e, u], 'dev dr', core.ue_dr.cpu().numpy()[e, u], 'oracle dr', ob.envs[e].state()['curr_dr'][u])
            p = pos_o[e][u]
            for b in range(B):
                if (int(st['conn'][e][u]) >> b) & 1:
                    others = [v for v in range(U) if (int(st['conn'][e][v]) >> b) & 1]
                    d2 = [(float((pos_o[e][v][0] - c['bs_xy'][b][0]) ** 2 + (pos_o[e][v][1] - c['bs_xy'][b][1]) ** 2), v) for v in others]
                    print('   bs', b, c['sh'][b], 'connected UEs (d2, ue):', sorted(d2)[:5])
        break
