"""Re-run ONE case of a tools/fuzz_parity.py run and, where the per-UE rates differ, dump the connected UEs of the env (positions, squared distances,
step of connection, both rates).  `python tools/fuzz_case.py <seed> <case index> <--many-stations fraction> <--many-ues fraction>` (GPU box);
e.g. `tools/fuzz_case.py 990002 92 0.6 0.0`: the max-cap tie of DESIGN.md section 6 / profiles/r06_maxcap_log10_ulp.txt."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tools'))
import torch
import fuzz_parity as F
from tests import parity

seed, idx, ms, mu = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
rng = np.random.default_rng(seed)
for i in range(idx + 1):
    spec = F.many_ues(F.many_stations(F.random_spec(rng), ms), mu)
c = F.build_case(spec)
print(F.describe(c))
orig = parity.assert_rates

def hook(core, ob, msg='', **kw):
    try:
        return orig(core, ob, msg, **kw)
    except AssertionError:
        r = ob.rates(want_dr_rel=True)
        E, U = r['curr_dr'].shape
        dr = core.ue_dr.cpu().numpy().reshape(E, U)
        bad = np.argwhere(np.abs(dr - r['curr_dr']) > 1e-5 * np.abs(r['curr_dr']) + 1e-30)
        st = core.state_host()
        print(msg, 'kernel', core.step_kernel_name, 'bad entries', bad.tolist())
        for e in sorted(set(int(b[0]) for b in bad)):
            pos = st['pos'][e]
            conn = st['conn'][e]
            print(' env', e, 'station', c['bs_xy'], 'sharing', c['sh'])
            cs = core.conn_since.cpu().numpy().reshape(E, U, -1)[e] if getattr(core, 'conn_since', None) is not None else None
            for u in range(U):
                if conn[u]:
                    d2 = [(pos[u][0] - bx) ** 2 + (pos[u][1] - by) ** 2 for bx, by in c['bs_xy']]
                    print('   ue', u, 'pos', pos[u].tolist(), 'conn', int(conn[u]), 'd2', [repr(x) for x in d2], 'hip dr', dr[e, u], 'oracle dr', r['curr_dr'][e, u], 'since', None if cs is None else cs[u].tolist())
        raise
parity.assert_rates = hook
F.parity.assert_rates = hook
F.run_case(c, torch)
print('case agrees')
