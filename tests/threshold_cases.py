"""Placements within ulps of the connect-threshold circle (round 6).

The reference decides `can_connect` as `snr(self.pos.distance(ue_pos)) > 2e-8` (station.py:122-127, 222-226) with
`distance = sqrt(dx*dx + dy*dy)`: two rounded squares, a rounded sum, a rounded root, then the channel formula.  With d_T the smallest
double whose computed snr is NOT above the threshold, that is

    in range  <=>  sqrt_rn(q) < d_T  <=>  q < X,     q = fl(fl(dx*dx) + fl(dy*dy)),   X = min{ q : sqrt_rn(q) >= d_T }.

A device path that compares a FUSED d^2 (`fma(dy, dy, dx*dx)`) against `fl(d_T * d_T)` differs twice: fl(d_T^2) is one ulp above X, and the
fused sum differs from the two-rounding sum in ~18 % of pairs.  This module builds stations AT that boundary, for the tests (oracle = checker)
and for tests/golden/gen_golden.py (reference-run fixtures `traj_threshold_ulps_*`).  Pure numpy / stdlib; no reference code.
"""
import math
from fractions import Fraction

import numpy as np


def boundary_q(d_t):
    """X = smallest double q with sqrt_rn(q) >= d_t (math.sqrt is correctly rounded)."""
    q = d_t * d_t
    while math.sqrt(q) >= d_t:
        q = math.nextafter(q, 0.0)
    while math.sqrt(q) < d_t:
        q = math.nextafter(q, math.inf)
    return q


def q_ref(px, py, bx, by):
    """The reference's squared distance: two rounded products, one rounded sum (GEOS / the oracle's point_distance)."""
    dx = float(bx) - float(px)
    dy = float(by) - float(py)
    return dx * dx + dy * dy


def q_fused(px, py, bx, by):
    """fma(dy, dy, fl(dx*dx)): what the round-5 kernels evaluated (exact rational arithmetic, one rounding)."""
    dx = float(bx) - float(px)
    dy = float(by) - float(py)
    return float(Fraction(dy) * Fraction(dy) + Fraction(dx * dx))


def ulp_step(x, k):
    """x moved by k representable doubles."""
    for _ in range(abs(int(k))):
        x = math.nextafter(x, math.inf if k > 0 else -math.inf)
    return x


def _bits(x):
    return int(np.float64(x).view(np.int64))


def _from_bits(i):
    return float(np.int64(i).view(np.float64))


def station_at_threshold(px, py, theta, k, d_t, X=None, j=0):
    """A station (bx, by) whose reference-form squared distance to the UE at (px, py) sits at the boundary.  With Q = X moved by j doubles,
    by* is the y coordinate nearest the UE (on theta's side) at which q_ref first reaches Q; the result is by* moved k doubles AWAY from the
    UE (k < 0: towards it).  j = 0, k = 0: the first coordinate that is NOT connectable; j = 0, k = -1: the last connectable one.
    |sin theta| small: thousands of neighbouring coordinates share one q, so (j, k) walks q over {X - 1ulp, X, X + 1ulp, ...} exactly;
    |sin theta| large: q moves by a few ulps per coordinate step."""
    if X is None:
        X = boundary_q(d_t)
    Q = ulp_step(X, j)
    bx = float(px) + d_t * math.cos(theta)
    dx = bx - float(px)
    rem = Q - dx * dx
    assert rem > 1e-7, 'theta too close to the x axis for a y-coordinate walk (keep |sin theta| >= 2e-5)'
    sgn = 1.0 if math.sin(theta) >= 0 else -1.0
    by0 = float(py) + sgn * math.sqrt(rem)
    assert abs(by0) > 1e-6, 'station y coordinate at zero: the bit walk would cross the sign'
    b0 = _bits(by0)
    away = (1 if by0 > py else -1) * (1 if by0 > 0 else -1)      # the bit direction that leads away from the UE (bits grow with |value|)
    span = 1 << 24
    a, b = b0 - away * span, b0 + away * span       # a: nearer the UE (q < Q), b: farther (q >= Q)
    assert q_ref(px, py, bx, _from_bits(a)) < Q <= q_ref(px, py, bx, _from_bits(b)), 'search window does not bracket the boundary'
    while abs(b - a) > 1:
        m = (a + b) // 2
        if q_ref(px, py, bx, _from_bits(m)) >= Q:
            b = m
        else:
            a = m
    return bx, _from_bits(b + away * int(k))


def classify(px, py, bx, by, d_t, X=None):
    """(reference decision, round-5 kernel decision, q_ref - X in ulps of X)."""
    if X is None:
        X = boundary_q(d_t)
    qr, qf = q_ref(px, py, bx, by), q_fused(px, py, bx, by)
    ref = qr < X
    old = qf < d_t * d_t
    return ref, old, (qr - X) / math.ulp(X)


def static_case(rng, U, B, width, height, d_t, k_choices=(-3, -2, -1, 0, 1, 2, 3), fine_share=0.5):
    """U static UEs on integer points; station b sits at the threshold circle of UE b % U.  Returns (ue_xy [U, 2] int, bs_pos [B, 2] float64,
    target_ue [B], info) -- info counts how many pairs the round-5 predicate decides differently from the reference form."""
    X = boundary_q(d_t)
    ue_xy = np.stack([rng.integers(0, int(width) + 1, U), rng.integers(0, int(height) + 1, U)], axis=1).astype(np.int64)
    bs = np.zeros((B, 2), dtype=np.float64)
    tgt = np.arange(B) % U
    n_diff = n_edge = 0
    for b in range(B):
        px, py = (float(v) for v in ue_xy[tgt[b]])
        # fine: |sin| small -> q moves by less than an ulp per coordinate step; coarse: anywhere on the circle
        if rng.random() < fine_share:
            theta = rng.choice([0.0, math.pi]) + rng.uniform(-0.03, 0.03)
            if abs(math.sin(theta)) < 2e-5:
                theta += 1e-3
        else:
            theta = rng.uniform(0, 2 * math.pi)
        k = int(rng.choice(k_choices))
        j = int(rng.integers(-2, 3))
        bx, by = station_at_threshold(px, py, theta, k, d_t, X, j)
        bs[b] = (bx, by)
        ref, old, du = classify(px, py, bx, by, d_t, X)
        n_diff += int(ref != old)
        n_edge += int(abs(du) <= 1.0)
    return ue_xy, bs, tgt, {'differs_from_round5_predicate': n_diff, 'within_one_ulp_of_X': n_edge, 'X': X}


def moving_case(positions, rng, B, d_t, t_lo=2, k_choices=(-2, -1, 0, 1, 2)):
    """positions: [T + 1, E, U, 2] FP64 trajectory (index 0 = after reset, t = after step t) of an env batch whose movement does not depend on
    the stations.  Station b sits at the threshold circle of the position ONE (env, UE) holds after step t_b (the drop decision of step t_b, and
    the connect decision of step t_b + 1, are taken there).  Returns (bs_pos [B, 2], actions [T, E, U] uint8, n_decisions)."""
    X = boundary_q(d_t)
    T = positions.shape[0] - 1
    E, U = positions.shape[1], positions.shape[2]
    bs = np.zeros((B, 2), dtype=np.float64)
    act = np.zeros((T, E, U), dtype=np.uint8)
    busy = set()                                     # (t, e, u) whose action slot is taken
    n_dec = 0
    for b in range(B):
        for _ in range(64):
            e, u, t = int(rng.integers(E)), int(rng.integers(U)), int(rng.integers(t_lo, T))
            if (t, e, u) not in busy:                # slot index t = action of step t + 1 (0-based step list)
                break
        px, py = (float(v) for v in positions[t, e, u])
        theta = rng.uniform(0, 2 * math.pi) if rng.random() < 0.5 else rng.choice([0.0, math.pi]) + rng.uniform(-0.03, 0.03)
        if abs(math.sin(theta)) < 2e-5:
            theta += 1e-3
        bx, by = station_at_threshold(px, py, theta, int(rng.choice(k_choices)), d_t, X, int(rng.integers(-2, 3)))
        bs[b] = (bx, by)
        # connect as early as the UE is comfortably in range before step t (so that step t's DROP decision happens at the boundary) ...
        for s in range(max(0, t - 6), t):
            qs = q_ref(*positions[s, e, u], bx, by)          # pre-move position of step s + 1 is positions[s]
            if qs < X * (1 - 1e-6) and (s, e, u) not in busy:
                act[s, e, u] = b + 1
                busy.add((s, e, u))
                n_dec += 1
                break
        # ... and toggle once more right at the boundary position (step t + 1 acts on positions[t]): disconnect if it survived, else a connect
        # attempt decided at the boundary
        if t < T and (t, e, u) not in busy:
            act[t, e, u] = b + 1
            busy.add((t, e, u))
            n_dec += 1
    return bs, act, n_dec
