"""Scenario geometry tables: map sizes, base-station coordinates and per-BS sharing models.

These are *data* restated from the reference's scenario factory (numbers only):
``deepcomp/util/env_setup.py:40-49`` (mixed sharing rule), ``:52-62`` (small), ``:87-104`` (medium, 3 BS on
an equilateral triangle), ``:107-142`` (large, <=7 BS hexagon), ``:164-176`` (custom, 4 BS) and ``:145-161``
(UE mix: static, then slow, then fast, ids '1'..'U'), ``:205-226`` (the named UE-arrival schedules).  ``grid_map`` is the
synthetic layout SURVEY.md §8d defines for BS counts the stock maps do not have (10, 32, ...).

Everything returns plain Python/NumPy values; ``deepcomp_amd.entities`` turns them into the
``Map``/``Basestation``/``User`` config objects the env constructors take.
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple

SHARING_MODELS = ('resource-fair', 'rate-fair', 'max-cap', 'proportional-fair')
_MIXED_CYCLE = ('resource-fair', 'rate-fair', 'proportional-fair')


def sharing_for_bs(sharing: str, bs_idx: int) -> str:
    """'mixed' cycles resource-/rate-/proportional-fair by BS index (env_setup.py:40-49)."""
    if sharing != 'mixed':
        if sharing not in SHARING_MODELS:
            raise AssertionError(f"sharing model {sharing!r} not supported: {SHARING_MODELS}")
        return sharing
    return _MIXED_CYCLE[bs_idx % len(_MIXED_CYCLE)]


@dataclass
class Scenario:
    width: float                      # raw (Map int()-truncates, map.py:20-21)
    height: float
    bs_ids: List[str]
    bs_pos: List[Tuple[float, float]]
    bs_sharing: List[str]
    name: str = ''
    ue_specs: list = field(default_factory=list)   # filled by with_ues()

    @property
    def num_bs(self):
        return len(self.bs_pos)

    def with_ues(self, num_static=0, num_slow=0, num_fast=0, util_func='log'):
        """UE list in the reference's order: static, slow, fast; ids '1'.. (env_setup.py:145-161)."""
        specs = []
        uid = 1
        for vel, n in ((0, num_static), ('slow', num_slow), ('fast', num_fast)):
            for _ in range(n):
                specs.append(dict(id=str(uid), pos_x='random', pos_y='random', velocity=vel,
                                  util_func=util_func, dr_req=1))
                uid += 1
        self.ue_specs = specs
        return self


def _ids(n):
    return [chr(ord('A') + i) if i < 26 else f'B{i}' for i in range(n)]


def small_map(sharing='mixed') -> Scenario:
    pos = [(50, 50), (100, 50)]
    return Scenario(150, 100, _ids(2), pos, [sharing_for_bs(sharing, i) for i in range(2)], 'small')


def medium_map(sharing='mixed', bs_dist=100, dist_to_border=10) -> Scenario:
    y_dist = math.sqrt(bs_dist ** 2 - (bs_dist / 2) ** 2)
    pos = [(dist_to_border, dist_to_border), (dist_to_border + bs_dist, dist_to_border),
           (dist_to_border + bs_dist / 2, dist_to_border + y_dist)]
    return Scenario(2 * dist_to_border + bs_dist, 2 * dist_to_border + y_dist, _ids(3), pos,
                    [sharing_for_bs(sharing, i) for i in range(3)], 'medium')


_LARGE_BS = [(115, 130), (30, 80), (115, 30), (200, 80), (200, 180), (115, 230), (30, 180)]


def large_map(sharing='mixed', num_bs=None, dist_to_border=10) -> Scenario:
    if num_bs is None:
        pos, w, h = list(_LARGE_BS), 230, 260
    else:
        assert 1 <= num_bs <= 7, "Only support 1-7 BS in large env"
        pos = _LARGE_BS[:num_bs]
        w = max(p[0] for p in pos) + dist_to_border
        h = max(p[1] for p in pos) + dist_to_border
    return Scenario(w, h, _ids(len(pos)), pos, [sharing_for_bs(sharing, i) for i in range(len(pos))], 'large')


def custom_map(sharing='mixed') -> Scenario:
    pos = [(10, 60), (97, 10), (184, 60), (97, 110)]
    return Scenario(194, 120, _ids(4), pos, [sharing_for_bs(sharing, i) for i in range(4)], 'custom')


def grid_map(num_bs, sharing='mixed', pitch=100, border=50) -> Scenario:
    """Synthetic square grid: cols=ceil(sqrt(B)), row-major fill, same layout in every env."""
    cols = int(math.ceil(math.sqrt(num_bs)))
    rows = int(math.ceil(num_bs / cols))
    pos = [(border + pitch * (i % cols), border + pitch * (i // cols)) for i in range(num_bs)]
    return Scenario(2 * border + pitch * (cols - 1), 2 * border + pitch * (rows - 1), _ids(num_bs), pos,
                    [sharing_for_bs(sharing, i) for i in range(num_bs)], f'grid{num_bs}')


# The CLI's named UE-arrival schedules (`--ue-arrival`, env_setup.py:205-226): {step: +n UEs arrive | -n UEs leave}, what
# env_config['ue_arrival'] carries (base.py:52-56, 433-443).  Data; tests/golden/ue_arrival_schedules.json holds what the reference's own
# get_ue_arrival returns.
UE_ARRIVAL = {
    'oneupdown': {10: 1, 30: -1},
    'updownupdown': {10: 1, 20: -1, 30: 1, 40: -1},
    '3up2down': {10: 3, 30: -2},
    'updown': {10: 1, 15: 1, 20: 1, 40: 1, 50: -1, 60: -1},
    'largeupdown': {20: 1, 30: -1, 40: 1, 45: 1, 50: 1, 55: 2, 60: 3, 65: 2, 70: 1, 75: -1, 80: -2, 85: -3, 90: -3, 95: -2},
}


def get_ue_arrival(name):
    """Named schedule -> the dict env_config['ue_arrival'] / BatchedMobileEnv(ue_arrival=...) take; None -> None (fixed UE list).
    A fresh dict every call (callers may edit theirs).  Unknown names: AssertionError, like the reference (env_setup.py:207)."""
    assert name is None or name in UE_ARRIVAL, f"UE arrival {name!r} is not one of {sorted(UE_ARRIVAL)} / None"
    return None if name is None else dict(UE_ARRIVAL[name])


def get_scenario(map_size, sharing='mixed', bs_dist=100, num_bs=None) -> Scenario:
    """Mirror of ``get_env``'s map switch (env_setup.py:179-202) plus 'grid' for synthetic sizes."""
    if map_size == 'small':
        return small_map(sharing)
    if map_size == 'medium':
        return medium_map(sharing, bs_dist=bs_dist)
    if map_size == 'large':
        return large_map(sharing, num_bs=num_bs)
    if map_size == 'custom':
        return custom_map(sharing)
    if map_size == 'grid':
        return grid_map(num_bs, sharing)
    raise AssertionError(f"Environment {map_size} is not one of small/medium/large/custom/grid")
