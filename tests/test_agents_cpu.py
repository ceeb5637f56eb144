"""Batched heuristic agents (deepcomp_amd/agents.py) against decisions recorded from the reference's own agents
(deepcomp/agent/heuristics.py run by tests/golden/gen_golden.py::gen_heuristics) on the same observations."""
import os

import numpy as np
import pytest
import torch

from deepcomp_amd import agents, scenarios
from deepcomp_amd.entities import build_from_scenario

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'heuristics.npz'))
OBS = {'dr': torch.from_numpy(G['obs_dr']), 'connected': torch.from_numpy(G['obs_connected'])}   # [T, U, B] float64


@pytest.mark.parametrize('name,agent', [('3gpp', agents.Heuristic3GPP()), ('fullcomp', agents.FullCoMP()),
                                        ('dynamic05', agents.DynamicSelection(0.5)), ('dynamic09', agents.DynamicSelection(0.9))])
def test_heuristic_matches_reference(name, agent):
    got = agent(OBS).numpy()
    assert got.dtype == np.uint8 and got.shape == G['act_' + name].shape
    assert np.array_equal(got, G['act_' + name])
    assert len(np.unique(got)) > 2          # the recorded episode exercises several branches


def test_static_clustering_matches_reference():
    _, bs_list, _ = build_from_scenario(scenarios.grid_map(10, 'mixed'))
    ag = agents.StaticClustering(3, bs_list, seed=1)
    assert np.array_equal(ag.member.numpy().astype(np.uint8), G['static3_member'])     # same clusters (same stdlib RNG draws)
    assert np.array_equal(ag(OBS).numpy(), G['act_static3'])


def test_tie_rules():
    obs = {'dr': torch.tensor([[1.0, 1.0, 0.5], [0.2, 1.0, 1.0], [1.0, 0.3, 0.3]]),
           'connected': torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [1.0, 1.0, 1.0]])}
    assert agents.Heuristic3GPP()(obs).tolist() == [1, 1, 0]          # first max; disconnect the other cell first; stay
    assert agents.FullCoMP()(obs).tolist() == [1, 2, 0]               # strict '>' scan keeps the first of equals
    assert agents.DynamicSelection(0.9)(obs).tolist() == [1, 1, 2]    # drop cells outside the set in index order
    a = agents.RandomAgent(3, seed=1)(obs)
    assert a.shape == (3,) and int(a.max()) <= 3
