"""CPU: the numpy model of the masked transform (tests/masked_model.py -- what the GPU tests compare the kernels of
fiesta_amd/csrc/mask_kernels.hpp with, voxel for voxel) against the reference's own order spread, no GPU needed.

Scenario: bench.py's C2-partial workload at a size the oracle runs in a second -- a map with a quarter of its 32^3 blocks never
observed, obstacles (also inside the unobserved blocks) inserted, then half of them replaced per step -- driven through K + 1
runs of the oracle in shuffled queue order (tests/scenarios.py: EnvelopeOracle); the model is judged like any engine: closer /
farther than every run on at most as many voxels as the runs disagree on (strict)."""
import os
import sys

import numpy as np

from scenarios import P_DEFAULT, EnvelopeOracle, assert_envelope

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_certificate_walk_visits_every_voxel_of_the_discrete_line():
    import masked_model
    obs = np.ones((24, 24, 24), bool)
    V = np.array([[2, 3, 4]] * 3)
    S = np.array([[20, 9, 4], [2, 3, 22], [15, 16, 17]])
    assert masked_model.certificate(obs, V, S).all()
    for v, s in zip(V, S):   # knocking out any voxel strictly between the two on the rounded line breaks the certificate
        d = s - v
        n = 2 * np.abs(d).max() + 1
        pts = {tuple(v + (2 * d * i + n) // (2 * n)) for i in range(1, n)} - {tuple(v)}
        for p in pts:
            o = obs.copy()
            o[p] = False
            assert not masked_model.certificate(o, v[None], s[None])[0], (v, s, p)
    # ... and a voxel off the line does not
    o = obs.copy()
    o[10, 20, 20] = False
    assert masked_model.certificate(o, V, S).all()


def test_isolated_obstacles_are_no_sites():
    import masked_model
    occ = np.zeros((16, 16, 16), bool)
    obs = np.zeros((16, 16, 16), bool)
    obs[:8] = True
    occ[3, 3, 3] = occ[12, 8, 8] = occ[9, 4, 4] = True    # the second: nobody around it observed; the third: two voxels from x = 7
    obs[12, 8, 8] = obs[9, 4, 4] = True
    eff = masked_model.effective_sites(occ, obs)
    assert eff[3, 3, 3] and not eff[12, 8, 8] and eff[9, 4, 4]   # (9,4,4) pushes to (7,4,4) along (-2,0,0)
    d2, W, st = masked_model.masked_engine(occ, obs)
    assert d2[12, 8, 8] == 0 and d2[9, 4, 4] == 0 and d2[7, 4, 4] == 4 and st["isolated_obstacles"] == 1
    assert d2[8, 4, 4] == -1   # never observed


def test_model_stays_inside_the_reference_envelope(oracle_libs, best_oracle_kind):
    import masked_model
    sys.path.insert(0, ROOT)
    import bench
    G, res = 96, 0.1
    keep = np.random.RandomState(2718).rand(3, 3, 3) >= 0.27
    env = EnvelopeOracle(lambda: oracle_libs.OracleMap((0, 0, 0), res, ((G - 0.5) * res,) * 3, kind=best_oracle_kind), k=3)
    env.SetParameters(*P_DEFAULT)
    env.SetOriginalRange()
    env.primary.SetOccupancyVox(np.argwhere(np.repeat(np.repeat(np.repeat(keep, 32, 0), 32, 1), 32, 2)).astype(np.int32), 0)
    env.UpdateOccupancy(True)
    env.UpdateESDF()
    w = bench.Workload(G, 330, seed=12345)
    for _ in range(3):
        env.primary.SetOccupancyVox(w.initial(), 1)
        env.UpdateOccupancy(True)
    W = None
    for step in range(3):
        if step:
            new, old = w.next_step()
            for c in range(3):
                env.primary.SetOccupancyVox(new, 1)
                if c == 2:
                    env.primary.SetOccupancyVox(old, 0)
                env.UpdateOccupancy(True)
        env.UpdateESDF()
        d = env.primary.dump_dense(("dist", "occ"))
        occ, obs = d["occ"].reshape(G, G, G) != 0, d["dist"].reshape(G, G, G) >= 0
        d2, W, st = masked_model.masked_engine(occ, obs, W)
        assert st["uncertified"] > 0 and st["isolated_obstacles"] > 0, st
        rep = env.judge(d2.reshape(-1))
        assert rep["inf_where_every_run_is_finite"] == 0 and rep["finite_where_every_run_is_inf"] == 0, rep
        assert_envelope(rep, f"masked model, 96^3 C2-partial, step {step}", strict=True)
    env.close()


def test_certificate_from_cell_crossings_equals_the_sample_walk():
    """k_mask_walk decides the straight certificate from the cells a segment crosses where no cell on its way is partly observed
    (mask_kernels.hpp: mask_segment_cells); the sample walk is the definition (mask_segment_samples, masked_model.certificate)."""
    import masked_model
    rng = np.random.RandomState(5)
    C = 12
    for density in (0.1, 0.3, 0.6):
        cellobs = (rng.rand(C, C, C) >= density).astype(np.uint8)
        obs = np.repeat(np.repeat(np.repeat(cellobs != 0, 8, 0), 8, 1), 8, 2)
        V = rng.randint(0, 8 * C, (6000, 3))
        S = np.clip(V + rng.randint(-40, 41, (6000, 3)), 0, 8 * C - 1)
        keep = obs[V[:, 0], V[:, 1], V[:, 2]] & (np.abs(S - V).max(1) > 0)
        V, S = V[keep], S[keep]
        a = masked_model.certificate(obs, V, S)
        b = masked_model.certificate_by_cells(cellobs, V, S)
        assert np.array_equal(a, b), int((a != b).sum())
        assert 0 < a.sum() < len(a)
