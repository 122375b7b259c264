"""The CPU restatement against the committed golden fixtures (tests/golden/*.npz), which were produced by
the reference's own sources compiled verbatim (tests/golden/make_golden.py).  This is the pin that travels:
it runs on any machine, including the GPU box, where /root/reference does not exist.
Bar: bit-exact in every stored array, including closest-obstacle ids and the printed counters, because the
restatement follows the reference's FIFO order literally.
"""
import os

import numpy as np
import pytest

from golden_programs import PROGRAMS, golden_rays

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_restatement_matches_reference_fixture(oracle_libs, name):
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))

    def make(origin, res, size):
        return oracle_libs.OracleMap(origin, res, size, kind="port")
    seen = 0
    for cp, m, extra in PROGRAMS[name](make):
        assert tuple(gold[f"{cp}/grid_size"]) == m.grid_size
        d = m.dump_dense()
        assert np.array_equal(d["dist"], gold[f"{cp}/dist"]), (name, cp, "distance_buffer_")
        assert np.array_equal(d["coc"], gold[f"{cp}/coc"].astype(np.int32)), (name, cp, "closest_obstacle_")
        assert np.array_equal(d["occ"], gold[f"{cp}/occ"]), (name, cp, "Exist")
        assert np.array_equal(d["logodds"], gold[f"{cp}/logodds"]), (name, cp, "occupancy_buffer_")
        for k, v in extra.items():
            if k == "stats":
                got = np.array([v["inserted"], v["deleted"], v["expanded"], v["change_num"]])
                assert np.array_equal(got, gold[f"{cp}/stats"]), (name, cp, "UpdateESDF counters")
            elif k == "pos":
                assert np.array_equal(m.GetDistancePos(v), gold[f"{cp}/GetDistance"])
                dist, grad = m.GetDistWithGradTrilinear(v)
                assert np.array_equal(dist, gold[f"{cp}/TrilinearDist"])
                assert np.array_equal(grad, gold[f"{cp}/TrilinearGrad"])
                assert np.array_equal(m.GetOccupancyPos(v), gold[f"{cp}/GetOccupancy"])
            else:
                assert np.array_equal(np.asarray(v), gold[f"{cp}/{k}"]), (name, cp, k)
        assert m.CheckConsistency()
        seen += 1
    assert seen >= 2


def test_raycast_known_answers(oracle_libs):
    gold = np.load(os.path.join(GOLD, "raycast_kat.npz"))
    rays, lo, hi = golden_rays()
    assert np.array_equal(lo, gold["lo"]) and np.array_equal(hi, gold["hi"])
    total = 0
    for i, (a, b) in enumerate(rays):
        assert np.array_equal(a, gold[f"a{i}"]) and np.array_equal(b, gold[f"b{i}"])
        got = oracle_libs.raycast(a, b, lo, hi, kind="port")
        assert np.array_equal(got, gold[f"v{i}"].astype(np.float64)), i
        total += len(got)
    assert total > 1000
