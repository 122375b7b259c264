"""dev: the sequence of tests/test_gpu_level_grid.py::test_a_grid_that_gives_up... under three engines, stage by stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
import test_gpu_level_grid as T
from scenarios import compare_dense

class Libs:
    OracleMap = staticmethod(pyoracle.OracleMap)

kind = "ref" if pyoracle.available("ref", "array") else "port"
for name, engine, spin in (("levels", "levels", -1), ("rounds", "rounds", -1), ("give-up", "levels", 0)):
    b = T._pair(pyoracle, kind, (64, 64, 48), engine, envelope=4, spin_limit=spin)
    st1, st2 = T._wide_update(b, 5, True)
    rep = compare_dense(b.gpu, b.cpu)
    e = rep["envelope"]
    print(name, "after wide:", {k: e[k] for k in ("disagree", "closer", "farther")}, "pairs", rep["pair_violations"], "levels", st1["levels"], st2["levels"])
    b.gpu.level_tuning(-1, 1 << 18)
    rng = np.random.RandomState(77)
    b.make_occupied(rng.randint(0, 48, (300, 3)).astype(np.int32))
    st3, _ = b.esdf()
    rep = compare_dense(b.gpu, b.cpu)
    e = rep["envelope"]
    print(name, "after 300 more:", {k: e[k] for k in ("disagree", "closer", "farther")}, "pairs", rep["pair_violations"], "levels", st3["levels"], "grid", st3["grid_levels"])
