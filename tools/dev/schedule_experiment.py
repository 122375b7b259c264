"""dev (CPU only): level-schedule model, schedule 1 (orphans wait for their first pull) against schedule 2 (the dead cell is
filled from its rim inwards before level 0, as the reference's list walk does) -- both judged by the reference's own order
spread on the scenarios where the level engine is outside the strict contract, and on the ordinary ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
from scenarios import P_DEFAULT, EnvelopeOracle, all_voxels, d2_from_dist, hash_key

KIND = "ref" if pyoracle.available("ref", "array") else "port"

def dense_wide(seed, sched, n=(64, 64, 48), k=4):
    res = 0.1
    size = tuple(np.asarray(n) * res)
    eng = pyoracle.OracleMap((0, 0, 0), res, size, kind="port"); eng.set_schedule(sched)
    env = EnvelopeOracle(lambda: pyoracle.OracleMap((0, 0, 0), res, size, kind=KIND), k=k)
    for m in (eng, env): m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
    def both(f): f(eng); f(env)
    rng = np.random.RandomState(seed)
    gs = np.array(eng.grid_size); g = all_voxels(eng.grid_size)
    blocks = rng.rand(*(gs // 4 + 1)) > 0.25
    g = g[blocks[g[:, 0] // 4, g[:, 1] // 4, g[:, 2] // 4]]
    both(lambda m: m.SetOccupancyVox(g, 0)); both(lambda m: m.UpdateOccupancy(True)); both(lambda m: m.UpdateESDF())
    S = g[rng.choice(len(g), 400, replace=False)]
    for _ in range(3): both(lambda m: m.SetOccupancyVox(S, 1)); both(lambda m: m.UpdateOccupancy(True))
    both(lambda m: m.UpdateESDF())
    out = []
    e = env.judge(d2_from_dist(eng.dump_dense(("dist",))["dist"], res)); out.append((e["closer"], e["farther"], e["disagree"]))
    T = g[rng.choice(len(g), 200, replace=False)]
    for _ in range(6):
        both(lambda m: m.SetOccupancyVox(T, 1)); both(lambda m: m.SetOccupancyVox(S[:200], 0)); both(lambda m: m.UpdateOccupancy(True))
    both(lambda m: m.UpdateESDF())
    e = env.judge(d2_from_dist(eng.dump_dense(("dist",))["dist"], res)); out.append((e["closer"], e["farther"], e["disagree"]))
    return out

def hash_fuzz(seed, sched):
    kind = KIND if pyoracle.available(KIND, "hash") else "port"
    rng = np.random.RandomState(seed)
    origin, res = tuple(float(v) for v in rng.uniform(-1, 1, 3)), float(rng.choice([0.05, 0.1]))
    rng.choice([0, 1000, 50000])
    eng = pyoracle.OracleMap(origin, res, reserve_size=1000, mode="hash", kind="port"); eng.set_schedule(sched)
    cpu = EnvelopeOracle(lambda: pyoracle.OracleMap(origin, res, reserve_size=1000, mode="hash", kind=kind), k=6)
    for m in (eng, cpu): m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
    centre = rng.randint(-30, 30, 3); live = np.zeros((0, 3), np.int32); outside = []
    for step in range(5):
        centre = centre + rng.randint(-6, 7, 3); ext = rng.randint(8, 22, 3)
        box = (all_voxels(tuple(int(v) for v in ext)) + (centre - ext // 2)).astype(np.int32)
        new = box[rng.rand(len(box)) < 0.01]; gone = live[rng.rand(len(live)) < 0.4]
        for k in range(3):
            if k == 0: eng.SetOccupancyVox(box, 0); cpu.SetOccupancyVox(box, 0)
            for vv, o in ((new, 1), (gone, 0)):
                if len(vv): eng.SetOccupancyVox(vv, o); cpu.SetOccupancyVox(vv, o)
            assert eng.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
        eng.UpdateESDF(); cpu.UpdateESDF()
        d = eng.dump_hash(); ok = d["vox"][:, 0] != -10000
        keys, d2 = hash_key(d["vox"][ok]), d2_from_dist(d["dist"][ok], res); o = np.argsort(keys)
        env = cpu.judge(d2[o], keys=keys[o]); outside.append((env["closer"], env["farther"], env["disagree"]))
        live = np.concatenate([live, new]); rng.uniform(-25, 25, (150, 3))
    return outside

if __name__ == "__main__":
    import contextlib, io
    for name, f in [("dense wide seed 9", lambda s: dense_wide(9, s)), ("dense wide seed 5", lambda s: dense_wide(5, s)), ("dense wide seed 3", lambda s: dense_wide(3, s)),
                    ("hash fuzz 63", lambda s: hash_fuzz(63, s)), ("hash fuzz 62", lambda s: hash_fuzz(62, s)), ("hash fuzz 61", lambda s: hash_fuzz(61, s))]:
        for sched in (1, 2, 3, 4, 5, 6):
            print(name, "schedule", sched, "(closer, farther, disagree) per state:", f(sched), flush=True)
