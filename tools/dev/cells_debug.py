import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from scipy import ndimage
import fiesta_amd
from scenarios import P_DEFAULT
n = 128
m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, ((n - .5) * .1,) * 3, update_engine="cells")
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
m.SetOccupancyBox((0, 0, 0), (n - 1,) * 3, 0); m.UpdateOccupancy(True); m.UpdateESDF()
S = np.random.RandomState(12345).randint(0, n, (1000, 3)).astype(np.int32)
for _ in range(3):
    m.SetOccupancy(S, 1, want_ret=False); m.UpdateOccupancy(True)
st = m.UpdateESDF()
print({k: st[k] for k in ("bulk", "cells", "nn_failed", "nn_entries")})
f = m.download_field(("d2", "occ"))
occ = f["occ"].reshape(n, n, n); d2 = f["d2"].reshape(n, n, n).astype(np.int64)
idx = ndimage.distance_transform_edt(occ == 0, return_distances=False, return_indices=True)
g = np.meshgrid(*[np.arange(n)] * 3, indexing="ij")
want = sum((idx[k] - g[k]) ** 2 for k in range(3))
bad = d2 != want
print("bad", bad.sum(), "of", bad.size)
bc = bad.reshape(16, 8, 16, 8, 16, 8).any(axis=(1, 3, 5))
print("bad cells", bc.sum(), "of", bc.size)
print("by cz", bc.sum(axis=(0, 1)))
print("by cy", bc.sum(axis=(0, 2)))
print("by cx", bc.sum(axis=(1, 2)))
w = np.argwhere(bad)[:5]
for v in w:
    print(v, "got", d2[tuple(v)], "want", want[tuple(v)])
# within bad cells: fraction of voxels bad, by x-slab
bx = bad.reshape(16, 8, 16, 8, 16, 8).sum(axis=(0, 2, 3, 4, 5))
print("bad voxels by x in cell", bx)
by = bad.reshape(16, 8, 16, 8, 16, 8).sum(axis=(0, 1, 2, 4, 5))
print("by y in cell", by)
bz = bad.reshape(16, 8, 16, 8, 16, 8).sum(axis=(0, 1, 2, 3, 4))
print("by z in cell", bz)
f2 = m.download_field(("coc",))["coc"].reshape(n, n, n, 3)
print("cell(0,0,0) bad mask by x-slab (rows y, cols z):")
for x in range(8):
    print("x", x, ["".join("X" if bad[x, y, z] else "." for z in range(8)) for y in range(8)])
print("coc at (0,0,0..7):", f2[0, 0, :8].tolist())
print("want site:", [(idx[0][0, 0, z], idx[1][0, 0, z], idx[2][0, 0, z]) for z in range(8)])
for cc in [(0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 0, 4), (0, 1, 0)]:
    sub = bad[cc[0]*8:cc[0]*8+8, cc[1]*8:cc[1]*8+8, cc[2]*8:cc[2]*8+8]
    print("cell", cc, "bad by x:", sub.sum(axis=(1, 2)).tolist(), "by y:", sub.sum(axis=(0, 2)).tolist(), "by z:", sub.sum(axis=(0, 1)).tolist())
cocg = f2.astype(np.int64)
wantsite = np.stack([idx[0], idx[1], idx[2]], -1).astype(np.int64)
bv = np.argwhere(bad)
rs = np.random.RandomState(0).choice(len(bv), 2000, replace=False)
hits = {}
for v in bv[rs]:
    got = cocg[tuple(v)]
    q = (v[2] // 32) * 32
    for dz in range(0, 32, 8):
        for dx in range(8):
            u = (v[0] // 8 * 8 + dx, v[1], q + dz + v[2] % 8)
            if np.array_equal(wantsite[u], got):
                key = (dx - v[0] % 8, (q + dz) // 8 - v[2] // 8)
                hits[key] = hits.get(key, 0) + 1
print("bad voxels whose value is the right answer of voxel (x + dx, same y, same z%8 in cell cz + dcz): ", sorted(hits.items(), key=lambda t: -t[1])[:12])
