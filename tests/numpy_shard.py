"""TEST INFRASTRUCTURE: a numpy stand-in for one shard engine, with the interface fiesta_amd.sharded drives.

It exists so that the multi-process protocol (exchange plan, three-phase ghost sweep, transition all-gather,
convergence all-reduce) can be exercised on CPU with the gloo backend -- the HIP engine has no CPU fallback and
never will.  The relaxation is a brute-force Jacobi iteration of the reference's 24-direction operator over the
local array (ghost cells are sources only); occupancy is deterministic (one majority vote per UpdateOccupancy),
which is all the protocol needs.  Word encoding = the engine's (fiesta_amd/csrc/common.hpp).
"""
import numpy as np

UNOBS, INF, ACT, NOCOC = 0xFFFFFFFF, 0x80000000, 0x40000000, 0x80000000
DIRS = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1), (-1, -1, 0), (1, 1, 0), (0, -1, -1),
        (0, 1, 1), (-1, 0, -1), (1, 0, 1), (-1, 1, 0), (1, -1, 0), (0, -1, 1), (0, 1, -1), (1, 0, -1), (-1, 0, 1),
        (-2, 0, 0), (2, 0, 0), (0, -2, 0), (0, 2, 0), (0, 0, -2), (0, 0, 2)]  # include/parameters.h:54-68


def pack(x, y, z):
    return (np.asarray(x, np.uint32) << 20) | (np.asarray(y, np.uint32) << 10) | np.asarray(z, np.uint32)


class NumpyShard:
    def __init__(self, origin, res, size_m, shard_lo, global_grid):
        gs = np.ceil(np.asarray(size_m) / res).astype(int)
        lo = np.asarray(shard_lo)
        gg = np.asarray(global_grid)
        glo = np.where(lo > 0, 2, 0)
        ghi = np.where(lo + gs < gg, 2, 0)
        self.dims = tuple(int(v) for v in gs + glo + ghi)
        self.g0 = lo - glo
        self.olo, self.ohi = glo, glo + gs - 1
        self.gg = tuple(int(v) for v in gg)
        self.coc = np.full(self.dims, UNOBS, np.uint32)
        self.occ = np.zeros(self.dims, bool)
        self.gocc = np.zeros(self.gg, bool)
        self.cnt_hit = np.zeros(self.dims, np.int32)
        self.cnt_all = np.zeros(self.dims, np.int32)
        self.ins, self.dels = [], []
        self.remote_del = False
        ix = np.indices(self.dims)
        self.G = [ix[a] + self.g0[a] for a in range(3)]  # global coordinates of every local cell
        self.owned = np.ones(self.dims, bool)
        for a in range(3):
            self.owned &= (ix[a] >= self.olo[a]) & (ix[a] <= self.ohi[a])
        self.last_insert = self.last_delete = 0

    # -- geometry ---------------------------------------------------------------------------------------------
    def shard_info(self):
        return {"local_dims": self.dims, "local_origin": tuple(int(v) for v in self.g0),
                "owned_lo": tuple(int(v) for v in self.olo), "owned_hi": tuple(int(v) for v in self.ohi),
                "global_grid": self.gg}

    def SetParameters(self, *p):
        pass

    def SetOriginalRange(self):
        pass

    # -- occupancy ----------------------------------------------------------------------------------------------
    def SetOccupancy(self, vox, occ, want_ret=False):
        v = np.asarray(vox).reshape(-1, 3) - self.g0
        occ = np.broadcast_to(np.asarray(occ), (len(v),))
        ok = np.all((v >= self.olo) & (v <= self.ohi), axis=1)
        np.add.at(self.cnt_all, tuple(v[ok].T), 1)
        np.add.at(self.cnt_hit, tuple(v[ok].T), occ[ok])

    def UpdateOccupancy(self, global_map=True):
        touched = self.cnt_all > 0
        now = np.where(touched, self.cnt_hit * 2 >= self.cnt_all, self.occ)
        self.coc[touched & (self.coc == UNOBS)] = INF
        for idx in np.argwhere(now & ~self.occ):
            self.ins.append(tuple(idx))
        for idx in np.argwhere(~now & self.occ):
            self.dels.append(tuple(idx))
        self.occ = now
        self.cnt_all[:] = 0
        self.cnt_hit[:] = 0
        self.last_insert, self.last_delete = len(self.ins), len(self.dels)
        return bool(self.ins or self.dels)

    def export_transitions(self):
        out = []
        for idx in self.ins + self.dels:
            g = np.array(idx) + self.g0
            out += [int(g[0]) | (int(g[1]) << 16), int(g[2]) | (0x80000000 if self.occ[idx] else 0)]
        return np.array(out, np.uint32)

    def apply_transitions(self, ent):
        ent = np.asarray(ent, np.uint32).reshape(-1, 2)
        for e0, e in ent:
            x, y, z = int(e0) & 0xFFFF, int(e0) >> 16, int(e) & 0x7FFFFFFF
            alive = bool(int(e) & 0x80000000)
            self.gocc[x, y, z] = alive
            if not alive:
                self.remote_del = True

    # -- ESDF -----------------------------------------------------------------------------------------------------
    def _d2(self, words):
        valid = (words & NOCOC) == 0
        c = words & 0x3FFFFFFF
        dx = self.G[0] - ((c >> 20) & 1023).astype(np.int64)
        dy = self.G[1] - ((c >> 10) & 1023).astype(np.int64)
        dz = self.G[2] - (c & 1023).astype(np.int64)
        return np.where(valid, dx * dx + dy * dy + dz * dz, np.iinfo(np.int64).max), valid

    def esdf_seed(self):
        st = {"inserted": len(self.ins), "deleted": len(self.dels)}
        for idx in self.ins:
            if self.occ[idx]:
                self.coc[idx] = int(pack(*(np.array(idx) + self.g0)))
        if self.dels or self.remote_del:
            w = self.coc
            valid = (w & NOCOC) == 0
            c = w & 0x3FFFFFFF
            alive = self.gocc[(c >> 20) & 1023, (c >> 10) & 1023, c & 1023]
            self.coc[valid & ~alive & self.owned] = INF
        self.ins, self.dels, self.remote_del = [], [], False
        return st

    def relax_pending(self):
        rounds = 0
        while True:
            w = self.coc
            d, _ = self._d2(w)
            observed = w != UNOBS
            best, bestd = w.copy(), d.copy()
            for dx, dy, dz in DIRS:
                src = np.full(self.dims, UNOBS, np.uint32)
                sl_dst = tuple(slice(max(0, -o), s - max(0, o)) for o, s in zip((dx, dy, dz), self.dims))
                sl_src = tuple(slice(max(0, o), s - max(0, -o)) for o, s in zip((dx, dy, dz), self.dims))
                src[sl_dst] = w[sl_src]  # src[v] = word of neighbour v + dir
                cd, cvalid = self._d2(src & ~np.uint32(ACT) | (src & np.uint32(NOCOC)))
                better = cvalid & (cd < bestd)
                best = np.where(better, src & np.uint32(0x3FFFFFFF), best)
                bestd = np.where(better, cd, bestd)
            upd = self.owned & observed & (bestd < d)
            if not upd.any():
                break
            self.coc = np.where(upd, best, w)
            rounds += 1
        st = {"invalidated": 0, "rounds": rounds, "tile_visits": 0, "relax_ms": 0.0}
        return rounds, st

    # -- ghosts -----------------------------------------------------------------------------------------------------
    def halo_pack(self, lo, hi):
        sl = tuple(slice(int(a), int(b) + 1) for a, b in zip(lo, hi))
        return np.ascontiguousarray(self.coc[sl])

    def halo_apply(self, lo, hi, words):
        sl = tuple(slice(int(a), int(b) + 1) for a, b in zip(lo, hi))
        mine = self.coc[sl]
        theirs = np.asarray(words, np.uint32).reshape(mine.shape)
        strip = lambda w: np.where(w == UNOBS, w, w & ~np.uint32(ACT))  # noqa: E731
        diff = strip(mine) != strip(theirs)
        self.coc[sl] = np.where(diff, strip(theirs), mine)
        return int(diff.sum())

    def download_field(self, want=("d2", "coc", "occ")):
        d, valid = self._d2(self.coc)
        d2 = np.where(self.coc == UNOBS, -1, np.where(valid, d, 0x7FFFFFFF)).astype(np.int32)
        c = self.coc & 0x3FFFFFFF
        coc = np.stack([(c >> 20) & 1023, (c >> 10) & 1023, c & 1023], -1).astype(np.int32)
        coc[~valid] = -10000
        return {"d2": d2.reshape(-1), "coc": coc.reshape(-1, 3), "occ": self.occ.astype(np.uint8).reshape(-1)}

    def close(self):
        pass
