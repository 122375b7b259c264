"""One ESDF map spread over several GPUs: spatial shards with a ghost-layer exchange.

The reference is single-process; this driver is what replaces its `ESDFMap` when one grid does not fit (or
should not sit on) one GPU.  The grid is cut into a regular shard grid (1 -> 1x1x1, 2 -> 2x1x1, 4 -> 2x2x1,
8 -> 2x2x2: with 8 GPUs every pair of shards is adjacent, matching the fully connected xGMI mesh); each shard is an
ordinary array-mode map (`fiesta_hip_create` with `global_grid` / `shard_lo`) that owns its box plus a 2-voxel ghost
layer -- the stencil radius of the reference's 24 directions (include/parameters.h:54-68).

One `UpdateESDF` of the whole grid:
    seed (insert drain + delete invalidation) on every shard
    exchange ghosts                                   <- neighbours' seeds / resets / newly observed cells
    repeat { relax every pending tile ; exchange ghosts } until no ghost cell changed on ANY shard
The exchange is the classic three-phase halo swap (x, then y including the x ghosts, then z including both), so
edges and corners travel without extra messages: 6 point-to-point messages per shard per sweep, each over its own
xGMI link.  `UpdateOccupancy` additionally all-gathers the occupancy transitions so that every shard's replica of the
global occupancy bitmap can tell whether ANY closest obstacle (it may live on another shard) still exists.

Transports
    LocalTransport   all shards live in this process (any mix of devices) -- tests, and N shards multiplexed on 1 GPU
    DistTransport    one shard per rank over torch.distributed: backend "nccl" (= RCCL over xGMI) with device
                     buffers on MI355X, "gloo" with host buffers on CPU (the CPU tests of the protocol)
The shard engine is duck-typed (`halo_pack/_apply`, `export/apply_transitions`, `esdf_seed`, `relax_pending`, ...):
the product uses fiesta_amd.ESDFMap; the CPU protocol tests plug in a numpy stand-in (tests/numpy_shard.py).
"""
from __future__ import annotations

import numpy as np

LAYOUTS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}
GHOST = 2  # stencil radius


def shard_layout(n_shards: int):
    if n_shards not in LAYOUTS:
        raise ValueError(f"unsupported shard count {n_shards} (1, 2, 4 or 8)")
    return LAYOUTS[n_shards]


def shard_boxes(global_grid, layout):
    """[(lo, size)] in rank order ((i*ly + j)*lz + k); remainders go to the last shard of an axis."""
    cuts = []
    for g, l in zip(global_grid, layout):
        base = g // l
        if base < 2 * GHOST:
            raise ValueError("shards thinner than two ghost layers are not supported")
        cuts.append([(i * base, base if i < l - 1 else g - base * (l - 1)) for i in range(l)])
    out = []
    for i in range(layout[0]):
        for j in range(layout[1]):
            for k in range(layout[2]):
                out.append((np.array([cuts[0][i][0], cuts[1][j][0], cuts[2][k][0]]),
                            np.array([cuts[0][i][1], cuts[1][j][1], cuts[2][k][1]])))
    return out


def rank_coords(rank, layout):
    return (rank // (layout[1] * layout[2]), (rank // layout[2]) % layout[1], rank % layout[2])


def coords_rank(c, layout):
    return (c[0] * layout[1] + c[1]) * layout[2] + c[2]


def exchange_plan(rank, layout, info):
    """Messages of one ghost sweep for `rank`: per axis a list of
    (peer_rank, send_lo, send_hi, recv_lo, recv_hi) in LOCAL array coordinates (inclusive boxes)."""
    c = rank_coords(rank, layout)
    dims, olo, ohi = np.array(info["local_dims"]), np.array(info["owned_lo"]), np.array(info["owned_hi"])
    plan = []
    for a in range(3):
        lo, hi = np.zeros(3, np.int64), np.zeros(3, np.int64)
        for b in range(3):  # axes already swept carry their ghosts along; later axes only their owned extent
            lo[b], hi[b] = (0, dims[b] - 1) if b < a else (olo[b], ohi[b])
        msgs = []
        for side in (-1, +1):
            n = list(c)
            n[a] += side
            if not 0 <= n[a] < layout[a]:
                continue
            slo, shi, rlo, rhi = lo.copy(), hi.copy(), lo.copy(), hi.copy()
            if side < 0:
                slo[a], shi[a] = olo[a], olo[a] + GHOST - 1
                rlo[a], rhi[a] = olo[a] - GHOST, olo[a] - 1
            else:
                slo[a], shi[a] = ohi[a] - GHOST + 1, ohi[a]
                rlo[a], rhi[a] = ohi[a] + 1, ohi[a] + GHOST
            msgs.append((coords_rank(n, layout), slo, shi, rlo, rhi))
        plan.append(msgs)
    return plan


# ---------------------------------------------------------------------------------------------------------
class LocalTransport:
    """All shards in one process: a 'message' is pack on the source shard, apply on the destination shard."""

    def __init__(self, n_shards):
        self.world = n_shards
        self.local_ranks = list(range(n_shards))

    def allreduce_sum(self, value):
        return int(value)

    def gather_transitions(self, per_shard, extra=()):
        """{rank: uint32 array} -> (concatenation of every shard's entries, element-wise sum of `extra` over all ranks)."""
        ent = np.concatenate([np.asarray(v, np.uint32) for v in per_shard.values()]) if per_shard else np.empty(0, np.uint32)
        return ent, [int(v) for v in extra]

    def exchange_axis(self, shards, plans, axis):
        bufs = {}
        for r, sh in shards.items():
            for peer, slo, shi, _, _ in plans[r][axis]:
                bufs[(r, peer)] = sh.halo_pack(slo, shi)
        changed = 0
        for r, sh in shards.items():
            for peer, _, _, rlo, rhi in plans[r][axis]:
                changed += sh.halo_apply(rlo, rhi, bufs[(peer, r)])
        return changed


class DistTransport:
    """One shard per rank over torch.distributed (RCCL on MI355X; gloo on CPU for the protocol tests)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.local_ranks = [self.rank]
        self.device = device if device is not None else torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"

    def broadcast_bytes(self, buf):
        """rank 0's bytes to every rank (the RCCL unique id of the native shard group)."""
        t = self.torch.from_numpy(np.ascontiguousarray(buf, dtype=np.uint8).copy()).to(self.device)
        self.dist.broadcast(t, src=0)
        return t.cpu().numpy()

    def allreduce_sum(self, value):
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather_transitions(self, per_shard, extra=()):
        """One all-reduce carries every rank's entry count AND the caller's scalars (queue sizes): a step of the
        sharded map makes a dozen of these calls, each collective is a fixed ~0.1 ms of latency."""
        torch, dist = self.torch, self.dist
        mine = np.asarray(per_shard[self.rank], np.uint32)
        head = np.zeros(self.world + len(extra), np.int64)
        head[self.rank] = len(mine)
        head[self.world:] = [int(v) for v in extra]
        counts = torch.from_numpy(head).to(self.device)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        counts = counts.cpu().numpy()
        sums = [int(v) for v in counts[self.world:]]
        counts = counts[: self.world]
        cap = int(counts.max())
        if cap == 0:
            return np.empty(0, np.uint32), sums
        buf = torch.zeros(cap, dtype=torch.int32, device=self.device)
        if len(mine):
            buf[: len(mine)] = torch.from_numpy(mine.view(np.int32)).to(self.device)
        out = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(out, buf)
        return np.concatenate([o.cpu().numpy().view(np.uint32)[: counts[r]] for r, o in enumerate(out)]), sums

    def exchange_axis(self, shards, plans, axis):
        torch, dist = self.torch, self.dist
        sh = shards[self.rank]
        ops, recvs, keep = [], [], []
        for peer, slo, shi, rlo, rhi in plans[self.rank][axis]:
            n = int(np.prod(shi - slo + 1))
            send = torch.empty(n, dtype=torch.int32, device=self.device)
            recv = torch.empty(n, dtype=torch.int32, device=self.device)
            if self.on_gpu:
                sh.halo_pack_dev(slo, shi, send.data_ptr())
            else:
                send.copy_(torch.from_numpy(sh.halo_pack(slo, shi).reshape(-1).view(np.int32)))
            ops.append(dist.P2POp(dist.isend, send, peer))
            ops.append(dist.P2POp(dist.irecv, recv, peer))
            recvs.append((rlo, rhi, recv))
            keep.append(send)
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if self.on_gpu:
                torch.cuda.synchronize(self.device)
        changed = 0
        for rlo, rhi, recv in recvs:
            if self.on_gpu:
                changed += sh.halo_apply_dev(rlo, rhi, recv.data_ptr())
            else:
                changed += sh.halo_apply(rlo, rhi, recv.numpy().view(np.uint32))
        return changed


# ---------------------------------------------------------------------------------------------------------
class ShardedESDFMap:
    """`ESDFMap` method names over a shard grid. `make_shard(rank, origin, res, owned_size_m, shard_lo, global_grid)`
    builds one shard engine; the default builds fiesta_amd.ESDFMap on `devices[rank % len(devices)]`."""

    def __init__(self, origin, resolution, global_grid, n_shards, transport=None, devices=(0,), make_shard=None,
                 update_engine=0, native=None, rccl_group_of_one=False):
        # rccl_group_of_one: a single shard whose native group is GIVEN an RCCL communicator (of one rank) and therefore
        # runs the protocol's collectives over it -- how the RCCL entry points are exercised on a one-GPU box
        self._rccl_group_of_one = bool(rccl_group_of_one) and n_shards == 1
        self.origin = np.asarray(origin, np.float64).reshape(3)
        self.resolution = float(resolution)
        self.global_grid = tuple(int(v) for v in global_grid)
        self.layout = shard_layout(n_shards)
        self.boxes = shard_boxes(self.global_grid, self.layout)
        self.transport = transport if transport is not None else LocalTransport(n_shards)
        self.n_shards = n_shards
        default_shards = make_shard is None
        if make_shard is None:
            from .esdf_map import ESDFMap

            def make_shard(rank, origin, res, size_m, lo, gg):
                return ESDFMap(origin, res, size_m, device=devices[rank % len(devices)], update_engine=update_engine,
                               shard_lo=lo, global_grid=gg)
        self.shards, self.plans, self.infos = {}, {}, {}
        for r in self.transport.local_ranks:
            lo, size = self.boxes[r]
            # map_size -> grid is ceil(size_m / res) (src/ESDFMap.cpp:175-176): aim at the middle of the last voxel
            sh = make_shard(r, self.origin, self.resolution, tuple((size - 0.5) * self.resolution), tuple(int(v) for v in lo),
                            self.global_grid)
            info = sh.shard_info()
            assert tuple(np.array(info["owned_hi"]) - np.array(info["owned_lo"]) + 1) == tuple(size), (info, size)
            self.shards[r], self.infos[r] = sh, info
            self.plans[r] = exchange_plan(r, self.layout, info)
        self.last_insert = self.last_delete = 0
        self.last_sweeps = 0
        self.last_entries_sent = 0
        # a wave crosses at most sum(layout) shard faces; anything far beyond that is a protocol bug
        self.max_sweeps = 16 + 4 * sum(self.layout)
        # The protocol runs natively (C++ over RCCL / device copies, fiesta_amd/csrc/shard_group.hip) whenever the shards
        # are real HIP maps reachable from one group: all of them in this process, or one per rank on the GPU transport.
        # The Python loop below is the same protocol spelled out over torch.distributed -- what the CPU (gloo) tests drive
        # with a numpy stand-in shard, and a cross-check for the native engine.
        self._group = None
        self.protocol = "python protocol over the transport (dense slabs; tests and debugging)"
        n_here = len(self.shards)
        can = default_shards and (n_here == n_shards or (n_here == 1 and getattr(self.transport, "on_gpu", False)))
        auto = native is None
        if native == "hosted":
            # the C++ protocol of shard_group.hip with THIS transport's torch.distributed group carrying its messages through
            # host buffers (fiesta_hip_shard_transport): one shard per rank, any back end -- gloo between processes that
            # share one GPU, where RCCL refuses to form a communicator
            if not (default_shards and n_here == 1 and n_shards > 1 and hasattr(self.transport, "dist")):
                raise ValueError("hosted shard group: needs one HIP shard per rank and a torch.distributed transport")
            self._open_hosted_group()
            native = False
        elif auto:
            native = can
        if native:
            if not can:
                raise ValueError("native shard group: needs HIP shards, all local or one per rank on the RCCL transport")
            try:
                self._open_native_group()
                self.protocol = "native C++ shard group, " + ("RCCL" if (n_here == 1 and n_shards > 1) or self._rccl_group_of_one else "all shards in this process")
                failed = 0
            except Exception as e:  # noqa: BLE001  (e.g. the RCCL communicator could not be formed)
                if not auto:
                    raise
                failed, self._group = 1, None
                import warnings
                warnings.warn(f"native shard group unavailable ({e}); the torch.distributed protocol takes over")
            # one shard per rank: every rank must drive the same protocol, or the collectives would not match up
            if auto and n_here == 1 and n_shards > 1 and self.transport.allreduce_sum(failed) and self._group is not None:
                self._glib.fiesta_hip_shard_group_destroy(self._group)
                self._group = None

    def _open_hosted_group(self):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        torch, dist, world = self.transport.torch, self.transport.dist, self.n_shards
        (rank,) = sorted(self.shards)

        def view(ptr, nbytes):
            return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)) if nbytes else np.empty(0, np.uint8)

        AG = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
        EX = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                         C.POINTER(C.c_void_p), C.POINTER(C.c_int64))

        def all_gather(_ctx, send, recv, nbytes):
            try:
                t = torch.from_numpy(view(send, nbytes).copy())
                out = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(out, t)
                view(recv, nbytes * world)[:] = np.concatenate([o.numpy() for o in out])
                return 0
            except Exception:  # noqa: BLE001  (nothing may propagate through the C frames)
                import traceback
                traceback.print_exc()
                return 1

        def exchange(_ctx, n, peers, send, sbytes, recv, rbytes):
            try:
                ops, landed, keep = [], [], []
                for k in range(n):
                    if sbytes[k]:
                        t = torch.from_numpy(view(send[k], sbytes[k]).copy())
                        keep.append(t)
                        ops.append(dist.P2POp(dist.isend, t, int(peers[k])))
                    if rbytes[k]:
                        r = torch.empty(int(rbytes[k]), dtype=torch.uint8)
                        landed.append((k, r))
                        ops.append(dist.P2POp(dist.irecv, r, int(peers[k])))
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                for k, r in landed:
                    view(recv[k], rbytes[k])[:] = r.numpy()
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        class Transport(C.Structure):
            _fields_ = [("ctx", C.c_void_p), ("all_gather", AG), ("exchange", EX)]

        self._hosted_cbs = (AG(all_gather), EX(exchange))     # (kept alive as long as the group)
        self._hosted_struct = Transport(None, *self._hosted_cbs)
        g = C.c_void_p()
        _lib.check(lib.fiesta_hip_shard_group_create_hosted(self.shards[rank]._h, rank, world, C.byref(self._hosted_struct), C.byref(g)))
        self._group, self._glib, self._check = g, lib, _lib.check
        self.protocol = "native C++ shard group over a hosted transport (torch.distributed through host buffers)"

    def _open_native_group(self):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        ranks = sorted(self.shards)
        rccl_id = None
        handles = (C.c_void_p * len(ranks))(*[self.shards[r]._h for r in ranks])
        rk = (C.c_int32 * len(ranks))(*ranks)
        use_rccl = (self.n_shards > 1 and len(ranks) == 1) or self._rccl_group_of_one
        # local preconditions first, agreed on by every rank BEFORE the collective communicator set-up: a rank that
        # fails here must not leave the others blocked inside ncclCommInitRank
        bad = int(lib.fiesta_hip_shard_group_precheck(handles, rk, len(ranks), self.n_shards, int(use_rccl)) != 0)
        why = _lib.last_error() if bad else ""
        if use_rccl and self.n_shards > 1:
            bad = int(self.transport.allreduce_sum(bad))
        if bad:
            raise RuntimeError(f"native shard group: local preconditions failed on {bad} rank(s) {why}")
        if use_rccl:
            buf = np.zeros(129, np.uint8)  # 128 bytes of id + "rank 0 got one" (every rank takes part in the broadcast)
            if self.n_shards == 1 or self.transport.rank == 0:
                buf[128] = lib.fiesta_hip_rccl_unique_id(buf.ctypes.data_as(C.c_void_p)) == 0
            got = buf if self.n_shards == 1 else self.transport.broadcast_bytes(buf)
            if not got[128]:
                raise RuntimeError("rank 0 could not create an RCCL unique id (librccl.so not loadable?)")
            rccl_id = np.ascontiguousarray(got[:128])
        g = C.c_void_p()
        idp = rccl_id.ctypes.data_as(C.c_void_p) if rccl_id is not None else None
        _lib.check(lib.fiesta_hip_shard_group_create(handles, rk, len(ranks), self.n_shards, idp, C.byref(g)))
        self._group, self._glib, self._check = g, lib, _lib.check

    def comm_info(self):
        """(ranks the RCCL communicator itself reports -- 0 on the in-process transport --, this process's rank in it)."""
        import ctypes as C
        if self._group is None:
            return 0, 0
        n, r = C.c_int32(0), C.c_int32(0)
        self._check(self._glib.fiesta_hip_shard_group_comm_info(self._group, C.byref(n), C.byref(r)))
        return n.value, r.value

    # -- plumbing ---------------------------------------------------------------------------------------------
    def _owner_masks(self, vox):
        for r in self.shards:
            lo, size = self.boxes[r]
            yield r, np.all((vox >= lo) & (vox < lo + size), axis=1)

    def close(self):
        if self._group is not None:
            self._glib.fiesta_hip_shard_group_destroy(self._group)
            self._group = None
        for sh in self.shards.values():
            if hasattr(sh, "close"):
                sh.close()
        self.shards = {}

    # -- reference API ------------------------------------------------------------------------------------------
    def SetParameters(self, *p):
        for sh in self.shards.values():
            sh.SetParameters(*p)

    def SetOriginalRange(self):
        for sh in self.shards.values():
            sh.SetOriginalRange()

    def SetOccupancy(self, vox, occ):
        """SetOccupancy(Vector3i, occ), global voxel coordinates; each observation goes to its owner shard."""
        vox = np.ascontiguousarray(vox, dtype=np.int32).reshape(-1, 3)
        occ = np.ascontiguousarray(np.broadcast_to(np.asarray(occ, dtype=np.int32), (len(vox),)))
        for r, m in self._owner_masks(vox):
            if m.any():
                self.shards[r].SetOccupancy(vox[m], occ[m], want_ret=False)

    def SetOccupancyBox(self, lo, hi, occ):
        for sh in self.shards.values():
            sh.SetOccupancyBox(lo, hi, occ)  # the device kernel keeps only owned voxels

    def UpdateOccupancy(self, global_map=True):
        if self._group is not None:
            import ctypes as C
            ni, nd, any_ = C.c_int64(0), C.c_int64(0), C.c_int32(0)
            self._check(self._glib.fiesta_hip_shard_group_update_occupancy(self._group, int(bool(global_map)), C.byref(ni),
                                                                           C.byref(nd), C.byref(any_)))
            self.last_insert, self.last_delete = ni.value, nd.value
            return bool(any_.value)
        any_local, ni, nd = False, 0, 0
        for sh in self.shards.values():
            any_local |= bool(sh.UpdateOccupancy(global_map))
            ni += sh.last_insert
            nd += sh.last_delete
        ent, (self.last_insert, self.last_delete) = self.transport.gather_transitions(
            {r: sh.export_transitions() for r, sh in self.shards.items()}, extra=(ni, nd))
        if len(ent):
            for sh in self.shards.values():
                sh.apply_transitions(ent)
        return self.last_insert + self.last_delete > 0

    def _sweep(self):
        return sum(self.transport.exchange_axis(self.shards, self.plans, a) for a in range(3))

    def UpdateESDF(self):
        if self._group is not None:
            import ctypes as C
            from ._lib import Stats
            st, sweeps, sent = Stats(), C.c_int32(0), C.c_int64(0)
            self._check(self._glib.fiesta_hip_shard_group_update_esdf(self._group, C.byref(st), C.byref(sweeps), C.byref(sent)))
            d = st.as_dict()
            d["sweeps"] = self.last_sweeps = sweeps.value
            d["halo_entries_sent"] = self.last_entries_sent = sent.value
            return d
        stats = {"inserted": 0, "deleted": 0, "invalidated": 0, "rounds": 0, "tile_visits": 0, "relax_ms": 0.0,
                 "sweeps": 0}
        for sh in self.shards.values():
            st = sh.esdf_seed()
            stats["inserted"] += st["inserted"]
            stats["deleted"] += st["deleted"]
        self._sweep()
        sweeps = 1
        while True:
            for sh in self.shards.values():
                _, st = sh.relax_pending()
                for k in ("invalidated", "rounds", "tile_visits", "relax_ms"):
                    stats[k] += st[k]
            changed = self.transport.allreduce_sum(self._sweep())
            sweeps += 1
            if changed == 0:
                break
            if sweeps > self.max_sweeps:
                raise RuntimeError(f"ghost exchange did not converge after {sweeps} sweeps ({changed} cells still changing)")
        stats["sweeps"] = self.last_sweeps = sweeps
        return stats

    # -- queries / whole field ------------------------------------------------------------------------------------
    def GetDistance(self, vox):
        vox = np.ascontiguousarray(vox, dtype=np.int32).reshape(-1, 3)
        out = np.full(len(vox), np.nan)
        for r, m in self._owner_masks(vox):
            if m.any():
                out[m] = self.shards[r].GetDistance(vox[m])
        return out

    def download_owned(self, want=("d2", "coc", "occ")):
        """{rank: (lo, size, fields cropped to the owned box)} for the shards of this process."""
        out = {}
        for r, sh in self.shards.items():
            info = self.infos[r]
            f = sh.download_field(want)
            dims = info["local_dims"]
            sl = tuple(slice(a, b + 1) for a, b in zip(info["owned_lo"], info["owned_hi"]))
            crop = {}
            for k in want:
                a = f[k].reshape(dims + ((3,) if k == "coc" else ()))
                crop[k] = np.ascontiguousarray(a[sl])
            out[r] = (self.boxes[r][0], self.boxes[r][1], crop)
        return out

    def assemble(self, want=("d2", "coc", "occ")):
        """Global dense field (only valid with a LocalTransport: every shard lives here)."""
        gg = self.global_grid
        full = {k: np.zeros(gg + ((3,) if k == "coc" else ()), np.int32 if k != "occ" else np.uint8) for k in want}
        for lo, size, crop in self.download_owned(want).values():
            sl = tuple(slice(int(a), int(a + s)) for a, s in zip(lo, size))
            for k in want:
                full[k][sl] = crop[k]
        return {k: v.reshape((-1, 3) if k == "coc" else (-1,)) for k, v in full.items()}
