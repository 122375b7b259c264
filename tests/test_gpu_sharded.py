"""One grid cut into 1/2/4/8 shards (multiplexed on the one visible GPU through LocalTransport -- the same shard
engine, halo kernels and driver loop the RCCL transport uses) against the single-grid CPU oracle
(SURVEY.md 4 (iv), 8e).  Contract as everywhere: d^2 / occupancy bit-exact, closest obstacle tie-equivalent.
"""
import numpy as np
import pytest

from scenarios import D2_INF, P_DEFAULT, oracle_d2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("engine")]


def drive(sharded, cpu, vox_occ_cycles):
    for occ_vox, free_vox, cycles in vox_occ_cycles:
        for _ in range(cycles):
            for v, o in ((occ_vox, 1), (free_vox, 0)):
                if len(v):
                    sharded.SetOccupancy(v, o)
                    cpu.SetOccupancyVox(v, o)
            a, b = sharded.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
            assert a == b and (sharded.last_insert, sharded.last_delete) == (cpu.last_insert, cpu.last_delete)
        sg, sc = sharded.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])


def compare(sharded, cpu, gs):
    f = sharded.assemble()
    o = cpu.dump_dense()
    od2, vox = oracle_d2(o, gs)
    assert np.array_equal(f["occ"], o["occ"])
    gd2 = f["d2"].astype(np.int64)
    assert np.array_equal(gd2 < 0, od2 < 0)
    assert int((gd2 != od2).sum()) == 0, np.flatnonzero(gd2 != od2)[:10]
    have = (gd2 >= 0) & (gd2 != D2_INF)
    gc = f["coc"].astype(np.int64)
    gi = (gc[have, 0] * gs[1] + gc[have, 1]) * gs[2] + gc[have, 2]
    assert np.all(f["occ"][gi] == 1)
    assert np.array_equal(((vox[have] - gc[have]) ** 2).sum(-1), gd2[have])


@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
@pytest.mark.parametrize("n_shards", [1, 2, 4, 8])
def test_sharded_matches_single_grid_oracle(hip_lib, oracle_libs, best_oracle_kind, n_shards, native):
    """native: the C++ protocol engine (shard_group.hip: sparse changed-entry halos, all 26 neighbours in one phase; the
    RCCL transport differs from this one only in how a message moves).  python: the same protocol spelled out in
    fiesta_amd/sharded.py (dense three-phase slabs), the form the CPU gloo tests drive."""
    from fiesta_amd.sharded import ShardedESDFMap
    gs, res = (72, 64, 80), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, n_shards, native=native)
    assert (sm._group is not None) == native
    cpu = oracle_libs.OracleMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res), kind=best_oracle_kind)
    assert cpu.grid_size == gs
    for m in (sm, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    allv = np.stack(np.meshgrid(*[np.arange(n) for n in gs], indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    rng = np.random.RandomState(3)
    S = (rng.rand(500, 3) * gs).astype(np.int32)
    # obstacles hugging the cuts: shard faces are at 36 / 32 / 40
    S[:60, 0] = rng.randint(34, 38, 60)
    S[60:120, 1] = rng.randint(30, 34, 60)
    S[120:180, 2] = rng.randint(38, 42, 60)
    drive(sm, cpu, [([], allv, 1), (S, [], 3)])
    compare(sm, cpu, gs)
    drive(sm, cpu, [((rng.rand(150, 3) * gs).astype(np.int32), S[:250], 6)])
    compare(sm, cpu, gs)
    occ = np.argwhere(cpu.dump_dense(("occ",))["occ"].reshape(gs) == 1).astype(np.int32)
    drive(sm, cpu, [([], occ, 6)])
    compare(sm, cpu, gs)
    sm.close()


@pytest.mark.parametrize("engine", ["rounds", "auto"])
def test_group_of_one_runs_its_protocol_over_an_rccl_communicator(hip_lib, oracle_libs, best_oracle_kind, engine):
    """The only RCCL a one-GPU box can run: a shard group of ONE that is given a communicator (ncclGetUniqueId,
    ncclCommInitRank with one rank) uses it -- the per-sweep row all-gather, the transition all-gather, the (empty)
    send/receive group -- instead of the shortcuts of a group on the local transport.  Same fields as the oracle, and the
    communicator itself reports one rank."""
    import torch  # noqa: F401  (first: the library then finds torch's librccl loaded instead of opening ROCm's own -- a later
    #                            `import torch` in this process would bring a SECOND copy, and the two abort at exit)
    from fiesta_amd.sharded import ShardedESDFMap
    gs, res = (72, 64, 80), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, 1, native=True, rccl_group_of_one=True, update_engine=engine)
    assert sm._group is not None and sm.comm_info() == (1, 0)
    cpu = oracle_libs.OracleMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res), kind=best_oracle_kind)
    for m in (sm, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    allv = np.stack(np.meshgrid(*[np.arange(n) for n in gs], indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    rng = np.random.RandomState(11)
    S = (rng.rand(400, 3) * gs).astype(np.int32)
    drive(sm, cpu, [([], allv, 1), (S, [], 3)])
    compare(sm, cpu, gs)
    drive(sm, cpu, [((rng.rand(100, 3) * gs).astype(np.int32), S[:200], 6)])
    compare(sm, cpu, gs)
    sm.close()


@pytest.mark.parametrize("n_shards", [2, 8])
def test_bulk_then_rounds_updates_next_to_a_cut(hip_lib, oracle_libs, best_oracle_kind, n_shards):
    """ADVICE r2 (shard_group.hip): a committed bulk transform rewrites owned and ghost cells behind the ghost exchange's
    back; the per-link "last sent" shadows must not survive it.  Sequence: frontier update inserting an obstacle A next to
    a cut (shadows = field with A), bulk update that deletes A, small frontier update that re-inserts A -- the owner
    relaxes straight back to the words it sent last, and only a forgotten shadow makes it resend them.  The engine is
    switched per update (fiesta_hip_set_update_engine): large deltas bulk, single-voxel ones the rounds -- what the
    engine choice does by itself on a large map."""
    from fiesta_amd.sharded import ShardedESDFMap
    gs, res = (64, 64, 64), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, n_shards)
    cpu = oracle_libs.OracleMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res), kind=best_oracle_kind)
    for m in (sm, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    allv = np.stack(np.meshgrid(*[np.arange(n) for n in gs], indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    rng = np.random.RandomState(5)
    A = np.array([[30, 20, 20]], np.int32)                 # two voxels from the cut at x = 32
    far = np.unique((rng.rand(90, 3) * gs).astype(np.int32), axis=0)
    far = far[np.abs(far[:, 0] - 32) > 14][:60]            # A's cell reaches well across the cut
    none = np.zeros((0, 3), np.int32)

    def step(occ, free, want_bulk):
        for sh in sm.shards.values():          # (on a map this small the cost model would always pick the transform)
            sh.set_update_engine("bulk" if want_bulk else "rounds")
        for _ in range(6 if len(free) else 3):
            for v, o in ((occ, 1), (free, 0)):
                if len(v):
                    sm.SetOccupancy(v, o)
                    cpu.SetOccupancyVox(v, o)
            assert sm.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
        sg, sc = sm.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
        assert bool(sg["bulk"]) == want_bulk, sg
        compare(sm, cpu, gs)
    drive(sm, cpu, [(none, allv, 1)])
    step(far, none, True)              # bulk: the whole obstacle set at once
    step(A, none, False)               # rounds: boundary words sent, shadows = the field with A
    step(far[:20] + 1, np.concatenate([A, far[:10]]), True)    # bulk: deletes A, rewrites the field on every shard
    step(A, none, False)               # rounds: the owner returns to the words it sent two updates ago
    step(none, A, False)               # rounds: ... and leaves them again
    step(far[:10], far[20:50], True)   # bulk once more, then a frontier update on top of it
    step(A, none, False)
    sm.close()


def test_single_obstacle_wave_crosses_every_shard(hip_lib, oracle_libs, best_oracle_kind, engine):
    """The adversarial case of SURVEY.md 8e: one obstacle in a corner, its wave must cross all 8 shards."""
    from fiesta_amd.sharded import ShardedESDFMap
    gs, res = (64, 64, 64), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, 8)
    cpu = oracle_libs.OracleMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res), kind=best_oracle_kind)
    for m in (sm, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    allv = np.stack(np.meshgrid(*[np.arange(n) for n in gs], indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    drive(sm, cpu, [([], allv, 1), (np.array([[1, 2, 3]], np.int32), [], 3)])
    compare(sm, cpu, gs)
    if engine == "rounds":
        assert sm.last_sweeps >= 3   # the wave needs a ghost exchange per shard face it crosses
        assert sm.last_entries_sent > 0
    drive(sm, cpu, [(np.array([[60, 61, 59]], np.int32), np.array([[1, 2, 3]], np.int32), 6)])
    compare(sm, cpu, gs)
    sm.close()


def test_device_pointer_halo_path_equals_host_path(hip_lib):
    """The RCCL transport hands device buffers to halo_pack_dev / halo_apply_dev; same result as the host forms."""
    import torch  # noqa: F401  (device buffers)
    from fiesta_amd.sharded import ShardedESDFMap
    gs, res = (40, 24, 36), 0.1
    maps = [ShardedESDFMap((0, 0, 0), res, gs, 2, native=False) for _ in range(2)]
    rng = np.random.RandomState(1)
    S = (rng.rand(80, 3) * gs).astype(np.int32)
    for sm in maps:
        sm.SetParameters(*P_DEFAULT)
        sm.SetOriginalRange()
        sm.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
        sm.UpdateOccupancy(True)
        for _ in range(3):
            sm.SetOccupancy(S, 1)
            sm.UpdateOccupancy(True)
        for sh in sm.shards.values():
            sh.esdf_seed()
    a, b = maps
    dev = torch.device("cuda", 0)
    changed = [0, 0]
    for r in (0, 1):
        for peer, slo, shi, rlo, rhi in a.plans[r][0]:
            # host path on map a
            changed[0] += a.shards[peer].halo_apply(a.plans[peer][0][0][3], a.plans[peer][0][0][4], a.shards[r].halo_pack(slo, shi))
            # device path on map b
            n = int(np.prod(shi - slo + 1))
            buf = torch.empty(n, dtype=torch.int32, device=dev)
            b.shards[r].halo_pack_dev(slo, shi, buf.data_ptr())
            torch.cuda.synchronize()
            changed[1] += b.shards[peer].halo_apply_dev(b.plans[peer][0][0][3], b.plans[peer][0][0][4], buf.data_ptr())
    assert changed[0] == changed[1] > 0
    for r in (0, 1):
        fa, fb = a.shards[r].download_field(("d2", "coc")), b.shards[r].download_field(("d2", "coc"))
        assert np.array_equal(fa["d2"], fb["d2"]) and np.array_equal(fa["coc"], fb["coc"])
    for sm in maps:
        sm.close()


def _dist_worker(rank, world, port, gs, q):
    try:
        import torch  # noqa: F401  (before the HIP library: one HIP runtime per process)
        import torch.distributed as dist
        import os
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        sys.path.insert(0, os.path.dirname(here))
        from fiesta_amd.sharded import DistTransport, ShardedESDFMap
        import fiesta_amd.esdf_map as em
        em.DEFAULT_UPDATE_ENGINE = int(os.environ.get("FIESTA_TEST_UPDATE_ENGINE", "0"))  # (the `engine` fixture of the parent)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sm = ShardedESDFMap((0, 0, 0), 0.1, gs, world, transport=DistTransport(), devices=(0,))
        sm.SetParameters(*P_DEFAULT)
        sm.SetOriginalRange()
        sm.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
        sm.UpdateOccupancy(True)
        sm.UpdateESDF()
        rng = np.random.RandomState(11)
        S = np.unique((rng.rand(120, 3) * gs).astype(np.int32), axis=0)
        S[:20, 0] = gs[0] // 2 - 2 + rng.randint(0, 4, 20)
        S = np.unique(S, axis=0)

        def check(obstacles, tag):
            ix = np.indices(gs).reshape(3, -1).T
            want = ((ix[:, None, :] - obstacles[None, :, :]) ** 2).sum(-1).min(1).reshape(gs)
            for lo, size, crop in sm.download_owned(("d2",)).values():
                sl = tuple(slice(int(a), int(a + s)) for a, s in zip(lo, size))
                assert np.array_equal(crop["d2"].astype(np.int64), want[sl]), (tag, rank)
        for _ in range(3):
            sm.SetOccupancy(S, 1)
            sm.UpdateOccupancy(True)
        assert sm.last_insert == len(S)
        sm.UpdateESDF()
        check(S, "insert")
        gone, new = S[:40], np.array([[1, 1, 1], [gs[0] - 2, gs[1] - 2, gs[2] - 2]], np.int32)
        for _ in range(6):
            sm.SetOccupancy(gone, 0)
            sm.SetOccupancy(new, 1)
            sm.UpdateOccupancy(True)
        assert sm.last_delete == len(gone)
        sm.UpdateESDF()
        check(np.concatenate([S[40:], new]), "mixed")
        q.put((rank, "ok", sm.last_sweeps))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


def test_dist_transport_with_hip_shards_two_ranks_one_gpu(hip_lib):
    """The torch.distributed driver path (what `bench.py --gpus N` runs over RCCL) with real HIP shards: two ranks
    share the one visible GPU, gloo carries the host-buffer forms of the same messages. Checked against brute force."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    gs = (48, 40, 40)
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, gs, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), [r for r in results if r[1] != "ok"]


def _hosted_worker(rank, world, port, gs, q):
    try:
        import torch  # noqa: F401  (before the HIP library: one HIP runtime per process)
        import torch.distributed as dist
        import os
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        sys.path.insert(0, os.path.dirname(here))
        from fiesta_amd.sharded import DistTransport, ShardedESDFMap
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sm = ShardedESDFMap((0, 0, 0), 0.1, gs, world, transport=DistTransport(), devices=(0,), native="hosted")
        assert sm._group is not None and "hosted" in sm.protocol
        (mine,) = sm.shards.values()
        sm.SetParameters(*P_DEFAULT)
        sm.SetOriginalRange()
        sm.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
        sm.UpdateOccupancy(True)
        sm.UpdateESDF()
        rng = np.random.RandomState(11)
        S = np.unique((rng.rand(160, 3) * gs).astype(np.int32), axis=0)
        S[:30, 0] = gs[0] // 2 - 2 + rng.randint(0, 4, 30)        # a cluster on the x cut
        S[30:50, 1] = gs[1] // 2 - 2 + rng.randint(0, 4, 20)      # ... and on the y cut (4 shards)
        S = np.unique(S, axis=0)

        def check(obstacles, tag):
            ix = np.indices(gs).reshape(3, -1).T
            want = ((ix[:, None, :] - obstacles[None, :, :]) ** 2).sum(-1).min(1).reshape(gs)
            for lo, size, crop in sm.download_owned(("d2",)).values():
                sl = tuple(slice(int(a), int(a + s)) for a, s in zip(lo, size))
                assert np.array_equal(crop["d2"].astype(np.int64), want[sl]), (tag, rank)
        # 1. the bulk path of the shard group (every shard eligible, fully observed): decided from the gathered table
        for _ in range(3):
            sm.SetOccupancy(S, 1)
            sm.UpdateOccupancy(True)
        assert sm.last_insert == len(S)
        st = sm.UpdateESDF()
        check(S, "insert / bulk")
        # 2. the frontier rounds with ghost sweeps: deletes and inserts next to the cuts, engine switched on the live maps
        mine.set_update_engine("rounds")
        gone = S[:60]
        new = np.array([[gs[0] // 2, 3, 3], [gs[0] // 2 - 1, gs[1] // 2, 5], [1, 1, 1], [gs[0] - 2, gs[1] - 2, gs[2] - 2]], np.int32)
        for _ in range(6):
            sm.SetOccupancy(gone, 0)
            sm.SetOccupancy(new, 1)
            sm.UpdateOccupancy(True)
        assert sm.last_delete == len(gone)
        sm.UpdateESDF()
        sweeps = sm.last_sweeps
        live = np.concatenate([S[60:], new])
        check(live, "mixed / rounds")
        # 3. back to the library's choice, a small delta at the cut
        mine.set_update_engine("auto")
        more = np.array([[gs[0] // 2 - 1, 7, 7], [gs[0] // 2, gs[1] - 3, 9]], np.int32)
        for _ in range(3):
            sm.SetOccupancy(more, 1)
            sm.UpdateOccupancy(True)
        sm.UpdateESDF()
        check(np.concatenate([live, more]), "small delta / auto")
        q.put((rank, "ok", (int(bool(st.get("bulk"))), sweeps, sm.last_entries_sent)))
        dist.barrier()
        sm.close()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cpp_protocol_across_processes_on_a_hosted_transport(hip_lib, world):
    """the C++ sweep loop, sparse diff / apply and convergence test of shard_group.hip -- the code
    `bench.py --gpus N` runs over RCCL -- ACROSS PROCESSES: 2, 4 and 8 processes share the one visible GPU (where RCCL refuses
    a communicator), each owns one shard, the group's all-gathers and neighbour exchanges travel through
    fiesta_hip_shard_transport bound to torch.distributed / gloo.  The sharded bulk path (decided from the gathered table),
    the frontier rounds with ghost sweeps at the cuts (2 x 1 x 1, 2 x 2 x 1 and config 5's 2 x 2 x 2: faces, edges and the corner
    where eight shards meet) and a small delta under the library's own
    engine choice, each against brute force on every owned voxel."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    gs = (48, 40, 32)
    procs = [ctx.Process(target=_hosted_worker, args=(r, world, port, gs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), [r for r in results if r[1] != "ok"]
    assert all(r[2][0] == 1 for r in results), "the first update must have gone through the sharded bulk path"
    assert all(r[2][1] >= 1 for r in results), "the rounds update must have needed ghost sweeps"


def _brute_d2(vox, obs):
    out = np.empty(len(vox), np.int64)
    for s in range(0, len(vox), 4096):
        out[s:s + 4096] = ((vox[s:s + 4096, None, :].astype(np.int64) - obs[None, :, :]) ** 2).sum(-1).min(1)
    return out


@pytest.mark.parametrize("n_shards", [1, 2])
def test_global_extent_2048_ids_beyond_30_bits(hip_lib, n_shards):
    """BASELINE config 5 needs a 2048-voxel axis: closest-obstacle ids are coordinates modulo 1024, decoded relative to
    the voxel that holds them on grids larger than 1024 (fiesta_amd/csrc/common.hpp: pack_coc).  A 2048 x 64 x 64 grid,
    unsharded and cut into 2 shards of 1024 x 64 x 64 (the reference cannot even index such a map along one axis of
    C5's size with its 48 B/voxel): fully observed, obstacles scattered along the whole length incl. around x = 1024
    (the cut, and the wrap of the id) -- every squared distance and every closest obstacle against brute force."""
    from fiesta_amd.sharded import ShardedESDFMap
    gs, res = (2048, 64, 64), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, n_shards)
    sm.SetParameters(*P_DEFAULT)
    sm.SetOriginalRange()
    sm.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
    sm.UpdateOccupancy(True)
    sm.UpdateESDF()
    rng = np.random.RandomState(9)
    S = (rng.rand(900, 3) * gs).astype(np.int32)
    S[:80, 0] = rng.randint(1015, 1034, 80)
    S[80:100, 0] = rng.randint(0, 6, 20)
    S[100:120, 0] = rng.randint(2042, 2048, 20)
    S = np.unique(S, axis=0)
    for _ in range(3):
        sm.SetOccupancy(S, 1)
        sm.UpdateOccupancy(True)
    st = sm.UpdateESDF()
    assert st["inserted"] == len(S)

    def check(obs):
        f = sm.assemble()
        assert int(f["occ"].sum()) == len(obs)
        idx = np.random.RandomState(2).randint(0, gs[0] * gs[1] * gs[2], 60000).astype(np.int64)
        idx = np.concatenate([idx, (np.arange(1000, 1048)[:, None] * gs[1] * gs[2] + np.arange(0, gs[1] * gs[2], 37)[None, :]).reshape(-1)])
        V = np.stack([idx // (gs[1] * gs[2]), (idx // gs[2]) % gs[1], idx % gs[2]], -1)
        want = _brute_d2(V, obs.astype(np.int64))
        got = f["d2"][idx].astype(np.int64)
        assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
        c = f["coc"][idx].astype(np.int64)
        assert np.array_equal(((V - c) ** 2).sum(-1), want)           # the decoded obstacle sits at that distance ...
        lin = (c[:, 0] * gs[1] + c[:, 1]) * gs[2] + c[:, 2]
        assert np.all(f["occ"][lin] == 1)                             # ... and is occupied
    check(S)
    gone = S[::2]
    new = (rng.rand(200, 3) * gs).astype(np.int32)
    for _ in range(6):
        sm.SetOccupancy(new, 1)
        sm.SetOccupancy(gone, 0)
        sm.UpdateOccupancy(True)
    sm.UpdateESDF()
    live = np.unique(np.concatenate([S[1::2], new]), axis=0)
    check(live)
    sm.close()


@pytest.mark.parametrize("n_shards", [1, 2])
def test_single_obstacle_in_a_2048_long_map_and_the_count_of_truncated_distances(hip_lib, engine, n_shards):
    """SURVEY.md 8e's adversarial case at config 5's axis length: ONE obstacle in an otherwise empty, fully observed
    2048 x 64 x 64 map (unsharded and as 2 shards of 1024): its wave has to cross the cut.  Within the reach of an id
    (512 voxels, common.hpp) every distance equals the closed form -- what the reference would hold; beyond it a voxel
    reads "no obstacle", and fiesta_hip_count_no_obstacle reports exactly how many do (the deviation is loud, not silent)."""
    from fiesta_amd.sharded import ShardedESDFMap
    gs, res = (2048, 64, 64), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, n_shards)
    sm.SetParameters(*P_DEFAULT)
    sm.SetOriginalRange()
    sm.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
    sm.UpdateOccupancy(True)
    sm.UpdateESDF()
    assert sum(sh.count_no_obstacle() for sh in sm.shards.values()) == gs[0] * gs[1] * gs[2]   # an empty map: everything
    obst = np.array([[900, 30, 31]], np.int32)               # 124 voxels from the cut at x = 1024
    for _ in range(3):
        sm.SetOccupancy(obst, 1)
        sm.UpdateOccupancy(True)
    sm.UpdateESDF()
    f = sm.assemble(("d2",))["d2"].reshape(gs).astype(np.int64)
    x, y, z = np.meshgrid(*[np.arange(d) for d in gs], indexing="ij", sparse=True)
    e = (x - 900) ** 2 + (y - 30) ** 2 + (z - 31) ** 2
    want = np.where(e < (1 << 18), e, D2_INF)
    assert np.array_equal(f, want)
    assert (want[1024:1400] < D2_INF).all() and (want[1500:] == D2_INF).all()               # across the cut, then out of reach
    assert sum(sh.count_no_obstacle() for sh in sm.shards.values()) == int((want == D2_INF).sum())
    sm.close()


def test_wrap_ids_reach_512_voxels(hip_lib):
    """On a grid larger than 1024 the reach of an id is 512 voxels (d^2 < 2^18): a voxel farther than that from every
    obstacle reads "no obstacle" (documented limit; the reference cannot hold such a grid)."""
    import fiesta_amd
    gs, res = (1400, 16, 32), 0.1
    m = fiesta_amd.ESDFMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res))
    assert m.grid_size == gs
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    m.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    for _ in range(3):
        m.SetOccupancy(np.array([[10, 8, 16]], np.int32), 1)
        m.UpdateOccupancy(True)
    m.UpdateESDF()
    f = m.download_field(("d2",))["d2"].reshape(gs)
    assert f[10 + 400, 8, 16] == 400 ** 2 and f[10 + 511, 8, 16] == 511 ** 2
    assert f[10 + 512, 8, 16] == D2_INF and f[1399, 0, 0] == D2_INF
    m.close()
