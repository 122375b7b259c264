"""CPU tests of the multi-GPU host path: shard layout, exchange plan, and the full driver loop over
torch.distributed with the gloo backend, world_size 2 and 4 -- the same ShardedESDFMap / DistTransport code that runs
over RCCL on MI355X, with a numpy stand-in for the shard engine (tests/numpy_shard.py; the HIP engine has no CPU
fallback).  Check: every rank's owned box equals the exact Euclidean distance transform of the global scene, through
insert, mixed insert+delete (deletes must invalidate on EVERY shard) and a wave that crosses shard faces.
"""
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from fiesta_amd import sharded  # noqa: E402


def test_layout_and_boxes_cover_the_grid():
    for n in (1, 2, 4, 8):
        layout = sharded.shard_layout(n)
        assert int(np.prod(layout)) == n
        boxes = sharded.shard_boxes((50, 41, 33), layout)
        cover = np.zeros((50, 41, 33), int)
        for r, (lo, size) in enumerate(boxes):
            assert sharded.coords_rank(sharded.rank_coords(r, layout), layout) == r
            cover[tuple(slice(a, a + s) for a, s in zip(lo, size))] += 1
        assert np.all(cover == 1)
    with pytest.raises(ValueError):
        sharded.shard_layout(3)
    with pytest.raises(ValueError):
        sharded.shard_boxes((6, 40, 40), (2, 1, 1))


def test_exchange_plan_is_symmetric_and_shapes_match():
    from numpy_shard import NumpyShard
    gg, layout = (24, 20, 28), sharded.shard_layout(8)
    boxes = sharded.shard_boxes(gg, layout)
    infos = [NumpyShard((0, 0, 0), 1.0, tuple(s - 0.5), tuple(lo), gg).shard_info() for lo, s in boxes]
    plans = [sharded.exchange_plan(r, layout, infos[r]) for r in range(8)]
    for r in range(8):
        for a in range(3):
            assert len(plans[r][a]) == 1  # 2x2x2: exactly one neighbour per axis
            peer, slo, shi, rlo, rhi = plans[r][a][0]
            back = [m for m in plans[peer][a] if m[0] == r]
            assert len(back) == 1
            _, pslo, pshi, prlo, prhi = back[0]
            assert tuple(shi - slo) == tuple(prhi - prlo) and tuple(rhi - rlo) == tuple(pshi - pslo)
            # what I send is, in global coordinates, exactly what the peer receives
            g_send = np.array(infos[r]["local_origin"]) + slo
            g_recv = np.array(infos[peer]["local_origin"]) + prlo
            assert np.array_equal(g_send, g_recv)


def _exact_d2(gg, obstacles):
    ix = np.indices(gg).reshape(3, -1).T
    if not len(obstacles):
        return np.full(len(ix), 0x7FFFFFFF, np.int64)
    O = np.asarray(obstacles)
    return ((ix[:, None, :] - O[None, :, :]) ** 2).sum(-1).min(1)


def _scenario(make_map, gg, check):
    m = make_map()
    m.SetParameters(0.7, 0.35, 0.12, 0.97, 0.8)
    m.SetOriginalRange()
    allv = np.indices(gg).reshape(3, -1).T.astype(np.int32)
    m.SetOccupancy(allv, 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    rng = np.random.RandomState(5)
    S = (rng.rand(14, 3) * gg).astype(np.int32)
    S[:4, 0] = gg[0] // 2 - 1 + rng.randint(0, 3, 4)  # hugging the x cut
    m.SetOccupancy(S, 1)
    assert m.UpdateOccupancy(True) and m.last_insert == len(np.unique(S, axis=0))
    m.UpdateESDF()
    check(m, np.unique(S, axis=0), "insert")
    keep = np.unique(S, axis=0)[5:]
    gone = np.unique(S, axis=0)[:5]
    new = np.array([[1, 1, 1], [gg[0] - 2, gg[1] - 2, gg[2] - 2]], np.int32)
    m.SetOccupancy(gone, 0)
    m.SetOccupancy(new, 1)
    m.UpdateOccupancy(True)
    assert m.last_delete == len(gone)
    m.UpdateESDF()
    check(m, np.concatenate([keep, new]), "mixed")
    m.SetOccupancy(np.concatenate([keep, new[1:]]), 0)   # one obstacle left in a corner: its wave crosses every face
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    check(m, new[:1], "corner")
    assert m.last_sweeps >= 2
    return m


def test_driver_with_local_transport_8_numpy_shards():
    from numpy_shard import NumpyShard
    gg = (16, 16, 16)

    def check(m, obstacles, tag):
        f = m.assemble(("d2",))
        assert np.array_equal(f["d2"].astype(np.int64), _exact_d2(gg, obstacles)), tag
    _scenario(lambda: sharded.ShardedESDFMap((0, 0, 0), 1.0, gg, 8, make_shard=lambda r, o, res, s, lo, g: NumpyShard(o, res, s, lo, g)),
              gg, check)


def _worker(rank, world, port, gg, q):
    try:
        import torch.distributed as dist
        from numpy_shard import NumpyShard
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)

        def check(m, obstacles, tag):
            want = _exact_d2(gg, obstacles).reshape(gg)
            for lo, size, crop in m.download_owned(("d2",)).values():
                sl = tuple(slice(int(a), int(a + s)) for a, s in zip(lo, size))
                assert np.array_equal(crop["d2"].astype(np.int64), want[sl]), (tag, rank)
        m = _scenario(lambda: sharded.ShardedESDFMap((0, 0, 0), 1.0, gg, world, transport=sharded.DistTransport(),
                                                     make_shard=lambda r, o, res, s, lo, g: NumpyShard(o, res, s, lo, g)),
                      gg, check)
        q.put((rank, "ok", m.last_sweeps))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world", [2, 4])
def test_driver_over_gloo(world):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    gg = (16, 12, 12)
    procs = [ctx.Process(target=_worker, args=(r, world, port, gg, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), [r for r in results if r[1] != "ok"]
