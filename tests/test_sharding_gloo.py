"""world_size-2 CPU test (gloo) of the N>1 logic of sws_scale_frames sharding: rank 0 builds the context and
broadcasts the table blob, every rank imports it and owns frames i with i % world == rank.  No compute here
(no GPU); the per-rank compute path is covered by the -m gpu tests."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import librempeg_amd as LA
    from librempeg_amd.multi import broadcast_context, shard_frames
    ctx = LA.SwsContext(1920, 1080, "nv12", 1280, 720, "bgr0", LA.SWS_LANCZOS | LA.SWS_BITEXACT) if rank == 0 else None
    ctx = broadcast_context(ctx, src=0, device="cpu")
    fs, taps, pos = ctx.filter(3)
    mine = shard_frames(11, rank, world)
    geom = (ctx.sw, ctx.sh, ctx.sfmt, ctx.dw, ctx.dh, ctx.dfmt, ctx.fields().src_h)   # the wrapper of an imported context knows its geometry
    q.put((rank, fs, int(taps.astype(np.int64).sum()), int(pos.sum()), ctx.tables()[0], mine, geom))
    dist.barrier()
    dist.destroy_process_group()


def test_context_broadcast_and_frame_sharding_world2():
    port = 29500 + (os.getpid() % 2000)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, fs0, t0, p0, tab0, m0, g0), (r1, fs1, t1, p1, tab1, m1, g1) = res
    assert (fs0, t0, p0, tab0) == (fs1, t1, p1, tab1)       # identical tables on both ranks
    # the wrapper of an imported context knows its geometry (rank 1 sees bgr0's stored alias bgra, utils.c:811-820)
    assert g0 == (1920, 1080, "nv12", 1280, 720, "bgr0", 1080) and g1 == (1920, 1080, "nv12", 1280, 720, "bgra", 1080)
    assert sorted(m0 + m1) == list(range(11)) and not set(m0) & set(m1)  # every frame owned exactly once
    assert m0 == [0, 2, 4, 6, 8, 10] and m1 == [1, 3, 5, 7, 9]


def test_partition_rule_of_sws_scale_frames():
    """sws_hip_plan_shards(): the in-library partition rule, exercised with fake device counts (no GPU needed)."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import librempeg_amd as LA
    L = LA.load_library()

    def plan(src, dst, ndev, home):
        n = len(src)
        out = (C.c_int * n)()
        r = L.sws_hip_plan_shards(n, (C.c_int * n)(*src), (C.c_int * n)(*dst), ndev, home, out)
        return r, list(out)

    # host frames: round-robin over the GPUs starting at the home GPU -> C4's 512 frames are 64 per GPU on 8 GPUs
    r, own = plan([-1] * 512, [-1] * 512, 8, 0)
    assert r == 0 and own[:10] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1] and all(own.count(g) == 64 for g in range(8))
    r, own = plan([-1] * 5, [-1] * 5, 4, 2)
    assert r == 0 and own == [2, 3, 0, 1, 2]
    # HBM-resident frames are converted where they live, whatever the round-robin position
    r, own = plan([3, 3, 1, -1, 0, -1], [3, 3, 1, -1, 0, -1], 4, 0)
    assert r == 0 and own == [3, 3, 1, 0, 0, 1]
    # one side in host memory: the GPU of the other side
    r, own = plan([-1, 2], [5, -1], 8, 0)
    assert r == 0 and own == [5, 2]
    # a frame whose two sides live on different GPUs is refused
    r, own = plan([0, 1], [0, 2], 4, 0)
    assert r < 0
    # single GPU: everything on it
    r, own = plan([-1, 0, -1], [-1, 0, 0], 1, 0)
    assert r == 0 and own == [0, 0, 0]
