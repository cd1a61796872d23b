"""The N>1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Views are dealt round-robin,
each rank produces its frames, the gather returns them in view order on every rank, timing is the max."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, n_views, q):
    sys.path.insert(0, ROOT)
    from pixelsynth_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        mine = D.shard_views(n_views)
        v_local = (n_views + world_size - 1) // world_size
        frames = torch.full((v_local, 3, 4, 4), -1.0)
        codes = torch.full((v_local, 2, 2), -1, dtype=torch.int32)
        for j, v in enumerate(mine):          # "render" view v: its index is the payload
            frames[j] = float(v)
            codes[j] = v
        D.barrier()
        all_frames = D.gather_frames(frames, n_views)
        all_codes = D.gather_frames(codes, n_views)
        # the asynchronous form bench.py pipelines (two collectives in flight, collected later, in any order): the same rows
        pend_f, pend_c = D.gather_frames_start(frames, n_views), D.gather_frames_start(codes, n_views)
        late_c, late_f = pend_c.result(), pend_f.result()
        assert torch.equal(late_f, all_frames) and torch.equal(late_c, all_codes)
        ok = all(bool((all_frames[v] == float(v)).all()) and bool((all_codes[v] == v).all()) for v in range(n_views))
        t = D.max_over_ranks(1.0 + rank)
        q.put((rank, mine, ok, tuple(all_frames.shape), t))
    finally:
        dist.destroy_process_group()


def _run(world_size, n_views):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world_size, port, n_views, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_two_ranks_round_robin_gather():
    res = _run(2, 16)
    seen = sorted(v for _, mine, _, _, _ in res for v in mine)
    assert seen == list(range(16))                      # every view rendered exactly once
    assert res[0][1] == list(range(0, 16, 2)) and res[1][1] == list(range(1, 16, 2))
    for rank, mine, ok, shape, t in res:
        assert ok and shape == (16, 3, 4, 4)            # gathered in view order on every rank
        assert t == 2.0                                 # max over ranks of (1.0, 2.0)


def test_ragged_view_count():
    res = _run(2, 5)                                    # 3 + 2 views: the short rank pads, the gather trims
    for rank, mine, ok, shape, t in res:
        assert ok and shape == (5, 3, 4, 4)


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from pixelsynth_amd import distributed as D
    x = torch.arange(6.0).view(3, 2)
    assert D.shard_views(7) == list(range(7))
    assert torch.equal(D.gather_frames(x), x) and D.max_over_ranks(0.25) == 0.25
