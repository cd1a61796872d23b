"""worker of tests/test_scorers_cpu.py::test_candidate_scores_gathered_across_ranks_rank_like_one_process (gloo, 2 ranks)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pixelsynth_amd import distributed as D  # noqa: E402
from pixelsynth_amd.z_buffermodel import rank_samples  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for n in (7, 2, 1):
    rs = np.random.RandomState(3 + n)
    disc, entr = rs.rand(n), rs.rand(n)
    mine = D.shard_views(n, rank, world)
    d_all, e_all = D.gather_scores([disc[i] for i in mine], [entr[i] for i in mine], n)
    assert np.allclose(d_all, disc) and np.allclose(e_all, entr), (d_all, disc)
    best = rank_samples(list(d_all), list(e_all)) if n > 1 else 0
    assert n == 1 or best == rank_samples(list(disc), list(entr))
    assert D.owner_of(best, world) == best % world
if rank == 0:
    print("ok")
dist.destroy_process_group()
