"""GPU: the RCCL code path executed for real, on the one GPU a test box has -- a single-rank "nccl" process group, device tensors,
asynchronous all_gather + Work.wait() stream semantics, broadcast, score gather, barrier (pixelsynth_amd/distributed.py; the
reference's counterpart: models/vqvae2/distributed/distributed.py:75-107), and one bench.py run through the `world > 1` branch of
back().  The multi-rank ROW ORDER is covered by the gloo tests (tests/test_distributed_cpu.py, tests/test_bench_gpu.py); this one
covers what those cannot: that the backend calls work on device tensors and streams."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_every_collective_of_the_path_on_a_one_rank_rccl_group():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank_worker.py")], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["backend"] == "nccl" and r["world_size"] == 1
    assert r["went_through_backend"] and r["rows_equal"] and r["sync_form_equal"]
    assert r["broadcast_equal"] and r["scores_equal"] and r["max_over_ranks"] == 0.75
    assert r["librccl_mapped"]


def test_bench_step_through_the_multi_rank_branch_on_rccl(tmp_path):
    """bench.py with PS_BENCH_FORCE_COLLECTIVE=1: the asynchronous gathers of a step issued on RCCL behind its AR run and collected
    between the NEXT step's prefix pass and its first column launch (outpaint_planned(between=...)) -- and what they deliver is what
    the same steps produce without any collective."""
    dump = str(tmp_path / "gather.npz")
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--views", "16",
            "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(args + ["--dump-gather", dump], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(_env(), PS_BENCH_FORCE_COLLECTIVE="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["collective"] == {"backend": "nccl", "world_size": 1, "forced_on_one_rank": True} and d["n_gpus"] == 1
    got = np.load(dump)
    assert got["all_codes"].reshape(16, -1).shape == (16, 1024) and got["all_features_u8"].shape == (16, 3, 256, 256)
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from pixelsynth_amd import distributed as D
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    dd, _ = bench.make_inputs(0, 16, dev)
    ref = bench.run_step(model, dd, 1)
    model.outpaint2.engine(32, 32, 16).check()
    assert np.array_equal(got["all_codes"].reshape(16, 1024), ref["codes"].reshape(16, 1024).cpu().numpy())
    assert np.array_equal(got["all_features_u8"], D.to_image_u8(ref["gen_fs"]).cpu().numpy())
