"""GPU: the contract of bench.py's single JSON line (what the driver parses), on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["unit"] == "frames/s"
    views = d["config"]["views_per_gpu"] if "views_per_gpu" in d["config"] else 16
    assert abs(d["value"] - views * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3       # whole-job frames / wall time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) / r["achieved"] < 1e-3
    # which library produced the number, and every matrix kernel of the step with its own event-timed figures, largest phase first
    assert "product build" in d["library"]["build"] and d["library"]["path"].endswith("libpixelsynth_hip.so")
    rows = r["kernels"]
    names = [row["kernel"] for row in rows]
    assert {"k_gemm_ws<0>", "k_gemm_ws<1>", "k_gemm_ws<2>"} <= set(names) and any(n.startswith("k_column_tp") for n in names), names
    assert [row["ms_per_step"] for row in rows] == sorted((row["ms_per_step"] for row in rows), reverse=True)
    assert r["kernel"].startswith(names[0] + ":")
    for row in rows:
        assert row["launches_per_step"] > 0 and row["avg_launch_us"] > 0
        assert abs(row["ms_per_step"] - row["launches_per_step"] * row["avg_launch_us"] * 1e-3) <= 0.02 * row["ms_per_step"] + 2e-3, row
        if "dense_flops_per_launch" in row:
            assert abs(row["frac_dense"] - row["dense_flops_per_launch"] / (row["avg_launch_us"] * 1e-6) / 1e12 / r["peak"]) < 2e-3, row
    ws = {row["kernel"]: row for row in rows}
    assert ws["k_gemm_ws<0>"]["launches_per_step"] == 28 and ws["k_gemm_ws<2>"]["launches_per_step"] == 8      # 14 / 4 stages x two frame ranges
    assert 0 < ws["k_gemm_ws<0>"]["items_evaluated_per_step"] <= 14 * 128 * 1024
    assert "executed_frac" not in r["step"]          # (no counter pass in this short run: nothing is made up)


def test_bench_line_with_the_counter_passes_says_what_is_executed():
    """With the run's own rocprofv3 --pmc passes (what the default `python bench.py` does): executed MFMA flops per kernel, both
    fractions, and the step's executed fraction beside the dense-equivalent one -- executed <= dense everywhere (closed taps and the
    prefix cone are skipped work)."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline", "--no-extra", "--live-pmc"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    r = d["roofline"]
    if r.get("pmc_live") is None:
        pytest.skip("the counter passes did not run on this box")
    step = r["step"]
    assert 0 < step["executed_frac"] < step["frac"] < 1
    ph = r["phases"]
    assert 0 < ph["column_launches_executed_frac"] < 1 and 0 < ph["prefix_pass_executed_frac"] < 1
    assert abs(ph["column_launches_ms"] + ph["prefix_pass_and_small_kernels_ms"] - d["ms_per_step"]) < 0.25 * d["ms_per_step"]
    for row in r["kernels"]:
        if row["kernel"].startswith("k_gemm_ws") or row["kernel"].startswith("k_column_tp"):
            assert 0 < row["frac_executed"] <= 1 and abs(row["executed_flops_per_launch"] - row["mfma_instructions_per_launch"] * 2048) <= 2048, row
        if row["kernel"].startswith("k_gemm_ws"):
            assert row["executed_flops_per_launch"] <= row["dense_flops_per_launch"] * 1.02, row      # per tile a tap is computed for all 16 items


def _torchrun(script_args, env_extra, nproc=2, timeout=900):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PS_DRYRUN_ONE_GPU="1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_two_rank_bench_gathers_what_single_rank_runs_produce(tmp_path):
    """The N > 1 path on one GPU (both ranks on cuda:0, gloo instead of RCCL -- control flow, not a scaling number): bench.py at
    --gpus 2 must gather, in rank order, exactly the frames and codes each rank's views give in a single-process run."""
    import numpy as np
    import torch
    dump = str(tmp_path / "gather.npz")
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--views", "16", "--no-cpu-baseline",
                     "--no-extra", "--dump-gather", dump], {})
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and abs(d["value"] - 2 * 16 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3   # whole-job frames / wall time
    got = np.load(dump)
    sys.path.insert(0, ROOT)
    import bench
    from pixelsynth_amd import distributed as D
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    for rank in range(2):
        dd, _ = bench.make_inputs(rank, 16, dev)
        o = bench.run_step(model, dd, 1)
        torch.cuda.synchronize()
        # gather_frames orders by view index under the round-robin deal: rank r's v-th view sits at row v * world + r
        assert np.array_equal(got["all_codes"][rank::2], o["codes"].cpu().numpy()), f"rank {rank}: gathered codes differ"
        assert np.array_equal(got["all_features_u8"][rank::2], D.to_image_u8(o["gen_fs"]).cpu().numpy())
    model.outpaint2.engine(32, 32, 16).check()


def test_strong_scaling_forms_one_rank_and_two(tmp_path):
    """The forms BASELINE.json names for several GPUs -- a job of a fixed TOTAL size dealt round-robin over the ranks: C5
    (--total-views) and C4 (--trajectory circle --frames 64).  One rank runs the whole 64-frame circle (C4 at its size); two ranks
    (both on cuda:0, gloo) gather, in view order, exactly the rows the whole job gives in one process."""
    import numpy as np
    import torch
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--trajectory", "circle",
                          "--frames", "64", "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["views_per_gpu"] == 64 and d["config"]["workload"].startswith("C4")
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    dump = str(tmp_path / "gather.npz")
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--total-views", "32", "--no-cpu-baseline",
                     "--no-extra", "--dump-gather", dump], {})
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["views_per_gpu"] == 16
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3          # the job's total / wall time
    got = np.load(dump)
    sys.path.insert(0, ROOT)
    import bench
    from pixelsynth_amd import distributed as D
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    dd, _ = bench.make_inputs(0, 32, dev)
    o = bench.run_step(model, dd, 1)
    torch.cuda.synchronize()
    model.outpaint2.engine(32, 32, 32).check()
    assert np.array_equal(got["all_codes"], o["codes"].cpu().numpy())
    assert np.array_equal(got["all_features_u8"], D.to_image_u8(o["gen_fs"]).cpu().numpy())


def test_two_rank_driver_writes_the_whole_circle(tmp_path):
    """C4 at its size, one rank and two ranks on one GPU: `driver --trajectory circle --frames 64` -- views dealt round-robin,
    frames gathered, rank 0 writes video/0.png (the source) .. video/64.png; every frame equal to what a single-rank run writes."""
    from PIL import Image
    import numpy as np
    outs = {}
    for nproc in (1, 2):
        od = str(tmp_path / f"n{nproc}")
        args = ["-m", "pixelsynth_amd.driver", "--trajectory", "circle", "--frames", "64", "--batch", "16", "--out", od]
        if nproc == 1:
            r = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=900, cwd=ROOT)
        else:
            r = _torchrun(args, {}, nproc=2)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[nproc] = [np.asarray(Image.open(os.path.join(od, "video", f"{i}.png"))) for i in range(65)]
    assert len({o.tobytes() for o in outs[1][1:]}) > 32     # the circle really moves
    for i in range(65):   # (a view's draws are seeded by the view, not by the rank or batch it falls in)
        assert outs[1][i].shape == (256, 256, 3)
        assert np.array_equal(outs[1][i], outs[2][i]), f"frame {i} differs between the 1-rank and the 2-rank run"
