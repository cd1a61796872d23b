#!/usr/bin/env python
"""bench.py -- novel-view frames/sec @256x256 (reproject + AR outpaint) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One STEP = one pass of the hot path over one batch of V independent novel views per GPU.  Default V = 128 = BASELINE.json's
config C5 -- Matterport-shaped 256x256 inputs, 8 source images x 16 novel views each -- which fits one GPU, so at N = 1
the step IS C5; under `torch.distributed.run` every rank renders its own 8 x 16 views (weak scaling).  A view =
reprojection + soft z-buffer splat, generation order + masks (a7-a9), autoregressive outpainting of
the 32x32 VQ code grid (a13).  Inputs are synthetic and already resident in HBM when the timed region starts; weights
are random-init with the reference's shapes.  The VQ-VAE / depth / refinement networks around the path are next-row
components, timed separately under `other_single_gpu_configs`; the codes of the reprojected view are
synthetic.  The only collective is the RCCL all_gather of the finished frames (no data-path exchange).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import gc
import ctypes
import json
import os
import sys
import time
import types

# More than four streams carry kernels at once when the ranks gather (main, the prefix pass's second frame range, the next step's splat,
# RCCL's own): the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default, and two of them
# sharing one serialise -- 19.5 instead of 18.1 ms per step on the collective path (tools/hwq_ab.sh).  Read when the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG", "False")

from pixelsynth_amd import _lib, distributed as D, synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2 peak


def make_opts():
    return types.SimpleNamespace(
        W=256, use_rgb_features=True, splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=128,
        tau=1.0, rad_pow=2, accumulation="alphacomposite", background_smoothing_kernel_size=13, min_z=1.0,
        max_z=100.0, rotation=0.6, direction="R", temperature=0.7, model_setting="gen_img", seed=0, homography=False)


def build_model(device):
    from pixelsynth_amd.z_buffermodel import ZbufferModelPts
    model = ZbufferModelPts(make_opts()).eval()
    model.outpaint2.load_state_dict({k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(0).items()})
    return model.to(device)


VIEWS_PER_SOURCE = 16   # config C5: 8 source images x 16 novel views each
# PS_BENCH_FORCE_COLLECTIVE=1: a single-rank run creates the RCCL process group all the same and sends its gathers, barrier and
# max-over-ranks through it -- the multi-GPU code path (`world > 1` in back()) executed on the one GPU a test box has
FORCE_COLLECTIVE = os.environ.get("PS_BENCH_FORCE_COLLECTIVE") == "1"
GATE_MIN_VIEWS = int(os.environ.get("PS_BENCH_GATE_MIN", "60"))     # batches from this size on: the next step's splat is gated behind the prefix pass
HOST_TIMES = [] if os.environ.get("PS_BENCH_HOST_TIMES") == "1" else None   # run_steps: host stamps per step, summarised on stderr


def make_inputs(rank, V, device, smooth=True, cameras="mp3d", ids=None, trajectory="sweep"):
    """V independent (source, target view) pairs: ceil(V / 16) source images, 16 target views each (a yaw sweep of +-0.6 rad,
    the reference's full angle, models/z_buffermodel.py:113).  cameras "mp3d": Matterport/Habitat-shaped (config C5:
    K = diag(1/tan(hfov/2), ., 1, 1) at hfov 90 deg, depth 0.5 .. 10, data/create_rgb_dataset.py:204-216); "demo": the demo /
    RealEstate10K-shaped cameras of demo.py:36-96 (K = I, P = diag(2,-2,-1,1)), depth 1 .. 100.
    trajectory "circle" (config C4): ONE source, V target poses on the 'C' circle of get_rt_from_rot (z_buffermodel.py:217-225),
    pose i = C_i/V, every view rendered from the source (create_vid.py-style playback order).
    ids: this rank's share of a job of V views in TOTAL (strong scaling: the job is built as a whole -- `rank` only seeds it --
    and the rows `ids` of it are put on the device)."""
    S = 256
    circle = trajectory == "circle"
    n_src = 1 if circle else max(1, -(-V // VIEWS_PER_SOURCE))
    per = -(-V // n_src)
    lo, hi = (0.5, 10.0) if cameras == "mp3d" else (1.0, 100.0)
    cam = (syn.mp3d_cameras if cameras == "mp3d" else syn.demo_cameras)(V)
    img = np.empty((V, 3, S, S), np.float32)
    depth = np.empty((V, 1, S, S), np.float32)
    yaws = np.empty(V, np.float64)
    for s_ in range(n_src):
        sl = slice(s_ * per, min(V, (s_ + 1) * per))
        n = sl.stop - sl.start
        img[sl] = syn.image(1000 + 64 * rank + s_, 1, 3, S)
        depth[sl] = (syn.depth_smooth if smooth else syn.depth_uniform)(2000 + 64 * rank + s_, 1, S, lo, hi)
        yaws[sl] = np.linspace(-0.6, 0.6, n) if n > 1 else np.array([0.6])
    RT2 = np.empty((V, 4, 4), np.float32)
    RT2inv = np.empty((V, 4, 4), np.float32)
    for v in range(V):
        inv, rt = syn.circle_pose(cam["P"][v:v + 1], v, V) if circle else syn.yaw_pose(cam["P"][v:v + 1], float(yaws[v]))
        RT2[v], RT2inv[v] = rt[0], inv[0]
    codes = syn.codes(3000 + rank, V)
    uniforms = np.random.RandomState(4000 + rank).rand(V, 1024).astype(np.float32)
    if ids is not None:   # this rank's rows of the whole job
        ids = np.asarray(ids, np.int64)
        img, depth, RT2, RT2inv, codes, uniforms, yaws = (a[ids] for a in (img, depth, RT2, RT2inv, codes, uniforms, yaws))
        cam = {k: a[ids] for k, a in cam.items()}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    dev = dict(img=t(img), depth=t(depth), K=t(cam["K"]), Kinv=t(cam["Kinv"]), P=t(cam["P"]), Pinv=t(cam["Pinv"]),
               RT2=t(RT2), RT2inv=t(RT2inv), codes=t(codes), uniforms=t(uniforms))
    host = dict(img=img, depth=depth, cam=cam, RT2=RT2, RT2inv=RT2inv, codes=codes, yaws=yaws, n_src=n_src, cameras=cameras,
                trajectory=trajectory, per_source=per)
    return dev, host


def front(model, d):
    """First half of a step: reproject + splat, background masks to the host, orders / masks / wavefront schedule built
    and uploaded (ends synchronised with the stream it ran on)."""
    return model.plan_views(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])


def back(model, d, planned, world, prev=None, pipelined=False):
    """Second half: the AR run (asynchronous), then the path's only collective.
    prev: the previous step's result, whose gathers are still in flight: the stream waits for them between this step's prefix
    pass and its first column launch (see finish_gathers)."""
    between = (lambda: finish_gathers(prev)) if prev is not None and "_gathers" in prev else None
    if pipelined:   # the result is the PREVIOUS step's batch, whose last wavefronts ran inside this step's first launches (None: first step)
        out = model.outpaint_pipelined(planned, d["codes"], temperature=0.7, uniforms=d["uniforms"], between=between)
    else:
        out = model.outpaint_planned(planned, d["codes"], temperature=0.7, uniforms=d["uniforms"], between=between)
    return start_gathers(out, world)


def start_gathers(out, world):
    if out is not None and (world > 1 or FORCE_COLLECTIVE):  # what the path produced on every rank -- the reprojected views as 8-bit images (the byte volume of finished
        # frames: the VQ-VAE decode that turns codes into pixels is a next-row component, timed under end_to_end_*) and the
        # completed 32x32 code grids -- RCCL all_gather over xGMI.  Started here as asynchronous collectives, collected by
        # finish_gathers() in the NEXT step, between its whole-grid prefix pass and its first column launch: 25 MB per rank and
        # step at 128 views take the ring a couple of ms, which then pass beside the prefix pass -- whose small workgroups fit
        # around the collective's kernels -- and never beside a column launch, which keeps one workgroup per compute unit
        # resident and would wait for the ones a late peer's collective still holds (round-3 advice).
        # PS_BENCH_SYNC_GATHER=1: collected at once, in front of the next step.
        out["_gathers"] = (D.gather_frames_start(D.to_image_u8(out["gen_fs"]), force_collective=FORCE_COLLECTIVE),
                           D.gather_frames_start(out["codes"].contiguous(), force_collective=FORCE_COLLECTIVE))
        if os.environ.get("PS_BENCH_SYNC_GATHER") == "1":
            finish_gathers(out)
    return out


def finish_gathers(out):
    """The step's collectives are through (for the current stream): their results into the step's dict."""
    g = out.pop("_gathers", None) if out is not None else None
    if g is not None:
        out["all_features_u8"], out["all_codes"] = g[0].result(), g[1].result()
    return out


def run_step(model, d, world):
    return finish_gathers(back(model, d, front(model, d), world))


def run_steps(model, d, world, n, side):
    """_run_steps; a run that dies on the way leaves no batch in flight for the next run to merge (z_buffermodel.outpaint_reset)."""
    try:
        return _run_steps(model, d, world, n, side)
    except BaseException:
        model.outpaint_reset()
        raise


def _run_steps(model, d, world, n, side):
    """n steps, software-pipelined: while the device runs the AR loop of step i (main stream), the host half of step
    i + 1 -- splat on the side stream, masks back, planning, uploads -- is already under way.  Same work per step, the
    steps are independent; results identical to n x run_step.
    """
    main = torch.cuda.current_stream()
    planned, out = None, None
    pipelined = ar_pipelined(d["codes"].shape[0])
    for i in range(n):
        if planned is None:
            planned = front(model, d)
        # The side stream's kernels of step i + 1 (the splat: a swarm of 64-thread workgroups) are held back until step i's
        # whole-grid prefix pass starts, i.e. until step i - 1's column launches are through: a column launch keeps one large
        # workgroup per compute unit resident, and the splat's workgroups, let loose beside the column launches, fill the
        # compute units between two launches and hold the next launch up until they retire (under rocprofv3: 194 instead of
        # 154 us per launch while the splat runs).  The step takes the same time either way -- the splat's 2.5 ms are paid
        # beside the prefix pass or beside the columns -- but no in-launch wait of a column launch is at the mercy of another
        # stream any more.
        gate = torch.cuda.Event()
        gate.record(main)
        t0 = time.perf_counter()
        prev, out = out, back(model, d, planned, world, prev=out, pipelined=pipelined)
        finish_gathers(prev)      # (a step without column launches has not collected them)
        t1 = time.perf_counter()
        planned = None
        if i + 1 < n:
            # (batches of the latency form -- a few ms per step -- are not gated: the chain gate -> splat -> masks to the host -> planning
            # -> enqueueing is then as long as the step itself and the host falls behind: 16 views 4.2 -> 4.0 ms per step without it)
            if not os.environ.get("PS_BENCH_NO_GATE") and d["codes"].shape[0] >= GATE_MIN_VIEWS:
                side.wait_event(gate)
            with torch.cuda.stream(side):
                planned = front(model, d)
            model.adopt_planned(planned, main)
            main.wait_stream(side)
        if HOST_TIMES is not None:   # PS_BENCH_HOST_TIMES=1: where the host thread spends a step (enqueueing the AR run; the next step's front)
            HOST_TIMES.append((t0, t1, time.perf_counter()))
    if pipelined:   # what is left of the batches in flight, as launches of their own: the n steps are complete when this call returns
        for o in model.outpaint_flush():
            finish_gathers(out)
            out = start_gathers(o, world)
    return finish_gathers(out)


def ar_pipelined(V):
    """The AR runs of consecutive steps overlapped (z_buffermodel.outpaint_pipelined): batches of at least two views (16 views: 5.55 ->
    4.30 ms per step, 64-frame circle 10.9 -> 9.1, 128 views 16.9 -> 12.8 with the per-frame prefixes); PS_BENCH_AR_PIPELINE=0 runs every
    step's AR run on its own."""
    return os.environ.get("PS_BENCH_AR_PIPELINE", "1") != "0" and V >= int(os.environ.get("PS_BENCH_AR_PIPELINE_MIN", "2"))


def measure_roofline(model, d, out, V, live_pmc=False):
    """Average column launch -- the dominant kernel: one launch per WAVEFRONT of independent columns (one column = one order
    position of one frame) -- over the column launches of one steady-state step (pipelined form: inside a two-batch run,
    ps_pixelcnn_time_ar_run_waves_range; otherwise a whole AR run of this step's views, ps_pixelcnn_time_ar_run_waves), measured with
    HIP events on the stream the launches go to (median of three runs), against the dense algorithmic fp32 work of its columns:
    10.42 MFLOP per column = the 32 matrix stages' centre taps (the chain role: 16-column MFMA tiles in the throughput form
    k_column_tp, fp32 FMA chains on the vector ALU in the latency form k_column_la) plus the neighbour-tap partial sums of all 32
    masked convs (fp32 MFMA in both forms, masked taps skipped) -- the u_init product is a gather and is not priced.  fp32 MFMA and
    packed fp32 FMA share gfx950's dense fp32 peak (157.3 TFLOP/s).  At C5's size every phase of the step is throughput-bound (the
    launches are full) -- the small batches under other_single_gpu_configs are bound by the latency of a launch's 33
    dependent stages; which kernel is the LARGEST phase of the step is `roofline.kernel` / `roofline.kernels` (measure_kernels), the
    top-level achieved / frac stay the column launch's for comparability with earlier rounds; `traffic` is measured in the run itself when live_pmc is set (live_pmc_traffic: two rocprofv3
    --pmc passes of this command), otherwise it, and always `mfma_counters` and `kernel_table`, come from the newest committed PMC record
    of the same workload (profiles/README.md names it) -- counter passes cannot run inside a timed bench."""
    plan = out["plan"]
    pipelined = ar_pipelined(V)
    cols, wave_start = plan.waves
    ncols = int(cols.shape[0])
    launches, total_ms, fpc = ctypes.c_int(0), ctypes.c_float(0.0), ctypes.c_double(0.0)
    us_list = []
    byref = lambda v: ctypes.cast(ctypes.byref(v), ctypes.c_void_p)
    if pipelined:
        # The column launches of steady-state steps as the timed region runs them (several batches in flight, every launch as full as they
        # make it: z_buffermodel.outpaint_pipelined), event-timed through the engine's launch profile (ps_pixelcnn_profile_begin / _end):
        # a long run minus a short one -- the pipeline's fill and the flush cancel.
        eng = model.outpaint2.engine(32, 32, model.pipe_frames(V))
        frames = getattr(plan, "waves_frames", None) if model.PER_FRAME_PREFIX else None
        ncols = int((frames[0] if frames is not None else cols).shape[0])
        n_short, n_long = 4, 12
        n_launch = 0
        for _ in range(3):
            run_steps(model, d, 1, 2, side_stream())
            torch.cuda.synchronize()
            profs = []
            for n_ in (n_short, n_long):
                eng.profile_begin()
                try:
                    run_steps(model, d, 1, n_, side_stream())
                finally:
                    profs.append(eng.profile_end())
            n_launch = sum(profs[1][k][0] - profs[0][k][0] for k in profs[1] if k.startswith("k_column"))
            ms = sum(profs[1][k][1] - profs[0][k][1] for k in profs[1] if k.startswith("k_column"))
            us_list.append(ms * 1e3 / max(1, n_launch))
        launches.value = int(round(n_launch / (n_long - n_short)))
        fpc.value = 10424320.0
        n_wavefronts = launches.value
        launches_exact = n_launch / (n_long - n_short)
    else:
        eng = model.outpaint2.engine(32, 32, V)
        for _ in range(3):
            c32 = d["codes"].reshape(V, 1024).to(torch.int32).contiguous().clone()
            rc = _lib.lib().ps_pixelcnn_time_ar_run_waves(
                eng.handle, _lib.ptr(c32), _lib.ptr(plan.order_loc), _lib.ptr(plan.region), _lib.ptr(plan.mask_init),
                _lib.ptr(plan.mask_undilated), _lib.ptr(plan.mask_dilated), _lib.ptr(d["uniforms"]), 0.7, V, plan.first_step,
                _lib.ptr(cols), _lib.ptr(wave_start), len(wave_start) - 1, byref(launches), byref(total_ms), byref(fpc), _lib.current_stream())
            _lib.check(rc, "ps_pixelcnn_time_ar_run_waves")
            us_list.append(total_ms.value * 1e3 / max(1, launches.value))
        n_wavefronts = len(wave_start) - 1
    us = sorted(us_list)[1]
    cols_per_launch = ncols / max(1e-9, launches_exact if pipelined else launches.value)
    fl = fpc.value * cols_per_launch
    tf = fl / (us * 1e-6) / 1e12
    traffic, traffic_src, mfma_util, kernel_table = None, None, None, None
    pmc = latest_pmc_record(V)   # PMC passes cannot run inside the timed bench: committed summary of the same workload
    if pmc:
        traffic, traffic_src, mfma_util = pmc.get("traffic_bytes_per_launch"), pmc.get("source"), pmc.get("mfma")
        kernel_table = pmc.get("kernel_table")
    live = live_pmc_traffic() if live_pmc else None
    if live:   # the driver's own run vouches for the traffic and the MFMA count; the per-kernel table stays the committed record's
        traffic, traffic_src = live["traffic_bytes_per_launch"], live["source"]
        # v_mfma_f32_16x16x4_f32: 32 cycles each on one of 1024 SIMDs, against the event-timed average launch at the nominal 2.4 GHz
        live["mfma_busy_fraction_of_all_simd_cycles"] = round(live["SQ_INSTS_MFMA_per_launch"] * 32 / 1024 / (us * 2.4e3), 4)
    tp = cols_per_launch > 128
    kernel = ("k_column_tp (throughput form of the column launch, one launch per wavefront of up to 1024 independent AR columns: "
              "16-column MFMA chain tiles + one wave per neighbour item, the neighbour role a launch ahead of the chain tiles; "
              "wavefronts of up to 256 columns as two launches of the latency form k_column_la -- the average is over all "
              "column launches of the AR run"
              + ("; the AR runs of consecutive steps share their launches -- up to four batches in flight, every launch taking what is left of "
                 "each batch's current wavefront (z_buffermodel.outpaint_pipelined, lmconv.model.pack_launches): the launches timed are those "
                 "of steady-state steps)" if pipelined else ")")
              if tp else
              "k_column (one launch per wavefront of independent AR columns: per-column centre-tap chains + "
              "neighbour-tap slots of all 32 masked convs)")
    return {"bound": "mfma", "column_launch_kernel": kernel,
            "achieved": round(tf, 4), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(tf / FP32_MFMA_PEAK_TF, 6), "traffic": traffic, "traffic_source": traffic_src, "mfma_counters": mfma_util,
            "algorithmic_flops_per_launch": round(fl), "avg_launch_us": round(us, 3),
            "flops_per_column": round(fpc.value), "columns_per_launch": round(cols_per_launch, 2),
            "launches_per_ar_run": launches.value, "wavefronts": n_wavefronts, "columns": ncols,
            "ar_runs_overlapped": bool(pipelined), "batches_in_flight": model.pipe_depth(V) if pipelined else 1,
            "wavefronts_of_a_batch_alone": len(wave_start) - 1,
            "per_frame_prefixes": bool(pipelined and getattr(plan, "waves_frames", None) is not None and model.PER_FRAME_PREFIX),
            "columns_with_one_prefix_for_the_batch": int(cols.shape[0]),
            "pmc_live": live, "traffic_committed_record": (pmc or {}).get("traffic_bytes_per_launch"),
            "kernel_table": kernel_table,
            "kernel_table_note": "the committed PMC record's table (every kernel alone on the chip: PMC and trace passes with "
                                 "PS_PREFIX_STREAMS=1; the default step deals the prefix pass to two frame ranges on two streams: twice the "
                                 "k_gemm_ws launches at half the items, overlapping -- z_buffermodel.PREFIX_STREAMS); THIS run's own table is "
                                 "`kernels`",
            "walk_positions_without_wavefronts": 1024 - plan.first_step,
            "reference_definition": {
                "what": "the same launch priced at what the reference schedules for its columns (SURVEY 8d): one whole-grid "
                        "forward, 11.43095 GFLOP, per frame and order position -- skipped redundant work is NOT utilisation, "
                        "this is shown for comparison only",
                "equivalent_tflops": round(11.43095e9 * cols_per_launch / (us * 1e-6) / 1e12, 1)}}


def live_pmc_traffic():
    """The column launches' FETCH_SIZE / WRITE_SIZE measured in THIS run (review, round 4: a committed record is reproducible but the
    driver's run cannot vouch for it): `rocprofv3 --pmc` passes -- one counter each, no trace beside them -- over a short run of this
    same command (2 steps + 1 warm-up, no side measurements), parsed like tools/pmc_record.py does: the average over ALL column launches
    (k_column_tp and k_column_la, weighted by dispatches), FETCH_SIZE doubled (gfx950 reports half of the bytes of wide coalesced
    reads, MI355X_MICROARCH.md); a third pass counts the MFMAs (SQ_INSTS_MFMA).  None when rocprofv3 is missing, a pass fails or takes
    more than four minutes, PS_BENCH_NO_LIVE_PMC=1, or inside such a pass."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get("PS_BENCH_NO_LIVE_PMC") == "1" or os.environ.get("PS_BENCH_PMC_CHILD") == "1":
        return None
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    acc, by_kind = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_MFMA"):
        tmp = tempfile.mkdtemp(prefix="ps_pmc_", dir="/tmp")
        cmd = ([exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)]
               + sys.argv[1:] + ["--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-extra"])
        proc = None
        try:
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", PS_BENCH_PMC_CHILD="1"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            if proc.wait(timeout=240) != 0:
                raise RuntimeError("rocprofv3 pass failed")
            for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as fh:
                    for row in csv.DictReader(fh):
                        name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                        if (row.get("Counter_Name") or row.get("Counter Name")) != counter:
                            continue
                        val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                        kern = "k_column_tp" if "k_column_tp" in name else "k_column_la" if "k_column_la" in name else None
                        if kern:
                            a = acc.setdefault((kern, counter), [0.0, 0])
                            a[0] += val
                            a[1] += 1
                        kind = kernel_kind_of(name)     # (every kernel that carries matrix work, by launch kind: roofline.kernels)
                        if kind:
                            a = by_kind.setdefault((kind, counter), [0.0, 0])
                            a[0] += val
                            a[1] += 1
        except Exception:
            if proc is not None and proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)   # (the session this call started, nothing else)
                except OSError:
                    pass
            shutil.rmtree(tmp, ignore_errors=True)
            return None
        shutil.rmtree(tmp, ignore_errors=True)
    kernels = sorted({k for k, _ in acc})
    if not kernels or any((k, c) not in acc for k in kernels for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_MFMA")):
        return None
    per = {k: {"dispatches": acc[(k, "FETCH_SIZE")][1], "FETCH_SIZE_KB_mean": round(acc[(k, "FETCH_SIZE")][0] / acc[(k, "FETCH_SIZE")][1], 1),
               "WRITE_SIZE_KB_mean": round(acc[(k, "WRITE_SIZE")][0] / acc[(k, "WRITE_SIZE")][1], 1),
               "SQ_INSTS_MFMA_mean": round(acc[(k, "SQ_INSTS_MFMA")][0] / acc[(k, "SQ_INSTS_MFMA")][1])} for k in kernels}
    n = sum(v["dispatches"] for v in per.values())
    total = sum((2 * v["FETCH_SIZE_KB_mean"] + v["WRITE_SIZE_KB_mean"]) * 1024 * v["dispatches"] for v in per.values())
    kinds = {}
    for (kind, counter), (tot, cnt) in by_kind.items():
        kinds.setdefault(kind, {})[counter + "_mean"] = tot / max(1, cnt)
        kinds[kind]["dispatches"] = cnt
    return {"traffic_bytes_per_launch": int(round(total / n)),
            "SQ_INSTS_MFMA_per_launch": int(round(sum(v["SQ_INSTS_MFMA_mean"] * v["dispatches"] for v in per.values()) / n)), "per_kernel": per,
            "by_launch_kind": kinds,
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc SQ_INSTS_MFMA, one pass each over this command "
                      "with --steps 6 --warmup 1 and nothing but the timed steps in it (no trace beside the counters; the child skips "
                      "the roofline's own measurement runs); average over all column launches; FETCH_SIZE x 2 (gfx950 correction)"}


def kernel_kind_of(name):
    """rocprofv3 kernel name -> the engine's launch kind (PixelCNNEngine.launch_kind_names), or None."""
    import re
    m = re.search(r"k_gemm_ws<\(?[^0-9>]*(\d)", name)
    if m:
        return f"k_gemm_ws<{m.group(1)}>"
    for k in ("k_column_tp8", "k_column_tp", "k_column_la", "k_gemm_wg"):
        if k in name:
            return k
    if re.search(r"\bk_column\b", name):
        return "k_column"
    if re.search(r"\bk_gemm\b", name):
        return "k_gemm"
    return None


# dense fp32 work of one (frame, location) at one stage of the network (2 x taps x Cin x Cout; the 8 gated blocks of the down
# pass add their nin_skip product, 2 x 160 x 80): what `roofline.kernels[].dense_flops_per_launch` prices an EVALUATED item at
STAGE_FLOPS = {"conv_out": 2 * 9 * 160 * 160, "conv_in": 2 * 9 * 160 * 80, "nin_skip": 2 * 160 * 80, "dilated": 2 * 9 * 80 * 80}
MFMA_FLOP = 2 * 16 * 16 * 4          # one v_mfma_f32_16x16x4_f32
N_SIMD, NOMINAL_MHZ = 1024, 2400.0


def measure_kernels(model, d, V, world, side, plan, live, ms_per_step, steps=12):
    """`roofline.kernels`: every kernel that carries matrix work, as the timed region runs it -- `steps` more pipelined steps with a HIP
    event pair around each such launch on the stream it goes to (ps_pixelcnn_profile_begin / _end): launches per step, event-timed
    average, ms per step (the prefix pass runs on two streams: its kernels overlap each other and the splat, so the column sums to more than
    the step), dense-equivalent flops (column launches: SURVEY 8d's 10.42 MFLOP per column; whole-grid launches: the items the pass
    really evaluates at that stage -- read back from the engine's dependency-cone table -- x the stage's dense flops), EXECUTED flops
    (SQ_INSTS_MFMA of this run's own --pmc pass x 2048; closed taps are skipped, so executed < dense) and both as fractions of the fp32
    matrix peak (`frac_executed` = the share of all SIMD cycles the matrix pipes are busy at the nominal 2.4 GHz).
    -> (kernels, dominant kernel name, executed flops per step)."""
    pipelined = ar_pipelined(V)
    eng = model.outpaint2.engine(32, 32, model.pipe_frames(V) if pipelined else V)
    run_steps(model, d, world, 2, side)
    torch.cuda.synchronize()
    # the two phases of a step on the main stream: [prefix pass + the small kernels around it | column launches]; events around the
    # column launches of every step (outpaint_pipelined -> _pipe_columns; outpaint_planned has no such seam and reports none)
    marks = []
    real_cols = getattr(model, "_pipe_columns")

    def cols(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_cols(*a, **k)
        e1.record()
        marks.append((e0, e1))
    # a run of n steps is the pipeline's fill, n steady-state steps' worth of launches and the flush of what is left (narrow launches of
    # their own): a SHORT run is profiled first and taken off the long one -- what remains are `steps - short` steady-state steps
    short = 4 if pipelined and steps > 6 else 0
    prof0 = None
    if short:
        eng.profile_begin()
        try:
            run_steps(model, d, world, short, side)
        finally:
            prof0 = eng.profile_end()
    model._pipe_columns = cols
    eng.profile_begin()
    try:
        run_steps(model, d, world, steps, side)
    finally:
        prof = eng.profile_end()
        model.__dict__.pop("_pipe_columns", None)          # (the instance attribute: the class's method is back)
    if prof0 is not None:
        prof = {k: (prof[k][0] - prof0[k][0], prof[k][1] - prof0[k][1]) for k in prof}
    nsteps = steps - short
    phases = None
    D = model.pipe_depth(V) if pipelined else 1
    if pipelined and len(marks) >= steps + 1 and steps > D + 1:     # (steps merged launches + the flush's)
        col_ms = [a.elapsed_time(b) for a, b in marks[D:steps]]                       # steady-state steps only (pipeline full, not the flush)
        pre_ms = [marks[i][1].elapsed_time(marks[i + 1][0]) for i in range(D - 1, steps - 1)]
        phases = {"column_launches_ms": round(float(np.mean(col_ms)), 3), "prefix_pass_and_small_kernels_ms": round(float(np.mean(pre_ms)), 3),
                  "what": "main-stream event marks around the column launches of the steady-state steps of this profile run: the step is the "
                          "prefix phase (whole-grid pass on two streams, the next step's splat beside it, item sort / cone / context kernels) "
                          "followed by the column phase"}
    # items the whole-grid pass evaluates per stage: ranks [start[stage][f], end[f]) of every frame
    N_EVAL = 33
    F = model.pipe_frames(V) if pipelined else V     # (the frame count of the run that filled the table: its row stride)
    _lib.lib().ps_pixelcnn_debug_cache.restype = ctypes.c_void_p
    ptr = _lib.lib().ps_pixelcnn_debug_cache(eng.handle, 5, 0)
    raw = type("Raw", (), {"__cuda_array_interface__": {"shape": (N_EVAL, F), "typestr": "<i4", "data": (ptr, False), "version": 2}})()
    starts = torch.as_tensor(raw, device=d["codes"].device).cpu().numpy()[:, :V].astype(np.int64)
    per_frame = pipelined and getattr(plan, "waves_frames", None) is not None and model.PER_FRAME_PREFIX
    ends = np.asarray(plan.first_steps, np.int64)[:V] if per_frame else np.full(V, plan.first_step, np.int64)
    items = np.maximum(ends[None, :] - np.minimum(starts, ends[None, :]), 0).sum(1)          # (33,) per prefix pass of V frames
    dense_grid = {"k_gemm_ws<0>": sum(items[15 + g] * STAGE_FLOPS["conv_out"] for g in range(14)),
                  "k_gemm_ws<1>": sum(items[1 + g] * (STAGE_FLOPS["conv_in"] + (STAGE_FLOPS["nin_skip"] if g >= 6 else 0)) for g in range(14)),
                  "k_gemm_ws<2>": sum(items[29 + k] * STAGE_FLOPS["dilated"] for k in range(4))}
    n_items = {"k_gemm_ws<0>": int(items[15:29].sum()), "k_gemm_ws<1>": int(items[1:15].sum()), "k_gemm_ws<2>": int(items[29:33].sum())}
    ncols = int((plan.waves_frames[0] if per_frame else plan.waves[0]).shape[0])
    steps = nsteps
    col_launches = sum(prof[k][0] for k in prof if k.startswith("k_column"))
    kinds = (live or {}).get("by_launch_kind", {})
    rows, executed_step = [], 0.0
    for name, (n, ms) in prof.items():
        if n == 0:
            continue
        per_step, us = n / steps, ms * 1e3 / n
        row = {"kernel": name, "launches_per_step": round(per_step, 2), "avg_launch_us": round(us, 2), "ms_per_step": round(ms / steps, 3)}
        if name.startswith("k_column"):
            # (a step's columns over its column launches of every form: the forms are not priced apart -- a schedule's wide wavefronts
            # take the throughput forms, its narrow ones the latency form)
            dense = 10424320.0 * ncols / max(1.0, col_launches / steps)
            row["columns_per_launch_mean_over_all_forms"] = round(ncols / max(1.0, col_launches / steps), 1)
        elif name in dense_grid:
            dense = float(dense_grid[name]) / max(1.0, per_step)      # (one pass per step; its launches are per stage AND per frame range)
            row["items_evaluated_per_step"] = n_items[name]
        else:
            dense = None
        if dense is not None:
            row["dense_flops_per_launch"] = round(dense)
            row["frac_dense"] = round(dense / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF, 4)
        k = kinds.get(name)
        if k and "SQ_INSTS_MFMA_mean" in k:
            mf = k["SQ_INSTS_MFMA_mean"]
            row["mfma_instructions_per_launch"] = round(mf)
            row["executed_flops_per_launch"] = round(mf * MFMA_FLOP)
            row["frac_executed"] = round(mf * 32.0 / N_SIMD / (us * NOMINAL_MHZ), 4)
            executed_step += mf * MFMA_FLOP * per_step
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    if phases is not None and kinds:
        ex = {"column": 0.0, "prefix": 0.0}
        for r in rows:
            if "executed_flops_per_launch" in r:
                ex["column" if r["kernel"].startswith("k_column") else "prefix"] += r["executed_flops_per_launch"] * r["launches_per_step"]
        phases["column_launches_executed_frac"] = round(ex["column"] / (phases["column_launches_ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)
        phases["prefix_pass_executed_frac"] = round(ex["prefix"] / (phases["prefix_pass_and_small_kernels_ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)
    return rows, (rows[0]["kernel"] if rows else None), (executed_step if kinds else None), phases


def latest_pmc_record(V):
    """The newest `*_k_column_pmc.json` under profiles/ that profiles/README.md lists for this number of views (the README's
    last matching line wins), or None -- so that the record follows the kernel instead of naming a file here."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        with open(os.path.join(root, "README.md")) as fh:
            names = [w.strip("`*,;()") for line in fh for w in line.split() if w.strip("`*,;()").endswith("_pmc.json")]
    except OSError:
        return None
    for name in reversed(names):
        path = os.path.join(root, name)
        if os.path.exists(path):
            with open(path) as fh:
                rec = json.load(fh)
            if rec.get("views") == V:
                return rec
    return None


_SIDE = {}


def side_stream():
    """ONE side stream per device for the whole process: the runtime deals streams onto a handful of hardware queues in turn, and
    which queue a configuration's side stream shares with whom should not depend on how many configurations ran before it."""
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream()
    return _SIDE[dev]


def cool_down(seconds=2.0):
    """Side configurations are measured one after the other on one GPU, and a heavy one leaves the board at its power limit: the
    64-frame circle measured right behind the 256-view batch took 15.9 ms per step instead of 13.0, the 256-view batch behind the
    end-to-end passes 45 ms instead of 35.  An idle moment in front of each gives every configuration the same start."""
    torch.cuda.synchronize()
    time.sleep(seconds)


def small_batch_config(device, V, cameras, steps=20, total=None, trajectory="sweep"):
    """frames/s and the column launch's roofline numbers of a smaller batch (pipelined steps like the headline).
    total: the batch is rank 0's share of a job of `total` views over total / V ranks (the 8-GPU forms of C4 / C5)."""
    cool_down()
    model = build_model(device)
    if total is None:
        d, _ = make_inputs(0, V, device, cameras=cameras, trajectory=trajectory)
    else:
        d, _ = make_inputs(0, total, device, cameras=cameras, trajectory=trajectory, ids=D.shard_views(total, 0, total // V))
    side = side_stream()
    out = run_steps(model, d, 1, 3, side)      # (a 6 ms step is close to what the host needs to plan and enqueue one: short runs scatter)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(model, d, 1, steps, side)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng = model.outpaint2.engine(32, 32, V)
    eng.check()
    r = measure_roofline(model, d, out, V)
    res = {"frames_per_s": round(V / dt, 1), "ms_per_step": round(dt * 1e3, 3), "views": V, "cameras": cameras,
           "sampled_codes_per_view_mean": round(float(np.mean(out["plan"].n_sampled)), 1),
           "column_launch": {k: r[k] for k in ("achieved", "frac", "avg_launch_us", "columns_per_launch", "launches_per_ar_run")}}
    # the engine's activation caches (27.8 MB per view) go back HERE: left to the garbage collector, their hipFree -- a device
    # synchronisation of several ms at 256 views -- lands in the timed region of whichever configuration runs next
    eng.close()
    del model, d, out, eng
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return res


def extra_configs(device):
    """The other single-GPU configurations BASELINE.json names, measured the same way (inputs resident, barrier-free
    single stream, wall clock around synchronised steps): C3 = ONE view end to end (latency-bound: the AR chain
    uses one CU), C2 = batch-32 reprojection + splat only."""
    res = {}
    # what ONE of eight GPUs does in the 8-GPU form of C5 (1 source x 16 views), and round 1's workload (16 views, demo /
    # RealEstate10K-shaped cameras): few columns per wavefront -> the latency form of the column launch (k_column)
    res["C5_one_source_16_views"] = small_batch_config(device, 16, "mp3d")
    res["RealEstate_shaped_16_views"] = small_batch_config(device, 16, "demo")
    # where the path's throughput saturates: 16 sources x 16 views per step (the column launches are bound by the latency of their
    # dependent stages, so a step's time grows more slowly than its batch up to ~1000 columns per launch; tools/batch_sweep.py).
    res["C5_shaped_16_sources_256_views"] = small_batch_config(device, 256, "mp3d", steps=8)
    # The STRONG-scaling forms BASELINE.json names for 8 GPUs, priced from one GPU's measured share (no 8-GPU node behind this
    # run: a projection, the path has no exchange besides the final gather).  C5 = 128 views in total -> 16 per GPU, dealt
    # round-robin (rank 0 renders views 0, 8, 16, ...: two of every source's sweep); C4 = the 64-frame circle -> 8 frames per GPU.
    c5 = small_batch_config(device, 16, "mp3d", total=128)
    c4 = small_batch_config(device, 8, "demo", total=64, trajectory="circle")
    c4_1gpu = small_batch_config(device, 64, "demo", total=None, trajectory="circle", steps=40)   # (four batches in flight: the fill and the flush weigh on a short run)
    res["C4_circle_64_frames_one_gpu"] = c4_1gpu
    res["projected_per_gpu"] = {
        "note": "what ONE of eight GPUs runs in the strong-scaling forms of C5 / C4 (python bench.py --total-views 128 | "
                "--trajectory circle --frames 64 under torch.distributed.run), measured here; x8 is a projection, not a measurement",
        "C5_total_128_views_8gpus": {"views_per_gpu": 16, "ms_per_step": c5["ms_per_step"], "column_launch": c5["column_launch"],
                                     "frames_per_s_8gpus_projected": round(128 / (c5["ms_per_step"] * 1e-3), 1)},
        "C4_circle_64_frames_8gpus": {"frames_per_gpu": 8, "ms_per_step": c4["ms_per_step"], "column_launch": c4["column_launch"],
                                      "frames_per_s_8gpus_projected": round(64 / (c4["ms_per_step"] * 1e-3), 1)}}
    m1 = build_model(device)
    d1, _ = make_inputs(0, 1, device, cameras="demo")
    for _ in range(2):
        o1 = run_step(m1, d1, 1)
    torch.cuda.synchronize()
    times = []
    for _ in range(7):  # latency of ONE view: median of single, synchronised runs (a host hiccup must not define it)
        t0 = time.perf_counter()
        o1 = run_step(m1, d1, 1)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    res["C3_single_view"] = {"frames_per_s": round(1.0 / dt, 3), "ms_per_frame": round(dt * 1e3, 3),
                             "sampled_codes": int(o1["plan"].n_sampled[0]), "ar_positions_walked": 1024 - o1["plan"].first_step}
    d32, _ = make_inputs(1, 32, device, cameras="demo")
    pm = m1.pts_transformer
    call = lambda: pm.forward_justpts(d32["img"], d32["depth"], d32["K"], d32["Kinv"], d32["P"], d32["Pinv"], d32["RT2"], d32["RT2inv"])
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    alg = 1900544.0 * 32  # algorithmic bytes per frame of the splat
    res["C2_splat_b32"] = {"frames_per_s": round(32 / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                           "algorithmic_GBs": round(alg / dt / 1e9, 2), "frac_hbm": round(alg / dt / 1e9 / HBM_PEAK_GBS, 5)}
    # C2 in idx-emitting mode: the (B,S,S,K) idx / zbuf / dist tensors PyTorch3D materialises, B = 4
    pts4 = pm.project_pts(d32["depth"][:4].reshape(4, 1, -1), d32["K"][:4], d32["Kinv"][:4], d32["P"][:4], d32["Pinv"][:4],
                          d32["RT2"][:4], d32["RT2inv"][:4]).permute(0, 2, 1).contiguous()
    src4 = d32["img"][:4].reshape(4, 3, -1).contiguous()
    dbg = lambda: pm.splatter(pts4.clone(), src4, return_debug=True)
    for _ in range(2):
        dbg()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        dbg()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    alg_dbg = (1900544.0 - 262144.0 + 786432.0 + 3 * 4 * 128 * 65536.0) * 4  # points in, + idx/zbuf/dist (S,S,K) out
    res["C2_splat_idx_emitting_b4"] = {"frames_per_s": round(4 / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                                       "algorithmic_GBs": round(alg_dbg / dt / 1e9, 2),
                                       "frac_hbm": round(alg_dbg / dt / 1e9 / HBM_PEAK_GBS, 5)}
    # C3 in reference-faithful mode (sample.py:54-57): one whole-grid forward per sampled code
    eng = m1.outpaint2.engine(32, 32, 1)
    plan = o1["plan"]
    c1 = o1["codes"].reshape(1, 1024).to(torch.int32).contiguous()
    fwd = lambda: eng.forward(c1, plan.mask_init, plan.mask_undilated, plan.mask_dilated)
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        fwd()
    torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / n
    n_s = int(plan.n_sampled[0])
    res["C3_reference_faithful_mode"] = {
        "ms_per_full_forward": round(t_fwd * 1e3, 3), "forward_tflops": round(11.43095e9 / t_fwd / 1e12, 3),
        "frac_mfma_fp32": round(11.43095e9 / t_fwd / 1e12 / FP32_MFMA_PEAK_TF, 4),
        "frames_per_s_extrapolated": round(1.0 / (n_s * t_fwd), 3),
        "note": f"{n_s} sampled codes x one whole-grid forward each (what the reference schedules); the incremental "
                "form above does the work of ONE such forward per frame"}
    # SURVEY 8f row 1, reported separately (not part of the metric): VQ-VAE-2 top level either side of the AR loop for the
    # 16 views of a step -- reprojected view -> top codes (convolutions: csrc/conv_f16x3.hip / conv1x1.hip / vq_ends.hip, quantiser = ps_vq_nearest_f32) and
    # sampled codes -> 256x256 image (ps_vq_embed_f32 + transposed convs), random-init weights
    from pixelsynth_amd.vqvae2 import VQVAETop
    vq = VQVAETop().eval()
    vq.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()})
    vq = vq.to(device)
    d16, _ = make_inputs(2, 16, device, cameras="demo")
    gen16 = pm.forward_justpts(d16["img"], d16["depth"], d16["K"], d16["Kinv"], d16["P"], d16["Pinv"], d16["RT2"], d16["RT2inv"])[0]
    timings = {}
    for name, fn in (("encode_codes", lambda: vq.encode_codes(gen16)), ("decode_code", lambda: vq.decode_code(d16["codes"]))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        timings[name] = (time.perf_counter() - t0) / n
    res["VQVAE_top_16_views"] = {"encode_codes_ms": round(timings["encode_codes"] * 1e3, 3),
                                 "decode_code_ms": round(timings["decode_code"] * 1e3, 3),
                                 "note": "next-row component (SURVEY 8f.1), outside the headline metric"}
    # SURVEY 8f row 2, also outside the metric: depth Unet on the 16 source images and refinement decoder on the 16 blended
    # views (fused noise-affine normalisation; the decoder's 3 x 3 convolutions through csrc/conv_f16x3.hip / conv_thin.hip), synthetic weights
    from pixelsynth_amd.networks import Unet, get_decoder
    nets = {}
    for name, mod in (("unet", Unet(channels_in=3, channels_out=1, opt=syn.network_opts())), ("decoder", get_decoder(syn.network_opts()))):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()})
        nets[name] = mod.to(device).eval()
    bg16 = torch.zeros(16, 256, 256, dtype=torch.bool, device=device)
    bg16[:, :, 160:] = True
    with torch.no_grad():
        for name, fn in (("depth_unet", lambda: nets["unet"](d16["img"])), ("refine_decoder", lambda: nets["decoder"](gen16, bg16))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            timings[name] = (time.perf_counter() - t0) / n
    # the decoder once more with every convolution through torch / MIOpen fp32, and its widest layer (128 -> 128 channels at 256 x 256,
    # three of the thirteen split-fp16 layers, a third of the decoder's arithmetic) alone on the fp16 pipe with HIP events
    from pixelsynth_amd.networks import architectures as arch
    mode = arch.DECODER_CONV
    conv_rec = None
    with torch.no_grad():
        arch.DECODER_CONV = "fp32"
        for _ in range(3):
            nets["decoder"](gen16, bg16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            nets["decoder"](gen16, bg16)
        torch.cuda.synchronize()
        timings["refine_decoder_miopen"] = (time.perf_counter() - t0) / 5
        arch.DECODER_CONV = mode
        arch.check_f16x3_overflow(device)
        if mode == "f16x3":
            conv = nets["decoder"].eblocks[6].ch_a[5]
            xx = torch.randn(16, 128, 256, 256, device=device).contiguous(memory_format=torch.channels_last)
            for _ in range(3):
                arch._f16x3_conv(conv, xx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                arch._f16x3_conv(conv, xx)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            fl = 2 * 9 * 128 * 128 * 256 * 256 * 16
            conv_rec = {"kernel": "k_conv3x3_f16x3 (csrc/conv_f16x3.hip): 128 -> 128 channels, 16 x 256 x 256, fp32 in / out, three fp16 MFMAs per product",
                        "ms": round(ms, 4), "fp32_equivalent_tflops": round(fl / ms / 1e9, 1), "mfma_tflops": round(3 * fl / ms / 1e9, 1),
                        "bound": "mfma", "peak": 2500.0, "unit": "TFLOP/s", "frac": round(3 * fl / ms / 1e9 / 2500.0, 4),
                        "sustained_peak_on_random_operands": 1774.0,
                        "note": "peak = dense fp16 MFMA at 2.4 GHz; under a pure stream of these MFMAs on random operands the chip holds "
                                "1.72 GHz = 1774 TFLOP/s (tools/mfma_f16_clock_probe.hip); MIOpen's fp32 implicit GEMM runs this layer at "
                                "~125 TFLOP/s"}
            del xx
    res["depth_and_refinement_16_views"] = {"depth_unet_ms": round(timings["depth_unet"] * 1e3, 3),
                                            "refine_decoder_ms": round(timings["refine_decoder"] * 1e3, 3),
                                            "refine_decoder_through_miopen_fp32_ms": round(timings["refine_decoder_miopen"] * 1e3, 3),
                                            "decoder_conv": mode, "widest_layer": conv_rec,
                                            "note": "next-row components (SURVEY 8f.2), outside the headline metric; the decoder's 3 x 3 "
                                                    "convolutions hand-written on the fp16 matrix pipe (split operands, fp32-grade results)"}
    # the whole pipeline as ONE timed configuration: what a user gets per frame
    res["end_to_end_16_views"] = end_to_end_config(device, 16)
    res["end_to_end_128_views"] = end_to_end_config(device, 128, steps=2)
    # the reference's own way of rendering a trajectory -- forward_scene, frames chained on one GPU
    # (every frame rendered from the previous one over the accumulated cloud, VQ-VAE in the loop, no sharding possible)
    import types
    from pixelsynth_amd.z_buffermodel import ZbufferModelPts
    o = vars(make_opts()).copy()
    o.update(model_setting="gen_scene", directions=["R", "L"], num_split=4, num_samples=1, sequential_outpainting=False,
             vqvae=True)
    ms = ZbufferModelPts(types.SimpleNamespace(**o)).eval()
    ms.outpaint2.load_state_dict(m1.outpaint2.state_dict())
    ms.vqvae.load_state_dict(vq.state_dict())
    ms = ms.to(device)
    batch = {"images": [d1["img"]], "cameras": [{"K": d1["K"], "Kinv": d1["Kinv"], "P": d1["P"], "Pinv": d1["Pinv"]}],
             "depth_fn": syn.depth_from_image}
    ms(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, outs = ms(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nfr = sum(1 for k in outs if k.startswith("PredImg_"))
    ms.outpaint2.engine(32, 32, 1).check()
    res["chained_scene_R_L_split4"] = {"frames": nfr, "frames_per_s": round(nfr / dt, 3), "ms_per_frame": round(dt / nfr * 1e3, 3),
                                       "note": "forward_scene (z_buffermodel.py:420-584) on one GPU: a state chain, "
                                               "replicas only across GPUs (SURVEY 8e)"}
    return res


def end_to_end_config(device, V, steps=3):
    """What a user of the reference's demo gets per frame (models/z_buffermodel.py:291-419), V views in one pass: depth Unet on
    the source images -> reproject + splat -> VQ-VAE top codes -> AR outpainting -> decode_code -> get_combined -> refinement
    decoder, every network in the loop (random-init weights of the reference's shapes), inputs resident, wall clock around
    synchronised passes.  The Unet's convolutions are MIOpen's, the VQ-VAE's and the decoder's hand-written -- next-row components; the
    hot path of the headline metric is the part `hot_path_ms` times inside this pass."""
    from pixelsynth_amd.z_buffermodel import ZbufferModelPts
    o = vars(make_opts()).copy()
    o.update(vars(syn.network_opts()))
    o.update(vqvae=True, min_z=0.5, max_z=10.0)
    m = ZbufferModelPts(types.SimpleNamespace(**o)).eval()
    m.outpaint2.load_state_dict({k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(0).items()})
    m.vqvae.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()})
    for mod in (m.pts_regressor, m.projector):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()})
    m = m.to(device)
    d, host = make_inputs(0, V, device, cameras="mp3d")
    per = host["per_source"]
    src = d["img"][::per].contiguous()
    view_src = torch.arange(V, device=device) // per
    run = lambda: m.synthesize_views(src, view_src, d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"], temperature=0.7,
                                     uniforms=d["uniforms"], check=False)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m.outpaint2.engine(32, 32, V).check()
    # the parts, each on its own (synchronised, so their sum exceeds the pass by what the pass overlaps)
    parts = {}
    def clock(name, fn, n=3):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        parts[name] = round((time.perf_counter() - t) / n * 1e3, 3)
        return r
    with torch.no_grad():
        depth_src = clock("depth_unet_ms", lambda: torch.sigmoid(m.pts_regressor(src)) * 9.5 + 0.5)
        depth = depth_src[view_src].contiguous()
        planned = clock("reproject_splat_plan_ms", lambda: m.plan_views(d["img"], depth, d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"]))
        codes = clock("vqvae_encode_codes_ms", lambda: m.vqvae.encode_codes(planned["gen_fs"]))
        o2 = clock("ar_outpaint_ms", lambda: m.outpaint_planned(dict(planned), codes, 0.7, d["uniforms"]))
        sample = clock("vqvae_decode_code_ms", lambda: m.vqvae.decode_code(o2["codes"]))
        comb = m.get_combined(o2["gen_fs"], sample, o2["background_mask"])
        clock("refine_decoder_ms", lambda: m.projector(comb, o2["background_mask"]))
    m.outpaint2.engine(32, 32, V).check()
    return {"frames_per_s": round(V / dt, 1), "ms_per_pass": round(dt * 1e3, 3), "views": V, "sources": host["n_src"],
            "sampled_codes_per_view_mean": round(float(np.mean(out["plan"].n_sampled)), 1), "parts_ms": parts,
            "hot_path_ms": round(parts["reproject_splat_plan_ms"] + parts["ar_outpaint_ms"], 3),
            "note": "every network of forward_image in the loop; depth from the (random-init) Unet, so the outpainting region is not "
                    "the headline's; the Unet's convolutions through MIOpen, the VQ-VAE's and the decoder's hand-written (3 x 3 and stride-2 layers on the fp16 pipe with split operands, 1 x 1 and 3-channel layers on the fp32 matrix pipe; SURVEY 8f next rows)"}


def cpu_baseline(host, out, V, budget_s=20.0):
    """The oracle (CPU restatement of the reference path, kind 'port') on this box's host cores, on a
    bounded sample: one view's project+splat+order/masks, plus a few reference-style AR steps (one full
    fp32 network forward per sampled code, models/lmconv/sample.py:54-57) extrapolated to the view's
    number of sampled codes."""
    from oracle import c_oracle, lmconv_oracle as lo
    v = min(V, VIEWS_PER_SOURCE) - 1  # the +0.6 rad view of the first source (largest outpainting region of its sweep)
    t0 = time.perf_counter()
    cam = {k: a[v:v + 1] for k, a in host["cam"].items()}
    sampler = c_oracle.project_pts(host["depth"][v:v + 1], cam["K"], cam["Kinv"], cam["Pinv"], host["RT2"][v:v + 1], 256)
    ref = c_oracle.splat_forward(np.ascontiguousarray(sampler.transpose(0, 2, 1)), host["img"][v:v + 1].reshape(1, 3, -1), 256)
    t_splat = time.perf_counter() - t0
    t0 = time.perf_counter()
    info = c_oracle.masks_for_background(ref["bg"][0], 32)
    t_plan = time.perf_counter() - t0
    n_sampled = int(info["bg32"].sum())
    sd = {k: torch.from_numpy(a) for k, a in syn.pixelcnn_state_dict(0).items()}
    masks = tuple(torch.from_numpy(info[k]) for k in ("mask_init", "mask_undilated", "mask_dilated"))
    codes = torch.from_numpy(host["codes"][v:v + 1])
    x = torch.nn.functional.one_hot(codes, 512).permute(0, 3, 1, 2).float()
    # the CPU path run sensibly: MKL oversubscribes on 1024-location tensors (128 threads: 540 ms per forward where 8 cores
    # of the survey box took 105 ms), so the thread count is swept and the best one is what is reported
    all_threads = int(torch.get_num_threads())
    sweep = {}
    with torch.no_grad():
        for nt in sorted({t_ for t_ in (4, 8, 16, 32, 64, all_threads) if t_ <= all_threads}):
            torch.set_num_threads(nt)
            lo.pixelcnn_forward(sd, x, *masks)  # warm-up
            t0 = time.perf_counter()
            for _ in range(3):
                lo.pixelcnn_forward(sd, x, *masks)
            sweep[nt] = (time.perf_counter() - t0) / 3
        best_nt = min(sweep, key=sweep.get)
        torch.set_num_threads(best_nt)
        t0 = time.perf_counter()
        n = 0
        while n < 8 or (time.perf_counter() - t0 < budget_s and n < 64):
            lo.pixelcnn_forward(sd, x, *masks)
            n += 1
        t_step = (time.perf_counter() - t0) / n
        torch.set_num_threads(all_threads)
    frame_s = t_splat + t_plan + n_sampled * t_step
    mean_sampled = float(np.mean(out["plan"].n_sampled))
    frame_mean_s = t_splat + t_plan + mean_sampled * t_step
    return {"value": round(1.0 / frame_mean_s, 5), "unit": "frames/s", "cores": best_nt,
            "kind": "port",
            "sample": (f"1 of {V} views on the host: oracle project+splat {t_splat:.3f}s (C/OpenMP) + order/masks "
                       f"{t_plan * 1e3:.1f}ms + {n} reference-style AR steps at {t_step * 1e3:.1f} ms/step (one full fp32 "
                       f"forward per sampled code, torch CPU on {best_nt} threads -- the best of the sweep), extrapolated to the "
                       f"sweep's mean of {mean_sampled:.0f} sampled codes/view (the +0.6 rad view alone: {n_sampled} codes, "
                       f"{frame_s:.1f} s/frame)"),
            "thread_sweep_ms_per_forward": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--views", type=int, default=128, help="independent novel views per GPU per step (128 = C5: 8 sources x 16 views)")
    ap.add_argument("--total-views", type=int, default=0, help="STRONG scaling: the job is this many views in TOTAL, dealt round-robin over the "
                    "ranks (C5 proper on 8 GPUs: --total-views 128 = 16 per GPU); overrides --views")
    ap.add_argument("--trajectory", choices=["sweep", "circle"], default="sweep", help="circle = config C4: one source, --frames poses of the "
                    "'C' circle (demo cameras), every frame rendered from the source, frames dealt round-robin over the ranks (strong scaling)")
    ap.add_argument("--frames", type=int, default=64, help="--trajectory circle: frames of the circle in total")
    ap.add_argument("--cameras", choices=["mp3d", "demo"], default=None, help="Matterport-shaped (C5, default) or demo / RealEstate10K-shaped inputs (default for the circle)")
    ap.add_argument("--depth", choices=["smooth", "uniform"], default="smooth")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC record only (no rocprofv3 --pmc passes)")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 single-view / C2 splat-only side measurements")
    ap.add_argument("--live-pmc", action="store_true", help="the rocprofv3 --pmc passes of this command even with --no-extra")
    ap.add_argument("--dump-gather", metavar="NPZ", help="rank 0 saves what the last step gathered from all ranks (tests)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...")
    # PS_BENCH_DRYRUN_ONE_GPU=1: every rank uses cuda:0 and the gloo backend -- exercises the multi-rank control
    # flow on a single-GPU box (the numbers of such a run mean nothing)
    dry = os.environ.get("PS_BENCH_DRYRUN_ONE_GPU") == "1" or os.environ.get("PS_DRYRUN_ONE_GPU") == "1"
    if dry:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or FORCE_COLLECTIVE:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    circle = args.trajectory == "circle"
    if args.cameras is None:
        args.cameras = "demo" if circle else "mp3d"
    total = args.frames if circle else args.total_views
    strong = total > 0
    if strong and total % world:
        raise SystemExit(f"{total} views do not deal evenly over {world} ranks")
    V = total // world if strong else args.views
    model = build_model(device)
    if strong:   # one job for all ranks (seeded as rank 0's), this rank's round-robin share of it
        d, host = make_inputs(0, total, device, smooth=args.depth == "smooth", cameras=args.cameras, trajectory=args.trajectory,
                              ids=D.shard_views(total, rank, world))
    else:
        d, host = make_inputs(rank, V, device, smooth=args.depth == "smooth", cameras=args.cameras)
    def barrier():
        D.barrier()
        torch.cuda.synchronize()

    side = side_stream()
    steps_fn = lambda k: run_steps(model, d, world, k, side)
    out = steps_fn(args.warmup) if args.warmup > 0 else None
    barrier()
    t0 = time.perf_counter()
    out = steps_fn(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if HOST_TIMES:   # the timed steps' host side: enqueueing the AR run (back), the next step's front, both in ms from the step's start
        for (a0, a1, a2), nxt in zip(HOST_TIMES[-args.steps:], HOST_TIMES[-args.steps + 1:] + [None]):
            print(f"host: back {1e3 * (a1 - a0):6.2f} ms, front of the next step {1e3 * (a2 - a1):6.2f} ms"
                  + (f", iteration {1e3 * (nxt[0] - a0):6.2f} ms" if nxt else ""), file=sys.stderr)
    model.outpaint2.engine(32, 32, V).check()  # (outside the timed region) no column launch gave up on an in-launch wait
    elapsed = D.max_over_ranks(dt, None if dry else device, force_collective=FORCE_COLLECTIVE)

    if rank == 0 and args.dump_gather and (world > 1 or FORCE_COLLECTIVE):
        np.savez_compressed(args.dump_gather, all_codes=out["all_codes"].cpu().numpy(), all_features_u8=out["all_features_u8"].cpu().numpy())
    if rank == 0:
        frames = V * world * args.steps   # (strong: V * world = the job's total)
        plan = out["plan"]
        res = {
            "metric": "novel-view frames/sec @256x256 (reproject+AR outpaint)",
            "value": round(frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ((f"C4: the {total}-frame 'C' circle from one source, {V} frame(s) per GPU per step (every frame rendered from the source, "
                                     if circle else
                                     f"{'C5' if total == 128 else 'C5-shaped job'}: {total} views in total = {host['n_src']} source image(s) x {host['per_source']} novel views, "
                                     f"dealt round-robin, {V} per GPU per step (yaw sweep +-0.6 rad per source, "
                                     if strong else
                                     f"{'C5 on every GPU' if V == 128 and args.cameras == 'mp3d' else 'C5-shaped batch on every GPU'}: {host['n_src']} source image(s) x "
                                     f"{host['per_source']} independent novel views = {V} views per GPU per step (yaw sweep +-0.6 rad per source, ")
                                    + ("Matterport-shaped cameras K = diag(1/tan(hfov/2)) at hfov 90 deg, " if args.cameras == "mp3d"
                                       else "demo / RealEstate10K-shaped cameras, ")
                                    + f"256x256 RGB features, {args.depth} depth {'0.5..10' if args.cameras == 'mp3d' else '1..100'}, "
                                    "K=128 r=4 alphacomposite splat, 13x13 mask dilation, "
                                    "custom generation order, exact incremental AR over the 32x32 code grid, T=0.7)"),
                       "sources_per_gpu": host["n_src"],
                       "views_per_gpu": V, "image": "256x256", "code_grid": "32x32", "num_classes": 512,
                       "ar_steps_walked": 1024 - plan.first_step,
                       "sampled_codes_per_view_mean": round(float(np.mean(plan.n_sampled)), 1),
                       "parallelism": f"views sharded over {world} GPU(s), RCCL all_gather of the reprojected views (8-bit) + completed code grids",
                       "step_pipeline": "the AR run on one stream, its whole-grid prefix pass dealt to two frame ranges on two streams; the host half of "
                                        "step i + 1 (splat, planning, uploads) on a side stream"
                                        + ("; up to four steps' AR runs in flight in one engine handle, every column launch taking what is left of each "
                                           "batch's current wavefront (the timed region ends with everything flushed: exactly `steps` complete steps); "
                                           "the whole-grid pass takes every frame up to ITS first sampled position (per-frame prefixes), the columns "
                                           "start there" if ar_pipelined(V) else "")},
        }
        # which library and which switches produced the number: a tuning / trace / experiment build (PS_HIP_LIB=...) says so itself
        res["library"] = {"path": os.path.relpath(_lib.LIB_PATH, os.path.dirname(os.path.abspath(__file__))), "build": _lib.lib().ps_build_info().decode(),
                          "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith("PS_") and k not in ("PS_BENCH_PMC_CHILD",)}}
        if torch.distributed.is_available() and torch.distributed.is_initialized():   # what the backend itself reports (tools/scale.sh)
            res["collective"] = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                                 "forced_on_one_rank": bool(FORCE_COLLECTIVE and world == 1)}
        if world == 1 and os.environ.get("PS_BENCH_PMC_CHILD") == "1":
            pass      # (a counter pass of live_pmc_traffic: only the timed steps' kernels are wanted in it)
        elif world == 1:
            try:
                res["roofline"] = measure_roofline(model, d, out, V, live_pmc=(not args.no_extra or args.live_pmc) and not args.no_live_pmc)
                # the step as a whole against the same peak: every view is ONE whole-grid forward's worth of matrix work (SURVEY 8d:
                # 11.43095 GFLOP, prefix pass + columns), whatever it is scheduled as; splat, planning and launch gaps count as time
                step_tf = V * 11.43095e9 / (elapsed / args.steps) / 1e12
                res["roofline"]["step"] = {"what": "views x 11.43095 GFLOP (one whole-grid forward's worth per view: SURVEY 8d's DENSE definition -- closed "
                                                   "taps, the prefix cone and the u_init gather are work the kernels legitimately skip, so this is a "
                                                   "dense-equivalent rate, NOT a utilisation; `executed_frac` is) / ms_per_step, the timed step with "
                                                   "everything in it", "achieved": round(step_tf, 3), "peak": FP32_MFMA_PEAK_TF,
                                           "unit": "TFLOP/s", "frac": round(step_tf / FP32_MFMA_PEAK_TF, 4)}
                try:
                    rows, dominant, executed, phases = measure_kernels(model, d, V, world, side, plan, res["roofline"].get("pmc_live"),
                                                               elapsed / args.steps * 1e3)
                    res["roofline"]["kernels"] = rows
                    res["roofline"]["kernels_note"] = ("event-timed per launch on the launch's own stream; the whole-grid pass runs as two frame ranges on two "
                                                       "streams, so two k_gemm_ws launches share the chip and stretch each other: their ms_per_step sum to more "
                                                       "than the phase's wall time and their frac_* are per-launch figures under that sharing -- the phase's own "
                                                       "busy fraction is `phases.prefix_pass_executed_frac`")
                    res["roofline"]["phases"] = phases
                    res["roofline"]["kernel"] = (f"{dominant}: the largest phase of the step by summed kernel time (`kernels`, sorted); the top-level "
                                                 "achieved / frac / avg_launch_us are the COLUMN launch's (`column_launch_kernel`), as in every round")
                    if executed is not None:
                        ex_tf = executed / (elapsed / args.steps) / 1e12
                        res["roofline"]["step"]["executed_frac"] = round(ex_tf / FP32_MFMA_PEAK_TF, 4)
                        res["roofline"]["step"]["executed_what"] = ("MFMA instructions of all matrix kernels of a step (this run's SQ_INSTS_MFMA pass x launches "
                                                                    "per step) x 2048 flop / ms_per_step / peak: the share of the fp32 matrix peak the step "
                                                                    "really keeps busy (the latency form's centre taps run on the vector ALU and are not in it)")
                except Exception as e:
                    res["roofline"]["kernels"] = {"error": repr(e)}
            except Exception as e:  # measurement aid must not sink the headline number
                res["roofline"] = {"error": repr(e)}
            if not args.no_cpu_baseline:
                try:
                    res["cpu_baseline"] = cpu_baseline(host, out, V)
                except Exception as e:
                    res["cpu_baseline"] = {"error": repr(e)}
            if not args.no_extra:
                try:
                    res["other_single_gpu_configs"] = extra_configs(device)
                except Exception as e:
                    res["other_single_gpu_configs"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if world > 1 or FORCE_COLLECTIVE:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
