"""Batches of novel views, software-pipelined over two compute-unit partitions of one MI355X.

One batch = reproject + splat, plan, whole-grid pass over the observed prefix (the "prefix pass"), column launches.
The column launches are bound by the latency of their 33 dependent stages, not by throughput: at 128 views a launch
keeps about a third of the chip idle.  The prefix pass is pure matrix throughput.  So while the
column launches of batch i run on compute units [0, cus_main) -- stream A --, the reprojection / splat / planning of
batch i + 1 and most of its prefix pass run on the other compute units -- stream B; the rest of that prefix pass
follows batch i's columns on A.  Both streams are confined with a compute-unit mask (ps_stream_create_cu_range): a
column launch keeps one workgroup per compute unit resident and must not find its compute units taken by another
stream's workgroups (which is also why round 2's unmasked side stream made an in-launch wait expire now and then).

MEASURED (round 3, C5's 128 views per batch): 24.4 ms per batch against 22.3 ms for the one-stream pipeline of bench.py -- the
column launches run at 200-217 us on 160 compute units where their share of the chip predicts 167, and stream B is saturated by
the splat and 60 % of a prefix pass.  An experiment that LOST (round 3): moved out of the package in round 4; it needs
`columns_on` support in ZbufferModelPts.outpaint_planned, which went with it (git history: round 3, pixelsynth_amd/pipeline.py).

Known weaknesses left as they were when the experiment lost (round-3 advice): the slot's engine may be rebuilt before the wait for
batch k - 2's column launches (safe only through hipFree's implicit synchronisation), and tensors allocated on streams A / B go back
to the caller's stream without record_stream.  Fix both before reviving it.

The batches are independent (the same work as `outpaint_views` batch by batch, bit-identical results: the prefix pass of
disjoint frame ranges is independent, and each batch has its own engine handle, i.e. its own activation caches).
There is no reference counterpart: the reference renders one view at a time (demo.py:247-251).
"""
import ctypes

import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pixelsynth_amd import _lib  # noqa: E402


from pixelsynth_amd.lmconv.model import CuRangeStream, VALIDATED_SPLITS  # noqa: F401


class OverlappedOutpainter:
    """model: ZbufferModelPts.  cus_main: compute units of stream A (the column launches; a multiple of 8); the others are
    stream B's.  prefix_share: fraction of a batch's frames whose prefix pass runs on B (beside the previous batch's column
    launches); the remaining frames' follows on A."""

    def __init__(self, model, cus_main=160, prefix_share=0.6, device=None):
        self.model = model
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        total = torch.cuda.get_device_properties(self.device).multi_processor_count
        if not (16 <= cus_main < total and cus_main % 8 == 0):
            raise ValueError(f"cus_main={cus_main}: a multiple of 8 in [16, {total})")
        self.cus_main, self.total = cus_main, total
        self.prefix_share = float(prefix_share)
        self.A = CuRangeStream(0, cus_main, self.device)
        self.B = CuRangeStream(cus_main, total - cus_main, self.device)
        self._engines = {}

    def close(self):
        for eng in self._engines.values():
            eng.set_compute_units(0)
        self.A.close()
        self.B.close()

    def _engine(self, slot, V):
        G = self.model.obs[1]
        eng = self.model.outpaint2.engine(G, self.model.obs[2], V, slot=slot)
        eng.set_compute_units(self.cus_main)          # (its column launches go to stream A)
        self._engines[slot] = eng
        return eng

    @torch.no_grad()
    def run(self, batches, temperature=0.7, after=None):
        """batches: list of dicts with the arguments of plan_views (img, depth, K, Kinv, P, Pinv, RT2, RT2inv) plus `codes`
        (V,32,32) or None (VQ-VAE top codes of the reprojected view) and `uniforms` (V,L) or None.  after(i, out): called on
        stream A once batch i's columns are enqueued (e.g. the gather of its results).  -> list of outputs as from
        outpaint_planned (gen_fs, background_mask, plan, codes), all work enqueued; the caller synchronises."""
        m = self.model
        A, B = self.A.stream, self.B.stream
        G, L = m.obs[1], m.obs[1] * m.obs[2]
        cur = torch.cuda.current_stream()
        A.wait_stream(cur)
        B.wait_stream(cur)
        outs = []
        cols_done = {}                               # batch -> event on A: its column launches are through (its engine is free again)

        def front_and_prefix(k):
            """batch k: front on B, its prefix pass split over B (first frames, now) and A (the rest, in A's order)."""
            d = batches[k]
            with torch.cuda.stream(B):
                planned = m.plan_views(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])
                plan, gen_fs = planned["plan"], planned["gen_fs"]
                V = gen_fs.shape[0]
                codes = d.get("codes")
                if codes is None:
                    codes = m.vqvae.encode_codes(gen_fs)
                c32 = codes.reshape(V, L).to(torch.int32).contiguous().clone()
                uniforms = d.get("uniforms")
                if uniforms is None:
                    uniforms = torch.rand(V, L, device=gen_fs.device, dtype=torch.float32)
                eng = self._engine(k & 1, V)
                if k - 2 in cols_done:               # this engine's caches were last read by batch k - 2's column launches
                    B.wait_event(cols_done.pop(k - 2))
                nb = min(V, max(0, int(round(self.prefix_share * V))))
                ready = torch.cuda.Event()
                ready.record(B)                      # plan, codes, uniforms are on the device
                if nb > 0:
                    eng.ar_prefix(c32, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated,
                                  plan.first_step, 0, nb)
                done_b = torch.cuda.Event()
                done_b.record(B)
            for t in (gen_fs, planned["background_mask"], plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated,
                      plan.mask_dilated, plan.waves[0], c32, uniforms):
                if t.numel():
                    t.record_stream(A)
            with torch.cuda.stream(A):
                A.wait_event(ready)
                if nb < V:
                    eng.ar_prefix(c32, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated,
                                  plan.first_step, nb, V)
            return dict(planned=planned, c32=c32, uniforms=uniforms, eng=eng, done_b=done_b, V=V)

        nxt = front_and_prefix(0) if batches else None
        for k in range(len(batches)):
            st = nxt
            plan = st["planned"]["plan"]
            with torch.cuda.stream(A):
                A.wait_event(st["done_b"])           # B's share of this batch's prefix pass
                if plan.waves[0].shape[0]:
                    st["eng"].ar_columns(st["c32"], plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated,
                                         plan.mask_dilated, plan.waves, temperature=temperature, uniforms=st["uniforms"],
                                         first_step=plan.first_step)
                cols_done[k] = torch.cuda.Event()
                cols_done[k].record(A)
                out = dict(st["planned"])
                out["codes"] = st["c32"].view(st["V"], G, m.obs[2])
                if after is not None:
                    after(k, out)
            outs.append(out)
            nxt = front_and_prefix(k + 1) if k + 1 < len(batches) else None
        cur.wait_stream(A)
        cur.wait_stream(B)
        return outs

    def check(self):
        for eng in self._engines.values():
            eng.check()
