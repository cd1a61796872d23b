"""Worker of tests/test_rccl_gpu.py: ONE rank, backend "nccl" (= RCCL on ROCm), every collective of pixelsynth_amd.distributed
on device tensors -- no gloo, no host staging.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelsynth_amd import distributed as D  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    res = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    try:
        frames = torch.rand(16, 3, 64, 64, device=dev) * 2 - 1
        u8 = D.to_image_u8(frames)
        codes = torch.randint(0, 512, (16, 1024), dtype=torch.int32, device=dev)
        # the asynchronous form bench.py pipelines: two collectives in flight, work on the current stream meanwhile, collected later
        pend_f = D.gather_frames_start(u8, force_collective=True)
        pend_c = D.gather_frames_start(codes, 16, force_collective=True)
        res["went_through_backend"] = pend_f.bufs is not None and pend_f.work is not None and pend_f.staged.is_cuda
        busy = torch.zeros(1024, 1024, device=dev)
        for _ in range(8):
            busy = busy @ busy + 1.0                      # something on the current stream the collectives run beside
        got_c, got_f = pend_c.result(), pend_f.result()   # Work.wait(): the current STREAM waits, the host does not
        tail = (got_f.float().sum() + got_c.float().sum()).item()   # consumed on the current stream, after the wait
        res["rows_equal"] = bool(torch.equal(got_f, u8) and torch.equal(got_c, codes) and got_f.is_cuda and np.isfinite(tail))
        res["sync_form_equal"] = bool(torch.equal(D.gather_frames(codes, 16, force_collective=True), codes))
        img = torch.rand(1, 3, 32, 32, device=dev)
        b = D.broadcast_from(img, 0, dev, force_collective=True)
        res["broadcast_equal"] = bool(torch.equal(b, img) and b.is_cuda)
        disc, entr = D.gather_scores([0.25, -1.5, 3.0], [1.0, 2.0, 0.5], 3, force_collective=True)
        res["scores_equal"] = bool(np.array_equal(disc, [0.25, -1.5, 3.0]) and np.array_equal(entr, [1.0, 2.0, 0.5]))
        res["max_over_ranks"] = D.max_over_ranks(0.75, dev, force_collective=True)
        D.barrier()
        torch.cuda.synchronize()
        with open("/proc/self/maps") as fh:
            res["librccl_mapped"] = any("librccl" in ln for ln in fh)
    finally:
        dist.destroy_process_group()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
