"""Multi-GPU sharding of the novel-view path: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The path shards without any exchange during compute: independent (source, target-view)
pairs are dealt round-robin to the ranks, every rank holds its own copy of the (small) weights, and the
only collective is the gather of finished frames.  Nothing here is specific to the GPU backend: the same
code runs under "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


import contextlib
import os


@contextlib.contextmanager
def shared_device_turn():
    """Column launches keep one workgroup per compute unit resident and wait, inside the launch, for each other (chain workgroups
    for neighbour slots, the look-ahead items of the neighbour role for the chains' published stores).  That is safe for ONE
    column-launching process per GPU -- the deployment: one process per GPU -- next to kernels that never wait (splat, GEMMs, other
    frameworks' kernels).  Two processes whose column launches meet on the same GPU can hold each other's compute units until the
    bounded waits give up (ps_pixelcnn_status then reports it).  The only place ranks share a GPU here is the single-GPU dry run of
    the multi-rank control flow (PS_DRYRUN_ONE_GPU=1, tests): there the ranks take turns -- a file lock around the AR run, released
    when the device is idle again."""
    if os.environ.get("PS_DRYRUN_ONE_GPU") != "1" and os.environ.get("PS_BENCH_DRYRUN_ONE_GPU") != "1":
        yield
        return
    import fcntl
    import tempfile
    with open(os.path.join(tempfile.gettempdir(), "pixelsynth_dryrun_gpu.lock"), "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
            torch.cuda.synchronize()
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_views(n_views, rank=None, world_size=None):
    """Indices of the views rank `rank` renders: r, r+W, r+2W, ...  (the reference renders them one by one on
    a single GPU: demo.py:247-251; docs/REALESTATE.md:74 recommends manual splits across GPUs)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    return list(range(rank, n_views, world_size))


class PendingGather:
    """An all_gather of finished frames under way (gather_frames_start); result() waits for it -- on the device, for the current
    stream, with RCCL -- and puts the rows in view order."""

    def __init__(self, local, n_views, work=None, bufs=None, staged=None):
        self.local, self.n_views, self.work, self.bufs, self.staged = local, n_views, work, bufs, staged

    def result(self):
        local = self.local
        if self.bufs is None:
            return local if self.n_views is None else local[:self.n_views]
        if self.work is not None:
            self.work.wait()
            self.work = None
        bufs = self.bufs
        if self.staged.device != local.device:
            bufs = [b.to(local.device) for b in bufs]
        stacked = torch.stack(bufs, 1)                      # (V_local, W, ...): row v_local*W + r = view index
        out = stacked.reshape(-1, *local.shape[1:])
        return out if self.n_views is None else out[:self.n_views]


def _collective(force):
    """Does a call go through the backend?  With more than one rank always; on a single rank only when asked to (`force_collective`:
    the one-rank RCCL execution of the -m gpu tests and of `PS_BENCH_FORCE_COLLECTIVE=1 python bench.py` -- the same calls, streams
    and waits a multi-GPU job makes, on the one GPU a test box has) and a process group exists."""
    return world()[1] > 1 or (force and dist.is_available() and dist.is_initialized())


def gather_frames_start(local, n_views=None, force_collective=False):
    """Start the all_gather of the finished frames and return at once (-> PendingGather): the collective runs on the backend's own
    stream, behind what the current stream has enqueued so far, and whatever the caller enqueues next -- the following batch's
    whole-grid pass -- runs beside it.  It must be COLLECTED (PendingGather.result(): the current stream waits for it) before the
    caller's next column launch is enqueued: a column launch keeps one workgroup per compute unit resident, and its in-launch
    waits are bounded -- a collective kernel still holding compute units when the launch starts would stall it.  bench.py does so
    through outpaint_planned(between=...), between the next step's prefix pass and its first column launch.
    local as for gather_frames."""
    rank, w = world()
    if not _collective(force_collective):
        return PendingGather(local, n_views)
    staged = local.contiguous()
    if staged.is_cuda and dist.get_backend() == "gloo":
        staged = staged.cpu()  # gloo has no device all_gather: host staging (tests / single-GPU dry runs only)
    bufs = [torch.empty_like(staged) for _ in range(w)]
    work = dist.all_gather(bufs, staged, async_op=True)
    return PendingGather(local, n_views, work, bufs, staged)


def gather_frames(local, n_views=None, force_collective=False):
    """all_gather of the finished frames.  local: (V_local, ...) tensor, same V_local on every rank (pad the
    last round when n_views is not a multiple of the world size).  Returns (W*V_local, ...) ordered by VIEW
    index when the views were dealt with shard_views (view v sits at row v); trimmed to n_views if given."""
    return gather_frames_start(local, n_views, force_collective).result()


def to_image_u8(frames):
    """Finished frames in [-1, 1] -> 8-bit images (what the reference's demo writes to disk, demo.py:100-178): a quarter
    of the bytes on the wire for the gather (786 KB -> 197 KB per 256x256 frame)."""
    return ((frames.clamp(-1.0, 1.0) * 0.5 + 0.5) * 255.0 + 0.5).to(torch.uint8)


def max_over_ranks(seconds, device=None, force_collective=False):
    """Slowest rank's wall time (bench.py contract: take the MAX over ranks)."""
    rank, w = world()
    if not _collective(force_collective):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def owner_of(index, world_size=None):
    """rank that rendered view / candidate `index` under shard_views' round-robin deal"""
    if world_size is None:
        world_size = world()[1]
    return index % world_size


def broadcast_from(tensor, src, device, force_collective=False):
    """Broadcast `tensor` from rank `src` to every rank; the other ranks pass None (or anything) and learn shape and dtype
    from the owner first -- a rank that holds no candidate of its own cannot know the shape of a decoded / refined image
    (it is not gen_fs's when the features are not RGB).  -> the tensor, on `device`, on every rank."""
    rank, w = world()
    if not _collective(force_collective):
        return tensor
    gloo = dist.get_backend() == "gloo"
    meta = torch.zeros(8, dtype=torch.int64)
    if rank == src:
        meta[0] = tensor.dim()
        meta[1:1 + tensor.dim()] = torch.tensor(tensor.shape, dtype=torch.int64)
        meta[7] = {torch.float32: 0, torch.uint8: 1, torch.float16: 2, torch.bfloat16: 3}[tensor.dtype]
    meta = meta if gloo else meta.to(device)
    dist.broadcast(meta, src=src)
    meta = meta.cpu()
    shape = tuple(int(x) for x in meta[1:1 + int(meta[0])])
    dtype = (torch.float32, torch.uint8, torch.float16, torch.bfloat16)[int(meta[7])]
    if rank == src:
        buf = tensor.contiguous()
        buf = buf.cpu() if gloo else buf
    else:
        buf = torch.empty(shape, dtype=dtype, device="cpu" if gloo else device)
    dist.broadcast(buf, src=src)
    return buf.to(device)


def gather_scores(disc_local, entr_local, n, force_collective=False):
    """Sample ranking across ranks: every rank scored the candidates shard_views(n) gave it -- discriminator score
    and classifier entropy, two scalars each -- and all ranks need all n of both to apply the rank rule.  One all_gather of a
    (2, ceil(n / W)) float64 block per rank.  -> (disc (n,), entr (n,)) numpy arrays, by candidate index."""
    import numpy as np
    rank, w = world()
    per = (n + w - 1) // w
    block = torch.zeros(2, per, dtype=torch.float64)
    block[0, :len(disc_local)] = torch.as_tensor(list(disc_local), dtype=torch.float64)
    block[1, :len(entr_local)] = torch.as_tensor(list(entr_local), dtype=torch.float64)
    if not _collective(force_collective):
        return block[0, :n].numpy().copy(), block[1, :n].numpy().copy()
    if dist.get_backend() != "gloo":
        block = block.cuda()
    bufs = [torch.empty_like(block) for _ in range(w)]
    dist.all_gather(bufs, block)
    stacked = torch.stack([b.cpu() for b in bufs], 2)            # (2, per, W): candidate k * W + r
    flat = stacked.reshape(2, -1)[:, :n].numpy()
    return flat[0].copy(), flat[1].copy()
