"""Driver of the novel-view path (SURVEY 8b "what calls it"): the counterpart of the reference's demo.py:181-270 /
create_vid.py for what this repository builds.

    python -m pixelsynth_amd.driver --trajectory circle --frames 64 --out results/      (one GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m pixelsynth_amd.driver --trajectory circle --frames 64 --out results/         (C4: views sharded over 8 GPUs)
    python -m pixelsynth_amd.driver --scene R L --num-split 4 --out results/             (chained, as the reference's gen_scene)

One source image (a PNG, or the synthetic RealEstate10K-shaped sample), the demo cameras of process_demo_data
(demo.py:36-96), target poses from ZbufferModelPts.get_rt_from_rot (directions 'R','L','U','D',... in `--frames`
equal steps, or the 'C' circle of z_buffermodel.py:217-225).  Every view is rendered independently FROM THE SOURCE
(the reference's demo chains frames on one GPU): reproject + splat, VQ-VAE top codes, AR outpainting,
decode, get_combined.  Ranks take views round-robin, finished frames are all-gathered (RCCL), rank 0 writes
<out>/video/%d.png in trajectory order -- the layout create_vid.py's ffmpeg call expects (demo.py:125-164).

--scene runs ZbufferModelPts.forward_scene instead (z_buffermodel.py:420-584): frames chained on ONE GPU, each rendered
from the previous generated frame on top of the accumulated point cloud; images go to <out>/scene/ and <out>/video/ in
the reference's save_scene / save_video layout (demo.py:100-164).  A chain does not shard: with several ranks, rank r
renders its own chain for direction list r (replicas).

The depth regressor, the refinement decoder and trained weights are not part of this repository: depth
is synthetic unless --depth-npy is given, weights are random-init unless --pixelcnn / --vqvae state dicts are given,
and the saved image is the un-refined composite (reprojected features where visible, decoded sample elsewhere).
"""
import argparse
import os
import types

import numpy as np
import torch

from . import distributed as D, synthetic as syn


def make_opts(**kw):
    o = dict(W=256, use_rgb_features=True, splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=128,
             tau=1.0, rad_pow=2, accumulation="alphacomposite", background_smoothing_kernel_size=13, min_z=1.0, max_z=100.0,
             rotation=0.6, direction="R", temperature=0.7, model_setting="gen_scene", seed=0, homography=False, vqvae=True)
    o.update(kw)
    return types.SimpleNamespace(**o)


def build_model(device, pixelcnn_sd=None, vqvae_sd=None):
    from .z_buffermodel import ZbufferModelPts
    model = ZbufferModelPts(make_opts()).eval()
    load = lambda path, fallback: torch.load(path, map_location="cpu") if path else {k: torch.from_numpy(v) for k, v in fallback.items()}
    model.outpaint2.load_state_dict(load(pixelcnn_sd, syn.pixelcnn_state_dict(0)))
    model.vqvae.load_state_dict(load(vqvae_sd, syn.vqvae_state_dict(0)))
    return model.to(device)


def trajectory(model, input_RT, kind, n):
    """-> list of (label, RTinv (1,4,4), RT (1,4,4)) target poses, in playback order."""
    poses = []
    for i in range(n):
        if kind == "circle":
            inv, rt = model.get_rt_from_rot("C", input_RT, i, n)
            poses.append((f"C_{i}", inv, rt))
        else:
            inv, rt = model.get_rt_from_rot(kind, input_RT, i + 1, n)
            poses.append((f"{kind}_{i + 1}", inv, rt))
    return poses


@torch.no_grad()
def render_views(model, img, depth, cam, poses, temperature=0.7, seed=0):
    """img (1,3,S,S) in [-1,1], depth (1,1,S,S), cam dict of (1,4,4) tensors, poses as from trajectory()
    -> dict(frames (V,3,S,S), features, background_mask, codes): the views of `poses`, batched through outpaint_views."""
    V = len(poses)
    rep = lambda t: t.expand(V, *t.shape[1:]).contiguous()
    RT2 = torch.cat([p[2] for p in poses]).contiguous()
    RT2inv = torch.cat([p[1] for p in poses]).contiguous()
    g = torch.Generator(device="cpu").manual_seed(seed)
    uniforms = torch.rand(V, 1024, generator=g).to(img.device)
    out = model.outpaint_views(rep(img), rep(depth), rep(cam["K"]), rep(cam["Kinv"]), rep(cam["P"]), rep(cam["Pinv"]), RT2, RT2inv,
                               None, temperature=temperature, uniforms=uniforms)
    model.outpaint2.engine(32, 32, V).check()
    sample = model.vqvae.decode_code(out["codes"])
    frames = model.get_combined(out["gen_fs"], sample, out["background_mask"])
    return dict(frames=frames, features=out["gen_fs"], background_mask=out["background_mask"], codes=out["codes"])


def scene_outputs_to_disk(outputs, directions, num_split, out_dir):
    """PredImg_<dir>_<i> of forward_scene -> scene/output_image_<dir>_%04d.png (demo.py:100-123) and video/%d.png in
    playback order: out along each direction, and back again for the rotational ones (demo.py:125-164).
    -> number of video frames written."""
    def splits(d):
        return num_split * 2 if d in ("S", "C") else max(num_split // 2, 1) if d in ("U", "D", "UL", "UR", "DR", "DL") else num_split
    scene, vid = os.path.join(out_dir, "scene"), os.path.join(out_dir, "video")
    os.makedirs(scene, exist_ok=True)
    os.makedirs(vid, exist_ok=True)
    for d in directions:
        if d in ("S", "C"):
            continue
        for i in range(1, splits(d) + 1):
            save_png(os.path.join(scene, "output_image_%s_%04d.png" % (d, i)), outputs[f"PredImg_{d}_{i}"][0])
    save_png(os.path.join(vid, "0.png"), outputs[f"PredImg_{directions[0]}_0"][0])
    n = 1
    for d in directions:
        order = list(range(1, splits(d)))
        if d not in ("S", "C"):
            order += list(range(splits(d) - 1, -1, -1))
        for i in order:
            save_png(os.path.join(vid, f"{n}.png"), outputs[f"PredImg_{d}_{i}"][0])
            n += 1
    return n


_SIDE = {}


def _side_stream():
    """ONE side stream per device and process.  The runtime deals the streams a process creates onto a few hardware queues in turn,
    and one of every eight shares the main stream's queue -- its kernels then run behind the main stream's instead of beside them
    (docs/LAB_NOTEBOOK.md, "Which stream the side stream is": +25 % per step).  A stream created once, early, is the same stream on every call."""
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream()
    return _SIDE[dev]


@torch.no_grad()
def render_pipelined(model, img, depth, cam, chunks, seeds, temperature=0.7):
    """render_views for several batches of poses (seeds: per batch, one seed per view), with the host half of batch i + 1 (splat on a side stream, masks back,
    orders / masks / wavefront schedule, uploads) overlapped with the AR run of batch i.  -> list of frames (V_i,3,S,S)."""
    main, side = torch.cuda.current_stream(), _side_stream()

    def inputs(chunk):
        V = len(chunk)
        rep = lambda t: t.expand(V, *t.shape[1:]).contiguous()
        return (rep(img), rep(depth), rep(cam["K"]), rep(cam["Kinv"]), rep(cam["P"]), rep(cam["Pinv"]),
                torch.cat([p[2] for p in chunk]).contiguous(), torch.cat([p[1] for p in chunk]).contiguous())

    frames, planned = [], None
    # batches of one size (a trajectory cut into equal chunks): their AR runs overlap -- the narrow last wavefronts of a batch inside the
    # launches of the next batch's first ones (outpaint_pipelined; the same codes); a batch then comes back one call late
    overlap = len(chunks) > 1 and len({len(c) for c in chunks}) == 1 and len(chunks[0]) >= 2

    def finish(out):
        if out is not None:
            sample = model.vqvae.decode_code(out["codes"])
            frames.append(model.get_combined(out["gen_fs"], sample, out["background_mask"]))
    try:
        for k, chunk in enumerate(chunks):
            if planned is None:
                planned = model.plan_views(*inputs(chunk))
            V = len(chunk)
            # the draws of a view are seeded by the VIEW (its index in the trajectory), not by where the sharding put it: a frame is
            # the same picture on one GPU or eight
            uniforms = torch.stack([torch.rand(1024, generator=torch.Generator(device="cpu").manual_seed(int(sd))) for sd in seeds[k]]).to(img.device)
            out = (model.outpaint_pipelined if overlap else model.outpaint_planned)(planned, None, temperature=temperature, uniforms=uniforms)
            planned = None
            if k + 1 < len(chunks):
                if k == 0:
                    side.wait_stream(main)      # (the shared inputs were produced on the main stream)
                with torch.cuda.stream(side):
                    planned = model.plan_views(*inputs(chunks[k + 1]))
                model.adopt_planned(planned, main)
                main.wait_stream(side)
            finish(out)
        if overlap:
            for out in model.outpaint_flush():
                finish(out)
    except BaseException:
        model.outpaint_reset()      # (a batch left in flight must not be merged into the next sequence's launches)
        raise
    if chunks:
        model.outpaint2.engine(32, 32, len(chunks[-1])).check()
    return frames


def save_png(path, chw):
    """chw: (3,S,S) uint8 image, or float in [-1, 1]."""
    from PIL import Image
    if chw.dtype != torch.uint8:
        chw = D.to_image_u8(chw)
    Image.fromarray(chw.permute(1, 2, 0).cpu().numpy()).save(path)


def load_image(path, S=256):
    from PIL import Image
    im = Image.open(path).convert("RGB").resize((S, S), Image.BICUBIC)
    return torch.from_numpy(np.asarray(im).astype(np.float32) / 127.5 - 1.0).permute(2, 0, 1)[None]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--image", help="source PNG/JPEG (default: the synthetic sample)")
    ap.add_argument("--depth-npy", help="(S,S) float32 depth in [min_z, max_z] (default: synthetic smooth depth)")
    ap.add_argument("--trajectory", default="circle", help="circle | R | L | U | D | UL | UR | DL | DR")
    ap.add_argument("--scene", nargs="+", metavar="DIR", help="chained mode: directions of forward_scene, e.g. R L C")
    ap.add_argument("--num-split", type=int, default=4, help="--scene: views per direction (num_split)")
    ap.add_argument("--sequential", action="store_true", help="--scene: sequential_outpainting")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16, help="views rendered together per rank")
    ap.add_argument("--out", default="results")
    ap.add_argument("--pixelcnn", help="state_dict of the reference's OurPixelCNN (torch.save)")
    ap.add_argument("--vqvae", help="state_dict of the reference's VQVAETop (torch.save)")
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # PS_DRYRUN_ONE_GPU=1: every rank uses cuda:0 and the gloo backend -- the multi-rank control flow (sharding, gather, who
    # writes what) on a single-GPU box
    dry = os.environ.get("PS_DRYRUN_ONE_GPU") == "1"
    if dry:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)
    model = build_model(device, args.pixelcnn, args.vqvae)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    img = (load_image(args.image) if args.image else torch.from_numpy(syn.image(1000, 1, 3, 256))).to(device)
    depth = t(np.load(args.depth_npy)[None, None].astype(np.float32)) if args.depth_npy else t(syn.depth_smooth(2000, 1, 256, 1.0, 100.0))
    cam = {k: t(v) for k, v in syn.demo_cameras(1).items()}
    if args.scene:
        model.opt.directions, model.opt.num_split = list(args.scene), args.num_split
        model.opt.sequential_outpainting, model.opt.num_samples = args.sequential, 1
        batch = {"images": [img], "cameras": [cam], "depth_fn": syn.depth_from_image}
        _, outputs = model(batch)
        model.outpaint2.engine(32, 32, 1).check()
        n = scene_outputs_to_disk(outputs, model.opt.directions, args.num_split, os.path.join(args.out, f"rank{rank}") if world > 1 else args.out)
        print(f"rank {rank}: chained scene {' '.join(args.scene)}: {n} video frames")
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    kind = "circle" if args.trajectory == "circle" else args.trajectory
    poses = trajectory(model, cam["P"], kind, args.frames)
    mine = D.shard_views(len(poses), rank, world)
    frames = render_pipelined(model, img, depth, cam, [[poses[i] for i in mine[s:s + args.batch]] for s in range(0, len(mine), args.batch)],
                              seeds=[[1000 + i for i in mine[s:s + args.batch]] for s in range(0, len(mine), args.batch)])
    local_frames = D.to_image_u8(torch.cat(frames)) if frames else torch.empty(0, 3, 256, 256, dtype=torch.uint8, device=device)
    per_rank = (len(poses) + world - 1) // world                         # gather_frames wants equal shards: pad the last round
    if local_frames.shape[0] < per_rank:
        pad = torch.zeros(per_rank - local_frames.shape[0], 3, 256, 256, dtype=torch.uint8, device=device)
        local_frames = torch.cat([local_frames, pad])
    all_frames = D.gather_frames(local_frames, len(poses))
    if rank == 0:
        vid = os.path.join(args.out, "video")
        os.makedirs(vid, exist_ok=True)
        save_png(os.path.join(vid, "0.png"), img[0])                      # frame 0 = the source (demo.py:133-137)
        for i in range(len(poses)):
            save_png(os.path.join(vid, f"{i + 1}.png"), all_frames[i])
        print(f"wrote {len(poses) + 1} frames to {vid}/%d.png  (ffmpeg -i {vid}/%d.png ... as create_vid.py does)")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
