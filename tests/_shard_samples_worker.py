"""worker of tests/test_zbuffermodel_gpu.py::test_sharded_sample_ranking_two_ranks_equals_one (2 ranks on cuda:0, gloo)"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG", "False")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pixelsynth_amd import synthetic as syn  # noqa: E402
from pixelsynth_amd.z_buffermodel import ZbufferModelPts, build_ar_plan  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
DEV = torch.device("cuda", 0)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


class Disc:   # deterministic stand-in scorers: the ranking rule and the transport are what is tested
    def run_discriminator_one_step(self, fake, real):
        return {"D_Fake": (fake * torch.linspace(-1, 1, fake.shape[-1], device=fake.device)).mean().reshape(1)}


class Cls(torch.nn.Module):
    def forward(self, x):
        return torch.cat([(x[:, :, ::7, ::5].mean() * k).reshape(1, 1) for k in range(1, 11)], 1)


for n in (3, 2):
    o = dict(W=256, use_rgb_features=True, splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=128, tau=1.0,
             rad_pow=2, accumulation="alphacomposite", background_smoothing_kernel_size=13, min_z=1.0, max_z=100.0, rotation=0.6,
             direction="R", temperature=0.7, model_setting="gen_img", seed=0, homography=False, vqvae=True, num_samples=n)
    m = ZbufferModelPts(types.SimpleNamespace(**o), classifier=Cls()).eval()
    m.outpaint2.load_state_dict({k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(0).items()})
    m.vqvae.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}, strict=True)
    m = m.to(DEV)
    img = tt(syn.image(31, 1, 3, 256))
    cam = {k: tt(v) for k, v in syn.demo_cameras(1).items()}
    RTinv, RT = m.get_rt_from_rot("R", cam["P"])
    gen_fs, bg = m.pts_transformer.forward_justpts(img, syn.depth_from_image(img), cam["K"], cam["Kinv"], cam["P"], cam["Pinv"], RT, RTinv)
    plan = build_ar_plan(bg, 32)
    codes = m.vqvae.encode_codes(gen_fs)
    uni = torch.rand(n, 1, 1024, generator=torch.Generator(device="cpu").manual_seed(5)).to(DEV)
    alone = m.get_best_sample(plan, codes, bg, gen_fs, Disc(), img, uniforms=uni)
    shared = m.get_best_sample(plan, codes, bg, gen_fs, Disc(), img, uniforms=uni, shard=True)
    assert shared.is_cuda and torch.equal(alone, shared), (rank, n)
    # the reference-shaped entry point routes there with opt.shard_samples
    m.opt.shard_samples = True
    batch = {"images": [img.cpu()], "cameras": [{k: v.cpu() for k, v in cam.items()}], "depths": [syn.depth_from_image(img).cpu()]}
    torch.manual_seed(0)
    _, out = m.forward_image(batch, netD=Disc())
    ref = [torch.empty_like(out["PredImg"].cpu()) for _ in range(world)]
    dist.all_gather(ref, out["PredImg"].cpu())
    assert all(torch.equal(ref[0], r) for r in ref)      # every rank holds the same winner
# the winner's transport on its own: a rank that holds nothing learns shape and dtype from the owner (the decoded / refined image
# is not shaped like gen_fs when the features are not RGB)
from pixelsynth_amd import distributed as D  # noqa: E402
for src, shape, dt in ((0, (2, 5, 7), torch.float32), (1, (1, 4, 16, 16), torch.uint8)):
    mine = (torch.arange(int(np.prod(shape)), device=DEV) % 251).reshape(shape).to(dt) if rank == src else None
    got = D.broadcast_from(mine, src, DEV)
    want = (torch.arange(int(np.prod(shape)), device=DEV) % 251).reshape(shape).to(dt)
    assert got.is_cuda and got.dtype == dt and tuple(got.shape) == shape and torch.equal(got, want), (rank, src)
m.outpaint2.engine(32, 32, 1).check()
if rank == 0:
    print("ok")
dist.destroy_process_group()
