"""GPU: the scorers of the sample ranking (models/z_buffermodel.py:254-262) ON THE DEVICE against fixed numbers --
the multiscale discriminator against the scores the reference's own DiscriminatorLoss produced (tests/golden/scorers.npz), the
ResNet-18 scene classifier against the functional twin oracle/resnet_oracle.py (torchvision is absent: published architecture).
Tolerance 1e-3: MIOpen sums the convolutions' products in another order than the CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import resnet_oracle as ro
from pixelsynth_amd import synthetic as syn
from test_scorers_cpu import make_netD

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_discriminator_scores_on_the_gpu_match_the_reference():
    net, fx = make_netD(DEV)
    cand = torch.from_numpy(syn.image(int(fx["image_seeds"][0]), 3, 3, 256)).to(DEV)
    real = torch.from_numpy(np.repeat(syn.image(int(fx["image_seeds"][1]), 1, 3, 256), 3, 0)).to(DEV)
    for i in range(3):
        out = net.run_discriminator_one_step(cand[i:i + 1], real[i:i + 1])
        assert out["D_Fake"].is_cuda
        for k, ref in (("D_Fake", fx["D_Fake"]), ("D_real", fx["D_real"]), ("Total Loss", fx["total"])):
            np.testing.assert_allclose(float(out[k].mean()), ref[i], rtol=1e-3, atol=1e-3, err_msg=f"{k}[{i}]")
    with torch.no_grad():
        feats = net.netD.netD(torch.cat([cand, real], 0))
    np.testing.assert_allclose(feats[0][-1].cpu().numpy()[:, :, ::4, ::4], fx["last0"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(feats[1][-1].cpu().numpy()[:, :, ::2, ::2], fx["last1"], rtol=1e-3, atol=1e-3)


def test_resnet18_and_entropy_score_on_the_gpu_match_the_functional_twin():
    from pixelsynth_amd.networks import resnet18
    from pixelsynth_amd.z_buffermodel import ZbufferModelPts
    net = resnet18(num_classes=365).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in syn.resnet_state_dict(shapes, 3).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    x = torch.from_numpy(syn.image(77, 2, 3, 224))
    with torch.no_grad():
        got = net(x.to(DEV)).cpu()
        want = ro.resnet18_forward(sd, x)
    assert float(want.std()) > 1e-2                       # the logits are not degenerate
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-3, atol=1e-3)
    # the score get_best_sample ranks with (reinterpreted 224x224 input, softmax entropy): product path vs twin
    holder = type("H", (), {"classifier": net, "_entropy_score": ZbufferModelPts._entropy_score})()
    for seed in (5, 6):
        img = torch.from_numpy(syn.image(seed, 1, 3, 256)).to(DEV)
        e_got, e_want = holder._entropy_score(img), ro.entropy_score(sd, img)
        assert 0.0 < e_want < np.log(365.0) + 1e-6
        assert abs(e_got - e_want) < 1e-3, (e_got, e_want)
