"""CPU-only: the reference's import paths resolve to the mirrors, the mirrors keep the reference's
parameter names/shapes (checkpoint compatibility), and the HIP ops refuse CPU tensors loudly."""
import subprocess
import sys
import os

import numpy as np
import pytest
import torch

from pixelsynth_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_aliases_resolve_in_a_fresh_interpreter():
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import pixelsynth_amd.compat as c; c.install_reference_aliases()\n"
        "from models.lmconv.model import OurPixelCNN\n"
        "from models.lmconv.sample import sample\n"
        "from models.projection.z_buffer_manipulator import PtsManipulator\n"
        "from models.layers.z_buffer_layers import RasterizePointsXYsBlending\n"
        "import models.lmconv.masking as m\n"
        "assert OurPixelCNN.__module__ == 'pixelsynth_amd.lmconv.model'\n"
        "assert 'pytorch3d' not in sys.modules and 'cv2' not in sys.modules\n"
        "print('ok')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="needs the reference checkout (build container only)")
def test_aliases_keep_the_reference_tree_importable():
    """INTEGRATION.md option A: demo.py run from a reference checkout.  With the reference on sys.path the aliased modules
    resolve to the mirrors AND everything that is not aliased (models.networks.configs, ...) still imports from the
    reference's own packages -- a synthesised empty parent package would hide them."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, '/root/reference')\n"
        "import pixelsynth_amd.compat as c; c.install_reference_aliases()\n"
        "import models.networks.configs as cfg\n"
        "assert cfg.__file__.startswith('/root/reference/'), cfg.__file__\n"
        "from models.lmconv.model import OurPixelCNN\n"
        "assert OurPixelCNN.__module__ == 'pixelsynth_amd.lmconv.model'\n"
        "import models.lmconv.masking as m\n"
        "assert m.__name__ == 'pixelsynth_amd.lmconv.masking'\n"
        "import models.vqvae2.vqvae as v\n"
        "assert v.__name__ == 'pixelsynth_amd.vqvae2.vqvae'\n"
        "import models.lmconv.average_checkpoints\n"       # a non-aliased sibling of aliased modules
        "print('ok')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_state_dict_keys_match_reference_order():
    from pixelsynth_amd.lmconv.layers import PONO
    from pixelsynth_amd.lmconv.model import PARAM_KEYS, OurPixelCNN
    net = OurPixelCNN(nr_resnet=2, nr_filters=80, input_channels=512, nr_logistic_mix=10, kernel_size=(3, 3),
                      max_dilation=2, weight_norm=False, feature_norm_op=lambda c: PONO(), dropout_prob=0, conv_bias=True,
                      conv_mask_weight=False, rematerialize=False, binarize=False)
    keys = list(net.state_dict().keys())
    assert keys == PARAM_KEYS and len(keys) == 93        # the order the reference printed (SURVEY, probe)
    ref = syn.pixelcnn_state_dict(0)
    assert list(ref.keys()) == keys
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == ref[k].shape, k
    assert sum(p.numel() for p in net.parameters()) == 5587584  # 5 587 584 parameters
    net.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)


def test_splat_and_lmconv_refuse_cpu_tensors():
    import types
    from pixelsynth_amd.layers.z_buffer_layers import RasterizePointsXYsBlending
    from pixelsynth_amd.lmconv.locally_masked_convolution import locally_masked_conv2d
    from pixelsynth_amd.projection.z_buffer_manipulator import PtsManipulator
    opt = types.SimpleNamespace(splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=8, tau=1.0,
                                rad_pow=2, accumulation="alphacomposite", background_smoothing_kernel_size=13)
    sp = RasterizePointsXYsBlending(3, True, 4, 16, 8, opt)
    assert "default_feature" in dict(sp.named_parameters())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sp(torch.zeros(1, 10, 3), torch.zeros(1, 3, 10))
    pm = PtsManipulator(16, C=3, opt=opt)
    assert set(pm.state_dict().keys()) == {"xyzs", "splatter.default_feature"}
    assert np.array_equal(pm.xyzs.numpy(), __import__("oracle.c_oracle", fromlist=["x"]).make_grid(16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pm.project_pts(torch.zeros(1, 1, 256), *[torch.eye(4)[None]] * 6)
    conv = locally_masked_conv2d(4, 6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        conv(torch.zeros(1, 4, 5, 5), torch.ones(1, 9, 25))


def test_masking_mirror_matches_oracle():
    from oracle import c_oracle
    from pixelsynth_amd.lmconv import masking
    D = dict(syn.distance_maps())["rand3"]
    d = D.copy()
    order = masking.get_generation_order_idx("custom", 32, 32, d, (16, 16))
    ref, _ = c_oracle.custom_idx(32, 32, D)
    assert np.array_equal(order, ref) and np.array_equal(d, D * 10000)
    km = masking.kernel_masks(order, 32, 32, 3, 2, "B")
    assert km.shape == (1024, 3, 3)
    unf = masking.get_unfolded_masks(order, 32, 32, 3, 2, "B")
    assert np.array_equal(unf.numpy(), c_oracle.unfolded_masks(ref, 32, 32, 3, 2, "B"))
    assert np.array_equal(km.reshape(1024, 9).T.astype(np.float32), unf[0].numpy())


def test_rank_samples_follows_the_reference_rule():
    """z_buffermodel.py:266-276, restated with its own loops: rank of sample i in each sorted list (np.where), total =
    .5*(n-1-entropy_rank) + .5*discriminator_rank, arg-max."""
    from pixelsynth_amd.z_buffermodel import rank_samples
    rs = np.random.RandomState(3)
    for n in (1, 2, 3, 5, 8):
        for _ in range(20):
            disc, entr = rs.rand(n).tolist(), rs.rand(n).tolist()
            sorted_disc, sorted_entr = np.array(disc).argsort(), np.array(entr).argsort()
            disc_ranks = [np.where(sorted_disc == i)[0][0] for i in range(n)]
            entr_ranks = [np.where(sorted_entr == i)[0][0] for i in range(n)]
            total = .5 * (n - 1 - np.array(entr_ranks)) + .5 * np.array(disc_ranks)
            assert rank_samples(disc, entr) == int(np.argmax(total))
    assert rank_samples([0.1, 0.9], [2.0, 1.0]) == 1     # highest discriminator score and lowest entropy wins


def test_wavefronts_with_a_first_step_per_frame():
    """ps_ar_wavefronts_frames: a first walked position per frame.  With the same value for every frame it is ps_ar_wavefronts_capped;
    with different ones every frame's positions from its own first step on appear once, and a column runs in a later wave than every
    column of its frame that it reads (3x3 neighbours at dilation 1 and 2 that come earlier in the order and are walked)."""
    from pixelsynth_amd.lmconv.model import wavefronts
    rs = np.random.RandomState(5)
    F_, G, L = 6, 8, 64
    order = np.stack([rs.permutation(L) for _ in range(F_)]).astype(np.int32)
    a = wavefronts(order, G, G, 20, None, max_cols=16)
    b = wavefronts(order, G, G, 20, None, max_cols=16, first_steps=np.full(F_, 20, np.int32))
    assert torch.equal(a[0], b[0]) and (a[1] == b[1]).all()
    fs = np.array([20, 33, 64, 21, 50, 20], np.int32)
    cols, ws = wavefronts(order, G, G, 20, None, max_cols=16, first_steps=fs)
    cols = cols.numpy()
    assert cols.shape[0] == int((L - fs).sum()) and np.diff(ws).max() <= 16
    wave_of = np.repeat(np.arange(len(ws) - 1), np.diff(ws))
    seen = {}
    for (f, i), w in zip(cols.tolist(), wave_of.tolist()):
        assert i >= fs[f] and (f, i) not in seen
        seen[(f, i)] = w
    assert len(seen) == int((L - fs).sum())
    for (f, i), w in seen.items():
        rank = np.empty(L, np.int64); rank[order[f]] = np.arange(L)
        q = int(order[f][i]); r, c = divmod(q, G)
        for dil in (1, 2):
            for t in range(9):
                if t == 4: continue
                rr, cc = r + (t // 3 - 1) * dil, c + (t % 3 - 1) * dil
                if 0 <= rr < G and 0 <= cc < G:
                    j = int(rank[rr * G + cc])
                    if fs[f] <= j < i:
                        assert seen[(f, j)] < w
