"""Tuning / debugging aid: run the same AR completion as the position-by-position walk and as a wavefront schedule whose
large waves go through the throughput form (k_column_tp); report where the logits differ (wave, column slot in the launch)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle  # noqa: E402
from pixelsynth_amd import synthetic as syn  # noqa: E402
from pixelsynth_amd.lmconv.layers import PONO  # noqa: E402
from pixelsynth_amd.lmconv.model import OurPixelCNN, wavefronts  # noqa: E402

DEV = "cuda:0"
F_, first, cap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
net = OurPixelCNN(nr_resnet=2, nr_filters=80, input_channels=512, nr_logistic_mix=10, kernel_size=(3, 3), max_dilation=2,
                  weight_norm=False, feature_norm_op=lambda c: PONO(), dropout_prob=0, conv_bias=True, conv_mask_weight=False,
                  rematerialize=False, binarize=False).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(3).items()})
net = net.to(DEV)
eng = net.engine(32, 32, F_)
bgs = syn.background_masks(256)
names = ["right_half", "half_plus_island", "ragged", "all", "top_band"]
infos = [c_oracle.masks_for_background(bgs[names[b % 5]], 32) for b in range(F_)]
order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
reg = np.zeros((F_, 1024), np.uint8)
ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
codes0 = syn.codes(13, F_).reshape(F_, 1024).astype(np.int32)
forced = tt(syn.codes(14, F_).reshape(F_, 1024).astype(np.int32))      # teacher-forced: a difference does not propagate through draws
for b in range(F_):
    reg[b, order_loc[b][first:]] = 1
c_walk, c_wave = tt(codes0.copy()), tt(codes0.copy())
l_walk = eng.ar_run(c_walk, tt(order_loc), tt(reg), *ms, temperature=0.7, forced=forced, first_step=first, want_logits=True)
eng.check()


class _Raw:
    """a device buffer by address, for torch.as_tensor"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}


def caches():
    """every activation cache of the engine, copied: name -> (F*L, ld) tensor"""
    from pixelsynth_amd import _lib
    out = {}
    torch.cuda.synchronize()
    for what, name, n, ld in ((0, "R", 19, 96), (2, "X", 14, 160)):
        for idx in range(n):
            p = _lib.lib().ps_pixelcnn_debug_cache(eng.handle, what, idx)
            out[f"{name}{idx}"] = torch.as_tensor(_Raw(p, (F_ * 1024, ld)), device=DEV).clone()
    return out


walk_caches = caches() if "-c" in sys.argv else None
waves = wavefronts(order_loc, 32, 32, first, DEV, max_cols=cap)
l_wave = eng.ar_run(c_wave, tt(order_loc), tt(reg), *ms, temperature=0.7, forced=forced, first_step=first, want_logits=True, waves=waves)
eng.check()
cols, ws = waves[0].cpu().numpy(), waves[1]
d = (l_walk - l_wave).abs().amax(-1).cpu().numpy()          # (F, L) by location
nz = np.isnan(l_wave.cpu().numpy()).any(-1)
print("waves", len(ws) - 1, "sizes", [int(ws[i + 1] - ws[i]) for i in range(len(ws) - 1)][:40])
for w in range(len(ws) - 1):
    sl = cols[ws[w]:ws[w + 1]]
    dd = np.array([d[f, order_loc[f, i]] for f, i in sl])
    nn = np.array([nz[f, order_loc[f, i]] for f, i in sl])
    if (dd > 0).any() or nn.any():
        bad = np.nonzero((dd > 0) | nn)[0]
        print(f"wave {w}: {len(sl)} columns, {len(bad)} differ, max {np.nanmax(dd):.3g}, nan {int(nn.sum())}, slots {bad[:48].tolist()}")
        if "-v" in sys.argv:
            print("   diffs", np.round(dd[bad[:16]], 6).tolist())
        if "-1" in sys.argv:
            break
print("total differing locations", int((d > 0).sum()), "of", int(reg.sum()))
if walk_caches is not None:
    # stage order of the network: the first cache that differs at a column names the stage that went wrong
    seq = ["R0", "X0", "R1", "X1", "R2", "R3", "X2", "R4", "X3", "R5", "R6", "X4", "R7", "X5", "R8", "X6", "R9", "X7", "R10", "R11",
           "X8", "R12", "X9", "R13", "X10", "R14", "R15", "X11", "R16", "X12", "R17", "X13", "R18"]
    wave_caches = caches()
    w = int(os.environ.get("TP_WAVE", "2"))
    sl = cols[ws[w]:ws[w + 1]]
    rows = np.array([f * 1024 + order_loc[f, i] for f, i in sl])
    for name in seq:
        a, b = walk_caches[name][rows], wave_caches[name][rows]
        nch = 80 if name[0] == "R" else 160
        dd = (a[:, :nch] - b[:, :nch]).abs().amax(1).cpu().numpy()
        bad = np.nonzero(dd > 0)[0]
        print(f"{name}: {len(bad)} of {len(rows)} columns differ, max {dd.max():.3g}, slots {bad[:40].tolist()}")
        if len(bad) and "-a" not in sys.argv:
            k = bad[0]
            ch = np.nonzero((a[k, :nch] != b[k, :nch]).cpu().numpy())[0]
            print("   slot", k, "channels differing:", ch[:40].tolist(), "walk", a[k, ch[:6]].tolist(), "wave", b[k, ch[:6]].tolist())
            break

if "-n" in sys.argv:
    # neighbour slots of the LAST launch (run with PS_TP_MIN_COLS=1) against a torch evaluation of stage 0, slot NA / NB:
    # sum over the slot's taps of mask * W_tap . concat_elu(u0)[neighbour]
    from pixelsynth_amd import _lib
    torch.cuda.synchronize()
    nbr = torch.as_tensor(_Raw(_lib.lib().ps_pixelcnn_debug_cache(eng.handle, 3, 0), (33, 2, 1024, 160)), device=DEV).clone()
    w = len(ws) - 2
    sl = cols[ws[w]:ws[w + 1]]
    E0 = torch.cat([wave_caches["R0"][:, :80]], 1)
    celu = torch.cat([torch.nn.functional.elu(E0), torch.nn.functional.elu(-E0)], 1)       # (F*L, 160)
    Wc = net.state_dict()["up_layers.0.u_stream.0.conv_input.weight"]                      # (80,160,3,3)
    mu = ms[1].view(F_, 9, 1024)
    for half, taps in ((0, (0, 1, 2, 3)), (1, (5, 6, 7, 8))):
        worst = []
        for k, (f, i) in enumerate(sl):
            q = int(order_loc[f, i]); r, c = divmod(q, 32)
            acc = torch.zeros(80, device=DEV)
            for t_ in taps:
                rr, cc = r + t_ // 3 - 1, c + t_ % 3 - 1
                if 0 <= rr < 32 and 0 <= cc < 32 and float(mu[f, t_, q]) != 0:
                    acc += Wc[:, :, t_ // 3, t_ % 3] @ celu[f * 1024 + rr * 32 + cc]
            worst.append(float((acc - nbr[0, half, k, :80]).abs().max()))
        worst = np.array(worst)
        print(f"stage 0 slot {'NA' if half == 0 else 'NB'}: max |torch - nbr_tp| per column slot:", np.round(worst, 4).tolist())
