"""Autoregressive sampling over the 32x32 code grid -- drop-in for the reference's
models/lmconv/sample.py:sample() (same signature, same return value).

Three evaluation modes (args.ar_mode or env PS_AR_MODE):
  "fused"       (default) the whole loop runs on the device through ps_pixelcnn_ar_run: exact
                incremental evaluation (one network COLUMN per location against cached activations,
                all columns of a dependency level in one launch) and an inverse-CDF categorical draw from
                softmax(logits/T) with uniforms taken from torch's generator after the reference's
                seeding rule.  Same distribution as the reference, different RNG stream.
  "multinomial" the incremental evaluation, but every draw is torch.multinomial called exactly as
                the reference calls it (1 + seed draws per position, sample.py:61-63).
  "reference"   the reference's loop verbatim: one FULL network forward per sampled position.
All three produce the same logits at every decided position (the masks only admit earlier locations,
so a location's activations never change once computed -- SURVEY.md 7).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .locally_masked_convolution import compact_mask


def _sample_indices(generation_idx, background_mask, obs):
    """sample.py:23-41: per image, the positions with background_mask == 1 in generation order."""
    out = []
    bm = background_mask.detach().cpu().numpy()
    for image_number in range(bm.shape[0]):
        sel = [[int(i), int(j)] for i, j in generation_idx[image_number] if bm[image_number, int(i), int(j)] == 1]
        out.append(np.array(sel, dtype=np.int64).reshape(-1, 2))
    return out


def sample(model, generation_idx, mask_init, mask_undilated, mask_dilated, batch_to_complete, obs, args, seed=0,
           temperature=1.0, background_mask=None):
    mode = getattr(args, "ar_mode", None) or os.environ.get("PS_AR_MODE", "fused")
    num_classes = args.num_classes
    batch_to_complete_full = torch.clone(batch_to_complete)
    dev = torch.device("cuda", torch.cuda.current_device())
    B = batch_to_complete.shape[0]
    H, W = int(obs[1]), int(obs[2])
    L = H * W

    np.random.seed(seed)
    seed2 = seed * 10 + np.random.randint(188)          # sample.py:13-15
    torch.manual_seed(seed2)
    model.eval()

    sample_indices = _sample_indices(generation_idx, background_mask, obs)
    codes = batch_to_complete.to(dev).to(torch.int64).clone()

    if mode == "reference":
        data = F.one_hot(codes, num_classes).permute(0, 3, 1, 2).to(torch.float32)
        for b in range(B):
            if len(sample_indices[b]) > 0:
                data[b, :, sample_indices[b][:, 0], sample_indices[b][:, 1]] = 0      # sample.py:47
        for n_pix in range(len(sample_indices[0])):                                    # sample.py:54
            out = model([data, mask_init, mask_undilated, mask_dilated], sample=True)
            for b in range(out.shape[0]):
                (i, j) = sample_indices[b][n_pix]
                prob = torch.softmax(out[:, :, i, j] / temperature, 1)
                new_samples = torch.multinomial(prob, 1).squeeze(-1)
                for _ in range(seed):
                    new_samples = torch.multinomial(prob, 1).squeeze(-1)
                data[b, :, i, j] = F.one_hot(new_samples[b], num_classes).to(torch.float32)
    else:
        eng = model.engine(H, W, B)
        m_i = compact_mask(mask_init.to(dev), B, num_classes + 1)
        m_u = compact_mask(mask_undilated.to(dev), B, 160)
        m_d = compact_mask(mask_dilated.to(dev), B, 80)
        if m_i.size(0) == 1 and B > 1:
            m_i, m_u, m_d = (m.expand(B, -1, -1).contiguous() for m in (m_i, m_u, m_d))
        order_np = np.stack([np.asarray(g, dtype=np.int64)[:, 0] * W + np.asarray(g, dtype=np.int64)[:, 1]
                             for g in generation_idx[:B]]).astype(np.int32)
        region_np = np.zeros((B, L), np.uint8)
        first = L
        for b in range(B):
            if len(sample_indices[b]) > 0:
                loc = sample_indices[b][:, 0] * W + sample_indices[b][:, 1]
                region_np[b, loc] = 1
                first = min(first, int(np.nonzero(region_np[b][order_np[b]])[0][0]))
        order = torch.from_numpy(order_np).to(dev)
        region = torch.from_numpy(region_np).to(dev)
        c32 = codes.view(B, L).to(torch.int32).contiguous()
        if mode == "fused":
            from .model import wavefronts
            uniforms = torch.rand(B, L, device=dev, dtype=torch.float32)
            eng.ar_run(c32, order, region, m_i, m_u, m_d, temperature=temperature, uniforms=uniforms,
                       first_step=first, waves=wavefronts(order_np, H, W, first, dev))
            eng.check()   # a column launch that gave up on an in-launch wait raises here instead of returning wrong codes
        elif mode == "multinomial":
            c32[region.bool()] = -1
            flat_region = region.bool()
            for step in range(first, L):
                logits = eng.ar_step(c32, order, m_i, m_u, m_d, step, first)
                for b in range(B):
                    q = int(order_np[b, step])
                    if not region_np[b, q]:
                        continue
                    prob = torch.softmax(logits / temperature, 1)
                    new_samples = torch.multinomial(prob, 1).squeeze(-1)
                    for _ in range(seed):
                        new_samples = torch.multinomial(prob, 1).squeeze(-1)
                    c32[b, q] = new_samples[b].to(torch.int32)
            del flat_region
            eng.check()
        else:
            raise ValueError(f"unknown AR mode {mode!r}")
        data = F.one_hot(c32.view(B, H, W).to(torch.int64), num_classes).permute(0, 3, 1, 2).to(torch.float32)

    loss_score = nn.CrossEntropyLoss()(data, batch_to_complete_full.to(dev).to(torch.int64))
    # revert seeding for dataloader (sample.py:70-71)
    torch.manual_seed(args.dataloader_seed)
    np.random.seed(args.dataloader_seed)
    return data, loss_score
