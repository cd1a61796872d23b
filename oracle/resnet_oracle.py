"""TEST INFRASTRUCTURE (oracle/): a functional CPU restatement of torchvision's ResNet-18 -- the scene classifier of the sample
ranking, which the reference builds as `torchvision.models.resnet18(num_classes=365)` (models/z_buffermodel.py:88) and applies
in get_best_sample (:256-262).  torchvision is not installed here and not vendored by the reference, so this twin restates the
PUBLISHED architecture (He et al. 2016, torchvision/models/resnet.py v0.x: 7x7/2 stem, 3x3/2 max-pool, four stages of two
BasicBlocks at 64/128/256/512 channels, stride-2 first block with a 1x1 projection shortcut from stage 2 on, global average
pool, linear head; inference-mode batch norm with eps 1e-5) directly from a torchvision-layout state_dict, without any Module
of the product package.  PARITY UNPINNED against torchvision itself (absent); it pins the product's mirror
(pixelsynth_amd/networks/resnet.py) numerically: the two are written independently and must agree."""
import torch
import torch.nn.functional as F


def _bn(x, sd, p, eps=1e-5):
    scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + eps)
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def resnet18_forward(sd, x):
    """sd: torchvision-layout state_dict (float32 tensors), x (B,3,H,W) -> logits (B,num_classes)."""
    x = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for stage in range(1, 5):
        for blk in range(2):
            p = f"layer{stage}.{blk}"
            stride = 2 if (stage > 1 and blk == 0) else 1
            out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1), sd, p + ".bn1"))
            out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], stride=1, padding=1), sd, p + ".bn2")
            if (p + ".downsample.0.weight") in sd:
                x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
            x = F.relu(out + x)
    x = x.mean(dim=(2, 3))
    return x @ sd["fc.weight"].t() + sd["fc.bias"]


def entropy_score(sd, gen_img):
    """The classifier-entropy score of get_best_sample (models/z_buffermodel.py:256-262) for gen_img (1,3,256,256) in [-1,1]:
    the tensor REINTERPRETED as (256,256,3) (a reshape, not a permute -- the reference's quirk), 8-bit, PIL-resized to 224x224
    bilinear, ImageNet-normalised, softmax, -sum p log p."""
    import numpy as np
    from PIL import Image
    raw = ((gen_img[0].reshape([256, 256, 3]).cpu().numpy() * .5 + .5) * 255).astype(np.uint8)
    im = np.asarray(Image.fromarray(raw).resize((224, 224), Image.BILINEAR), np.float32) / 255.0
    im = (im - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    x = torch.from_numpy(im).permute(2, 0, 1)[None]
    probs = torch.softmax(resnet18_forward(sd, x), 1).squeeze().numpy()
    probs = np.sort(probs)[::-1]
    return float(-np.sum(probs * np.log(probs)))
