"""Inference mirror of the reference's models/losses/gan_loss.py for what the novel-view path calls:
`netD.run_discriminator_one_step(pred, real)["D_Fake"]` (models/z_buffermodel.py:254) -- the discriminator-side GAN loss of the
multiscale discriminator on a candidate.  Same class names and state_dict layout (`netD.netD.discriminator_<i>...`)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..networks import discriminators


class GANLoss(nn.Module):
    """gan_loss.py:20-113: hinge (default), ls, original or wgan loss of a prediction or of the list of lists a multiscale
    discriminator returns (the last entry of every scale counts; the mean over scales)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor, opt=None):
        super().__init__()
        if gan_mode not in ("ls", "original", "w", "hinge"):
            raise ValueError("Unexpected gan_mode {}".format(gan_mode))
        self.real_label, self.fake_label, self.gan_mode, self.opt = target_real_label, target_fake_label, gan_mode, opt

    def loss(self, input, target_is_real, for_discriminator=True):
        if self.gan_mode in ("original", "ls"):
            target = torch.full_like(input, self.real_label if target_is_real else self.fake_label)
            return F.binary_cross_entropy_with_logits(input, target) if self.gan_mode == "original" else F.mse_loss(input, target)
        if self.gan_mode == "hinge":
            if for_discriminator:
                return -torch.mean(torch.min((input if target_is_real else -input) - 1, torch.zeros_like(input)))
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            return -torch.mean(input)
        return -input.mean() if target_is_real else input.mean()

    def __call__(self, input, target_is_real, for_discriminator=True):
        if isinstance(input, list):
            loss = 0
            for pred_i in input:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                loss_tensor = self.loss(pred_i, target_is_real, for_discriminator)
                bs = 1 if len(loss_tensor.size()) == 0 else loss_tensor.size(0)
                loss = loss + torch.mean(loss_tensor.view(bs, -1), dim=1)
            return loss / len(input)
        return self.loss(input, target_is_real, for_discriminator)


class BaseDiscriminator(nn.Module):
    """gan_loss.py:116-238, discriminator mode: fake and real go through D in one batch, D_Fake / D_real / Total Loss."""

    def __init__(self, opt, name):
        super().__init__()
        if name == "pix2pixHD":
            self.netD = discriminators.define_D(opt)
        self.criterionGAN = GANLoss(opt.gan_mode, opt=opt)
        self.opt = opt

    def discriminate(self, fake_image, real_image):
        out = self.netD(torch.cat([fake_image, real_image], dim=0))
        fake = [[t[: t.size(0) // 2] for t in p] for p in out]
        real = [[t[t.size(0) // 2:] for t in p] for p in out]
        return fake, real

    @torch.no_grad()
    def forward(self, fake_image, real_image, mode="discriminator"):
        if mode != "discriminator":
            raise NotImplementedError("the inference mirror scores candidates (discriminator mode); training is out of scope")
        pred_fake, pred_real = self.discriminate(fake_image, real_image)
        losses = {"D_Fake": self.criterionGAN(pred_fake, False, for_discriminator=True),
                  "D_real": self.criterionGAN(pred_real, True, for_discriminator=True)}
        losses["Total Loss"] = sum(losses.values()).mean()
        return losses


class DiscriminatorLoss(nn.Module):
    """gan_loss.py:241-288: what train / demo hand to the model as `netD`."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.netD = BaseDiscriminator(opt, name=opt.discriminator_losses)

    def run_discriminator_one_step(self, pred_img, gt_img):
        return self.netD(pred_img, gt_img, mode="discriminator")
