"""Scorers of the sample ranking: the discriminator loss wrapper get_best_sample calls."""
from .gan_loss import DiscriminatorLoss, GANLoss  # noqa: F401
