from .vqvae import VQVAETop, Quantize  # noqa: F401
