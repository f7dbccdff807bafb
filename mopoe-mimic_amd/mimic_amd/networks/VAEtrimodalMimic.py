"""Same import path as the reference's mimic/networks/VAEtrimodalMimic.py."""
from ..mmvae import VAEtrimodalMimic  # noqa: F401
