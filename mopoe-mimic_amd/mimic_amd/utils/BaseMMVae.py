"""Same import path as the reference's mimic/utils/BaseMMVae.py."""
from ..mmvae import BaseMMVae  # noqa: F401
