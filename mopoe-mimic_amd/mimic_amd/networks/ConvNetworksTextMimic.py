"""Same import path as the reference's mimic/networks/ConvNetworksTextMimic.py."""
from ..nets import DecoderText, EncoderText  # noqa: F401
