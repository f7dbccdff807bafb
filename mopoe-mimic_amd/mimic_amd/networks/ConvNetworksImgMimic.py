"""Same import path as the reference's mimic/networks/ConvNetworksImgMimic.py."""
from ..nets import DecoderImg, EncoderImg  # noqa: F401
