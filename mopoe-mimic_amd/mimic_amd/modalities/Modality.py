"""Same import path as the reference's mimic/modalities/Modality.py."""
from ..plugins import Modality, ModalityIMG  # noqa: F401
