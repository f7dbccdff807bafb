from ..plugins import MimicText  # noqa: F401
