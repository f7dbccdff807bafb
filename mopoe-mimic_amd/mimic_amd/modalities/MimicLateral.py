from ..plugins import MimicLateral  # noqa: F401
