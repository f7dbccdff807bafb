from ..plugins import MimicPA  # noqa: F401
