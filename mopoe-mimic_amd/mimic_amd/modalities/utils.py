from ..plugins import get_likelihood  # noqa: F401
