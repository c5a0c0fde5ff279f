from .registry import Register  # noqa: F401
