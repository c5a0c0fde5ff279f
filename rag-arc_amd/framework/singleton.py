from .registry import singleton  # noqa: F401
