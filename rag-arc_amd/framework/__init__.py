"""Tagged-config -> module factory and the process-wide registry (mirrors the reference's framework/)."""
from .config import AbstractConfig  # noqa: F401
from .module import AbstractModule  # noqa: F401
from .register import Register  # noqa: F401
from .singleton import singleton  # noqa: F401
