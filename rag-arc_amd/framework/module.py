from .registry import AbstractModule  # noqa: F401  (kept as a module for import-path parity)
