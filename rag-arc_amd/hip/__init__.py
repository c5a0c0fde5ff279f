from .binding import RarcError, load_library, library_path  # noqa: F401
