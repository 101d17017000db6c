from .config import face  # noqa: F401
