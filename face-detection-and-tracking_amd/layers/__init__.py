from .functions import Detect, PriorBoxLayer  # noqa: F401

__all__ = ['Detect', 'PriorBoxLayer']
