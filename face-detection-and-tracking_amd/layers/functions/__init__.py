from .detection import Detect
from .prior_box import PriorBoxLayer

__all__ = ['Detect', 'PriorBoxLayer']
