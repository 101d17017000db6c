"""`from layers.functions import Detect, PriorBoxLayer` of the reference (layers/functions/__init__.py), backed by the
HIP post-processing ops (fdt_detect / fdt_priorbox in include/fdt.h)."""
from .prior_box import PriorBoxLayer  # noqa: F401
from .detection import Detect  # noqa: F401

__all__ = ("PriorBoxLayer", "Detect")
