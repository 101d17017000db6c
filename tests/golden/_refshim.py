"""Harness-side shims that let the *unmodified* reference import in the build container.

Only used by tests/golden/make_golden.py (run once, here, to produce the committed
fixtures).  Never imported by tests, bench or the product; /root/reference does not
exist on the GPU box.  Shims: stub cv2/torchvision modules that the
inference path imports but never calls, `time.clock`, and identity `.cuda()`.
"""
import os
import sys
import time
import types

REFERENCE = os.environ.get("FDT_REFERENCE", "/root/reference")


def install():
    sys.dont_write_bytecode = True
    for m in ("cv2", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(m, types.ModuleType(m))
    if not hasattr(time, "clock"):
        time.clock = time.perf_counter
    import torch
    import torch.nn as nn
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
