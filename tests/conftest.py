import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared library is a build product (git-ignored): on a fresh checkout compile it once (hipcc cross-compiles
    gfx950 without a GPU, ~25 s) so that the C-ABI tests do not depend on who ran build() before."""
    lib = os.path.join(ROOT, "face-detection-and-tracking_amd", "csrc", "libfdt_hip.so")
    if not os.path.exists(lib):
        try:
            importlib.import_module("__graft_entry__").build()
        except Exception as e:          # the tests that need the library will say so
            print("conftest: building libfdt_hip.so failed: %r" % (e,))


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = json.loads(bytes(d.pop("meta_json")).decode()) if "meta_json" in d else {}
    return d, meta


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("face-detection-and-tracking_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("face-detection-and-tracking_amd.synth")


@pytest.fixture(scope="session")
def res50_sd(synth):
    return synth.make_state_dict("res50", seed=0)


@pytest.fixture(scope="session")
def try3_sd(synth):
    return synth.make_state_dict("try3", seed=0)
