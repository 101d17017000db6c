"""The floating-point half of the oracle (oracle/pyramidbox.py) calls ATen through torch.nn.functional; upstream
has no test that pins those functions.  oracle/c/conv_ref.c restates them as plain C loop nests (double
accumulation, one rounding); here the torch calls the oracle makes are checked against that C code on the layer
geometries the detectors use, so the oracle does not rest on PyTorch alone."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "oracle", "c")


@pytest.fixture(scope="module")
def cref():
    subprocess.run(["make", "-C", CDIR], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return C.CDLL(os.path.join(CDIR, "liboracle_ref.so"))


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


# (Cin, Cout, K, stride, pad, dil, groups): every convolution geometry of the detector graphs
GEOMS = [(24, 40, 1, 1, 0, 1, 1), (24, 40, 1, 2, 0, 1, 1), (16, 24, 3, 1, 1, 1, 1), (16, 24, 3, 1, 2, 2, 1),
         (16, 24, 3, 2, 1, 1, 1), (3, 16, 7, 2, 3, 1, 1), (3, 12, 7, 4, 3, 1, 1), (6, 10, 5, 2, 2, 1, 1),
         (3, 8, 7, 2, 1, 1, 1), (20, 20, 3, 1, 1, 1, 20), (20, 20, 3, 2, 1, 1, 20), (12, 12, 5, 2, 2, 1, 12),
         (12, 12, 3, 1, 2, 2, 12), (3, 3, 7, 2, 3, 1, 3), (16, 16, 1, 1, 0, 1, 4), (8, 8, 1, 1, 1, 1, 1)]


@pytest.mark.parametrize("g", GEOMS)
def test_torch_conv2d_matches_c_loops(cref, g):
    Cin, Cout, K, s, p, d, groups = g
    rng = np.random.default_rng(sum(g))
    x = rng.standard_normal((2, Cin, 19, 23)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin // groups, K, K)) / np.sqrt(Cin // groups * K * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    exp = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s, p, d, groups).numpy()
    got = np.empty_like(exp)
    cref.oracle_conv2d(fp(x), 2, Cin, 19, 23, fp(w), fp(b), Cout, K, s, p, d, groups, fp(got))
    # the C value is correctly rounded to <= 1 ulp; f32 accumulation in ATen may drift a few ulp of the largest term
    np.testing.assert_allclose(exp, got, rtol=2e-5, atol=2e-6)


def test_torch_bn_relu_pool_upsample_match_c_loops(cref):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 13, 17)).astype(np.float32) * 3
    gam, bet = rng.uniform(0.5, 1.5, 6).astype(np.float32), rng.standard_normal(6).astype(np.float32)
    mu, var = rng.standard_normal(6).astype(np.float32), rng.uniform(0.5, 2, 6).astype(np.float32)
    for act, fn in ((0, lambda t: t), (1, F.relu), (2, F.relu6)):
        exp = fn(F.batch_norm(torch.from_numpy(x), torch.from_numpy(mu), torch.from_numpy(var), torch.from_numpy(gam),
                              torch.from_numpy(bet), False, 0.0, 1e-5)).numpy()
        got = x.copy()
        cref.oracle_bn_act(fp(got), 2, 6, 13 * 17, fp(gam), fp(bet), fp(mu), fp(var), C.c_double(1e-5), act)
        np.testing.assert_allclose(exp, got, rtol=2e-6, atol=2e-6)
    for s in (1, 2):
        exp = F.max_pool2d(torch.from_numpy(x), 3, s, 1).numpy()
        got = np.empty_like(exp)
        cref.oracle_maxpool3(fp(x), 12, 13, 17, s, fp(got))
        assert np.array_equal(exp, got)
    exp = F.interpolate(torch.from_numpy(x), scale_factor=2, mode="bilinear", align_corners=False).numpy()
    got = np.empty_like(exp)
    cref.oracle_upsample2x(fp(x), 12, 13, 17, fp(got))
    np.testing.assert_allclose(exp, got, rtol=1e-6, atol=1e-6)
