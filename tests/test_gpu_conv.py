"""Numerics of the HIP convolution kernel against a plain PyTorch fp32 reference (F.conv2d on CPU) for
EVERY instantiated (class, tile) variant, with and without split-K, on shapes that exercise the edge
handling: odd sizes (W % 4 != 0 -> scalar staging/epilogue), tiles hanging over the image, Cin not a
multiple of the stage depth, Cout not a multiple of 32, fused residual / upsample-add / ReLU6."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KINDS = {  # name: (k, stride, pad, dil)
    "1x1s1": (1, 1, 0, 1), "1x1s2": (1, 2, 0, 1), "3x3s1": (3, 1, 1, 1), "3x3d2": (3, 1, 2, 2),
    "3x3s2": (3, 2, 1, 1), "7x7s2": (7, 2, 3, 1), "7x7s4": (7, 4, 3, 1), "5x5s2": (5, 2, 2, 1),
    "7x7s2p1": (7, 2, 1, 1)}
N_TILES = 37      # 14 direct + 11 Winograd F(2x2,3x3) (3x3 s1 / d2 only) + 4 ring-of-four (1x1 only) + 2 quarter-split Winograd
                  # + 1 packed-f32 VALU tile for narrow heads (3x3 s1 only) + 2 Winograd F(4x4,3x3) (3x3 s1 only)
                  # + 2 persistent-tile 1x1 (1x1 s1 only) + the 4x32-px x 32-ch tile of the raw-frame FaceBoxes stem
T_P64, T_P128 = 34, 35           # conv.h: TILE_P_128x64 / TILE_P_128x128 (kernel classes CONV_1x1_S1_P16 = 16, _P32 = 17)
T_WINO44 = 32
T_WINO44B = 33     # its twelve-wave form
WINO44_TOL = 1e-4  # F(4x4,3x3): factors up to 8 in A^T / 5 in B^T amplify the f32 rounding of the transforms (2.7e-6 relative RMS
                  # per layer on post-ReLU data, up to ~3e-5 of the output's maximum on N(0,1) inputs)


def lib():
    return importlib.import_module("face-detection-and-tracking_amd._lib")


def run_conv(x, w, b, k, s, p, d, res=None, up=None, act=0, tile=-1, split=0):
    L = lib()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    rc = L.lib().fdt_conv2d(L.ptr(x), B, Cin, H, W, L.ptr(w), L.ptr(b) if b is not None else None, Cout, k, s, p, d,
                            L.ptr(res) if res is not None else None, L.ptr(up) if up is not None else None,
                            up.shape[2] if up is not None else 0, up.shape[3] if up is not None else 0, act, tile,
                            split, L.ptr(out))
    return rc, out


def reference(x, w, b, k, s, p, d, res=None, up=None, act=0):
    y = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b) if b is not None else None, s, p, d)
    if up is not None:
        u = F.interpolate(torch.from_numpy(up), scale_factor=2, mode="bilinear", align_corners=False)
        y = y + u[:, :, :y.shape[2], :y.shape[3]]
    if res is not None:
        y = y + torch.from_numpy(res)
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.relu6(y)
    return y.numpy()


def rel_err(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("kind", list(KINDS))
def test_every_tile_variant_matches_torch(kind):
    k, s, p, d = KINDS[kind]
    rng = np.random.default_rng(sum(map(ord, kind)))
    L = lib()
    tested = 0
    # (Cin, H, W, Cout): aligned; odd / overhanging
    for (Cin, H, W, Cout) in ((40, 36, 48, 72), (19, 27, 37, 45)):
        if k >= 5:
            Cin = min(Cin, 5)
        x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        exp = reference(x, w, b, k, s, p, d, act=1)
        for tile in range(N_TILES):
            for split in (1, 2):
                rc, got = run_conv(x, w, b, k, s, p, d, act=1, tile=tile, split=split)
                if rc != 0:
                    msg = L.lib().fdt_last_error()
                    assert b"not instantiated" in msg or b"bad split-K" in msg, msg
                    continue
                tested += 1
                tol = WINO44_TOL if tile in (T_WINO44, T_WINO44B) else 3e-5 if tile >= 14 else 1e-5   # Winograd: rounding of the transforms
                assert rel_err(got, exp) < tol, (kind, tile, split, (Cin, H, W, Cout), rel_err(got, exp))
    assert tested >= 4


@pytest.mark.parametrize("mode", [1, 2, 3])   # conv.h: CONV_MAP_XCD_SPATIAL / CONV_MAP_XCD_CHANNEL / CONV_MAP_XCD_REGION
def test_xcd_aware_workgroup_maps(mode, monkeypatch):
    """The XCD-aware workgroup -> tile maps (incl. the padding workgroups that exit at once) compute the same
    convolution as the row-major map."""
    monkeypatch.setenv("FDT_CONV_MAP", str(mode))
    rng = np.random.default_rng(mode)
    for (k, s, p, d, tiles) in ((1, 1, 0, 1, (0, 3, 9, 25)), (3, 1, 1, 1, (1, 3, 24, 30)), (3, 1, 2, 2, (1, 24, 29)), (3, 2, 1, 1, (1,))):
        for (Cin, H, W, Cout) in ((24, 52, 76, 200), (17, 9, 11, 40)):      # 28 / 1 spatial tiles; 7 / 2 channel tiles
            x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
            w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
            b = rng.standard_normal(Cout).astype(np.float32)
            exp = reference(x, w, b, k, s, p, d, act=1)
            for tile in tiles:
                for split in (1, 2):
                    rc, got = run_conv(x, w, b, k, s, p, d, act=1, tile=tile, split=split)
                    assert rc == 0, lib().lib().fdt_last_error()
                    assert rel_err(got, exp) < 3e-5, (mode, k, tile, split, rel_err(got, exp))


@pytest.mark.parametrize("W", [64, 50])       # vector and scalar epilogue
@pytest.mark.parametrize("tile", [0, 3, 5, 7, 11])
def test_fused_epilogues(tile, W):
    rng = np.random.default_rng(tile * 100 + W)
    Cin, Cout, H = 64, 136, 40
    x = rng.standard_normal((1, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / 8).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((1, Cout, H, W)).astype(np.float32)
    up = rng.standard_normal((1, Cout, (H + 1) // 2, (W + 1) // 2)).astype(np.float32)
    for kw in (dict(res=res, act=1), dict(up=up, act=0), dict(res=res, up=up, act=2), dict(act=2)):
        for split in (1, 2):
            rc, got = run_conv(x, w, b, 1, 1, 0, 1, tile=tile, split=split, **kw)
            assert rc == 0, lib().lib().fdt_last_error()
            assert rel_err(got, reference(x, w, b, 1, 1, 0, 1, **kw)) < 1e-5, (tile, W, split, list(kw))


@pytest.mark.parametrize("tile", list(range(14, 25)) + [29, 30])
@pytest.mark.parametrize("shape", [(64, 40, 48, 96), (37, 31, 45, 70)])
def test_winograd_variants(tile, shape):
    """Winograd F(2x2,3x3): 2.25x fewer multiplies, same result up to f32 rounding of the transforms
    (tolerance 3x the direct kernel's), incl. odd sizes, residual, ReLU and split-K."""
    Cin, H, W, Cout = shape
    rng = np.random.default_rng(tile * 7 + H)
    x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((2, Cout, H, W)).astype(np.float32)
    for kw in (dict(act=0), dict(res=res, act=1)):
        exp = reference(x, w, b, 3, 1, 1, 1, **kw)
        for split in (1, 2):
            rc, got = run_conv(x, w, b, 3, 1, 1, 1, tile=tile, split=split, **kw)
            if rc != 0 and b"not instantiated" in lib().lib().fdt_last_error():
                pytest.skip("variant needs more LDS than a CU has")
            assert rc == 0, lib().lib().fdt_last_error()
            assert rel_err(got, exp) < 3e-5, (tile, shape, split, rel_err(got, exp))


@pytest.mark.parametrize("shape", [(64, 70, 132, 8), (37, 33, 50, 5), (6, 64, 64, 20), (3, 5, 7, 8)])
def test_narrow_head_valu_kernel(shape):
    """conv_n8.h: the loc + conf heads (8 output channels) on v_pk_fma_f32 -- several workgroup tiles, odd sizes, an odd
    channel count (half-empty last stage), more than one 8-channel tile, residual / ReLU6 and split-K."""
    Cin, H, W, Cout = shape
    rng = np.random.default_rng(H * 31 + W)
    x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((2, Cout, H, W)).astype(np.float32)
    for kw in (dict(act=0), dict(res=res, act=2)):
        exp = reference(x, w, b, 3, 1, 1, 1, **kw)
        for split in (1, 2, 4):
            rc, got = run_conv(x, w, b, 3, 1, 1, 1, tile=31, split=split, **kw)
            if rc != 0 and b"bad split-K" in lib().lib().fdt_last_error():
                continue
            assert rc == 0, lib().lib().fdt_last_error()
            assert rel_err(got, exp) < 1e-5, (shape, split, rel_err(got, exp))


@pytest.mark.parametrize("variant", [10, 11])      # CONV_1x1_S1_K32 / _K64: 32 / 64 channels per LDS stage
def test_deep_stage_1x1_variants(variant):
    rng = np.random.default_rng(variant)
    tested = 0
    for (Cin, H, W, Cout) in ((200, 24, 32, 72), (70, 19, 21, 40)):
        x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        res = rng.standard_normal((2, Cout, H, W)).astype(np.float32)
        exp = reference(x, w, b, 1, 1, 0, 1, res=res, act=1)
        for tile in list(range(14)) + [25, 26, 27, 28]:
            for split in (1, 2):
                rc, got = run_conv(x, w, b, 1, 1, 0, 1, res=res, act=1, tile=variant * 100 + tile, split=split)
                if rc != 0:
                    msg = lib().lib().fdt_last_error()
                    assert b"not instantiated" in msg or b"bad split-K" in msg, msg
                    continue
                tested += 1
                assert rel_err(got, exp) < 1e-5, (variant, tile, split, rel_err(got, exp))
    assert tested >= 8


def test_deep_reduction_split_k_is_deterministic():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 512, 16, 16)).astype(np.float32)
    w = (rng.standard_normal((64, 512, 3, 3)) / 68).astype(np.float32)
    exp = reference(x, w, None, 3, 1, 1, 1)
    outs = []
    for _ in range(2):
        rc, got = run_conv(x, w, None, 3, 1, 1, 1, tile=3, split=16)
        assert rc == 0
        outs.append(got)
    assert np.array_equal(outs[0], outs[1])          # fixed-order reduce: bitwise reproducible
    assert rel_err(outs[0], exp) < 1e-5


COMBINE = 0x1000   # include/fdt.h FDT_SPLIT_COMBINE: the last workgroup to arrive at an output tile sums the split-K slabs


@pytest.mark.parametrize("case", [
    # (k, s, p, d, tile, split, B, Cin, H, W, Cout)                       kernel the (class, tile) selects
    (1, 1, 0, 1, 2, 4, 2, 256, 24, 40, 136),      # direct 1x1, tiles hanging over the map and over Cout
    (1, 1, 0, 1, 1006, 8, 1, 512, 16, 16, 64),    # deep-stage 1x1 (K32)
    (3, 1, 1, 1, 3, 16, 1, 512, 16, 16, 64),      # direct 3x3
    (3, 2, 1, 1, 2, 4, 1, 128, 33, 64, 96),       # strided 3x3 (Wout = 32)
    (3, 1, 2, 2, 2, 8, 1, 256, 8, 8, 64),         # dilated direct
    (3, 1, 1, 1, 14, 2, 2, 64, 20, 36, 40),       # Winograd F(2x2), 4 waves (8-byte slab stores)
    (3, 1, 1, 1, 22, 4, 1, 128, 32, 32, 64),      # ... half-split 8-wave form
    (3, 1, 1, 1, 30, 8, 1, 256, 32, 32, 128),     # ... quarter-split form
    (3, 1, 2, 2, 29, 4, 1, 128, 24, 32, 64),      # ... dilated quarter-split form (4-byte slab stores)
    (3, 1, 2, 2, 24, 2, 1, 64, 16, 64, 64),       # ... dilated half-split form
    (3, 1, 1, 1, 32, 4, 1, 64, 40, 64, 72),       # Winograd F(4x4), 8 waves
    (3, 1, 1, 1, 33, 2, 2, 32, 16, 32, 64),       # ... 12 waves
    (3, 1, 2, 2, 32, 4, 1, 64, 24, 32, 64),       # ... dilated
])
def test_in_kernel_split_k_combine_equals_the_reduce_pass(case):
    """conv.h splitk_combine_tile: same slabs, same summation order, same epilogue as splitk_reduce_kernel -> the same bits,
    whoever arrives last (two runs), with bias + residual + ReLU on top; 1x1 also with the fused upsample-add."""
    k, s, p, d, tile, split, B, Cin, H, W, Cout = case
    rng = np.random.default_rng(hash(case) & 0xffff)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
    assert Wo % 4 == 0
    res = rng.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
    up = rng.standard_normal((B, Cout, (Ho + 1) // 2, (Wo + 1) // 2)).astype(np.float32) if (k == 1 and tile < 100) else None
    rc, two_pass = run_conv(x, w, b, k, s, p, d, res=res, up=up, act=1, tile=tile, split=split)
    assert rc == 0, lib().lib().fdt_last_error()
    for _ in range(2):
        rc, fused = run_conv(x, w, b, k, s, p, d, res=res, up=up, act=1, tile=tile, split=split | COMBINE)
        assert rc == 0, lib().lib().fdt_last_error()
        assert np.array_equal(fused, two_pass)
    tol = WINO44_TOL if tile in (T_WINO44, T_WINO44B) else 2e-5
    assert rel_err(fused, reference(x, w, b, k, s, p, d, res=res, up=up, act=1)) < tol


def test_in_kernel_combine_refuses_what_it_cannot_do():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 64, 9, 9)).astype(np.float32)          # Wout % 4 != 0: 4-byte rows, not offered
    w = rng.standard_normal((32, 64, 1, 1)).astype(np.float32)
    rc, _ = run_conv(x, w, None, 1, 1, 0, 1, tile=2, split=2 | COMBINE)
    assert rc != 0 and b"combine" in lib().lib().fdt_last_error()
    x = rng.standard_normal((1, 64, 8, 8)).astype(np.float32)          # nothing to combine
    rc, _ = run_conv(x, w, None, 1, 1, 0, 1, tile=2, split=1 | COMBINE)
    assert rc != 0


def test_unknown_class_is_an_error():
    x = np.zeros((1, 4, 8, 8), np.float32)
    w = np.zeros((4, 4, 3, 3), np.float32)
    rc, _ = run_conv(x, w, None, 3, 3, 1, 1)
    assert rc == lib().FDT_ERR_ARG


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("shape", [(16, 96, 40, 64), (24, 144, 37, 50), (32, 100, 9, 33), (64, 384, 16, 32)])
def test_fused_expand_depthwise(stride, shape):
    """conv[0..5] of an InvertedResidual (pyramid_mb2_try3.py:96-114, BatchNorms folded) as one kernel vs torch: sizes that
    are not multiples of the 32-column tile / 4-row strips, hidden widths that are not multiples of 32, batch 2."""
    Cin, hid, H, W = shape
    rng = np.random.default_rng(Cin * 7 + H + stride)
    x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
    w1 = (rng.standard_normal((hid, Cin)) / np.sqrt(Cin)).astype(np.float32)
    b1 = rng.standard_normal(hid).astype(np.float32)
    wd = (rng.standard_normal((hid, 9)) / 3).astype(np.float32)
    bd = rng.standard_normal(hid).astype(np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = np.empty((2, hid, Ho, Wo), np.float32)
    L = lib()
    rc = L.lib().fdt_expand_dw(L.ptr(x), 2, Cin, H, W, L.ptr(w1), L.ptr(b1), L.ptr(wd), L.ptr(bd), hid, stride, L.ptr(out))
    assert rc == 0, L.lib().fdt_last_error()
    h = F.relu6(F.conv2d(torch.from_numpy(x), torch.from_numpy(w1)[:, :, None, None], torch.from_numpy(b1)))
    y = F.relu6(F.conv2d(h, torch.from_numpy(wd).reshape(hid, 1, 3, 3), torch.from_numpy(bd), stride, 1, 1, hid)).numpy()
    assert out.shape == y.shape
    assert rel_err(out, y) < 1e-5, (shape, stride, rel_err(out, y))


@pytest.mark.parametrize("stride,residual", [(1, 1), (1, 0), (2, 0)])
@pytest.mark.parametrize("shape", [(16, 96, 24, 40, 64), (24, 144, 24, 37, 50), (24, 144, 32, 64, 96), (32, 100, 17, 9, 33)])
def test_whole_inverted_residual_in_one_launch(stride, residual, shape):
    """fused_ir.hip with PROJ: expand + depthwise + 1x1 project (+ residual) of an InvertedResidual (pyramid_mb2_try3.py:96-134,
    BatchNorms folded) as ONE kernel: against torch, and bit-identical to the two-launch form (fdt_expand_dw, then the
    stand-alone 1x1 conv with the residual fused) -- same k pairing and order in the project GEMM.  Odd sizes, hidden widths that
    are not multiples of 32 (a partly filled last chunk), output widths below 32, batch 2."""
    Cin, hid, oup, H, W = shape
    if residual:
        oup = Cin
    rng = np.random.default_rng(Cin * 11 + H + stride + residual)
    x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
    w1 = (rng.standard_normal((hid, Cin)) / np.sqrt(Cin)).astype(np.float32)
    b1 = rng.standard_normal(hid).astype(np.float32)
    wd = (rng.standard_normal((hid, 9)) / 3).astype(np.float32)
    bd = rng.standard_normal(hid).astype(np.float32)
    wp = (rng.standard_normal((oup, hid)) / np.sqrt(hid)).astype(np.float32)
    bp = rng.standard_normal(oup).astype(np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = np.empty((2, oup, Ho, Wo), np.float32)
    L = lib()
    rc = L.lib().fdt_ir_block(L.ptr(x), 2, Cin, H, W, L.ptr(w1), L.ptr(b1), L.ptr(wd), L.ptr(bd), hid, stride, L.ptr(wp), L.ptr(bp),
                              oup, residual, L.ptr(out))
    assert rc == 0, L.lib().fdt_last_error()
    h = F.relu6(F.conv2d(torch.from_numpy(x), torch.from_numpy(w1)[:, :, None, None], torch.from_numpy(b1)))
    d = F.relu6(F.conv2d(h, torch.from_numpy(wd).reshape(hid, 1, 3, 3), torch.from_numpy(bd), stride, 1, 1, hid))
    y = F.conv2d(d, torch.from_numpy(wp)[:, :, None, None], torch.from_numpy(bp))
    if residual:
        y = y + torch.from_numpy(x)
    assert rel_err(out, y.numpy()) < 1e-5, (shape, stride, residual, rel_err(out, y.numpy()))
    # the two-launch form of the same block
    dw = np.empty((2, hid, Ho, Wo), np.float32)
    assert L.lib().fdt_expand_dw(L.ptr(x), 2, Cin, H, W, L.ptr(w1), L.ptr(b1), L.ptr(wd), L.ptr(bd), hid, stride, L.ptr(dw)) == 0
    rc, two = run_conv(dw, np.ascontiguousarray(wp[:, :, None, None]), bp, 1, 1, 0, 1, res=x if residual else None, act=0, tile=6)
    assert rc == 0, L.lib().fdt_last_error()
    assert np.array_equal(out, two), float(np.abs(out - two).max())


@pytest.mark.parametrize("shape", [(64, 40, 48, 96), (37, 31, 45, 70), (7, 16, 32, 64), (256, 64, 64, 128), (2, 5, 3, 3)])
@pytest.mark.parametrize("tile", [T_WINO44, T_WINO44B])
def test_winograd_f4x4(shape, tile):
    """Winograd F(4x4,3x3) (conv_wino44.h): 4x fewer multiplies than the direct form.  Aligned and odd sizes (W % 4 != 0 ->
    scalar stores; tiles hanging over the image; an odd number of input channels -> a zero-padded k-step; fewer k-steps
    than the pipeline is deep), residual + ReLU, split-K 1 / 2 / 3, and the RMS error on a deep reduction."""
    Cin, H, W, Cout = shape
    rng = np.random.default_rng(H * 131 + Cin)
    x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((2, Cout, H, W)).astype(np.float32)
    for kw in (dict(act=0), dict(res=res, act=1), dict(act=2)):
        exp = reference(x, w, b, 3, 1, 1, 1, **kw)
        for split in (1, 2, 3):
            if split > max(1, (Cin + 1) // 2):
                continue
            rc, got = run_conv(x, w, b, 3, 1, 1, 1, tile=tile, split=split, **kw)
            assert rc == 0, lib().lib().fdt_last_error()
            assert rel_err(got, exp) < WINO44_TOL, (shape, split, list(kw), rel_err(got, exp))
            rms = float(np.sqrt(((got.astype(np.float64) - exp) ** 2).mean()) / np.sqrt((exp.astype(np.float64) ** 2).mean()))
            assert rms < 1e-5, (shape, split, rms)


@pytest.mark.parametrize("shape", [(64, 40, 48, 96), (37, 28, 44, 70), (7, 16, 32, 64), (256, 64, 64, 128)])
def test_winograd_f4x4_dilated(shape):
    """The dilation-2 form of the F(4x4,3x3) kernel (the SSH context convs, pyramid.py:36,38): Winograd on the four parity
    sub-lattices of 8x8-pixel cells.  Widths are multiples of 4 (the only widths it is instantiated for; others are refused)."""
    Cin, H, W, Cout = shape
    rng = np.random.default_rng(H * 17 + Cin)
    x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((2, Cout, H, W)).astype(np.float32)
    for kw in (dict(act=0), dict(res=res, act=1)):
        exp = reference(x, w, b, 3, 1, 2, 2, **kw)
        for split in (1, 2, 3):
            if split > max(1, (Cin + 1) // 2):
                continue
            rc, got = run_conv(x, w, b, 3, 1, 2, 2, tile=T_WINO44, split=split, **kw)
            assert rc == 0, lib().lib().fdt_last_error()
            assert rel_err(got, exp) < WINO44_TOL, (shape, split, list(kw), rel_err(got, exp))
    xo = rng.standard_normal((1, 8, 12, 30)).astype(np.float32)            # Win % 4 != 0: refused, not mis-computed
    rc, _ = run_conv(xo, w[:8, :8].copy(), b[:8].copy(), 3, 1, 2, 2, tile=T_WINO44)
    assert rc != 0 and b"not instantiated" in lib().lib().fdt_last_error()


@pytest.mark.parametrize("k,p,d,tile", [(3, 1, 1, T_WINO44), (3, 2, 2, T_WINO44), (3, 1, 1, 30), (1, 0, 1, 6), (3, 1, 1, 1)])
def test_channels_past_cin_read_as_zero(k, p, d, tile):
    """An odd / non-multiple Cin leaves the last LDS stage with channels that do not exist.  Their weights are zero, but 0 x NaN
    is NaN: the kernels must stage ZEROS there, not whatever follows the image in memory.  Image 1 of the batch is all NaN, so
    anything image 0's workgroups fetch past their own Cin channels would poison image 0's output."""
    rng = np.random.default_rng(k * 10 + d)
    Cin, H, W, Cout = 5, 20, 36, 40
    x = rng.standard_normal((2, Cin, H, W)).astype(np.float32)
    x[1] = np.nan
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    exp = reference(x[:1], w, None, k, 1, p, d)
    rc, got = run_conv(x, w, None, k, 1, p, d, tile=tile, split=1)
    assert rc == 0, lib().lib().fdt_last_error()
    assert np.isfinite(got[0]).all() and np.isnan(got[1]).all()
    assert rel_err(got[:1], exp) < WINO44_TOL


@pytest.mark.parametrize("variant", [16 * 100 + T_P64, 16 * 100 + T_P128, 17 * 100 + T_P64])
@pytest.mark.parametrize("shape", [(1, 64, 256, 256, 256), (2, 40, 250, 252, 136), (1, 256, 128, 160, 64)])
def test_persistent_1x1_walks_tiles(variant, shape):
    """conv_1x1p.h: the grids of these shapes give every workgroup 2..4 consecutive output tiles (1024+ tiles for 256 CUs x 2..4
    resident workgroups), so the ring runs across tile boundaries, the next tile's stages are in flight under an epilogue and
    stores are still draining when the next tile's MFMAs start.  Bit-identical to the one-tile-per-workgroup kernel of the
    same class (same MFMA order, same epilogue arithmetic), and within f32 rounding of torch; bias / residual / ReLU, channel
    tiles hanging over Cout, tiles hanging over the image, two images."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(variant + H)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    for kw in (dict(res=res, act=1), dict(act=0), dict(act=2)):
        rc, got = run_conv(x, w, b, 1, 1, 0, 1, tile=variant, **kw)
        assert rc == 0, lib().lib().fdt_last_error()
        rc, base = run_conv(x, w, b, 1, 1, 0, 1, tile=6, **kw)            # TILE_128x64W of the direct class
        assert rc == 0, lib().lib().fdt_last_error()
        assert np.array_equal(got, base), (variant, shape, list(kw), float(np.abs(got - base).max()))
        assert rel_err(got, reference(x, w, b, 1, 1, 0, 1, **kw)) < 1e-5
    rc, got = run_conv(x, w, None, 1, 1, 0, 1, tile=variant, act=0)       # no bias tensor
    assert rc == 0 and rel_err(got, reference(x, w, None, 1, 1, 0, 1)) < 1e-5


@pytest.mark.parametrize("tile", [7, 8, 11, 12, 5, 6, 37, 38, 39, 40])    # class 21 (conv.h: CONV_1x1_S1_B3, conv_b3.h); 5, 6: waves 4 x 1; 37-40: long-row tiles
@pytest.mark.parametrize("shape", [(1, 64, 64, 64, 256), (2, 40, 36, 48, 72), (1, 19, 27, 36, 45), (1, 1024, 32, 32, 256)])
def test_split_bf16_1x1(tile, shape):
    """conv_b3.h: 1x1 convolution as split-bf16 products on v_mfma_f32_32x32x16_bf16 (three bf16 planes per operand, six plane
    products, f32 accumulate).  Same tolerance against torch's f32 convolution as the f32-MFMA classes (1e-5 of the output's
    maximum), with bias / residual / fused bilinear upsample-add / ReLU6, split-K, channels past Cin (Cin % 16 != 0), couts past
    Cout -- and an error against an f64 convolution that is no larger than the f32 class's (x1.5 + 1e-7 slack)."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(tile * 1000 + Cin)
    x = np.maximum(rng.standard_normal((B, Cin, H, W)), 0).astype(np.float32) * np.exp(rng.standard_normal((B, Cin, 1, 1))).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    up = rng.standard_normal((B, Cout, (H + 1) // 2, (W + 1) // 2)).astype(np.float32)
    v = 21 * 100 + tile
    for kw in (dict(res=res, act=1), dict(act=0), dict(up=up, act=2)):
        exp = reference(x, w, b, 1, 1, 0, 1, **kw)
        for split in (1, 2):
            if split > (Cin + 15) // 16:
                continue
            rc, got = run_conv(x, w, b, 1, 1, 0, 1, tile=v, split=split, **kw)
            assert rc == 0, lib().lib().fdt_last_error()
            assert rel_err(got, exp) < 1e-5, (tile, shape, list(kw), split, rel_err(got, exp))
    # against f64: not worse than the f32-MFMA class on the same data
    ref64 = np.einsum("oc,bchw->bohw", w[:, :, 0, 0].astype(np.float64), x.astype(np.float64)) + b[None, :, None, None]
    rc, got = run_conv(x, w, b, 1, 1, 0, 1, tile=v)
    rc2, f32 = run_conv(x, w, b, 1, 1, 0, 1, tile=6)
    assert rc == 0 and rc2 == 0
    e_b3 = float(np.sqrt(((got - ref64) ** 2).mean()) / np.sqrt((ref64 ** 2).mean()))
    e_f32 = float(np.sqrt(((f32 - ref64) ** 2).mean()) / np.sqrt((ref64 ** 2).mean()))
    print("split-bf16 vs f64: %.3e   f32 MFMA vs f64: %.3e   (tile %d, %s)" % (e_b3, e_f32, tile, shape))
    assert e_b3 <= 1.5 * e_f32 + 1e-7, (e_b3, e_f32)
    rc, _ = run_conv(x[:, :, :, :W - 1].copy(), w, b, 1, 1, 0, 1, tile=v)     # Win % 4 != 0: not this class
    assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [5, 6])
@pytest.mark.parametrize("shape", [(1, 64, 64, 64, 256), (2, 40, 35, 47, 72), (1, 19, 27, 39, 45), (1, 512, 32, 32, 128)])
def test_split_bf16_1x1_stride2(tile, shape):
    """Class 23 (conv.h: CONV_1x1_S2_B3): conv_b3.h with a dword gather of every second pixel in its staging -- the bottleneck's
    downsample branch (pyramid.py:87-91) as split-bf16 products.  Same tolerance as the f32 class of the same layer; odd input
    sizes, channels past Cin, couts past Cout, residual / activation, split-K; Wout % 4 != 0 is refused."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(tile * 1000 + Cin + 7)
    x = np.maximum(rng.standard_normal((B, Cin, H, W)), 0).astype(np.float32) * np.exp(rng.standard_normal((B, Cin, 1, 1))).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    res = rng.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
    v = 23 * 100 + tile
    for kw in (dict(act=0), dict(res=res, act=1)):
        exp = reference(x, w, b, 1, 2, 0, 1, **kw)
        for split in (1, 2):
            if split > (Cin + 15) // 16 // 2:
                continue
            rc, got = run_conv(x, w, b, 1, 2, 0, 1, tile=v, split=split, **kw)
            assert rc == 0, lib().lib().fdt_last_error()
            assert got.shape == exp.shape and rel_err(got, exp) < 1e-5, (tile, shape, list(kw), split, rel_err(got, exp))
    rc, _ = run_conv(x[:, :, :, :W - 2].copy(), w, b, 1, 2, 0, 1, tile=v)      # Wout % 4 != 0: not this class
    assert rc != 0


@pytest.mark.parametrize("tile", [34, 35])
@pytest.mark.parametrize("shape", [(1, 64, 64, 64, 256), (2, 56, 36, 48, 72), (3, 147, 27, 36, 200), (1, 1024, 32, 32, 256), (4, 64, 128, 128, 64)])
def test_persistent_split_bf16_1x1(tile, shape):
    """Class 26 (conv.h: CONV_1x1_S1_PB3, conv_1x1p_b3.h): the split-bf16 1x1 convolution as a persistent-tile kernel -- a workgroup
    walks consecutive output tiles, the LDS ring and the operand pipeline run across tile boundaries, register epilogue.  The same
    products in the same order as class 21 (tiles 5 / 6, waves 4 x 1): BIT-identical outputs, with bias / residual / activations,
    ragged tiles, channels past Cin, couts past Cout, several images (a workgroup's tile range crosses image boundaries); torch
    tolerance 1e-5.  Refused: fewer than three stages, split-K, the fused upsample-add, W % 4 != 0."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(tile * 1000 + Cin + B)
    x = np.maximum(rng.standard_normal((B, Cin, H, W)), 0).astype(np.float32) * np.exp(rng.standard_normal((B, Cin, 1, 1))).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    v, v21 = 26 * 100 + tile, 21 * 100 + (6 if tile == 34 else 5)
    for kw in (dict(res=res, act=1), dict(act=0), dict(act=2), dict(res=res, act=0)):
        exp = reference(x, w, b, 1, 1, 0, 1, **kw)
        rc, got = run_conv(x, w, b, 1, 1, 0, 1, tile=v, **kw)
        assert rc == 0, lib().lib().fdt_last_error()
        assert rel_err(got, exp) < 1e-5, (tile, shape, list(kw), rel_err(got, exp))
        rc, same = run_conv(x, w, b, 1, 1, 0, 1, tile=v21, **kw)
        assert rc == 0 and np.array_equal(got, same), (tile, shape, list(kw), float(np.abs(got - same).max()))
    rc, got = run_conv(x, w, None, 1, 1, 0, 1, tile=v)                        # no bias
    assert rc == 0 and rel_err(got, reference(x, w, None, 1, 1, 0, 1)) < 1e-5
    up = rng.standard_normal((B, Cout, (H + 1) // 2, (W + 1) // 2)).astype(np.float32)
    assert run_conv(x, w, b, 1, 1, 0, 1, tile=v, up=up)[0] != 0
    assert run_conv(x, w, b, 1, 1, 0, 1, tile=v, split=2)[0] != 0
    assert run_conv(x[:, :32].copy(), w[:, :32].copy(), b, 1, 1, 0, 1, tile=v)[0] != 0     # two stages
    assert run_conv(x[:, :, :, :W - 1].copy(), w, b, 1, 1, 0, 1, tile=v)[0] != 0


@pytest.mark.parametrize("tile", [5, 6])
@pytest.mark.parametrize("shape", [(1, 64, 64, 64, 128), (2, 40, 35, 47, 72), (1, 19, 27, 39, 45), (1, 128, 32, 32, 128), (2, 256, 16, 16, 256)])
def test_split_bf16_3x3_stride2(tile, shape):
    """Class 27 (conv.h: CONV_3x3_S2_B3): the 3x3 / stride-2 / padding-1 convolution of the first bottleneck of layer2-4
    (pyramid.py:99) as split-bf16 products -- conv_b3.h with nine tap stages per 16-channel group, the gather of a tap = the centre
    tap's offsets plus a constant, padding as out-of-range offsets.  Torch tolerance 1e-5 (the f32 class's), every image border
    (odd and even input sizes), channels past Cin, couts past Cout, bias / residual / activation, split-K between channel groups;
    error against an f64 convolution not above the f32 class's; Wout % 4 != 0 is refused."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(tile * 1000 + Cin + 3)
    x = np.maximum(rng.standard_normal((B, Cin, H, W)), 0).astype(np.float32) * np.exp(rng.standard_normal((B, Cin, 1, 1))).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    res = rng.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
    v = 27 * 100 + tile
    for kw in (dict(act=1), dict(res=res, act=0), dict(act=2)):
        exp = reference(x, w, b, 3, 2, 1, 1, **kw)
        for split in (1, 2, 4):
            if split > (Cin + 15) // 16:
                continue
            rc, got = run_conv(x, w, b, 3, 2, 1, 1, tile=v, split=split, **kw)
            assert rc == 0, lib().lib().fdt_last_error()
            assert got.shape == exp.shape and rel_err(got, exp) < 1e-5, (tile, shape, list(kw), split, rel_err(got, exp))
    ref64 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), 2, 1).numpy()
    rc, got = run_conv(x, w, b, 3, 2, 1, 1, tile=v)
    rc2, f32 = run_conv(x, w, b, 3, 2, 1, 1, tile=1)          # the f32-MFMA class of the same layer
    assert rc == 0 and rc2 == 0
    e_b3 = float(np.sqrt(((got - ref64) ** 2).mean()) / np.sqrt((ref64 ** 2).mean()))
    e_f32 = float(np.sqrt(((f32 - ref64) ** 2).mean()) / np.sqrt((ref64 ** 2).mean()))
    assert e_b3 <= 1.5 * e_f32 + 1e-7, (e_b3, e_f32)
    rc, _ = run_conv(x[:, :, :, :W - 2].copy(), w, b, 3, 2, 1, 1, tile=v)      # Wout % 4 != 0: not this class
    assert rc != 0


def test_persistent_1x1_refuses_what_it_is_not_built_for():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 40, 20, 30)).astype(np.float32)           # W % 4 != 0
    w = rng.standard_normal((64, 40, 1, 1)).astype(np.float32)
    rc, _ = run_conv(x, w, None, 1, 1, 0, 1, tile=16 * 100 + T_P64)
    assert rc != 0 and b"not instantiated" in lib().lib().fdt_last_error()
    x = rng.standard_normal((1, 40, 20, 32)).astype(np.float32)
    rc, _ = run_conv(x, w, None, 1, 1, 0, 1, tile=16 * 100 + T_P64, split=2)   # no split-K
    assert rc != 0 and b"not instantiated" in lib().lib().fdt_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, 128, 24), (1, 1024, 1024, 24), (3, 45, 77, 24), (2, 130, 250, 45), (1, 7, 9, 5)])
def test_facebox_stem_k168(shape):
    """conv_stem_s4.h (CONV_7x7_S4_K168): 7x7 / stride 4 / pad 3 on three input channels as 84 k-pairs (3 x 7 x 8 columns, the
    eighth column's weights zero) from a column-phase de-interleaved patch staged by LDS-DMA -- against torch, incl. odd sizes,
    images smaller than a tile, two channel tiles, every activation."""
    B, H, W, Cout = shape
    rng = np.random.default_rng(H * 1000 + W)
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    for act, bias in ((1, b), (0, None), (2, b)):
        rc, got = run_conv(x, w, bias, 7, 4, 3, 1, act=act, tile=36)
        assert rc == 0, lib().lib().fdt_last_error()
        exp = reference(x, w, bias, 7, 4, 3, 1, act=act)
        assert rel_err(got, exp) < 1e-5, (shape, act, rel_err(got, exp))
    x5 = rng.standard_normal((1, 5, 32, 32)).astype(np.float32)                    # not the three-channel stem: refused
    rc, _ = run_conv(x5, rng.standard_normal((8, 5, 7, 7)).astype(np.float32), None, 7, 4, 3, 1, tile=36)
    assert rc != 0 and b"not instantiated" in lib().lib().fdt_last_error()


@pytest.mark.parametrize("shape", [(2, 64, 128, 24), (1, 1024, 1024, 24), (3, 45, 76, 24), (2, 130, 252, 45), (1, 7, 12, 5)])
def test_facebox_stem_b3(shape):
    """conv_stem_b3.h (CONV_7x7_S4_B3, class 22): FaceBoxes' 7x7 / stride 4 / pad 3 stem as split-bf16 products -- 11 k-steps of
    two (channel, tap row) pairs x 8 columns, three bf16 planes per operand, six plane products per k-step -- against torch at the
    f32 classes' tolerance, incl. odd heights, images smaller than a tile, two channel tiles, every activation; and an error
    against f64 no larger than the f32 kernel's (x1.5)."""
    B, H, W, Cout = shape
    rng = np.random.default_rng(H * 1000 + W)
    x = rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32)                 # what FaceBoxes feeds: BGR / 255
    w = (rng.standard_normal((Cout, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    for act, bias in ((1, b), (0, None), (2, b)):
        rc, got = run_conv(x, w, bias, 7, 4, 3, 1, act=act, tile=22 * 100 + 36)
        assert rc == 0, lib().lib().fdt_last_error()
        exp = reference(x, w, bias, 7, 4, 3, 1, act=act)
        assert rel_err(got, exp) < 1e-5, (shape, act, rel_err(got, exp))
    ref64 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, 4, 3).numpy()
    rc, got = run_conv(x, w, None, 7, 4, 3, 1, tile=22 * 100 + 36)
    rc2, f32 = run_conv(x, w, None, 7, 4, 3, 1, tile=36)
    assert rc == 0 and rc2 == 0
    e_b3 = float(np.sqrt(((got - ref64) ** 2).mean()) / np.sqrt((ref64 ** 2).mean()))
    e_f32 = float(np.sqrt(((f32 - ref64) ** 2).mean()) / np.sqrt((ref64 ** 2).mean()))
    print("stem: split-bf16 vs f64 %.3e, f32 MFMA vs f64 %.3e (%s)" % (e_b3, e_f32, shape))
    assert e_b3 <= 1.5 * e_f32 + 1e-7
    rc, _ = run_conv(x[:, :, :, :W - 2].copy(), w, None, 7, 4, 3, 1, tile=22 * 100 + 36)     # Win % 4 != 0: not this class
    assert rc != 0
