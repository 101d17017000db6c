"""Deterministic synthetic weights and frames for the PyramidBox hot path.

The reference ships no PyramidBox weights (reference README.md:35-39 points at a
Baidu-pan download), so parity fixtures, smoke() and bench.py all use a seeded
state-dict with the *reference's own key names and shapes*:

* Res50:  reference pyramid.py:106-198 (`SFD.__init__`)
* try3:   reference pyramid_mb2_try3.py:137-216 (`SFD_mobile.__init__`)

Every tensor is drawn from its own `numpy.random.default_rng([seed, crc32(name)])`
stream, so the dict is independent of iteration order and can be rebuilt on the
GPU box without shipping 256 MB (or anything from the reference).
"""
import zlib
from collections import OrderedDict

import numpy as np

__all__ = ["res50_schema", "try3_schema", "make_state_dict", "make_frames",
           "RES50_CONF_SHIFT", "TRY3_CONF_SHIFT"]

# Constant added to the "face" logit of every face_conf head so that a seeded
# random-weight net yields a few hundred candidates per 1024x1024 frame instead
# of tens of thousands.  Found with tools in tests/golden/.
RES50_CONF_SHIFT = -6.65
TRY3_CONF_SHIFT = -4.72
TRY12_CONF_SHIFT = -3.0


def _conv(out, name, cin, cout, k, bias, groups=1, kind="conv_w"):
    # conv_w: He init (followed by ReLU); conv_w_lin: 1/fan_in (no ReLU after it);
    # conv_w_head: small, keeps logits / box deltas at O(1).
    out.append((name + ".weight", (cout, cin // groups, k, k), kind))
    if bias:
        out.append((name + ".bias", (cout,), "conv_b"))


def _bn(out, name, c, stem=False, res=False):
    # stem: running stats sized for raw pixel input (|x| ~ 74) so activations enter the
    # net at O(1); res: small gamma on the last BN of a residual branch so 16 stacked
    # blocks do not double the variance each time.
    out.append((name + ".weight", (c,), "bn_w_res" if res else "bn_w"))
    out.append((name + ".bias", (c,), "bn_b"))
    out.append((name + ".running_mean", (c,), "bn_mean_stem" if stem else "bn_mean"))
    out.append((name + ".running_var", (c,), "bn_var_stem" if stem else "bn_var"))
    out.append((name + ".num_batches_tracked", (), "bn_nbt"))


def _ssh(out, name, c, xc):
    _conv(out, name + ".conv1", c, xc, 3, True)
    _conv(out, name + ".conv2", c, xc // 2, 3, True)
    _conv(out, name + ".conv2_1", xc // 2, xc // 2, 3, True)
    _conv(out, name + ".conv2_2", xc // 2, xc // 2, 3, True)
    _conv(out, name + ".conv2_2_1", xc // 2, xc // 2, 3, True)


def _ct(out, name, up, main):
    _conv(out, name + ".up_conv", up, main, 1, True, kind="conv_w_lin")
    _conv(out, name + ".main_conv", main, main, 1, True, kind="conv_w_lin")


def res50_schema():
    """(name, shape, kind) in the order of reference `SFD.state_dict()`."""
    s = []
    _conv(s, "conv1", 3, 64, 7, False)
    _bn(s, "bn1", 64, stem=True)
    in_planes = 64
    for li, (planes, nblk, stride) in enumerate(
            [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], start=1):
        for b in range(nblk):
            st = stride if b == 0 else 1
            p = "layer%d.%d" % (li, b)
            _conv(s, p + ".conv1", in_planes, planes, 1, False)
            _bn(s, p + ".bn1", planes)
            _conv(s, p + ".conv2", planes, planes, 3, False)
            _bn(s, p + ".bn2", planes)
            _conv(s, p + ".conv3", planes, planes * 4, 1, False)
            _bn(s, p + ".bn3", planes * 4, res=True)
            if st != 1 or in_planes != planes * 4:
                _conv(s, p + ".downsample.0", in_planes, planes * 4, 1, False)
                _bn(s, p + ".downsample.1", planes * 4)
            in_planes = planes * 4
    _conv(s, "layer5.0", 2048, 512, 1, True)
    _bn(s, "layer5.1", 512)
    _conv(s, "layer5.3", 512, 512, 3, True)
    _bn(s, "layer5.4", 512)
    _conv(s, "layer6.0", 512, 128, 1, True)
    _bn(s, "layer6.1", 128)
    _conv(s, "layer6.3", 128, 256, 3, True)
    _bn(s, "layer6.4", 256)
    _ct(s, "conv3_ct_py", 512, 256)
    _ct(s, "conv4_ct_py", 1024, 512)
    _ct(s, "conv5_ct_py", 2048, 1024)
    _conv(s, "latlayer_fc", 2048, 2048, 1, True, kind="conv_w_lin")
    _conv(s, "latlayer_c6", 512, 512, 1, True, kind="conv_w_lin")
    _conv(s, "latlayer_c7", 256, 256, 1, True, kind="conv_w_lin")
    _conv(s, "smooth_c3", 256, 256, 3, True, kind="conv_w_lin")
    _conv(s, "smooth_c4", 512, 512, 3, True, kind="conv_w_lin")
    _conv(s, "smooth_c5", 1024, 1024, 3, True, kind="conv_w_lin")
    for n, c in zip(range(2, 8), (256, 512, 1024, 2048, 512, 256)):
        _ssh(s, "conv%d_SSH" % n, c, 256)
    for i in range(6):
        _conv(s, "face_loc.%d" % i, 512, 4, 3, True, kind="conv_w_head")
    for i in range(6):
        _conv(s, "face_conf.%d" % i, 512, 4, 3, True, kind="conv_w_head")
    for i in range(5):
        _conv(s, "head_loc.%d" % i, 512, 4, 3, True, kind="conv_w_head")
    for i in range(5):
        _conv(s, "head_conf.%d" % i, 512, 2, 3, True, kind="conv_w_head")
    return s


def _ir(out, name, inp, oup, t, stride=1):
    hid = int(round(inp * t))
    i = 0
    if t != 1:
        _conv(out, "%s.conv.%d" % (name, i), inp, hid, 1, False)
        _bn(out, "%s.conv.%d" % (name, i + 1), hid)
        i += 3
    _conv(out, "%s.conv.%d" % (name, i), hid, hid, 3, False, groups=hid)
    _bn(out, "%s.conv.%d" % (name, i + 1), hid)
    i += 3
    _conv(out, "%s.conv.%d" % (name, i), hid, oup, 1, False)
    _bn(out, "%s.conv.%d" % (name, i + 1), oup, res=(stride == 1 and inp == oup))


TRY3_CFGS = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
             (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def try3_blocks():
    """[(feature index, inp, oup, stride, t)] of reference pyramid_mb2_try3.py:150-168."""
    blocks = []
    inp, idx = 32, 1
    for t, c, n, st in TRY3_CFGS:
        for i in range(n):
            blocks.append((idx, inp, c, st if i == 0 else 1, t))
            inp = c
            idx += 1
    return blocks


def try3_schema(variant=3):
    """(name, shape, kind) in the order of reference `SFD_mobile.state_dict()`; variant 4 / 5 = the try4 / try5
    siblings (pyramid_mb2_try4.py:16,184-191, pyramid_mb2_try5.py:184-191)."""
    s = []
    _conv(s, "features.0.0", 3, 32, 7 if variant == 4 else 3, False)
    _bn(s, "features.0.1", 32, stem=True)
    for idx, inp, oup, st, t in try3_blocks():
        _ir(s, "features.%d" % idx, inp, oup, t, st)
    _ir(s, "layer6", 320, 160, 6, 2)
    _ct(s, "conv2_ct_py", 32, 24)
    _ct(s, "conv3_ct_py", 96, 32)
    _ct(s, "conv4_ct_py", 320, 96)
    for n, c, t in zip(range(2, 7), (24, 32, 96, 320, 160), (4, 4, 2, 0, 0)):
        if variant != 3 and t:
            _ir(s, "smooth_c%d.0" % n, c, c, t, 1)
            _conv(s, "smooth_c%d.1" % n, c, c, 3, True, kind="conv_w_lin")
        else:
            k = 1 if (variant != 3 and n == 6) or (variant == 4 and n == 5) else 3
            _conv(s, "smooth_c%d" % n, c, c, k, True, kind="conv_w_lin")
    for n, c in zip(range(2, 7), (24, 32, 96, 320, 160)):
        _ssh(s, "conv%d_SSH" % n, c, 128)
    for i in range(6):
        _conv(s, "face_loc.%d" % i, 256, 4, 3, True, kind="conv_w_head")
    for i in range(6):
        _conv(s, "face_conf.%d" % i, 256, 4, 3, True, kind="conv_w_head")
    for i in range(5):
        _conv(s, "head_loc.%d" % i, 256, 4, 3, True, kind="conv_w_head")
    for i in range(5):
        _conv(s, "head_conf.%d" % i, 256, 2, 3, True, kind="conv_w_head")
    return s


# pyramid_mobile_try1.py:160-181 / pyramid_mobile_try2.py:163-189: (inp, oup, dw kernel, stride, t, pad, dil, side_way)
TRY1_LAYERS = [[(64, 64, 3, 1, 2, 1, 1, 1), (64, 64, 3, 1, 2, 1, 1, 1), (64, 256, 3, 1, 2, 1, 1, 0)],
               [(256, 64, 5, 2, 2, 2, 1, 0), (64, 512, 3, 1, 2, 2, 2, 0)],
               [(512, 256, 5, 2, 2, 2, 1, 0), (256, 256, 5, 1, 2, 2, 1, 1), (256, 1024, 3, 1, 2, 2, 2, 0)],
               [(1024, 256, 5, 2, 2, 2, 1, 0), (256, 2048, 3, 1, 2, 1, 1, 0)]]
TRY2_LAYERS = [[(64, 64, 3, 1, 4, 1, 1, 1)] * 3,
               [(64, 64, 3, 2, 4, 1, 1, 0), (64, 64, 3, 1, 4, 1, 1, 1), (64, 64, 3, 1, 4, 1, 1, 1),
                (64, 128, 3, 1, 4, 1, 1, 0)],
               [(128, 128, 3, 2, 2, 1, 1, 0)] + [(128, 128, 3, 1, 2, 1, 1, 1)] * 4 + [(128, 256, 3, 1, 2, 1, 1, 0)],
               [(256, 256, 3, 2, 4, 1, 1, 0), (256, 256, 3, 1, 4, 1, 1, 1), (256, 512, 3, 1, 4, 1, 1, 0)]]


def _mbv2(out, p, inp, oup, k, t, side, dw_bias=False):
    hid = inp * t
    _conv(out, p + ".conv1", inp, hid, 1, False)
    _bn(out, p + ".bn1", hid)
    _conv(out, p + ".conv2", hid, hid, k, dw_bias, groups=hid)
    _bn(out, p + ".bn2", hid)
    _conv(out, p + ".conv3", hid, oup, 1, False, kind="conv_w_lin")
    _bn(out, p + ".bn3", oup, res=bool(side))


def _mbv1(out, p, cin, cout, k, dw_bias=False, stem=False, kind="conv_w"):
    _conv(out, p + ".conv1", cin, cin, k, dw_bias, groups=cin)
    _bn(out, p + ".bn", cin, stem=stem)
    _conv(out, p + ".conv2", cin, cout, 1, False, kind=kind)


def try12_schema(variant):
    """(name, shape, kind) in the order of reference pyramid_mobile_try1.py / _try2.py `SFD_mobile.state_dict()`."""
    s = []
    _mbv1(s, "conv1_my", 3, 64, 7, stem=True)
    _bn(s, "bn1", 64)
    layers = TRY1_LAYERS if variant == 1 else TRY2_LAYERS
    adj = (256, 512, 1024, 2048)
    for li, blocks in enumerate(layers, start=1):
        for bi, (inp, oup, k, st, t, pad, dil, side) in enumerate(blocks):
            _mbv2(s, "layer%d_my.%d" % (li, bi), inp, oup, k, t, side)
        if variant == 2:
            _conv(s, "layer%d_adj" % li, blocks[-1][1], adj[li - 1], 1, False, kind="conv_w_lin")
    t56 = 2 if variant == 1 else 4
    _mbv2(s, "layer5_my", 2048 if variant == 1 else 512, 512, 3, t56, 0, dw_bias=(variant == 2))
    _mbv2(s, "layer6_my", 512, 256, 3, t56, 0, dw_bias=(variant == 2))
    for n, c in ((3, 256), (4, 512), (5, 1024)):
        _mbv1(s, "smooth_c%d_my" % n, c, c, 3, dw_bias=(variant == 2), kind="conv_w_lin")
    _conv(s, "latlayer_fc_my", 2048, 2048, 1, True, groups=4, kind="conv_w_lin")
    _conv(s, "latlayer_c6_my", 512, 512, 1, True, groups=2, kind="conv_w_lin")
    _conv(s, "latlayer_c7_my", 256, 256, 1, True, kind="conv_w_lin")
    _ct(s, "conv3_ct_py", 512, 256)
    _ct(s, "conv4_ct_py", 1024, 512)
    _ct(s, "conv5_ct_py", 2048, 1024)
    for n, c in zip(range(2, 8), (256, 512, 1024, 2048, 512, 256)):
        _ssh(s, "conv%d_SSH" % n, c, 256)
    for i in range(6):
        _conv(s, "face_loc.%d" % i, 512, 4, 3, True, kind="conv_w_head")
    for i in range(6):
        _conv(s, "face_conf.%d" % i, 512, 4, 3, True, kind="conv_w_head")
    for i in range(5):
        _conv(s, "head_loc.%d" % i, 512, 4, 3, True, kind="conv_w_head")
    for i in range(5):
        _conv(s, "head_conf.%d" % i, 512, 2, 3, True, kind="conv_w_head")
    return s


def _draw(name, shape, kind, seed):
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    if kind == "conv_w":
        fan_in = shape[1] * shape[2] * shape[3]
        return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    if kind == "conv_w_lin":
        fan_in = shape[1] * shape[2] * shape[3]
        return (rng.standard_normal(shape) * np.sqrt(1.0 / fan_in)).astype(np.float32)
    if kind == "conv_w_head":
        fan_in = shape[1] * shape[2] * shape[3]
        return (rng.standard_normal(shape) * (0.5 * np.sqrt(1.0 / fan_in))).astype(np.float32)
    if kind == "conv_b":
        return (rng.standard_normal(shape) * 0.05).astype(np.float32)
    if kind == "bn_w":
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if kind == "bn_w_res":
        return rng.uniform(0.2, 0.3, shape).astype(np.float32)
    if kind == "bn_mean_stem":
        return (rng.standard_normal(shape) * 10.0).astype(np.float32)
    if kind == "bn_var_stem":
        return rng.uniform(9000.0, 13000.0, shape).astype(np.float32)
    if kind == "bn_b":
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if kind == "bn_mean":
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if kind == "bn_var":
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if kind == "bn_nbt":
        return np.array(0, dtype=np.int64)
    raise ValueError(kind)


def make_state_dict(arch="res50", seed=0, conf_shift=None):
    """OrderedDict[name -> numpy array] with the reference's keys/shapes.

    `conf_shift` is added to the face logit of every `face_conf.N.bias`
    (channel 3 on level 0, channels 1..3 on the others: the max-in-out layout of
    reference pyramid.py:291-305).
    """
    if arch == "res50":
        schema, default_shift = res50_schema(), RES50_CONF_SHIFT
    elif arch == "try3":
        schema, default_shift = try3_schema(), TRY3_CONF_SHIFT
    elif arch in ("try4", "try5"):
        schema, default_shift = try3_schema(int(arch[3])), TRY3_CONF_SHIFT
    elif arch in ("try1", "try2"):
        schema, default_shift = try12_schema(int(arch[3])), TRY12_CONF_SHIFT
    else:
        raise ValueError("unknown arch %r" % (arch,))
    if conf_shift is None:
        conf_shift = default_shift
    sd = OrderedDict()
    for name, shape, kind in schema:
        sd[name] = _draw(name, shape, kind, seed)
    for i in range(6):
        b = sd["face_conf.%d.bias" % i]
        if i == 0:
            b[3] += np.float32(conf_shift)
        else:
            b[1:4] += np.float32(conf_shift)
    return sd


def make_frames(n, height, width, seed=1234):
    """uint8 BGR HWC frames, i.i.d. uniform (SURVEY.md 8(d) synthetic input)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, height, width, 3), dtype=np.uint8)
