"""GPU parity of the detector forward (through the C ABI) against the oracle and the golden
fixtures produced by the reference.  Convolutions accumulate in exact f32 on the matrix cores in a
different order than oneDNN, so stage tensors are compared with a relative-RMS tolerance and final
boxes by IoU (north_star: boxes within 1e-3 IoU of the reference PyTorch-CPU path)."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_npz
from oracle import postproc as opp
from oracle import pyramidbox as opb

pytestmark = pytest.mark.gpu

STAGE_RTOL = 2e-5      # relative RMS error allowed per stage tensor (f32 accumulation-order noise)
BOX_IOU_TOL = 1e-3     # north_star
SCORE_ATOL = 1e-4


def M(name):
    return importlib.import_module("face-detection-and-tracking_amd." + name)


@pytest.fixture(scope="module")
def res50(res50_sd):
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(res50_sd)
    net.cuda(); net.eval()
    yield net
    net.close()


@pytest.fixture(scope="module")
def try3(try3_sd):
    net = M("pyramid_mb2_try3").build_sfd_mobile('test', 640, 2)
    net.load_state_dict(try3_sd)
    yield net
    net.close()


def rel_rms(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


def match_detections(got, exp, n_exp):
    """Same count; every reference box has a GPU box within the IoU / score tolerance.  Matching is by
    best IoU, not by row: two detections whose scores differ by less than the f32 accumulation noise
    (~3e-6) may legitimately swap places in the score-ordered output."""
    n_got = int((got[:, 0] > 0).sum())
    assert n_got == n_exp, (n_got, n_exp)
    if n_exp == 0:
        return 0.0, 0.0
    iou = opp.calculate_iou(exp[:n_exp, 1:].astype(np.float64), got[:n_exp, 1:].astype(np.float64))
    j = iou.argmax(1)
    assert len(set(j.tolist())) == n_exp, "two reference boxes matched the same detection"
    d_sc = np.abs(got[j, 0] - exp[:n_exp, 0]).max()
    # rows may only move among (near-)equal scores
    assert np.abs(got[:n_exp, 0] - exp[:n_exp, 0]).max() <= SCORE_ATOL
    return float((1 - iou[np.arange(n_exp), j]).max()), float(d_sc)


RES50_STAGES = ["stem", "pool", "c2", "c3", "c4", "c5", "c6", "c7", "c4_ct", "c3_ct", "c2_ct",
                "c2_smooth", "c3_smooth", "c4_smooth", "src0", "src1", "src2", "src3", "src4", "src5"]


@pytest.mark.parametrize("H,W,seed", [(64, 64, 7), (136, 200, 8), (256, 256, 21)])
def test_res50_stages_vs_oracle(res50, res50_sd, synth, H, W, seed):
    frame = synth.make_frames(1, H, W, seed=seed)[0]
    x = opb.preprocess(frame)
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.02, 0.35)
    y = res50(x).numpy()
    o = opb.res50_forward(res50_sd, x, want=RES50_STAGES + ["conf_logits"])
    for st in RES50_STAGES:
        got = res50.get_tensor(st)
        assert got.shape == o[st].shape, st
        assert rel_rms(got, o[st]) < STAGE_RTOL, (st, rel_rms(got, o[st]))
    assert rel_rms(res50.get_tensor("loc"), o["loc"]) < STAGE_RTOL
    np.testing.assert_allclose(res50.get_tensor("conf"), o["conf"], atol=SCORE_ATOL, rtol=0)
    pri = opp.build_priors(opp.PriorBoxLayer(W, H), H, W)
    assert np.array_equal(res50.priors.numpy(), pri)
    exp = opp.Detect(2, 0, 750, 0.02, 0.35)(o["loc"], o["conf"], pri)
    n = int((exp[0, 1, :, 0] > 0).sum())
    d_iou, d_sc = match_detections(y[0, 1], exp[0, 1], n)
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL
    assert not y[:, 0].any()


@pytest.mark.parametrize("key", ["res50_64x64", "res50_136x200", "res50_480x640", "res50_1024x1024"])
def test_res50_vs_reference_fixture(res50, synth, key):
    """Final detections vs what the reference itself produced (tests/golden/nets.npz)."""
    d, meta = load_npz("nets")
    m = meta[key]
    H, W = m["H"], m["W"]
    frame = synth.make_frames(1, H, W, seed=m["frame_seed"])[0]
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    y = res50(frame).numpy()          # uint8 frame in: mean subtraction on the GPU
    exp = d[key + "_out"]
    d_iou, d_sc = match_detections(y[0, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]),
                                   m["n_out"])
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL
    if key + "_sel" in d:
        sel = d[key + "_sel"]
        np.testing.assert_allclose(res50.get_tensor("loc")[0, sel], d[key + "_loc_s"], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(res50.get_tensor("conf")[0, sel], d[key + "_conf_s"], atol=SCORE_ATOL, rtol=0)


def test_res50_batch2_equals_two_singles(res50, synth):
    frames = synth.make_frames(2, 128, 160, seed=33)
    res50.priorbox = M("layers").PriorBoxLayer(160, 128); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    yb = res50(frames).numpy()
    y0 = res50(frames[0]).numpy(); y1 = res50(frames[1]).numpy()
    # the (tile, split-K) plan depends on the batch size, so sums are re-associated: tolerance, not bits
    for got, exp in ((yb[0, 1], y0[0, 1]), (yb[1, 1], y1[0, 1])):
        n = int((exp[:, 0] > 0).sum())
        assert n > 5
        d_iou, d_sc = match_detections(got, exp, n)
        assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


def test_cuda_tensor_inputs_equal_host_inputs(res50, synth):
    """`net(x.cuda())` as the reference writes it (iouTracke_cal.py:49-50): f32 NCHW and u8 HWC tensors that already live on
    the GPU are consumed there and give the bits of the host-array call."""
    H, W = 128, 160
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames = synth.make_frames(2, H, W, seed=21)
    x = np.stack([opb.preprocess(f)[0] for f in frames])              # [2,3,H,W] f32, mean-subtracted BGR
    want_f32, want_u8 = res50(x).numpy(), res50(frames).numpy()
    dev = torch.device("cuda", 0)
    assert np.array_equal(res50(torch.from_numpy(x).to(dev)).numpy(), want_f32)
    assert np.array_equal(res50(torch.from_numpy(frames).to(dev)).numpy(), want_u8)
    assert np.array_equal(res50(torch.from_numpy(x)).numpy(), want_f32)                     # CPU tensor
    y = res50(torch.from_numpy(x[:1]).to(dev).double())                                     # converted like the host path
    assert np.array_equal(y.numpy(), res50(x[:1]).numpy()) and not y.is_cuda and res50.last_counts.shape == (1, 2)


def test_one_handle_through_many_configurations(res50, res50_sd, synth):
    """The mutable attributes the reference's callers poke (My_test.py:31-36: priorbox, firstTime, detect) changed between
    calls on ONE handle -- record size (top_k 750 / 200 / 30 / 1200), thresholds, frame size down to 40x56 (1x1 maps at the
    last levels), an odd batch -- each configuration run three times (eager, captured, replayed) and compared with the oracle."""
    def check(H, W, topk, conf, nms, B=1):
        res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
        res50.detect = M("layers").Detect(2, 0, topk, conf, nms)
        fr = synth.make_frames(B, H, W, seed=3)
        for _ in range(3):
            y = res50(fr).numpy()
        assert y.shape == (B, 2, topk, 5)
        for b in range(B):
            exp = opb.detect_frame(res50_sd, fr[b], "res50", detect=opp.Detect(2, 0, topk, conf, nms))
            n = int((exp[0, 1, :, 0] > 0).sum())
            d_iou, d_sc = match_detections(y[b, 1], exp[0, 1], n)
            assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL, (H, W, topk, b, d_iou, d_sc)
        return n
    assert check(128, 160, 750, 0.05, 0.35) > 10
    check(128, 160, 200, 0.05, 0.35)                    # smaller record, same plan
    assert check(128, 160, 30, 0.02, 0.5) == 30         # more survivors than top_k keeps
    check(128, 160, 750, 0.05, 0.35)                    # and back
    check(40, 56, 750, 0.05, 0.35)
    check(128, 160, 750, 0.05, 0.35, B=5)
    assert check(128, 160, 1200, 0.01, 0.6) > 100       # a record larger than the usual 750


def test_load_state_dict_strictness(res50_sd):
    net = M("pyramid").build_sfd('test', 640, 2)
    sd = dict(res50_sd); sd.pop("layer1.0.conv1.weight")
    with pytest.raises(RuntimeError, match="Missing key"):
        net.load_state_dict(sd)
    sd = dict(res50_sd); sd["bogus.weight"] = np.zeros(3, np.float32)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        net.load_state_dict(sd)
    assert M("pyramid").build_sfd('test', 512, 2) is None
    assert M("pyramid").build_sfd('bogus', 640, 2) is None
    net.close()


TRY3_STAGES = ["stem", "c2", "c3", "c4", "c5", "c6", "c2_smooth", "c3_smooth", "c4_smooth", "c5_smooth",
               "c6_smooth", "src0", "src1", "src2", "src3", "src4"]


@pytest.mark.parametrize("H,W,seed", [(64, 64, 7), (136, 200, 8)])
def test_try3_stages_vs_oracle(try3, try3_sd, synth, H, W, seed):
    frame = synth.make_frames(1, H, W, seed=seed)[0]
    x = opb.preprocess(frame)
    PB = M("layers").PriorBoxLayer
    try3.priorbox = PB(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256)); try3.firstTime = True
    try3.detect = M("layers").Detect(2, 0, 750, 0.02, 0.35)
    y = try3(x).numpy()
    o = opb.try3_forward(try3_sd, x, want=TRY3_STAGES)
    for st in TRY3_STAGES:
        got = try3.get_tensor(st)
        assert got.shape == o[st].shape, st
        assert rel_rms(got, o[st]) < STAGE_RTOL, (st, rel_rms(got, o[st]))
    pri = opp.build_priors(opp.PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256)),
                           H, W, "try3")
    exp = opp.Detect(2, 0, 750, 0.02, 0.35)(o["loc"], o["conf"], pri)
    n = int((exp[0, 1, :, 0] > 0).sum())
    d_iou, d_sc = match_detections(y[0, 1], exp[0, 1], n)
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


@pytest.mark.parametrize("key", ["try3_64x64", "try3_136x200", "try3_480x640", "try3_1024x1024"])
def test_try3_vs_reference_fixture(try3, synth, key):
    d, meta = load_npz("nets")
    m = meta[key]
    H, W = m["H"], m["W"]
    frame = synth.make_frames(1, H, W, seed=m["frame_seed"])[0]
    PB = M("layers").PriorBoxLayer
    try3.priorbox = PB(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256)); try3.firstTime = True
    try3.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    y = try3(frame).numpy()
    exp = d[key + "_out"]
    d_iou, d_sc = match_detections(y[0, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]),
                                   m["n_out"])
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


def test_autotuned_plan_keeps_parity(res50, res50_sd, synth):
    """The autotuner may pick any instantiated (tile, split-K) variant per layer (ring / wide tiles,
    vector staging): the tuned forward must stay inside the same tolerance as the default plan."""
    H, W = 192, 256
    frame = synth.make_frames(1, H, W, seed=41)[0]
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    res50(frame)
    res50.autotune(2)
    y = res50(frame).numpy()
    o = opb.res50_forward(res50_sd, opb.preprocess(frame), want=["c2", "c5", "src0", "src3", "src5"])
    for st in ["c2", "c5", "src0", "src3", "src5"]:
        assert rel_rms(res50.get_tensor(st), o[st]) < STAGE_RTOL, st
    exp = opp.Detect(2, 0, 750, 0.05, 0.35)(o["loc"], o["conf"], opp.build_priors(opp.PriorBoxLayer(W, H), H, W))
    n = int((exp[0, 1, :, 0] > 0).sum())
    d_iou, d_sc = match_detections(y[0, 1], exp[0, 1], n)
    assert n > 10 and d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


# ------------------------------------------------------------------ try4 / try5
@pytest.fixture(scope="module", params=["try4", "try5"])
def try45(request, synth):
    arch = request.param
    net = M("pyramid_mb2_" + arch).build_sfd_mobile('test', 640, 2)
    sd = synth.make_state_dict(arch, seed=0)
    net.load_state_dict(sd)
    yield arch, net, sd
    net.close()


@pytest.mark.parametrize("H,W,seed", [(64, 64, 27), (136, 200, 28)])
def test_try45_stages_vs_oracle(try45, synth, H, W, seed):
    """7x7/pad-1 stem (try4), InvertedResidual smooth layers, 1x1 convs with padding 1 (sources grow by a
    zero-padded pixel per side) -- stage tensors, priors (bit-exact) and detections against the oracle."""
    arch, net, sd = try45
    frame = synth.make_frames(1, H, W, seed=seed)[0]
    x = opb.preprocess(frame)
    PB = M("layers").PriorBoxLayer
    net.priorbox = PB(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256)); net.firstTime = True
    net.detect = M("layers").Detect(2, 0, 750, 0.02, 0.35)
    y = net(x).numpy()
    o = opb.try3_forward(sd, x, want=TRY3_STAGES, variant=int(arch[3]))
    assert [tuple(s) for s in net.source_sizes(H, W)] == [tuple(s) for s in o["source_sizes"]]
    for st in TRY3_STAGES:
        got = net.get_tensor(st)
        assert got.shape == o[st].shape, (st, got.shape, o[st].shape)
        assert rel_rms(got, o[st]) < STAGE_RTOL, (st, rel_rms(got, o[st]))
    pri = opp.build_priors(opp.PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256)),
                           H, W, arch, sizes=o["source_sizes"])
    assert np.array_equal(net.priors.numpy(), pri)
    loc, conf = net.forward_raw(x)
    assert loc.shape == o["loc"].shape and rel_rms(loc, o["loc"]) < STAGE_RTOL
    exp = opp.Detect(2, 0, 750, 0.02, 0.35)(o["loc"], o["conf"], pri)
    n = int((exp[0, 1, :, 0] > 0).sum())
    d_iou, d_sc = match_detections(y[0, 1], exp[0, 1], n)
    assert n > 5 and d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


@pytest.mark.parametrize("size", ["64x64", "136x200", "480x640"])
def test_try45_vs_reference_fixture(try45, synth, size):
    """Final detections vs what the reference modules pyramid_mb2_try4.py / _try5.py produced here
    (tests/golden/nets45.npz)."""
    arch, net, _ = try45
    d, meta = load_npz("nets45")
    key = "%s_%s" % (arch, size)
    m = meta[key]
    H, W = m["H"], m["W"]
    frame = synth.make_frames(1, H, W, seed=m["frame_seed"])[0]
    PB = M("layers").PriorBoxLayer
    net.priorbox = PB(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256)); net.firstTime = True
    net.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    y = net(frame).numpy()
    assert net.priors.shape[0] == m["P"]
    exp = d[key + "_out"]
    d_iou, d_sc = match_detections(y[0, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]),
                                   m["n_out"])
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


# ------------------------------------------------------------------ try1 / try2
TRY12_STAGES = ["stem", "c2", "c3", "c4", "c5", "c6", "c7", "c4_ct", "c3_ct", "c2_ct", "c2_smooth", "c3_smooth",
                "c4_smooth", "src0", "src1", "src2", "src3", "src4", "src5"]


@pytest.fixture(scope="module", params=["try1", "try2"])
def try12(request, synth):
    arch = request.param
    net = M("pyramid_mobile_" + arch).build_sfd_mobile('test', 640, 2)
    sd = synth.make_state_dict(arch, seed=0)
    net.load_state_dict(sd)
    yield arch, net, sd
    net.close()


@pytest.mark.parametrize("H,W,seed", [(64, 64, 47), (136, 200, 48)])
def test_try12_stages_vs_oracle(try12, synth, H, W, seed):
    """Mobilenetv1/v2 blocks: 7x7/s2 depthwise stem on the raw image channels, 5x5/s2 and dilated 3x3 depthwise,
    grouped (4 / 2) 1x1 lateral layers run as block-diagonal dense convs -- stage tensors and detections vs oracle."""
    arch, net, sd = try12
    frame = synth.make_frames(1, H, W, seed=seed)[0]
    x = opb.preprocess(frame)
    net.priorbox = M("layers").PriorBoxLayer(W, H); net.firstTime = True
    net.detect = M("layers").Detect(2, 0, 750, 0.02, 0.35)
    y = net(x).numpy()
    o = opb.try12_forward(sd, x, want=TRY12_STAGES, variant=int(arch[3]))
    for st in TRY12_STAGES:
        got = net.get_tensor(st)
        assert got.shape == o[st].shape, (st, got.shape, o[st].shape)
        assert rel_rms(got, o[st]) < STAGE_RTOL, (st, rel_rms(got, o[st]))
    pri = opp.build_priors(opp.PriorBoxLayer(W, H), H, W)
    assert np.array_equal(net.priors.numpy(), pri)
    exp = opp.Detect(2, 0, 750, 0.02, 0.35)(o["loc"], o["conf"], pri)
    n = int((exp[0, 1, :, 0] > 0).sum())
    d_iou, d_sc = match_detections(y[0, 1], exp[0, 1], n)
    assert n > 5 and d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


@pytest.mark.parametrize("size", ["64x64", "136x200", "480x640"])
def test_try12_vs_reference_fixture(try12, synth, size):
    """Final detections vs what reference pyramid_mobile_try1.py / _try2.py produced here (tests/golden/nets12.npz)."""
    arch, net, _ = try12
    d, meta = load_npz("nets12")
    key = "%s_%s" % (arch, size)
    m = meta[key]
    H, W = m["H"], m["W"]
    frame = synth.make_frames(1, H, W, seed=m["frame_seed"])[0]
    net.priorbox = M("layers").PriorBoxLayer(W, H); net.firstTime = True
    net.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    y = net(frame).numpy()
    exp = d[key + "_out"]
    d_iou, d_sc = match_detections(y[0, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]),
                                   m["n_out"])
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


# ------------------------------------------------------------------ round-2 fixtures: C1 640x640 and C3 batch 8
def test_res50_640x640_vs_reference_fixture(res50, synth):
    """Config 1 of BASELINE.json (the size My_test.py feeds, reference My_test.py:31-36)."""
    d, meta = load_npz("nets_r2")
    key = "res50_640x640_b1"
    m = meta[key]
    frame = synth.make_frames(1, 640, 640, seed=m["frame_seeds"][0])[0]
    res50.priorbox = M("layers").PriorBoxLayer(640, 640); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    y = res50(frame).numpy()
    exp = d[key + "_out0"]
    d_iou, d_sc = match_detections(y[0, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]),
                                   m["n_out"][0])
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL
    sel = d[key + "_sel"]
    np.testing.assert_allclose(res50.get_tensor("loc")[0, sel], d[key + "_loc_s0"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(res50.get_tensor("conf")[0, sel], d[key + "_conf_s0"], atol=SCORE_ATOL, rtol=0)


def test_res50_native_1080p_vs_reference_fixture(res50, synth):
    """BASELINE config 4's source size run natively (no resize): Res50 at 1080 x 1920 against the reference's own forward of
    the same seeded frame (tests/golden/nets_r3.npz: P = 172 845 priors, 602 candidates, 390 detections), with the
    committed plan bench.py uses for this shape."""
    import os
    d, meta = load_npz("nets_r3")
    key = "res50_1080x1920_b1"
    m = meta[key]
    frame = synth.make_frames(1, 1080, 1920, seed=m["frame_seeds"][0])[0]
    res50.priorbox = M("layers").PriorBoxLayer(1920, 1080); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    plan = res50.tuned_plan_text(1080, 1920, 1)
    assert plan is not None
    res50.import_plan(plan)
    y = res50(frame).numpy()
    assert int(res50.get_tensor("loc").shape[1]) == m["P"]
    exp = d[key + "_out0"]
    d_iou, d_sc = match_detections(y[0, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]),
                                   m["n_out"][0])
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL, (d_iou, d_sc)
    sel = d[key + "_sel"]
    np.testing.assert_allclose(res50.get_tensor("loc")[0, sel], d[key + "_loc_s0"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(res50.get_tensor("conf")[0, sel], d[key + "_conf_s0"], atol=SCORE_ATOL, rtol=0)
    assert np.array_equal(res50(frame).numpy(), y)              # graph replay: same bits


def test_plan_with_in_kernel_split_k_combine_gives_the_same_bits(res50, synth):
    """The optional sixth plan column (fdt_model_export_plan / _import_plan): split-K layers combined inside the conv kernel by
    the last workgroup to arrive (csrc/conv.h) instead of by splitk_reduce_kernel.  Same slabs, same order -> the whole
    forward (640 x 480, every split layer the kernels support switched over) is bit-identical, eagerly and from the graph."""
    frame = synth.make_frames(1, 480, 640, seed=77)[0]
    res50.priorbox = M("layers").PriorBoxLayer(640, 480); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.3)
    plan = res50.tuned_plan_text(480, 640, 1)
    assert plan is not None
    res50.import_plan(plan)
    y = res50(frame).numpy()
    loc, conf = res50.get_tensor("loc").copy(), res50.get_tensor("conf").copy()
    lines, n_split = [], 0
    for ln in plan.splitlines():
        f = ln.split()
        if len(f) == 5 and f[0] != "shape" and int(f[3]) > 1:
            ln += " 1"
            n_split += 1
        lines.append(ln)
    assert n_split >= 20
    res50.import_plan("\n".join(lines) + "\n")
    for _ in range(3):                                             # eager pass, graph capture, replay
        assert np.array_equal(res50(frame).numpy(), y)
    assert np.array_equal(res50.get_tensor("loc"), loc) and np.array_equal(res50.get_tensor("conf"), conf)
    combined = [ln for ln in res50.export_plan().splitlines() if len(ln.split()) == 6]
    assert len(combined) >= 10, len(combined)                      # (the 8-channel VALU heads and Wout % 4 != 0 maps stay two-pass)
    res50.import_plan(plan)


@pytest.mark.parametrize("arch", ["res50", "try3"])
def test_lazy_reduce_passes_read_nothing_stale(arch, res50, try3, synth):
    """csrc/model.hip plan_reduces: split-K layers leave their slabs in the workspace and their reduce passes run, grouped, in
    front of the first op that needs one of them.  A missed dependency would make that op read what the PREVIOUS forward left
    in the tensor -- invisible when the same frame is run twice.  So: frame A then frame B on one handle against frame B on a
    clone whose activations have never been written; bit for bit, detections and both head tensors."""
    net = res50 if arch == "res50" else try3
    H, W = 480, 640
    PB = M("layers").PriorBoxLayer
    net.priorbox = PB(W, H) if arch == "res50" else PB(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
    net.firstTime = True
    net.detect = M("layers").Detect(2, 0, 750, 0.05, 0.3)
    plan = net.tuned_plan_text(H, W, 1)
    if plan is not None:
        net.import_plan(plan)
    a = synth.make_frames(1, H, W, seed=5)[0]
    b = synth.make_frames(1, H, W, seed=6)[0]
    net(a)
    y1 = net(b).numpy()
    loc1, conf1 = net.get_tensor("loc").copy(), net.get_tensor("conf").copy()
    fresh = net.clone()
    y2 = fresh(b).numpy()
    assert np.array_equal(y1, y2)
    assert np.array_equal(fresh.get_tensor("loc"), loc1) and np.array_equal(fresh.get_tensor("conf"), conf1)
    assert not np.array_equal(net(a).numpy(), y1)                  # (the two frames do differ)
    lazy = [ln for ln in net.export_plan().splitlines() if len(ln.split()) >= 5 and ln.split()[0] != "shape" and int(ln.split()[3]) > 1]
    assert len(lazy) >= 10                                         # there are split layers to be lazy about


def test_try3_1024_batch8_vs_reference_fixture(try3, synth):
    """Config 3 of BASELINE.json: ONE batched forward of eight 1024x1024 frames (the depthwise / batched conv plan that
    bench.py --arch try3 --batch 8 times) against the reference's own batch-8 forward, per image."""
    d, meta = load_npz("nets_r2")
    key = "try3_1024x1024_b8"
    m = meta[key]
    frames = np.stack([synth.make_frames(1, 1024, 1024, seed=s)[0] for s in m["frame_seeds"]])
    PB = M("layers").PriorBoxLayer
    try3.priorbox = PB(1024, 1024, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256)); try3.firstTime = True
    try3.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    import os
    plan = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "face-detection-and-tracking_amd",
                        "tuned", "try3_1024x1024_b8.plan")
    try3.import_plan(open(plan).read())           # the committed plan bench.py uses for this config
    y = try3(frames).numpy()
    assert y.shape == (8, 2, 750, 5) and not y[:, 0].any()
    sel = d[key + "_sel"]
    loc, conf = try3.get_tensor("loc"), try3.get_tensor("conf")
    for b in range(8):
        exp = d["%s_out%d" % (key, b)]
        d_iou, d_sc = match_detections(y[b, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]),
                                       m["n_out"][b])
        assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL, (b, d_iou, d_sc)
        np.testing.assert_allclose(loc[b, sel], d["%s_loc_s%d" % (key, b)], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(conf[b, sel], d["%s_conf_s%d" % (key, b)], atol=SCORE_ATOL, rtol=0)
    # and the second (graph-replayed) batched forward returns the same bits
    assert np.array_equal(try3(frames).numpy(), y)


def test_try3_1024_heads_on_the_vector_alu_kernel_without_a_plan(try3_sd, synth):
    """Without a tuned plan the builder's own rule puts the 8-channel loc/conf heads of the large maps on the vector-ALU
    kernel (csrc/conv_n8.h); frame 0 of the batch-8 fixture, run alone, still matches the reference's numbers."""
    d, meta = load_npz("nets_r2")
    key = "try3_1024x1024_b8"
    m = meta[key]
    frame = synth.make_frames(1, 1024, 1024, seed=m["frame_seeds"][0])[0]
    net = M("pyramid_mb2_try3").build_sfd_mobile('test', 640, 2)
    net.load_state_dict(try3_sd)
    net.priorbox = M("layers").PriorBoxLayer(1024, 1024, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
    net.detect = M("layers").Detect(2, 0, 750, m["conf_t"], m["nms_t"])
    y = net(frame).numpy()
    names = [n for n, _, _ in (net.profile(True), net(frame), net.profile_read())[2]]
    assert any(n.startswith("face_loc.0#k13t31s") for n in names), [n for n in names if n.startswith("face_loc")]
    exp = d[key + "_out0"]
    d_iou, d_sc = match_detections(y[0, 1], np.vstack([exp, np.zeros((750 - exp.shape[0], 5), np.float32)]), m["n_out"][0])
    assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL, (d_iou, d_sc)
    sel = d[key + "_sel"]
    np.testing.assert_allclose(net.get_tensor("loc")[0, sel], d[key + "_loc_s0"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(net.get_tensor("conf")[0, sel], d[key + "_conf_s0"], atol=SCORE_ATOL, rtol=0)
    net.close()


@pytest.mark.parametrize("H,W,B", [(512, 512, 2), (522, 520, 1)])
def test_try3_streaming_stem_and_depthwise_project_kernels(try3_sd, synth, monkeypatch, H, W, B):
    """csrc/stream_ir.hip: features.0 (3x3 / 2 stem, pyramid_mb2_try3.py:162) on the raw uint8 frame and features.1 (the t = 1
    InvertedResidual, :84-94) as depthwise + project in one pass, both on the vector ALU, against the MFMA / two-launch forms
    (FDT_STREAM_IR=0 at create time) and against the oracle's stages; an odd row count exercises the two-row strips' tail."""
    frames = synth.make_frames(B, H, W, seed=31)
    PB = M("layers").PriorBoxLayer
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FDT_STREAM_IR", mode)
        net = M("pyramid_mb2_try3").build_sfd_mobile('test', 640, 2)
        net.load_state_dict(try3_sd)
        net.priorbox = PB(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
        net.detect = M("layers").Detect(2, 0, 750, 0.02, 0.35)
        y = net(frames).numpy()
        names = [n for n, _, _ in (net.profile(True), net(frames), net.profile_read())[2]]
        net.profile(False)
        res[mode] = (y, net.get_tensor("stem"), net.get_tensor("features.1.conv.3"), names)
        net.close()
    monkeypatch.delenv("FDT_STREAM_IR")
    assert any(n == "features.0.0.u8_stream" for n in res["1"][3]) and any(n == "features.1.dw_project" for n in res["1"][3])
    assert not any(n.endswith((".u8_stream", ".dw_project")) for n in res["0"][3])
    assert len(res["0"][3]) - len(res["1"][3]) == 1                      # depthwise + project: one launch instead of two
    o = opb.try3_forward(try3_sd, opb.preprocess(frames[0]), want=["stem", "c2"])
    for mode in ("1", "0"):
        assert rel_rms(res[mode][1][0], o["stem"][0]) < STAGE_RTOL, mode
    assert rel_rms(res["1"][1], res["0"][1]) < 2e-6 and rel_rms(res["1"][2], res["0"][2]) < 2e-6
    for b in range(B):
        n = int((res["0"][0][b, 1, :, 0] > 0).sum())
        d_iou, d_sc = match_detections(res["1"][0][b, 1], res["0"][0][b, 1], n)
        assert n > 5 and d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


def test_try3_with_fused_inverted_residual_heads(try3_sd, synth, monkeypatch):
    """The fused expand + depthwise kernel (csrc/fused_ir.hip) normally takes over only on maps of 256^2 and more; forced on
    for every block that fits (FDT_FUSE_IR=1), the try3 stages at a small odd size still match the oracle, and the forced-off
    graph gives the same detections."""
    H, W = 136, 200
    frame = synth.make_frames(1, H, W, seed=8)[0]
    x = opb.preprocess(frame)
    PB = M("layers").PriorBoxLayer
    outs = {}
    for mode in ("1", "0", "2"):                        # expand + depthwise fused / separate launches / whole blocks in one launch
        monkeypatch.setenv("FDT_FUSE_IR", mode)
        net = M("pyramid_mb2_try3").build_sfd_mobile('test', 640, 2)
        net.load_state_dict(try3_sd)
        net.priorbox = PB(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
        net.detect = M("layers").Detect(2, 0, 750, 0.02, 0.35)
        outs[mode] = net(x).numpy()
        if mode == "1":
            o = opb.try3_forward(try3_sd, x, want=TRY3_STAGES)
            for st in TRY3_STAGES:
                got = net.get_tensor(st)
                assert got.shape == o[st].shape and rel_rms(got, o[st]) < STAGE_RTOL, (st, rel_rms(got, o[st]))
            names = [n for n, _, _ in (net.profile(True), net(x), net.profile_read())[2]]
            # every block whose staged patch leaves room for two workgroups per CU (features.2 .. features.6)
            assert sum(n.endswith(".expand_dw") for n in names) == 5
            net.profile(False)
        if mode == "2":
            o = opb.try3_forward(try3_sd, x, want=TRY3_STAGES)
            for st in TRY3_STAGES:
                got = net.get_tensor(st)
                assert got.shape == o[st].shape and rel_rms(got, o[st]) < STAGE_RTOL, (st, rel_rms(got, o[st]))
            names = [n for n, _, _ in (net.profile(True), net(x), net.profile_read())[2]]
            assert sum(n.endswith(".expand_dw_project") for n in names) >= 3, names     # the blocks with <= 32 output channels
            net.profile(False)
        net.close()
    n = int((outs["0"][0, 1, :, 0] > 0).sum())
    d_iou, d_sc = match_detections(outs["1"][0, 1], outs["0"][0, 1], n)
    assert n > 5 and d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL


@pytest.mark.parametrize("H,W,B", [(136, 200, 1), (256, 256, 2), (480, 640, 1), (100, 202, 1), (101, 36, 3), (67, 132, 1)])
def test_raw_frame_stem_on_the_bf16_pipe(res50_sd, synth, H, W, B):
    """conv_stem_u8b.h (class 24 = CONV_7x7_S2_U8B, the default raw-frame stem where W % 4 == 0): (float)u8 - mean is an integer of
    magnitude <= 255, exact in ONE bf16; the weights are split into three bf16 planes; three exact plane products per k-step, f32
    accumulate.  Against the ingest kernel + planar f32-MFMA stem (class 5): the stem tensor to f32 rounding (1e-6 of its
    maximum; every product is exact in both forms, only the order of the f32 additions differs), the same detections; the
    "input" tensor formed lazily is the oracle's; image borders (zero padding of the NORMALISED image), tiles hanging over the
    map, two images, device frames, the resize in front.  W % 4 != 0: the f32 raw-frame class 18 runs instead."""
    L = M("_lib")
    nets = []
    for fuse in (1, 0):
        n = M("pyramid").build_sfd('test', 640, 2)
        n.load_state_dict(res50_sd)
        n.priorbox = M("layers").PriorBoxLayer(W, H)
        n.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
        L.check(L.lib().fdt_model_fuse_ingest(n._h, fuse))
        nets.append(n)
    frames = synth.make_frames(B, H, W, seed=3 * H + B)
    frames[0, :3, :5] = 255                                   # extremes at an image corner
    frames[0, -2:, -7:] = 0
    outs = []
    for n in nets:
        y = n(frames if B > 1 else frames[0]).numpy()
        outs.append((y, n.get_tensor("stem"), n.get_tensor("input")))
    want = "#k24t" if W % 4 == 0 else "#k18t"
    for n, k in zip(nets, (want, "#k5t")):
        n.profile(True)
        n(frames if B > 1 else frames[0])
        assert n.profile_read()[0][0].startswith("conv1" + k), n.profile_read()[0][0]
        n.profile(False)
    a, b = outs[0][1], outs[1][1]
    assert a.shape == b.shape and np.isfinite(a).all()
    assert float(np.abs(a.astype(np.float64) - b).max()) <= 1e-6 * float(np.abs(b).max()), float(np.abs(a - b).max() / np.abs(b).max())
    assert np.array_equal(outs[0][2], np.concatenate([opb.preprocess(f) for f in frames]))
    for bi in range(B):
        nn = int((outs[1][0][bi, 1, :, 0] > 0).sum())
        d_iou, d_sc = match_detections(outs[0][0][bi, 1], outs[1][0][bi, 1], nn)
        assert d_iou <= BOX_IOU_TOL and d_sc <= SCORE_ATOL, (bi, nn, d_iou, d_sc)
    fd = torch.from_numpy(frames).cuda()
    ydev = nets[0](fd if B > 1 else fd[0]).numpy()
    assert np.array_equal(ydev, outs[0][0])                  # device frames: the same kernel on the caller's buffer
    # ... also when the caller's device buffer starts at an odd address (the kernel loads 12-byte groups with dword instructions)
    raw = torch.zeros(frames.size + 8, dtype=torch.uint8, device="cuda")
    odd = raw[1:1 + frames.size].view(frames.shape)
    odd.copy_(fd)
    assert odd.data_ptr() % 4 == 1
    yodd = nets[0](odd if B > 1 else odd[0]).numpy()
    assert np.array_equal(yodd, outs[0][0])
    big = synth.make_frames(B, 2 * H + 6, 2 * W + 10, seed=5)
    yr = [n.forward_resized(big, (W, H)).numpy() for n in nets]
    sr = [n.get_tensor("stem") for n in nets]
    assert float(np.abs(sr[0].astype(np.float64) - sr[1]).max()) <= 1e-6 * float(np.abs(sr[1]).max())
    assert np.array_equal(nets[0].get_tensor("input"), nets[1].get_tensor("input"))
    for n in nets:
        n.close()


@pytest.mark.parametrize("H,W,B", [(136, 200, 1), (256, 256, 2), (480, 640, 1)])
def test_fused_ingest_gives_the_bits_of_the_two_launch_form(res50_sd, synth, H, W, B, monkeypatch):
    """conv_stem_u8.h: the 7x7 stem reading the raw uint8 frame (mean subtraction in its staging) against preprocess_kernel +
    the planar stem conv: same accumulation order -> the stem tensor, the Detect record and the (lazily formed) "input" tensor
    are bit-identical; host frames, device frames, and the device-side resize in front (tiles hanging over the map, the
    zero padding of the converted domain at every image border, two images).  FDT_STEM_B3=0: the f32-MFMA raw-frame class (the
    default where the width allows is the bf16-pipe class 24, another summation order: test_raw_frame_stem_on_the_bf16_pipe)."""
    import ctypes
    monkeypatch.setenv("FDT_STEM_B3", "0")
    L = M("_lib")
    nets = []
    for fuse in (1, 0):
        n = M("pyramid").build_sfd('test', 640, 2)
        n.load_state_dict(res50_sd)
        n.priorbox = M("layers").PriorBoxLayer(W, H)
        n.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
        L.check(L.lib().fdt_model_fuse_ingest(n._h, fuse))
        nets.append(n)
    frames = synth.make_frames(B, H, W, seed=3 * H + B)
    outs = []
    for n in nets:
        y = n(frames if B > 1 else frames[0]).numpy()
        outs.append((y, n.get_tensor("stem"), n.get_tensor("input")))
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][2], outs[1][2])
    assert np.array_equal(outs[0][2], np.concatenate([opb.preprocess(f) for f in frames]))
    # the fused handle really ran the raw-frame kernel (kernel class 18 = CONV_7x7_S2_U8), the other one the planar class 5
    for n, k in zip(nets, ("#k18t", "#k5t")):
        n.profile(True)
        n(frames if B > 1 else frames[0])
        assert n.profile_read()[0][0].startswith("conv1" + k), n.profile_read()[0][0]
        n.profile(False)
    # device frames (zero copy) and the resize ingest
    fd = torch.from_numpy(frames).cuda()
    ydev = [n(fd if B > 1 else fd[0]).numpy() for n in nets]
    assert np.array_equal(ydev[0], ydev[1]) and np.array_equal(ydev[0], outs[0][0])
    big = synth.make_frames(B, 2 * H + 6, 2 * W + 10, seed=5)
    yr = [n.forward_resized(big, (W, H)).numpy() for n in nets]
    assert np.array_equal(yr[0], yr[1])
    assert np.array_equal(nets[0].get_tensor("input"), nets[1].get_tensor("input"))
    for n in nets:
        n.close()
