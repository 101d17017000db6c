"""GPU parity of the FaceBoxes path (config 5) with the REAL weights that ship in the reference tree
(tests/golden/faceboxes_weights.npz) against fixtures produced by the reference and the oracle."""
import hashlib
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz
from oracle import facebox as ofb
from oracle import postproc as opp

pytestmark = pytest.mark.gpu


def M(name):
    return importlib.import_module("face-detection-and-tracking_amd." + name)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def fb_sd():
    z = np.load(os.path.join(GOLDEN, "faceboxes_weights.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def net(fb_sd):
    n = M("FACEBOX.networks").FaceBox()
    n.load_state_dict(fb_sd)
    n.cuda(); n.eval()
    yield n
    n.close()


def rel_rms(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


def test_anchors_bit_exact():
    d, _ = load_npz("facebox")
    enc = M("FACEBOX.encoderl").DataEncoder()
    assert sha(enc.default_boxes_np) == bytes(d["anchors_sha"]).decode()


def test_decode_np_vs_reference_fixture():
    d, meta = load_npz("facebox")
    enc = M("FACEBOX.encoderl").DataEncoder()
    boxes, probs = enc.decode_np(d["dec_loc"], d["dec_conf"])
    assert len(probs) == meta["dec"]["n"]
    assert np.array_equal(probs, d["dec_probs"])                   # keep set + order: exact
    assert np.allclose(boxes, d["dec_boxes"], rtol=3e-7, atol=3e-7)


def test_forward_stages_vs_oracle(net, fb_sd, synth):
    d, _ = load_npz("facebox")
    fr = d["img1_frame"]
    x = torch.from_numpy(np.ascontiguousarray(fr.transpose(2, 0, 1))).float().div(255)[None]
    loc, conf = net(x)
    want = ["crelu_pool1", "crelu_pool2", "inception1", "inception2", "hs0", "hs1", "hs2"]
    o = ofb.forward(fb_sd, x, want=want)
    for st in want:
        g = net.get_tensor(st)
        assert g.shape == o[st].shape and rel_rms(g, o[st]) < 2e-5, (st, rel_rms(g, o[st]))
    assert rel_rms(loc.numpy(), o["loc"]) < 2e-5 and rel_rms(conf.numpy(), o["conf"]) < 2e-5
    # u8 frame in (the raw-frame stem on the bf16 pipe, conv_stem_u8b.h: bytes x three weight planes, / 255 behind the sum) against
    # the oracle too, and against the f32 tensor in to f32 rounding
    loc2, conf2 = net(fr[None])
    assert rel_rms(loc2.numpy(), o["loc"]) < 2e-5 and rel_rms(conf2.numpy(), o["conf"]) < 2e-5
    assert rel_rms(loc2.numpy(), loc.numpy()) < 2e-6 and rel_rms(conf2.numpy(), conf.numpy()) < 2e-6


@pytest.mark.parametrize("i", [0, 1, 2])
def test_detect_vs_reference_fixture(net, synth, i):
    d, meta = load_npz("facebox")
    key = "img%d" % i
    fr = d[key + "_frame"] if i < 2 else synth.make_frames(1, 1024, 1024, seed=55)[0]
    mt = M("FACEBOX.My_test_facebox")
    mt.net = net
    boxes, probs = mt.detect(fr)
    assert len(probs) == meta[key]["n"]
    if len(probs):
        iou = opp.calculate_iou(d[key + "_boxes"].astype(np.float64), boxes.astype(np.float64)).max(1)
        assert (1 - iou).max() <= 1e-3 and np.abs(probs - d[key + "_probs"]).max() <= 1e-4
    sel = d[key + "_sel"]
    loc, conf = net(fr[None])
    np.testing.assert_allclose(loc.numpy()[0, sel], d[key + "_loc_s"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(conf.numpy()[0, sel], d[key + "_conf_s"], rtol=1e-4, atol=2e-4)


def test_batch16_detect(net, synth):
    """Config 5 shape: a batch of 16 frames; image b of the batch equals image b alone."""
    d, _ = load_npz("facebox")
    frames = np.stack([d["img0_frame"], d["img1_frame"]] * 8)
    res = net.detect_frames(frames)
    one0 = net.detect_frames(frames[:1])[0]
    one1 = net.detect_frames(frames[1:2])[0]
    for b, (bx, pr) in enumerate(res):
        ref = one0 if b % 2 == 0 else one1
        assert len(pr) == len(ref[1]) and np.allclose(pr, ref[1], atol=1e-5) and np.allclose(bx, ref[0], atol=1e-5)
    with pytest.raises(ValueError):
        net(np.zeros((1, 512, 512, 3), np.uint8))


@pytest.mark.parametrize("i", range(6))
def test_detect_six_more_reference_images(net, i):
    """Six of the reference's sample JPEGs with 3..12 faces each (tests/golden/facebox_r2.npz: the reference's own
    FaceBox + decode_np output, FACEBOX/My_test_facebox.py:12-36 after the resize): same faces, boxes within 1e-3 IoU,
    probabilities within 1e-4."""
    d, meta = load_npz("facebox_r2")
    key = "img%d" % i
    mt = M("FACEBOX.My_test_facebox")
    mt.net = net
    boxes, probs = mt.detect(d[key + "_frame"])
    assert len(probs) == meta[key]["n"] >= 3
    iou = opp.calculate_iou(d[key + "_boxes"].astype(np.float64), boxes.astype(np.float64))
    j = iou.argmax(1)
    assert len(set(j.tolist())) == len(probs)
    assert (1 - iou.max(1)).max() <= 1e-3 and np.abs(probs[j] - d[key + "_probs"]).max() <= 1e-4
    # keep order = descending probability like nms_np; img0 holds an exact f32 tie whose order is numpy's unstable
    # argsort (FACEBOX/encoderl.py:234) -- unpinned upstream, so rows may swap only inside a tie
    if meta[key]["min_gap"] > 1e-4:
        assert np.array_equal(j, np.arange(len(probs)))
    else:
        assert np.array_equal(np.sort(probs)[::-1], probs)


def test_batch16_of_distinct_images_vs_reference(net):
    """Config 5 shape with 16 frames cycling over the six multi-face images: image b of the batch == reference."""
    d, meta = load_npz("facebox_r2")
    frames = np.stack([d["img%d_frame" % (b % 6)] for b in range(16)])
    res = net.detect_frames(frames)
    for b, (bx, pr) in enumerate(res):
        key = "img%d" % (b % 6)
        assert len(pr) == meta[key]["n"]
        assert np.abs(pr - d[key + "_probs"]).max() <= 1e-4
        iou = opp.calculate_iou(d[key + "_boxes"].astype(np.float64), bx.astype(np.float64)).max(1)
        assert (1 - iou).max() <= 1e-3


def test_detect_from_4k_source_matches_oracle(net, fb_sd):
    """BASELINE config 5 as stated: a 2160x3840 u8 source is resized on the GPU (line :13 of the reference's detect(),
    cv2.resize -> restated in oracle/ingest.py, cv2 parity unpinned), /255, FaceBox, decode_np + nms_np.  The resized
    input tensor is bit-exact against the oracle's resize; faces match the oracle end to end."""
    from oracle import ingest as oin
    d, meta = load_npz("facebox_r2")
    SH, SW = 2160, 3840
    yi = (np.arange(SH) * 1024) // SH
    xi = (np.arange(SW) * 1024) // SW
    for i in (0, 2):
        src = np.ascontiguousarray(d["img%d_frame" % i][yi][:, xi])
        (boxes, probs), = net.detect_frames(src[None])
        small = oin.resize_linear_u8(src, 1024, 1024)
        x = net.get_tensor("input")
        assert np.array_equal(x[0], small.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
        rb, rp = ofb.detect(fb_sd, small)
        assert len(probs) == len(rp) >= 3
        iou = opp.calculate_iou(rb.astype(np.float64), boxes.astype(np.float64))
        assert (1 - iou.max(1)).max() <= 1e-3 and np.abs(probs[iou.argmax(1)] - rp).max() <= 1e-4
        # My_test_facebox.detect() takes the raw frame like the reference does
        mt = M("FACEBOX.My_test_facebox")
        mt.net = net
        b2, p2 = mt.detect(src)
        assert np.array_equal(p2, probs) and np.array_equal(b2, boxes)


def test_traffic_and_ingest_profile(net):
    d, _ = load_npz("facebox")
    net.detect_frames(d["img0_frame"][None])
    act, wts, per = net.traffic()
    assert 30e6 < act < 120e6 and 3.9e6 < wts < 4.3e6          # 1 008 810 parameters, a few tens of MB of activations
    net.profile(True)
    net.detect_frames(d["img0_frame"][None]); net.detect_frames(d["img0_frame"][None])
    prof = net.profile_read()
    net.profile(False)
    assert [p[0] for p in prof[-2:]] == ["detect", "ingest"] and len(prof) == len(per)
    assert all(ms >= 0 for _, ms, _ in prof) and prof[-1][1] > 0


def test_three_batches_in_flight_on_cloned_handles(net):
    """What bench.py --arch facebox times: several batches in flight, each on its own fdt_model_clone() handle and stream,
    4K sources resized on the GPU -- every slot returns the faces of the single-handle synchronous path, bit for bit."""
    import ctypes
    L = M("_lib")
    lib = L.lib()
    d, meta = load_npz("facebox_r2")
    SH, SW, B = 540, 960, 4
    yi = (np.arange(SH) * 1024) // SH
    xi = (np.arange(SW) * 1024) // SW
    frames = np.stack([np.ascontiguousarray(d["img%d_frame" % (b % 6)][yi][:, xi]) for b in range(B)])
    want = net.detect_frames(frames)
    dev = torch.device("cuda", 0)
    fd = torch.from_numpy(frames).to(dev)
    clones = [net.clone(), net.clone()]
    nets = [net] + clones
    streams = [torch.cuda.Stream(device=dev) for _ in nets]
    counts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in nets]
    for rep in range(3):
        for k, n in enumerate(nets):
            L.check(lib.fdt_model_detect_facebox_resized(n._h, ctypes.c_void_p(fd.data_ptr()), 1, B, SH, SW, 0.35, 0.5, None,
                                                         None, ctypes.c_void_p(counts[k].data_ptr()),
                                                         ctypes.c_void_p(streams[k].cuda_stream)))
    torch.cuda.synchronize()
    for k, n in enumerate(nets):
        c = counts[k].cpu().numpy()
        assert c.tolist() == [len(p) for _, p in want]
        probs = n.get_tensor("fb_probs").reshape(B, -1)
        boxes = n.get_tensor("fb_boxes").reshape(B, -1, 4)
        for b in range(B):
            assert np.array_equal(probs[b, :c[b]], want[b][1]) and np.array_equal(boxes[b, :c[b]], want[b][0])
    for c_ in clones:
        c_.close()


def test_fused_ingest_facebox_bits(fb_sd, monkeypatch):
    """FaceBoxes: conv1 (7x7 / 4, 3 -> 24) on the raw uint8 frame with the /255 in its staging (conv_stem_u8.h, class 19) == the
    ingest kernel + planar f32-MFMA conv1 (class 20; FDT_STEM_B3=0: the default planar stem is the split-bf16 class 22, which has
    another summation order), bit for bit, also behind the 4K resize."""
    monkeypatch.setenv("FDT_STEM_B3", "0")
    L = M("_lib")
    d, _ = load_npz("facebox_r2")
    frames = np.stack([d["img%d_frame" % i] for i in (0, 1)])
    res = []
    for fuse in (2, 0):                                  # 2: the stride-4 stem too (not the default: slower)
        n = M("FACEBOX.networks").FaceBox()
        n.load_state_dict(fb_sd)
        L.check(L.lib().fdt_model_fuse_ingest(n._h, fuse))
        r = n.detect_frames(frames)
        SH, SW = 1080, 1920
        yi = (np.arange(SH) * 1024) // SH
        xi = (np.arange(SW) * 1024) // SW
        r2 = n.detect_frames(np.ascontiguousarray(frames[:, yi][:, :, xi]))
        res.append((r, r2, n.get_tensor("input").copy()))
        n.close()
    for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and len(a[1]) >= 3
    assert np.array_equal(res[0][2], res[1][2])


def test_raw_frame_stem_on_the_bf16_pipe_facebox(fb_sd):
    """conv_stem_u8b.h, class 25 (the default on uint8 frames): conv1 reads the bytes -- exact in one bf16 --, multiplies them with
    the three bf16 planes of the weights and divides the SUM by 255 (My_test_facebox.py:14-15 divides the pixels: a scalar that
    commutes with the convolution up to f32 rounding).  Against the ingest kernel + planar f32 conv1 (FDT_STEM_B3=0 handle,
    class 20): conv1's output to 2e-6 of its maximum, the same faces (boxes to 1e-3 px, scores to 1e-5), also behind the resize."""
    import os
    L = M("_lib")
    d, _ = load_npz("facebox_r2")
    frames = np.stack([d["img%d_frame" % i] for i in (0, 1)])
    res = []
    for b3 in ("1", "0"):
        os.environ["FDT_STEM_B3"] = b3
        try:
            n = M("FACEBOX.networks").FaceBox()
        finally:
            del os.environ["FDT_STEM_B3"]
        n.load_state_dict(fb_sd)
        if b3 == "0":
            L.check(L.lib().fdt_model_fuse_ingest(n._h, 0))
        r = n.detect_frames(frames)
        c1 = n.get_tensor("conv1").copy()
        n.profile(True)
        n.detect_frames(frames)
        first = n.profile_read()
        n.profile(False)
        SH, SW = 1080, 1920
        yi = (np.arange(SH) * 1024) // SH
        xi = (np.arange(SW) * 1024) // SW
        r2 = n.detect_frames(np.ascontiguousarray(frames[:, yi][:, :, xi]))
        res.append((r, r2, c1, [x[0] for x in first]))
        n.close()
    assert any(x.startswith("conv1#k25t") for x in res[0][3]), res[0][3][:3]
    assert any(x.startswith("conv1#k20t") for x in res[1][3]), res[1][3][:3]
    a, b = res[0][2], res[1][2]
    assert a.shape == b.shape and float(np.abs(a.astype(np.float64) - b).max()) <= 2e-6 * float(np.abs(b).max())
    for x, y in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert len(x[1]) == len(y[1]) and len(x[1]) >= 3
        assert np.abs(x[0] - y[0]).max() <= 1e-3 and np.abs(x[1] - y[1]).max() <= 1e-5


@pytest.mark.parametrize("B", [1, 16])
def test_inception_branches_in_one_launch_equal_the_eight_launch_form(fb_sd, monkeypatch, B):
    """FACEBOX/networks.py:43-57 with the branches that share an input as single launches (conv1 | conv3 | conv5 on x: one
    1x1 launch with two destinations; conv4 | conv6: one block-diagonal 3x3 launch) against the one-launch-per-layer form
    (FDT_FB_FUSE=0 at create time) and the 1x1-only form (1): the same sums per output channel -- bit-equal wherever neither
    form splits a reduction, else to f32 rounding --, the same faces, and fewer launches."""
    d, _ = load_npz("facebox_r2")
    frames = np.stack([d["img%d_frame" % (i % 6)] for i in range(B)])
    out = {}
    for fuse in ("0", "1", "2"):
        monkeypatch.setenv("FDT_FB_FUSE", fuse)
        n = M("FACEBOX.networks").FaceBox()
        n.load_state_dict(fb_sd)
        loc, conf = n(frames)
        res = n.detect_frames(frames)
        n.profile(True)
        n.detect_frames(frames)
        launches = len(n.profile_read())
        n.profile(False)
        stages = {st: n.get_tensor(st) for st in ("inception1", "inception2", "hs0", "hs1", "hs2")}
        out[fuse] = (loc.numpy(), conf.numpy(), res, launches, stages)
        n.close()
    monkeypatch.delenv("FDT_FB_FUSE")
    assert out["0"][3] - out["1"][3] == 6 and out["1"][3] - out["2"][3] == 3, [out[k][3] for k in "012"]
    assert out["2"][3] <= 33                                    # ops + "detect" + "ingest" entries
    for fuse in ("1", "2"):
        for st, g in out[fuse][4].items():
            assert g.shape == out["0"][4][st].shape and rel_rms(g, out["0"][4][st]) < 2e-6, (fuse, st)
        assert rel_rms(out[fuse][0], out["0"][0]) < 2e-6 and rel_rms(out[fuse][1], out["0"][1]) < 2e-6
        for (bx, pr), (bx0, pr0) in zip(out[fuse][2], out["0"][2]):
            assert len(pr) == len(pr0) >= 3 and np.allclose(pr, pr0, atol=1e-6) and np.allclose(bx, bx0, atol=1e-6)


def test_config5_as_baseline_words_it_4k_sources_and_batch_16_in_one_call(net, fb_sd):
    """BASELINE.json configs[4] in ONE call: sixteen 2160x3840 uint8 sources (the six multi-face images, pixel-replicated like
    bench.py builds its frames) resized on the GPU and detected as one batch.  Equal images of the batch give equal bits, every image
    matches its single-image call to f32 rounding, the resized input of every image is bit-exact against the oracle's resize, and
    the faces match the oracle end to end."""
    from oracle import ingest as oin
    d, meta = load_npz("facebox_r2")
    SH, SW, B = 2160, 3840, 16
    yi = (np.arange(SH) * 1024) // SH
    xi = (np.arange(SW) * 1024) // SW
    uniq = [np.ascontiguousarray(d["img%d_frame" % i][yi][:, xi]) for i in range(6)]
    frames = np.stack([uniq[b % 6] for b in range(B)])                 # 398 MB of sources
    plan = net.tuned_plan_text(1024, 1024, B)                          # tuned/facebox_1024x1024_b16.plan: what bench.py --arch facebox runs
    assert plan is not None and plan.split()[:4] == ["shape", "16", "1024", "1024"]
    net.import_plan(plan)
    res = net.detect_frames(frames)
    ran = {ln.split()[0]: ln.split() for ln in net.export_plan().strip().splitlines()[1:]}
    assert all(ran[ln.split()[0]][:len(ln.split())] == ln.split() for ln in plan.strip().splitlines()[1:]), "the batch did not run the committed plan"
    x = net.get_tensor("input")
    assert x.shape == (B, 3, 1024, 1024)
    singles = [net.detect_frames(u[None])[0] for u in uniq]
    for b, (bx, pr) in enumerate(res):
        # same image twice in the batch: same bits (images of a batch do not see each other); against the batch-1 call only to
        # f32 rounding -- a batch-1 plan may pick other tiles / split-K for a layer, i.e. another summation order
        assert np.array_equal(pr, res[b % 6][1]) and np.array_equal(bx, res[b % 6][0]) and len(pr) >= 3
        sb, sp = singles[b % 6]
        assert len(pr) == len(sp) and np.allclose(pr, sp, atol=1e-5) and np.allclose(bx, sb, atol=1e-5)
    for i in range(6):
        small = oin.resize_linear_u8(uniq[i], 1024, 1024)
        assert np.array_equal(x[i], small.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
        assert np.array_equal(x[i + 6], x[i])
        rb, rp = ofb.detect(fb_sd, small)
        bx, pr = res[i]
        assert len(pr) == len(rp)
        iou = opp.calculate_iou(rb.astype(np.float64), bx.astype(np.float64))
        assert (1 - iou.max(1)).max() <= 1e-3 and np.abs(pr[iou.argmax(1)] - rp).max() <= 1e-4
