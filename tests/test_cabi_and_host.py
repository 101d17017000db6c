"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol include/fdt.h
declares, validates arguments before touching the GPU, and fails loudly without one.  No compute."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "fdt.h")


def M(name):
    return importlib.import_module("face-detection-and-tracking_amd." + name)


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fdt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = M("_lib")
    L = lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), "include/fdt.h declares %s but libfdt_hip.so does not export it" % s
    assert set(syms) == set(lib.SIGNATURES), set(syms) ^ set(lib.SIGNATURES)
    assert L.fdt_version() >= 100


def test_header_cites_reference_lines():
    src = open(HEADER).read()
    for ref in ("prior_box.py:28-44", "box_utils.py:238-258", "box_utils.py:275-340", "detection.py:15-84",
                "calc_performance.py:54-74", "iouTracke_cal.py:113-156", "pyramid.py:218-351"):
        assert ref in src, ref


def test_argument_errors_do_not_need_a_gpu():
    lib = M("_lib")
    L = lib.lib()
    z = np.zeros((1, 4, 4), np.float32)
    out = np.zeros((1, 2, 750, 5), np.float32)
    rc = L.fdt_detect(lib.ptr(z), lib.ptr(np.zeros((1, 4, 2), np.float32)), lib.ptr(z[0]), 1, 4, 2, 750,
                      0.3, 0.0, 5000, 0.1, 0.2, lib.ptr(out), None)
    assert rc == lib.FDT_ERR_ARG
    assert b"nms_threshold must be non negative" in L.fdt_last_error()
    assert L.fdt_detect_workspace_bytes(1, 87360, 5000) > 4 * 1024 * 1024
    assert L.fdt_pairwise_iou(None, 3, None, 3, 7, None) == lib.FDT_ERR_ARG
    cnt = ctypes.c_int(-1)
    assert L.fdt_nms(None, None, 0, 0.5, 10, lib.ptr(np.zeros(1, np.int64)), ctypes.byref(cnt)) == 0
    assert cnt.value == 0            # empty input: reference box_utils.py:290-291 returns (keep, 0)
    # tracker: one frame's detections and association state live in LDS -> max_dets is bounded (include/fdt.h)
    assert not L.fdt_tracker_create(0.4, 0.6, 5, 4096, 64)
    assert b"does not fit the LDS-resident frame state" in L.fdt_last_error()
    assert not L.fdt_tracker_create(0.4, 0.6, 5, 0, 64)
    # unknown architecture ids are refused before any device work
    assert not L.fdt_model_create(99, 0)
    assert b"unknown arch" in L.fdt_last_error()


def test_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    lib = M("_lib")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(lib.FdtError):
        M("pyramid").build_sfd('test', 640, 2)
    with pytest.raises(lib.FdtError):
        M("layers").PriorBoxLayer(640, 640)(0, 4, 4)
    with pytest.raises(lib.FdtError):
        M("tracker").IouTracker()
    # a missing shared object is an error, never a CPU fallback
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libfdt_hip.so")
    with pytest.raises(lib.FdtError, match="no CPU fallback"):
        lib.lib()


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may touch oracle/: neither the package nor
    tools/ does, and bench.py imports it only inside the cpu-baseline branches."""
    for sub in ("face-detection-and-tracking_amd", "tools"):
        for dp, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h", ".sh")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)
    bench = open(os.path.join(ROOT, "bench.py")).read().split("\n")
    for i, line in enumerate(bench):
        if "from oracle" in line or "import oracle" in line:
            ctx = "\n".join(bench[max(0, i - 6):i])
            assert "cpu_frames > 0" in ctx, "bench.py line %d imports the oracle outside a cpu-baseline branch" % (i + 1)


def test_reference_api_surface():
    assert M("pyramid").build_sfd('bogus', 640, 2) is None            # pyramid.py:368-370
    assert M("pyramid").build_sfd('test', 300, 2) is None             # pyramid.py:371-373
    assert M("pyramid_mb2_try3").build_sfd_mobile('test', 300, 2) is None
    with pytest.raises(ValueError):
        M("layers").Detect(2, 0, 750, 0.3, 0)                         # detection.py:28-29
    d = M("layers").Detect(2, 0, 750, 0.05, 0.3)
    assert (d.top_k, d.nms_top_k, d.variance) == (750, 5000, [0.1, 0.2])
    pb = M("layers").PriorBoxLayer(640, 480)
    assert (pb.width, pb.height, tuple(pb.stride)) == (640, 480, (4, 8, 16, 32, 64, 128))


def test_synthetic_schema_counts(synth):
    r = synth.res50_schema()
    assert len(r) == 474
    n_params = sum(int(np.prod(s)) for _, s, k in r if k != "bn_nbt" and "running" not in _)
    assert n_params == 67269390
    t = synth.try3_schema()
    assert len([1 for _, _, k in t if k != "bn_nbt"]) == 386
    a = synth.make_state_dict("try3", 3)["features.5.conv.3.weight"]
    b = synth.make_state_dict("try3", 3)["features.5.conv.3.weight"]
    assert np.array_equal(a, b) and a.shape == (192, 1, 3, 3)


def test_round2_entry_points_validate_arguments_without_a_gpu():
    lib = M("_lib")
    L = lib.lib()
    assert not L.fdt_model_clone(None)
    assert b"fdt_model_clone" in L.fdt_last_error()
    assert L.fdt_model_enable_graph(None, 1) == lib.FDT_ERR_ARG
    assert L.fdt_tracker_step_dev_multi(None, None, 8, 7500, 2, 750, 640, 480, 0.4, None) == lib.FDT_ERR_ARG
    t = ctypes.c_int(-1)
    assert L.fdt_model_forward_async(None, None, 0, 1, 8, 8, 0, 0, ctypes.byref(t)) == lib.FDT_ERR_ARG
    assert L.fdt_model_wait(None, 0, None, None, None) == lib.FDT_ERR_ARG
    assert L.fdt_model_async_record(None, 0, None, None) == lib.FDT_ERR_ARG
    assert L.fdt_model_release(None, 0, None) == lib.FDT_ERR_ARG
    assert L.fdt_comm_unique_id(None) == lib.FDT_ERR_ARG
    assert not L.fdt_comm_init_rank(0, 0, None, 0) and b"fdt_comm_init_rank" in L.fdt_last_error()
    assert not L.fdt_comm_init_all(0, None)
    assert L.fdt_allgather_dets(None, 0, None, None, 1, None) == lib.FDT_ERR_ARG
    assert L.fdt_comm_world(None, None, None) == lib.FDT_ERR_ARG
    assert L.fdt_model_detect_facebox_resized(None, None, 0, 1, 2160, 3840, 0.35, 0.5, None, None, None, None) == lib.FDT_ERR_ARG
    assert L.fdt_model_traffic(None, None, None, 0, None, None) == lib.FDT_ERR_ARG
    L.fdt_comm_destroy(None)          # no-ops on NULL like free()
    L.fdt_model_destroy(None)


def test_header_documents_the_round2_surface():
    src = open(HEADER).read()
    for word in ("FDT_COMM_ID_BYTES 128", "fdt_tracker_step_dev_multi", "fdt_model_clone", "fdt_model_forward_async",
                 "MyTrain_repo.py:71", "iouTracke_cal.py:119-124", "FACEBOX/My_test_facebox.py:12-36"):
        assert word in src, word


def test_bench_host_helpers_and_committed_plans():
    """Host-side bookkeeping of bench.py (no GPU): op-name parsing, kernel labels, CPU description; and the committed
    plans still contain the layers whose kernels tools/refresh_profiles.sh profiles one by one (it reads kind / tile /
    split-K from the plan, so a re-tune cannot leave the PMC files describing a kernel that is no longer used)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_op("layer3.1.conv1#k10t27s1") == ("layer3.1.conv1", 10, 27, 1)
    assert bench.parse_op("pool") == ("pool", None, None, None)
    assert bench.kernel_label(8, 30).startswith("conv_wino4_kernel<3x3s1_wino")
    assert bench.kernel_label(9, 22).startswith("conv_wino2_kernel<3x3d2_wino")
    assert bench.kernel_label(0, 12) == "conv_kernel<1x1s1, tile 12>"
    assert bench.kernel_label(13, 31) == "conv_n8_kernel<3x3s1_n8, tile 31>"
    assert bench.kernel_label(16, 34) == "conv1x1p_kernel<1x1s1_p16, tile 34>"     # persistent-tile 1x1 (conv_1x1p.h)
    # the watchdog around the multi-GPU communicator set-up: result, exception and timeout
    import time
    assert bench.run_with_timeout(lambda: 7, 5.0) == (True, 7)
    done, res = bench.run_with_timeout(lambda: (_ for _ in ()).throw(ValueError("x")), 5.0)
    assert done and isinstance(res, ValueError)
    assert bench.run_with_timeout(lambda: time.sleep(2.0), 0.1) == (False, None)
    model, phys, logical = bench.cpu_info()
    assert phys >= 1 and logical >= phys and isinstance(model, str)
    plan_dir = os.path.join(ROOT, "face-detection-and-tracking_amd", "tuned")
    plan = {}
    for ln in open(os.path.join(plan_dir, "res50_1024x1024_b1.plan")):
        f = ln.split()
        if len(f) >= 4 and f[0] != "shape":
            plan[f[0]] = tuple(int(v) for v in f[1:4])
    for layer in ("conv2_SSH.conv1", "layer3.1.conv1", "layer1.0.conv3", "layer2.1.conv3"):
        assert layer in plan, layer
    assert plan["conv2_SSH.conv1"][0] == 14 and plan["conv2_SSH.conv1"][1] in (32, 33)    # Winograd F(4x4,3x3)
    assert plan["conv2_SSH.conv2"][0] == 15 and plan["conv2_SSH.conv2"][1] == 32          # dilated F(4x4,3x3) on parity sub-lattices
    assert bench.kernel_label(14, 32).startswith("conv_wino44_kernel<3x3s1_wino44") and bench.WINO_RATIO[14] == 4.0
    # the HBM-heavy 256^2 expand layers: a 1x1 class without split-K (persistent f32 16, split-bf16 21, persistent split-bf16 26)
    assert plan["layer1.1.conv3"][0] in (16, 21, 26) and plan["layer1.1.conv3"][2] == 1
    assert bench.kernel_label(26, 35) == "conv1x1p_b3_kernel<1x1s1_pb3, tile 35>" and bench.kernel_label(24, 6).startswith("conv_stem_u8b_kernel<7x7s2_u8b")
    assert len(plan) == 105                                                               # every conv layer of Res50
    sh = open(os.path.join(ROOT, "tools", "refresh_profiles.sh")).read()
    assert all(("plan_of " + layer) in sh for layer in ("conv2_SSH.conv1", "conv2_SSH.conv2", "layer3.1.conv1", "layer1.0.conv3"))
