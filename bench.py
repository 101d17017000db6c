#!/usr/bin/env python3
"""Benchmark of the hot path: detect + track on synthetic 1024x1024 frames (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL when N > 1).  A *step* is one batch-1 frame per GPU:
u8 frame (already resident in HBM) -> PyramidBox-Res50 forward -> decode + NMS + top-k on device ->
(N > 1: one RCCL all-gather of the fixed-size detection records) -> sequential IoU-tracker
association of the N frames, device resident.  Rank 0 prints ONE JSON line.

`roofline`: dominant kernel = the f32-MFMA implicit-GEMM convolution.  achieved = algorithmic conv
FLOPs per frame (2*MAC of the live convs, SURVEY.md 8(d): 755.751 GFLOP @1024x1024) divided by the
summed conv-launch durations per frame, measured with HIP events around every launch on the stream
the kernels run on (a profiled pass of the same frames right after the timed region).
`cpu_baseline`: the CPU oracle (oracle/, PyTorch-CPU convs + numpy post-processing + tracker) timed on
this host's cores on a bounded sample of the same frames; rank 0, N = 1 only.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak


def facebox_main(args):
    """Config 5 of BASELINE.json: FaceBoxes, 1024x1024 frames, `--batch` frames per step on one GPU; /255 + forward +
    softmax + decode_np + nms_np all on device (reference FACEBOX/My_test_facebox.py:12-36 after the resize).  Real
    weights (tests/golden/faceboxes_weights.npz = the reference's FACEBOX/faceboxes.pt stored as plain arrays)."""
    import torch
    B = args.batch if args.batch > 1 else 16
    lib = importlib.import_module("face-detection-and-tracking_amd._lib")
    FaceBox = importlib.import_module("face-detection-and-tracking_amd.FACEBOX.networks").FaceBox
    z = np.load(os.path.join(ROOT, "tests", "golden", "faceboxes_weights.npz"))
    sd = {k: z[k] for k in z.files}
    net = FaceBox()
    net.load_state_dict(sd)
    g = np.load(os.path.join(ROOT, "tests", "golden", "facebox.npz"))
    real = [g["img0_frame"], g["img1_frame"]]
    frames_h = np.stack([real[i % 2] for i in range(B)])
    dev = torch.device("cuda", 0)
    frames_d = torch.from_numpy(frames_h).to(dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    sp = ctypes.c_void_p(st.cuda_stream)
    L = lib.lib()
    net.detect_frames(frames_h)          # plan
    net.autotune(3)

    def step():
        lib.check(L.fdt_model_detect_facebox_dev(net._h, ctypes.c_void_p(frames_d.data_ptr()), lib.FRAME_U8_HWC_BGR,
                                                 B, 1024, 1024, 0.35, 0.5, ctypes.c_void_p(counts.data_ptr()), sp))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = net.detect_frames(frames_h)
    cpu = None
    if args.cpu_frames > 0:
        from oracle import facebox as ofb
        times = []
        for i in range(args.cpu_frames):
            t1 = time.perf_counter()
            rb, rp = ofb.detect(sd, frames_h[i % 2])
            times.append(time.perf_counter() - t1)
            gb, gp = res[i % 2]
            assert len(gp) == len(rp) and np.abs(gp - rp).max() < 1e-4 and np.abs(gb - rb).max() < 1e-4
        per = float(np.mean(times[1:])) if len(times) > 1 else times[0]
        cpu = {"value": round(1 / per, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": "%d frames, oracle/facebox.py (boxes/probs of these frames within 1e-4 of the GPU path)"
                         % max(len(times) - 1, 1)}
    print(json.dumps({"metric": "frames/sec (FaceBoxes detect) at 1024x1024", "value": round(B * args.steps / dt, 2),
                      "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "f32",
                      "data": "2 reference sample images (post-resize) tiled to the batch",
                      "config": {"workload": "FaceBoxes 1024x1024 batch=%d, decode_np+nms_np on device" % B,
                                 "weights": "reference FACEBOX/faceboxes.pt",
                                 "faces_per_image": [int(c) for c in counts.cpu()[:2]]},
                      "roofline": None, "cpu_baseline": cpu}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--height", type=int, default=0, help="frame height (default: --size)")
    ap.add_argument("--width", type=int, default=0, help="frame width (default: --size)")
    ap.add_argument("--arch", default="res50", choices=["res50", "try3", "facebox"],
                    help="facebox = config 5 of BASELINE.json (FaceBoxes, 1024x1024, --batch 16), single GPU")
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step (one batched forward)")
    ap.add_argument("--source", default="", help="HxW of raw source frames (e.g. 1080x1920): the frames are resized on the "
                    "GPU to --height x --width inside the timed step like iouTracke_cal.py:123 does with cv2.resize")
    ap.add_argument("--unique-frames", type=int, default=8)
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--profile-frames", type=int, default=4)
    ap.add_argument("--autotune", type=int, default=1,
                    help="1: use the committed tuned plan for this shape if there is one, else autotune (tile, split-K) "
                         "per conv layer at start-up; 2: always autotune; 0: analytic model only")
    ap.add_argument("--save-plan", type=int, default=0, help="write the autotuned plan under tuned/")
    ap.add_argument("--inflight", type=int, default=3,
                    help="frames in flight per GPU: consecutive batch-1 steps overlap on separate HIP streams "
                         "(detection of frame i+1 runs beside the tail / tracker step of frame i)")
    args = ap.parse_args()
    if args.arch == "facebox":
        return facebox_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (WORLD_SIZE=%d)"
                             % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    # FDT_BENCH_BACKEND=gloo lets the N > 1 path be rehearsed with several ranks sharing one GPU
    backend = os.environ.get("FDT_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = importlib.import_module("face-detection-and-tracking_amd")
    synth = importlib.import_module("face-detection-and-tracking_amd.synth")
    layers = importlib.import_module("face-detection-and-tracking_amd.layers")
    trk = importlib.import_module("face-detection-and-tracking_amd.tracker")
    par = importlib.import_module("face-detection-and-tracking_amd.parallel")
    lib = pkg._lib

    H = args.height or args.size
    W = args.width or args.size
    sd = synth.make_state_dict(args.arch, seed=0)
    NF = max(1, args.inflight)
    B = max(1, args.batch)
    nets = []
    for _ in range(NF):     # one handle (own activations + stream) per frame in flight; weights replicated
        if args.arch == "res50":
            n = importlib.import_module("face-detection-and-tracking_amd.pyramid").SFD(device=local_rank)
            n.priorbox = layers.PriorBoxLayer(W, H)
        else:
            n = importlib.import_module("face-detection-and-tracking_amd.pyramid_mb2_try3").SFD_mobile(device=local_rank)
            n.priorbox = layers.PriorBoxLayer(W, H, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
        n.load_state_dict(sd)
        n.cuda(); n.eval()
        n._sync_attributes(H, W)
        plan_file = os.path.join(ROOT, "face-detection-and-tracking_amd", "tuned",
                                 "%s_%dx%d_b%d.plan" % (args.arch, W, H, B))
        if args.autotune == 1 and os.path.exists(plan_file):
            n.import_plan(open(plan_file).read())      # committed result of an earlier autotune on MI355X
            plan_src = "tuned/" + os.path.basename(plan_file)
        elif args.autotune:
            # plan-time measurement of every (tile, split-K) variant per layer; outside the timed region
            n(synth.make_frames(B, H, W, seed=99) if B > 1 else synth.make_frames(1, H, W, seed=99)[0])
            n.autotune(3)
            plan_src = "autotuned at start-up"
            if args.save_plan and not nets:
                with open(plan_file, "w") as f:
                    f.write(n.export_plan())
        else:
            plan_src = "analytic model"
        nets.append(n)
    net = nets[0]
    top_k = net.detect.top_k

    # synthetic frames, resident in HBM before the timed region
    U = (max(args.unique_frames, B) + B - 1) // B * B          # whole batches
    SH, SW = (int(v) for v in args.source.lower().split("x")) if args.source else (H, W)
    frames_h = synth.make_frames(U, SH, SW, seed=1234 + rank)
    frames_d = torch.from_numpy(frames_h).to(dev)
    REC = 2 * top_k * 5                                       # one frame's Detect record
    fps = [par.FrameParallel(rank, world, B * REC, dev) for _ in range(NF)]
    counts = torch.zeros(2 * B, dtype=torch.int32, device=dev)
    tracker = trk.IouTracker(0.4, 0.6, 5, max_dets=2 * top_k, log_frames=64)
    # non-default torch streams: their handles go through the C ABI, so torch.cuda.Event brackets and the
    # RCCL collective are ordered with the library's launches.  One stream per frame in flight for the
    # detector, one for the (strictly sequential) exchange + association.
    det_streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
    trk_stream = torch.cuda.Stream(device=dev)
    sp_det = [ctypes.c_void_p(s_.cuda_stream) for s_ in det_streams]
    sp_trk = ctypes.c_void_p(trk_stream.cuda_stream)
    assert all(p_.value for p_ in sp_det) and sp_trk.value, "need real stream handles"
    det_done = [torch.cuda.Event() for _ in range(NF)]
    trk_done = [torch.cuda.Event() for _ in range(NF)]
    L = lib.lib()

    def step(i):
        k = i % NF
        f = frames_d[(i * B) % U:(i * B) % U + B]
        fp = fps[k]
        with torch.cuda.stream(det_streams[k]):
            det_streams[k].wait_event(trk_done[k])        # slot k's record was consumed (step i - NF)
            if args.source:     # raw source frames: resize + mean subtraction in one kernel, then the forward
                lib.check(L.fdt_model_forward_resized(nets[k]._h, ctypes.c_void_p(f.data_ptr()), 1, B, SH, SW, H, W,
                                                      ctypes.c_void_p(fp.mine.data_ptr()),
                                                      ctypes.c_void_p(counts.data_ptr()), sp_det[k]))
            else:
                lib.check(L.fdt_model_forward_dev(nets[k]._h, ctypes.c_void_p(f.data_ptr()), lib.FRAME_U8_HWC_BGR, B,
                                                  H, W, ctypes.c_void_p(fp.mine.data_ptr()),
                                                  ctypes.c_void_p(counts.data_ptr()), sp_det[k]))
            det_done[k].record(det_streams[k])
        with torch.cuda.stream(trk_stream):
            trk_stream.wait_event(det_done[k])
            # the one exchange step of the path: fixed-size per-frame box lists, rank order == frame order
            g = fp.exchange()
            for r in range(world):            # rank order == frame order; B consecutive frames per rank
                for b in range(B):
                    tracker.step_dev(ctypes.c_void_p(g[r].data_ptr() + 4 * b * REC), 2, top_k, W, H, 0.4, sp_trk)
            trk_done[k].record(trk_stream)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(trk_stream)
    for i in range(args.steps):
        step(args.warmup + i)
    e1.record(trk_stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_cand_last = int(counts.cpu()[1])
    tracks = tracker.finish()
    mine = fps[0].mine
    stream = sp_det[0]
    torch.cuda.set_stream(det_streams[0])

    # ---- per-launch timing of the dominant kernel (HIP events on the same stream) ---------------
    roof = None
    if rank == 0:
        net.profile(True)
        conv_ms, other_ms, flops = [], [], 0.0
        dw_stats = []
        for i in range(args.profile_frames + 1):
            if args.source:
                lib.check(L.fdt_model_forward_resized(net._h, ctypes.c_void_p(frames_d[(i * B) % U:].data_ptr()), 1, B,
                                                      SH, SW, H, W, ctypes.c_void_p(mine.data_ptr()),
                                                      ctypes.c_void_p(counts.data_ptr()), stream))
            else:
                lib.check(L.fdt_model_forward_dev(net._h, ctypes.c_void_p(frames_d[(i * B) % U:].data_ptr()),
                                                  lib.FRAME_U8_HWC_BGR, B, H, W, ctypes.c_void_p(mine.data_ptr()),
                                                  ctypes.c_void_p(counts.data_ptr()), stream))
            torch.cuda.synchronize()
            prof = net.profile_read()
            if i == 0:
                continue      # first profiled frame creates the events
            conv_ms.append(sum(ms for nm, ms, fl in prof if "#k" in nm))
            other_ms.append(sum(ms for nm, ms, fl in prof if "#k" not in nm))
            flops = sum(fl for nm, ms, fl in prof if "#k" in nm)
            n_conv = sum(1 for nm, ms, fl in prof if "#k" in nm)
            if args.arch == "try3":
                # the depthwise 3x3 layers (HBM-bound by construction: in + out bytes, nothing else).  A depthwise op
                # has 18 FLOP per output element, so its output size follows from its FLOPs; the input is stride^2
                # times that (strides: pyramid_mb2_try3.py:150-168, layer6 :178).
                s2 = {"features.%d.conv.3" % i for i, _, _, st, _ in synth.try3_blocks() if st == 2} | {"layer6.conv.3"}
                dwb = dwm = 0.0
                for nm, ms, fl in prof:
                    if "#k" not in nm and (nm.startswith("features.") or nm.startswith("layer6.")) and fl > 0:
                        out_el = fl / 18.0
                        dwb += 4.0 * out_el * (1 + (4 if nm in s2 else 1))
                        dwm += ms
                dw_stats.append((dwb, dwm))
        net.profile(False)
        cms = float(np.mean(conv_ms))
        achieved = flops / (cms * 1e-3) / 1e12
        # HBM bytes per launch come from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
        # command (PMC cannot be read from inside the process); the committed summary is quoted here.
        traffic, traffic_src, traffic_cal = None, None, None
        tj = os.path.join(ROOT, "profiles", "r01", "conv_hbm_traffic.json")
        if args.arch == "res50" and H == 1024 and os.path.exists(tj):
            tdata = json.load(open(tj))
            traffic = round(tdata["hbm_bytes_per_launch"])
            if tdata.get("hbm_bytes_per_launch_calibrated"):
                traffic_cal = round(tdata["hbm_bytes_per_launch_calibrated"])
            traffic_src = "profiles/r01/conv_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "traffic_calibrated": traffic_cal,     # FETCH_SIZE divided by its measured per-byte reading on this
                                                       # code's LDS-DMA pattern instead of doubled (see the JSON note)
                "algorithmic_bytes_per_launch": round((3.52e9 + 0.2688e9) / 105) if args.arch == "res50" and H == 1024 and W == 1024 else None,
                "kernel": "conv_kernel (f32 MFMA implicit GEMM)", "launches_per_frame": n_conv,
                "avg_launch_us": round(cms * 1e3 / n_conv, 2), "conv_ms_per_frame": round(cms / B, 3),
                "other_ms_per_frame": round(float(np.mean(other_ms)) / B, 3),
                "algorithmic_gflop_per_frame": round(flops / B / 1e9, 3)}
        if dw_stats and dw_stats[-1][1] > 0:      # config 3: state the HBM side too
            dwb = float(np.mean([d[0] for d in dw_stats]))
            dwm = float(np.mean([d[1] for d in dw_stats]))
            roof["hbm_side"] = {"bound": "hbm", "kernel": "dwconv3_vec_kernel (depthwise 3x3 + BN + ReLU6)",
                                "achieved": round(dwb / (dwm * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                "frac": round(dwb / (dwm * 1e-3) / 1e9 / 8000.0, 4),
                                "algorithmic_bytes_per_frame": round(dwb / B), "ms_per_frame": round(dwm / B, 4)}

    # ---- CPU baseline: the oracle on this host's cores, bounded sample ---------------------------
    cpu, parity = None, None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import postproc as opp
        from oracle import pyramidbox as opb
        from oracle import ingest as oin
        ncores = torch.get_num_threads()
        ref_trk = opp.IouTracker(0.4, 0.6, 5)
        times, ref_dets, gpu_dets = [], [], []
        for i in range(args.cpu_frames):
            t1 = time.perf_counter()
            src = frames_h[i % U]
            if args.source:       # the oracle's restatement of cv2.resize(image, (W, H)) (8-bit INTER_LINEAR)
                src = oin.resize_linear_u8(src, W, H)
            y = opb.detect_frame(sd, src, args.arch)
            det_ref = opp.unpack_detections(y, W, H, 0.4)
            with np.errstate(all="ignore"):
                ref_trk.step(det_ref)
            times.append(time.perf_counter() - t1)
            # parity of the same frames on the GPU path (checker only; not timed)
            yg = (net.forward_resized(frames_h[i % U], (W, H)) if args.source else net(frames_h[i % U])).numpy()
            ref_dets.append(det_ref)
            gpu_dets.append(opp.unpack_detections(yg, W, H, 0.4))
        ap, n_truth, n_pred = opp.ap_against_reference(gpu_dets, ref_dets, 0.5)
        iou_def = 0.0
        for g, r in zip(gpu_dets, ref_dets):
            if g.shape == r.shape and r.shape[0]:
                with np.errstate(all="ignore"):
                    iou_def = max(iou_def, float((1 - opp.calculate_iou(r[:, :4].astype(np.float64),
                                                                         g[:, :4].astype(np.float64)).max(1)).max()))
        parity = {"ap_vs_cpu_ref": round(ap, 6), "ref_boxes": n_truth, "gpu_boxes": n_pred,
                  "max_iou_deficit": float("%.3g" % iou_def), "frames": len(ref_dets)}
        per = float(np.mean(times[1:])) if len(times) > 1 else times[0]
        cpu = {"value": round(1.0 / per, 4), "unit": "frames/s", "cores": ncores, "kind": "port",
               "sample": "%d frames of the same %dx%d workload after 1 warm-up (oracle/: torch-CPU convs + numpy "
                         "Detect + tracker), %.2f s/frame" % (max(len(times) - 1, 1), H, W, per)}

    if rank == 0:
        frames = args.steps * world * B
        line = {
            "metric": "frames/sec (detect+track) at %dx%d" % (W, H),
            "value": round(frames / dt, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "PyramidBox-%s %dx%d synthetic u8 frames%s, batch=%d per GPU, decode+NMS+IoU-tracker "
                                   "on device" % ("Res50" if args.arch == "res50" else "MobileNetV2-try3", W, H,
                                                  " resized on the GPU from %dx%d sources" % (SW, SH) if args.source else "",
                                                  B),
                       "frames_per_step": world * B, "frames_in_flight_per_gpu": NF, "kernel_plan": plan_src, "parallelism": "frame-parallel x%d%s" % (
                           world, ", RCCL all-gather of box lists" if world > 1 else ""),
                       "weights": "seeded synthetic (seed 0)", "detections_last_frame": n_cand_last,
                       "tracks": len(tracks), "gpu_ms_per_step_events": round(gpu_ms / args.steps, 4),
                       "device": pkg.device_name(local_rank)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
