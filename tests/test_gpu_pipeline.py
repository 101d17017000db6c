"""The timed path of bench.py, pinned: several model handles (fdt_model_clone: one weight copy) on their own HIP streams,
captured HIP graphs, a tracker stream ordered by events, the step's frames associated in one launch
(fdt_tracker_step_dev_multi), the RCCL all-gather behind the C ABI.  Everything is compared bit-for-bit with the
sequential synchronous path (`net(frame)`) and the oracle's IouTracker (reference iouTracke_cal.py:117-156,174-177)."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

from oracle import postproc as opp

pytestmark = pytest.mark.gpu


def M(name):
    return importlib.import_module("face-detection-and-tracking_amd." + name)


def tracks_key(tracks):
    return [(t["start_frame"], float(t["max_score"]), [list(map(float, b)) for b in t["bboxes"]]) for t in tracks]


@pytest.fixture(scope="module")
def res50(res50_sd):
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(res50_sd)
    yield net
    net.close()


def moving_frames(synth, n, H, W, seed):
    """n frames that change slowly (a base frame plus a drifting bright block), so that detections persist and tracks
    actually form under the synthetic weights."""
    base = synth.make_frames(1, H, W, seed=seed)[0]
    out = np.repeat(base[None], n, 0).copy()
    for i in range(n):
        out[i, 8 + i:40 + i, 16 + 2 * i:64 + 2 * i] = 255 - out[i, 8 + i:40 + i, 16 + 2 * i:64 + 2 * i]
    return out


def test_clone_shares_weights_and_matches(res50, synth):
    L = M("_lib")
    H, W = 128, 160
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frame = synth.make_frames(1, H, W, seed=3)[0]
    y0 = res50(frame).numpy()
    free0, tot = ctypes.c_longlong(0), ctypes.c_longlong(0)
    L.check(L.lib().fdt_device_mem_info(ctypes.byref(free0), ctypes.byref(tot)))
    c = res50.clone()
    y1 = c(frame).numpy()
    free1 = ctypes.c_longlong(0)
    L.check(L.lib().fdt_device_mem_info(ctypes.byref(free1), ctypes.byref(tot)))
    assert np.array_equal(y0, y1)                        # same weights, same plan -> same bits
    # a clone costs activations + workspaces at this size (tens of MB), not another ~270 MB of weights (+ Winograd copies)
    assert free0.value - free1.value < 200e6, (free0.value - free1.value)
    # weights are read-only while shared
    w = np.zeros((64, 3, 7, 7), np.float32)
    dims = (ctypes.c_longlong * 4)(64, 3, 7, 7)
    rc = L.lib().fdt_model_set_tensor(c._h, b"conv1.weight", L.ptr(w), 4, dims)
    assert rc == L.FDT_ERR_STATE
    c.close()
    y2 = res50(frame).numpy()                            # the original survives its clone
    assert np.array_equal(y0, y2)


def test_distinct_handles_on_distinct_host_threads(res50, res50_sd, synth):
    """include/fdt.h: a handle is not thread-safe, but DISTINCT handles may be driven from distinct host threads at the same
    time (SURVEY.md 8(b) Threading) -- two clones sharing one weight copy plus one independent model, each on its own
    thread, first forwards (plan, weight upload, graph capture) racing each other; every result equals the one the same
    handle gives single-threaded, and fdt_last_error is per thread."""
    import threading
    L = M("_lib")
    H, W = 128, 160
    PB, Det = M("layers").PriorBoxLayer, M("layers").Detect
    res50.priorbox = PB(W, H); res50.firstTime = True
    res50.detect = Det(2, 0, 750, 0.05, 0.35)
    frames = [synth.make_frames(6, H, W, seed=40 + t) for t in range(3)]
    res50(frames[0][0])                                   # the parent must have run before it can be cloned
    want = [[res50(f).numpy().copy() for f in fr] for fr in frames]       # same weights for all three handles
    other = M("pyramid").build_sfd('test', 640, 2)
    other.load_state_dict(res50_sd)
    other.priorbox = PB(W, H); other.detect = Det(2, 0, 750, 0.05, 0.35)
    nets = [res50.clone(), res50.clone(), other]           # none of them has run a forward yet
    got, errs, msgs = [None] * 3, [], [None] * 3
    start = threading.Barrier(3)

    def work(t):
        try:
            start.wait()
            outs = []
            for rep in range(3):                           # eager first pass, then graph replays
                outs = [nets[t](f).numpy().copy() for f in frames[t]]
                if t == 2:                                 # a host-pointer entry point between the forwards: it must not
                    a = np.array([[0, 0, 2, 2], [1, 1, 3, 3]], np.float64)    # touch the legacy stream while the other
                    iou = np.empty((2, 2), np.float64)                          # threads capture their graphs
                    L.check(L.lib().fdt_pairwise_iou(L.ptr(a), 2, L.ptr(a), 2, 1, L.ptr(iou)))   # 1 = FDT_F64
                    assert iou[0, 0] == 1.0 and iou[0, 1] == 1.0 / 7.0
            got[t] = outs
            if t == 1:                                     # this thread fails a call; the others must not see its message
                rc = L.lib().fdt_priorbox(0, 0, 0, 0, 0, None, 0, 0, 0, None)
                assert rc != 0
            start.wait()
            msgs[t] = L.lib().fdt_last_error() or b""
        except Exception as e:                             # noqa: BLE001 -- re-raised on the main thread
            errs.append((t, e))
            start.abort()

    th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join(120)
    assert not errs, errs
    for t in range(3):
        assert got[t] is not None and all(np.array_equal(a, b) for a, b in zip(got[t], want[t])), t
    assert msgs[1] and not msgs[0] and not msgs[2], msgs
    for n in nets:
        n.close()


def test_graph_replay_equals_eager(res50, synth):
    H, W = 136, 200
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames = synth.make_frames(3, H, W, seed=11)
    res50.enable_graph(False)
    eager = [res50(f).numpy() for f in frames]
    res50.enable_graph(True)
    res50(frames[0])                                     # eager run of the plan, then capture, then replays
    got = [res50(f).numpy() for f in frames] + [res50(f).numpy() for f in frames]
    for i, g in enumerate(got):
        assert np.array_equal(g, eager[i % 3]), i
    assert int((eager[0][0, 1, :, 0] > 0).sum()) > 5


@pytest.mark.parametrize("inflight,multi", [(3, True), (3, False), (1, True)])
def test_pipelined_streams_match_sequential_and_oracle(res50, synth, inflight, multi):
    """>= 24 frames through `inflight` handles x streams + the tracker stream exactly as bench.py runs them: per-frame
    Detect records and the final tracks are bit-equal to the sequential path + the oracle tracker."""
    H, W, N = 128, 160, 26
    dev = torch.device("cuda", 0)
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames = moving_frames(synth, N, H, W, seed=5)
    # sequential, synchronous reference run of the same handles' weights
    seq = [res50(f).numpy() for f in frames]
    ref = opp.IouTracker(0.4, 0.6, 5)
    for y in seq:
        with np.errstate(all="ignore"):
            ref.step(opp.unpack_detections(y, W, H, 0.4))
    want = tracks_key(ref.finish())
    assert len(want) >= 2 and max(len(t[2]) for t in want) >= 8

    pipe = M("pipeline").DetectTrackPipeline(res50, H, W, dev, inflight=inflight, multi_step=multi, log_frames=8)
    frames_d = torch.from_numpy(frames).to(dev)
    recs = {}
    for i in range(N):
        pipe.step(i, frames_d[i:i + 1])
        if i % 5 == 4 or i == N - 1:                     # sample records without disturbing the overlap every step
            recs[i] = pipe.record_of_slot(i % pipe.NF)[0].copy()
    got = tracks_key(pipe.finish())
    for i, r in recs.items():
        assert np.array_equal(r, seq[i][0]), i
    assert got == want
    pipe.close()


@pytest.mark.parametrize("group,inflight,N", [(4, 2, 26), (2, 3, 25)])
def test_frames_handed_over_one_at_a_time_run_as_grouped_launches(res50, synth, group, inflight, N):
    """DetectTrackPipeline.step_frame: single frames in, `group` consecutive frames per launch (the handle runs its batch-`group`
    plan), tracker in frame order; a partly filled last group is flushed by finish().  Same tracks as the one-frame-at-a-time
    sequential path + the oracle tracker, bit for bit (frames are independent until the tracker)."""
    H, W = 128, 160
    dev = torch.device("cuda", 0)
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames = moving_frames(synth, N, H, W, seed=11)
    # the reference run uses the same batch-`group` forwards, synchronously (another batch size may pick other kernels for a
    # layer, i.e. another summation order); images of a batch do not see each other, so the tail is padded with any frame
    seq = []
    for g0 in range(0, N, group):
        chunk = frames[g0:g0 + group]
        n = len(chunk)
        if n < group:
            chunk = np.concatenate([chunk, np.repeat(frames[:1], group - n, 0)])
        y = res50(chunk).numpy()
        seq += [y[j:j + 1] for j in range(n)]
    ref = opp.IouTracker(0.4, 0.6, 5)
    for y in seq:
        with np.errstate(all="ignore"):
            ref.step(opp.unpack_detections(y, W, H, 0.4))
    want = tracks_key(ref.finish())
    assert len(want) >= 2
    pipe = M("pipeline").DetectTrackPipeline(res50, H, W, dev, inflight=inflight, batch=group, log_frames=8)
    frames_d = torch.from_numpy(frames).to(dev)
    for i in range(N):
        pipe.step_frame(i, frames_d[i:i + 1])
    got = tracks_key(pipe.finish())                     # N is not a multiple of the group: the tail is flushed here
    assert got == want
    pipe.close()
    res50.firstTime = True                              # the handle is back to batch-1 plans for the tests that follow


def test_cu_partitioned_streams_give_the_same_tracks(res50, synth, monkeypatch):
    """fdt_stream_create_partition (include/fdt.h): detector streams confined to halves of every XCD's compute units.  A speed
    experiment (slower: docs/EXPERIMENTS.md R3-6) -- the placement must not change a single bit."""
    H, W, N = 128, 160, 14
    dev = torch.device("cuda", 0)
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames_d = torch.from_numpy(moving_frames(synth, N, H, W, seed=9)).to(dev)
    out = []
    for parts in ("1", "2"):
        monkeypatch.setenv("FDT_EXPERIMENTS", "1")        # pipeline.py honours its experiment hooks only with this
        monkeypatch.setenv("FDT_CU_PARTS", parts)
        pipe = M("pipeline").DetectTrackPipeline(res50, H, W, dev, inflight=4, log_frames=8)
        for i in range(N):
            pipe.step(i, frames_d[i:i + 1])
        out.append((tracks_key(pipe.finish()), pipe.record_of_slot((N - 1) % pipe.NF)[0].copy()))
        pipe.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    L = M("_lib")
    st = ctypes.c_void_p()
    assert L.lib().fdt_stream_create_partition(4, 4, ctypes.byref(st)) != 0     # partition index out of range
    assert L.lib().fdt_stream_create_partition(0, 3, ctypes.byref(st)) != 0     # 1, 2 or 4 partitions


def test_step_dev_multi_equals_sequential_steps():
    """G = 8 gathered records in one launch == 8 launches == the oracle (random records incl. empty frames)."""
    TOP_K, G, STEPS = 40, 8, 9
    rng = np.random.default_rng(17)
    dev = torch.device("cuda", 0)
    trk = M("tracker")
    a = trk.IouTracker(0.4, 0.6, 5, max_dets=2 * TOP_K, log_frames=16)
    b = trk.IouTracker(0.4, 0.6, 5, max_dets=2 * TOP_K, log_frames=16)
    ref = opp.IouTracker(0.4, 0.6, 5)
    faces = rng.uniform(0.1, 0.6, (6, 2))
    for s in range(STEPS):
        rec = np.zeros((G, 2, TOP_K, 5), np.float32)
        for g in range(G):
            f = s * G + g
            n = 0 if f % 13 == 5 else 6
            sc = np.sort(rng.uniform(0.41, 1.0, n))[::-1]
            xy = faces[:n] + 0.003 * f + rng.uniform(-0.002, 0.002, (n, 2))
            rec[g, 1, :n, 0] = sc
            rec[g, 1, :n, 1:3] = xy
            rec[g, 1, :n, 3:5] = xy + 0.15
        d = torch.from_numpy(rec).to(dev)
        a.step_dev_multi(ctypes.c_void_p(d.data_ptr()), G, 2 * TOP_K * 5, 2, TOP_K, 640, 480, 0.4, None)
        for g in range(G):
            b.step_dev(ctypes.c_void_p(d.data_ptr() + 4 * g * 2 * TOP_K * 5), 2, TOP_K, 640, 480, 0.4, None)
            with np.errstate(all="ignore"):
                ref.step(opp.unpack_detections(rec[g][None], 640, 480, 0.4))
        torch.cuda.synchronize()
    ta, tb, tr = tracks_key(a.finish()), tracks_key(b.finish()), tracks_key(ref.finish())
    assert len(tr) >= 4
    assert ta == tr and tb == tr
    a.close(); b.close()


def test_tracker_steps_on_alternating_streams_are_ordered():
    """A caller that alternates streams still gets the sequential association (the tracker chains its steps)."""
    TOP_K = 16
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    t = M("tracker").IouTracker(0.4, 0.6, 3, max_dets=2 * TOP_K, log_frames=64)
    ref = opp.IouTracker(0.4, 0.6, 3)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    recs = []
    for f in range(40):
        rec = np.zeros((2, TOP_K, 5), np.float32)
        xy = np.array([[0.2, 0.2], [0.6, 0.5]]) + 0.004 * f
        rec[1, :2, 0] = [0.9, 0.7]
        rec[1, :2, 1:3] = xy
        rec[1, :2, 3:5] = xy + 0.2 + rng.uniform(0, 0.01)
        recs.append(torch.from_numpy(rec).to(dev))
        with np.errstate(all="ignore"):
            ref.step(opp.unpack_detections(rec[None], 640, 480, 0.4))
    torch.cuda.synchronize()
    for f, d in enumerate(recs):
        s = streams[f % 3]
        t.step_dev(ctypes.c_void_p(d.data_ptr()), 2, TOP_K, 640, 480, 0.4, ctypes.c_void_p(s.cuda_stream))
    torch.cuda.synchronize()
    assert tracks_key(t.finish()) == tracks_key(ref.finish())
    t.close()


def test_rccl_allgather_through_the_c_abi_single_rank():
    """World of one: the communicator builds (ncclCommInitRank and ncclCommInitAll) and the all-gather is the identity."""
    L = M("_lib")
    lib = L.lib()
    dev = torch.device("cuda", 0)
    idb = ctypes.create_string_buffer(128)
    L.check(lib.fdt_comm_unique_id(idb))
    for make in (lambda: lib.fdt_comm_init_rank(1, 0, idb, 0), lambda: lib.fdt_comm_init_all(1, (ctypes.c_int * 1)(0))):
        c = make()
        assert c, lib.fdt_last_error()
        w, nl = ctypes.c_int(0), ctypes.c_int(0)
        L.check(lib.fdt_comm_world(c, ctypes.byref(w), ctypes.byref(nl)))
        assert (w.value, nl.value) == (1, 1)
        src = torch.arange(7500, dtype=torch.float32, device=dev)
        dst = torch.zeros(7500, dtype=torch.float32, device=dev)
        st = torch.cuda.Stream(device=dev)
        L.check(lib.fdt_allgather_dets(c, 0, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), 7500,
                                       ctypes.c_void_p(st.cuda_stream)))
        st.synchronize()
        assert torch.equal(src, dst)
        lib.fdt_comm_destroy(c)
    assert lib.fdt_allgather_dets(None, 0, None, None, 1, None) == L.FDT_ERR_ARG


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_rccl_allgather_two_devices_single_process():
    L = M("_lib")
    lib = L.lib()
    c = lib.fdt_comm_init_all(2, (ctypes.c_int * 2)(0, 1))
    assert c, lib.fdt_last_error()
    bufs = []
    for d in range(2):
        dev = torch.device("cuda", d)
        bufs.append((torch.full((7500,), float(d + 1), device=dev), torch.zeros(15000, device=dev)))
    L.check(lib.fdt_comm_group_begin())
    for d in range(2):
        L.check(lib.fdt_allgather_dets(c, d, ctypes.c_void_p(bufs[d][0].data_ptr()),
                                       ctypes.c_void_p(bufs[d][1].data_ptr()), 7500, None))
    L.check(lib.fdt_comm_group_end())
    for d in range(2):
        torch.cuda.synchronize(d)
        assert float(bufs[d][1][:7500].sum()) == 7500.0 and float(bufs[d][1][7500:].sum()) == 15000.0
    lib.fdt_comm_destroy(c)


def test_async_host_ingest_matches_sync(res50, synth):
    """fdt_model_forward_async / fdt_model_wait (pinned ring, copy stream, two tickets in flight per handle) returns the
    same records as the synchronous host path, also through the device-side hand-over to the tracker."""
    L = M("_lib")
    lib = L.lib()
    H, W, N = 128, 160, 9
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames = moving_frames(synth, N, H, W, seed=9)
    seq = [res50(f).numpy() for f in frames]
    ref = opp.IouTracker(0.4, 0.6, 3)
    for y in seq:
        with np.errstate(all="ignore"):
            ref.step(opp.unpack_detections(y, W, H, 0.4))
    trk = M("tracker").IouTracker(0.4, 0.6, 3, max_dets=1500, log_frames=64)
    st = torch.cuda.Stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    pending, got = [], []

    def retire():
        t = pending.pop(0)
        rec = ctypes.c_void_p(0)
        L.check(lib.fdt_model_async_record(res50._h, t, ctypes.byref(rec), sp))
        trk.step_dev(rec, 2, 750, W, H, 0.4, sp)
        out = np.empty((1, 2, 750, 5), np.float32)
        cnt = np.zeros((1, 2), np.int32)
        L.check(lib.fdt_model_wait(res50._h, t, L.ptr(out), L.ptr(cnt), sp))
        got.append(out)

    for i in range(N):
        if len(pending) == 2:
            retire()
        t = ctypes.c_int(-1)
        buf = frames[i].copy()
        L.check(lib.fdt_model_forward_async(res50._h, L.ptr(buf), L.FRAME_U8_HWC_BGR, 1, H, W, 0, 0, ctypes.byref(t)))
        buf[:] = 0                                        # the caller's buffer is free as soon as the call returns
        pending.append(t.value)
    # a third ticket on one handle is refused, not queued
    t = ctypes.c_int(-1)
    assert lib.fdt_model_forward_async(res50._h, L.ptr(frames[0]), L.FRAME_U8_HWC_BGR, 1, H, W, 0, 0,
                                       ctypes.byref(t)) == L.FDT_ERR_STATE
    while pending:
        retire()
    for i in range(N):
        assert np.array_equal(got[i], seq[i]), i
    assert tracks_key(trk.finish()) == tracks_key(ref.finish())
    assert lib.fdt_model_wait(res50._h, 0, None, None, None) == L.FDT_ERR_ARG      # already retired
    trk.close()


def test_async_ingest_release_without_host_wait(res50, synth):
    """fdt_model_release: tickets retired with no host wait (the record is consumed on the device by the tracker only); the
    host runs ahead of the GPU, slots are re-issued while their previous forward may still be queued, and the tracks still
    equal the oracle's on the synchronous records.  Two handles, so four tickets are in flight."""
    L = M("_lib")
    lib = L.lib()
    H, W, N = 128, 160, 40
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames = moving_frames(synth, N, H, W, seed=11)
    seq = [res50(f).numpy() for f in frames]
    ref = opp.IouTracker(0.4, 0.6, 3)
    for y in seq:
        with np.errstate(all="ignore"):
            ref.step(opp.unpack_detections(y, W, H, 0.4))
    nets = [res50, res50.clone()]
    trk = M("tracker").IouTracker(0.4, 0.6, 3, max_dets=1500, log_frames=64)
    st = torch.cuda.Stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    pending = []

    def retire():
        k, t = pending.pop(0)
        rec = ctypes.c_void_p(0)
        L.check(lib.fdt_model_async_record(nets[k]._h, t, ctypes.byref(rec), sp))
        trk.step_dev(rec, 2, 750, W, H, 0.4, sp)
        L.check(lib.fdt_model_release(nets[k]._h, t, sp))

    for i in range(N):
        if len(pending) == 4:
            retire()
        t = ctypes.c_int(-1)
        buf = frames[i].copy()
        L.check(lib.fdt_model_forward_async(nets[i % 2]._h, L.ptr(buf), L.FRAME_U8_HWC_BGR, 1, H, W, 0, 0, ctypes.byref(t)))
        buf[:] = 0
        pending.append((i % 2, t.value))
    while pending:
        retire()
    assert tracks_key(trk.finish()) == tracks_key(ref.finish())
    assert lib.fdt_model_release(res50._h, 0, None) == L.FDT_ERR_ARG                 # already retired
    # a released slot can be waited on again after its next use (mixing the two retirements is allowed)
    t = ctypes.c_int(-1)
    L.check(lib.fdt_model_forward_async(res50._h, L.ptr(frames[0]), L.FRAME_U8_HWC_BGR, 1, H, W, 0, 0, ctypes.byref(t)))
    out = np.empty((1, 2, 750, 5), np.float32)
    L.check(lib.fdt_model_wait(res50._h, t.value, L.ptr(out), None, None))
    assert np.array_equal(out, seq[0])
    trk.close()
    nets[1].close()


def test_async_ingest_with_device_resize(res50, synth):
    L = M("_lib")
    lib = L.lib()
    H, W, SH, SW = 96, 128, 270, 480
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    src = synth.make_frames(2, SH, SW, seed=21)
    want = [res50.forward_resized(f, (W, H)).numpy() for f in src]
    for i in range(2):
        t = ctypes.c_int(-1)
        L.check(lib.fdt_model_forward_async(res50._h, L.ptr(src[i]), L.FRAME_U8_HWC_BGR, 1, H, W, SH, SW, ctypes.byref(t)))
        out = np.empty((1, 2, 750, 5), np.float32)
        L.check(lib.fdt_model_wait(res50._h, t.value, L.ptr(out), None, None))
        assert np.array_equal(out, want[i])


# ------------------------------------------------------------------ N > 1 rehearsal: two ranks share the one GPU
def _rank_worker(rank, world, port, H, W, n_steps, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    synth = importlib.import_module("face-detection-and-tracking_amd.synth")
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(synth.make_state_dict("res50", seed=0))
    net.priorbox = M("layers").PriorBoxLayer(W, H)
    net.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    dev = torch.device("cuda", 0)
    frames = moving_frames(synth, n_steps * world, H, W, seed=5)
    pipe = M("pipeline").DetectTrackPipeline(net, H, W, dev, inflight=2, world=world, rank=rank, log_frames=8)
    mine = torch.from_numpy(np.ascontiguousarray(frames[rank::world])).to(dev)
    for s in range(n_steps):
        pipe.step(s, mine[s:s + 1])
    q.put((rank, tracks_key(pipe.finish())))
    dist.barrier()
    pipe.close()
    net.close()
    dist.destroy_process_group()


def test_two_ranks_frame_parallel_device_tracker(res50, synth):
    """The N > 1 path with the PRODUCT tracker: frame f = step * 2 + rank, host (gloo) all-gather of the Detect records,
    fdt_tracker_step_dev_multi over the gathered records in rank order -- every rank ends with the track list of a single
    process that sees the frames in order."""
    import os
    import torch.multiprocessing as mp
    H, W, STEPS, WORLD = 128, 160, 13, 2
    frames = moving_frames(synth, STEPS * WORLD, H, W, seed=5)
    res50.priorbox = M("layers").PriorBoxLayer(W, H); res50.firstTime = True
    res50.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    ref = opp.IouTracker(0.4, 0.6, 5)
    for f in frames:
        with np.errstate(all="ignore"):
            ref.step(opp.unpack_detections(res50(f).numpy(), W, H, 0.4))
    want = tracks_key(ref.finish())
    assert len(want) >= 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_worker, args=(r, WORLD, port, H, W, STEPS, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, got in res:
        assert got == want, rank
