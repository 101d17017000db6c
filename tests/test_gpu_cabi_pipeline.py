"""The timed path without Python in the process: a plain-C program (C99, links libfdt_hip.so and nothing else) loads the
detector's weights and a few frames from flat files, builds an fdt_pipeline with two frames in flight, runs detect + track
step by step and prints the tracks; the test compares them with the sequential Python path + the CPU oracle tracker
(reference iouTracke_cal.py:117-156,174-177), bit for bit.  A second test drives the same object from Python (CabiPipeline)
against DetectTrackPipeline, incl. the one-frame-at-a-time grouped hand-over."""
import importlib
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from oracle import postproc as opp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "face-detection-and-tracking_amd", "csrc")


def M(name):
    return importlib.import_module("face-detection-and-tracking_amd." + name)


def tracks_key(tracks):
    return [(t["start_frame"], float(t["max_score"]), [list(map(float, b)) for b in t["bboxes"]]) for t in tracks]


def moving_frames(synth, n, H, W, seed):
    base = synth.make_frames(1, H, W, seed=seed)[0]
    out = np.repeat(base[None], n, 0).copy()
    for i in range(n):
        out[i, 8 + i:40 + i, 16 + 2 * i:64 + 2 * i] = 255 - out[i, 8 + i:40 + i, 16 + 2 * i:64 + 2 * i]
    return out


C_DRIVER = r"""
/* detect + track from plain C: weights file = { int32 name_len, name, int32 ndim, int64 dims[ndim], float data[] }* */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fdt.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != FDT_OK) { fprintf(stderr, "%s failed: %d %s\n", #x, rc_, fdt_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const char* wfile = argv[1];
  const char* ffile = argv[2];
  const int n_frames = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]), inflight = atoi(argv[6]);
  const int host_mode = argc > 7 && atoi(argv[7]);   /* 1: frames stay in host memory, fdt_pipeline_step_host uploads them */
  fdt_model* m = fdt_model_create(FDT_ARCH_RES50, 0);
  if (!m) { fprintf(stderr, "create: %s\n", fdt_last_error()); return 1; }
  FILE* f = fopen(wfile, "rb");
  if (!f) return 3;
  for (;;) {
    int nl, nd, i;
    long long dims[8], cnt = 1;
    char name[256];
    float* data;
    if (fread(&nl, 4, 1, f) != 1) break;
    if (nl <= 0 || nl >= 256 || fread(name, 1, (size_t)nl, f) != (size_t)nl) return 4;
    name[nl] = 0;
    if (fread(&nd, 4, 1, f) != 1 || nd < 0 || nd > 8) return 4;
    for (i = 0; i < nd; ++i) { if (fread(&dims[i], 8, 1, f) != 1) return 4; cnt *= dims[i]; }
    data = (float*)malloc((size_t)cnt * 4);
    if (fread(data, 4, (size_t)cnt, f) != (size_t)cnt) return 4;
    { int rc = fdt_model_set_tensor(m, name, data, nd, dims); if (rc != FDT_OK && rc != FDT_ERR_NAME) { fprintf(stderr, "set_tensor %s: %s\n", name, fdt_last_error()); return 1; } }
    free(data);
  }
  fclose(f);
  CHECK(fdt_model_finalize(m));
  { const int stride[6] = {4, 8, 16, 32, 64, 128}, box[6] = {16, 32, 64, 128, 256, 512};
    CHECK(fdt_model_set_priorbox(m, W, H, 6, stride, box)); }
  CHECK(fdt_model_set_detect(m, 750, 0.05f, 0.35f, 5000));
  /* frames: n_frames x H x W x 3 u8, uploaded once */
  const long long fb = (long long)H * W * 3;
  unsigned char* host = (unsigned char*)malloc((size_t)(fb * n_frames));
  f = fopen(ffile, "rb");
  if (!f || fread(host, 1, (size_t)(fb * n_frames), f) != (size_t)(fb * n_frames)) return 5;
  fclose(f);
  void* dev = NULL;
  if (!host_mode) {
    CHECK(fdt_dev_malloc(&dev, fb * n_frames));
    CHECK(fdt_dev_upload(dev, host, fb * n_frames));
  }
  fdt_pipeline* p = fdt_pipeline_create(m, 0, H, W, inflight, 1, NULL, NULL, 0, 1, 0, 0, 0.4f, 0.4, 0.6, 5, 8);
  if (!p) { fprintf(stderr, "pipeline: %s\n", fdt_last_error()); return 1; }
  if (!host_mode) CHECK(fdt_pipeline_prime(p, dev));
  CHECK(fdt_pipeline_mark(p, 0));
  { int i;
    for (i = 0; i < n_frames; ++i) {
      if (host_mode) CHECK(fdt_pipeline_step_host(p, i, host + fb * i, 1));
      else CHECK(fdt_pipeline_step(p, i, (unsigned char*)dev + fb * i));
    } }
  CHECK(fdt_pipeline_mark(p, 1));
  CHECK(fdt_pipeline_sync(p));
  { float ms = 0; CHECK(fdt_pipeline_elapsed_ms(p, &ms)); fprintf(stderr, "%d frames, %d in flight: %.3f ms\n", n_frames, inflight, ms); }
  fdt_tracker* t = fdt_pipeline_tracker(p);
  CHECK(fdt_tracker_finish(t));
  { int n = 0, i, j;
    CHECK(fdt_tracker_num_tracks(t, &n));
    printf("tracks %d\n", n);
    for (i = 0; i < n; ++i) {
      int nb = 0, sf = 0; double ms = 0; double* boxes;
      CHECK(fdt_tracker_track_info(t, i, &nb, &ms, &sf));
      boxes = (double*)malloc((size_t)nb * 32);
      CHECK(fdt_tracker_track_boxes(t, i, boxes));
      printf("track %d %d %.17g", sf, nb, ms);
      for (j = 0; j < nb * 4; ++j) printf(" %.17g", boxes[j]);
      printf("\n");
      free(boxes);
    } }
  fdt_pipeline_destroy(p);
  if (dev) CHECK(fdt_dev_free(dev));
  fdt_model_destroy(m);
  free(host);
  printf("ok\n");
  return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_plain_c_program_runs_two_frames_in_flight_detect_and_track(tmp_path, res50_sd, synth):
    H, W, N = 128, 160, 14
    frames = moving_frames(synth, N, H, W, seed=5)
    # reference: sequential Python path + the oracle tracker
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(res50_sd)
    net.priorbox = M("layers").PriorBoxLayer(W, H)
    net.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    ref = opp.IouTracker(0.4, 0.6, 5)
    for f in frames:
        with np.errstate(all="ignore"):
            ref.step(opp.unpack_detections(net(f).numpy(), W, H, 0.4))
    want = tracks_key(ref.finish())
    net.close()
    assert len(want) >= 1
    wfile, ffile = tmp_path / "weights.bin", tmp_path / "frames.bin"
    with open(wfile, "wb") as f:
        for k, v in res50_sd.items():
            a = np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v), dtype=np.float32)
            kb = k.encode()
            f.write(struct.pack("<i", len(kb)) + kb + struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())
    frames.tofile(ffile)
    src, exe = tmp_path / "drive.c", tmp_path / "drive"
    src.write_text(C_DRIVER)
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", INC, str(src), "-o", str(exe), "-L", LIBDIR, "-lfdt_hip",
                    "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    for host_mode in ("0", "1"):      # frames uploaded once (fdt_dev_*) / frames in host memory (fdt_pipeline_step_host)
        r = subprocess.run([str(exe), str(wfile), str(ffile), str(N), str(H), str(W), "2", host_mode], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-2000:])
        got = []
        for ln in r.stdout.splitlines():
            p = ln.split()
            if p and p[0] == "track":
                sf, nb, ms = int(p[1]), int(p[2]), float(p[3])
                v = [float(x) for x in p[4:]]
                got.append((sf, ms, [v[4 * i:4 * i + 4] for i in range(nb)]))
        assert got == want, host_mode


@pytest.mark.parametrize("batch,inflight", [(1, 3), (2, 2)])
def test_cabi_pipeline_equals_the_python_pipeline(res50_sd, synth, batch, inflight):
    H, W, N = 128, 160, 18
    dev = torch.device("cuda", 0)
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(res50_sd)
    net.priorbox = M("layers").PriorBoxLayer(W, H)
    net.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    frames = moving_frames(synth, N, H, W, seed=7)
    fd = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()
    py = M("pipeline").DetectTrackPipeline(net, H, W, dev, inflight=inflight, batch=batch, log_frames=8)
    for i in range(N // batch):
        py.step(i, fd[i * batch:(i + 1) * batch])
    want = tracks_key(py.finish())
    rec_want = py.record_of_slot((N // batch - 1) % inflight).copy()
    py.close()
    c = M("pipeline").CabiPipeline(net, H, W, 0, inflight=inflight, batch=batch, log_frames=8)
    c.prime(fd[:batch])
    for i in range(N // batch):
        c.step(i, fd[i * batch:(i + 1) * batch])
    rec_got = c.record_of_slot((N // batch - 1) % inflight)
    got = tracks_key(c.finish())
    c.close()
    assert np.array_equal(rec_got, rec_want)
    assert got == want and len(want) >= 1
    # one frame at a time, `batch` per launch (+ a partly filled last batch)
    c = M("pipeline").CabiPipeline(net, H, W, 0, inflight=inflight, batch=batch, log_frames=8)
    for i in range(N - 1):
        c.step_frame(i, fd[i:i + 1])
    got2 = tracks_key(c.finish())
    c.close()
    py = M("pipeline").DetectTrackPipeline(net, H, W, dev, inflight=inflight, batch=batch, log_frames=8)
    for i in range(N - 1):
        py.step_frame(i, fd[i:i + 1])
    want2 = tracks_key(py.finish())
    py.close()
    assert got2 == want2
    net.close()


# ------------------------------------------------------------------ world > 1 behind the C ABI (loop-back communicator)
def _run_ranks(world, fn):
    """fn(rank) on one host thread per rank (ctypes releases the GIL inside the library); re-raises the first failure."""
    import threading
    out, err = [None] * world, []

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:      # noqa: BLE001 -- handed to the main thread
            err.append((r, e))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not any(t.is_alive() for t in th), "a rank thread did not finish"
    if err:
        raise err[0][1]
    return out


def _local_id():
    import ctypes
    lib = M("_lib")
    buf = ctypes.create_string_buffer(128)
    lib.check(lib.lib().fdt_comm_unique_id_local(buf))
    return buf


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode,batch,inflight", [("step", 1, 3), ("step", 2, 2), ("step_frame", 2, 2), ("step_frame", 4, 3)])
def test_cabi_pipeline_at_world_gt_1_through_the_loopback_communicator(res50_sd, synth, world, mode, batch, inflight):
    """fdt_pipeline_create(comm, rank, world > 1): `world` ranks = host threads sharing the one GPU, the exchange through
    fdt_allgather_dets on a loop-back communicator (fdt_comm_unique_id_local + fdt_comm_init_rank) -- the gathered-record
    indexing of track_slot (rank-major for step, (group, entry, rank) for step_frame incl. a partly filled last group), the
    collective on the tracker stream beside the detector streams, the event chain.  Every rank must end with the track list of
    ONE process that saw the same frames in video order (bit-equal), which is itself the oracle tracker's list on those
    detections (reference iouTracke_cal.py:117-156,174-177)."""
    lib = M("_lib")
    L = lib.lib()
    H, W = 128, 160
    per_rank = 9 if mode == "step_frame" else 4             # step_frame: 9 frames per rank = a partly filled last group
    n_items = per_rank if mode == "step_frame" else per_rank * batch
    N = n_items * world
    frames = moving_frames(synth, N, H, W, seed=11)
    dev = torch.device("cuda", 0)
    fd = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(res50_sd)
    net.priorbox = M("layers").PriorBoxLayer(W, H)
    net.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)

    # one process, video order, same batch size (the same kernel plan): the list every rank must reproduce
    one = M("pipeline").CabiPipeline(net, H, W, 0, inflight=inflight, batch=batch, log_frames=8)
    one.prime(fd[:batch])
    if mode == "step":
        for i in range(N // batch):
            one.step(i, fd[i * batch:(i + 1) * batch])
    else:
        for i in range(N):
            one.step_frame(i, fd[i:i + 1])
    recs = None
    want = tracks_key(one.finish())
    one.close()
    assert len(want) >= 1
    # ... which is the oracle tracker's list on the detections of the sequential path (batch 1 only: same plan, same bits)
    if batch == 1:
        ref = opp.IouTracker(0.4, 0.6, 5)
        for f in frames:
            with np.errstate(all="ignore"):
                ref.step(opp.unpack_detections(net(f).numpy(), W, H, 0.4))
        assert tracks_key(ref.finish()) == want

    nets = [net.clone() for _ in range(world)]
    uid = _local_id()

    def rank_main(r):
        comm = L.fdt_comm_init_rank(world, r, uid, 0)
        assert comm, (L.fdt_last_error() or b"").decode()
        try:
            p = M("pipeline").CabiPipeline(nets[r], H, W, 0, inflight=inflight, batch=batch, comm=comm, world=world, rank=r,
                                           log_frames=8)
            try:
                p.prime(fd[:batch])
                if mode == "step":
                    for s in range(per_rank):
                        o = (s * world + r) * batch                      # rank r's batch of step s: consecutive frames
                        p.step(s, fd[o:o + batch])
                else:
                    for i in range(per_rank):
                        p.step_frame(i, fd[i * world + r:i * world + r + 1])     # frame i of rank r = frame i * world + r
                return tracks_key(p.finish())
            finally:
                p.close()
        finally:
            L.fdt_comm_destroy(comm)

    got = _run_ranks(world, rank_main)
    for c in nets:
        c.close()
    net.close()
    for r in range(world):
        assert got[r] == want, "rank %d of %d" % (r, world)


def test_step_frame_refuses_to_continue_a_flushed_group_and_step_host_a_partial_batch_at_world_2(res50_sd, synth):
    lib = M("_lib")
    L = lib.lib()
    H, W = 128, 160
    frames = moving_frames(synth, 8, H, W, seed=3)
    fd = torch.from_numpy(frames).to(torch.device("cuda", 0))
    torch.cuda.synchronize()
    net = M("pyramid").build_sfd('test', 640, 2)
    net.load_state_dict(res50_sd)
    net.priorbox = M("layers").PriorBoxLayer(W, H)
    net.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    for cls in ("CabiPipeline", "DetectTrackPipeline"):
        p = (M("pipeline").CabiPipeline(net, H, W, 0, inflight=2, batch=4, log_frames=8) if cls == "CabiPipeline" else
             M("pipeline").DetectTrackPipeline(net, H, W, torch.device("cuda", 0), inflight=2, batch=4, log_frames=8))
        p.step_frame(0, fd[0:1])
        p.step_frame(1, fd[1:2])
        with pytest.raises(lib.FdtError) as e:
            p.step_frame(3, fd[3:4])                          # skips entry 2
        assert e.value.code == lib.FDT_ERR_STATE
        with pytest.raises(lib.FdtError):
            p.step_frame(4, fd[4:5])                          # opens group 1 while group 0 is open
        p.flush()                                             # group 0 runs with two frames and is closed
        with pytest.raises(lib.FdtError) as e:
            p.step_frame(2, fd[2:3])                          # would continue the flushed group
        assert e.value.code == lib.FDT_ERR_STATE
        p.step_frame(4, fd[4:5])                              # the next group is fine
        assert len(p.finish()) >= 0
        p.close()
    # step_host: a partly filled batch is defined at world 1 only
    nets = [net.clone() for _ in range(2)]
    uid = _local_id()

    def rank_main(r):
        comm = L.fdt_comm_init_rank(2, r, uid, 0)
        assert comm
        p = M("pipeline").CabiPipeline(nets[r], H, W, 0, inflight=2, batch=2, comm=comm, world=2, rank=r, log_frames=8)
        with pytest.raises(lib.FdtError) as e:
            p.step_host(0, frames[0:2], n_valid=1)
        code = e.value.code
        p.step_host(0, frames[2 * r:2 * r + 2])               # a full batch is fine (collective: both ranks call it)
        n = len(p.finish())
        p.close()
        L.fdt_comm_destroy(comm)
        return code, n

    out = _run_ranks(2, rank_main)
    assert [o[0] for o in out] == [lib.FDT_ERR_ARG] * 2 and out[0][1] == out[1][1]
    for c in nets:
        c.close()
    net.close()
