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
