"""Offline multi-face IoU tracker entry point, mirroring reference iouTracke_cal.py.

Same module-level configuration names (reference iouTracke_cal.py:22-31), same `detect_face(x, shrink)`
contract (:36-84), same per-frame association and finalisation (:126-156, :174-177) and the same
`.npy` track schema (:150-154,:177) -- but detection, the host unpack and the association run on the
MI355X (`fdt_model_forward*`, `fdt_tracker_step*`).  Video decoding / display are not part of the path
(cv2 is not a dependency): frames come from any iterable of uint8 BGR HWC arrays, either already at
network resolution or -- `track(frames, size=(640, 480))` -- as raw source frames that the GPU resizes
like the reference's `cv2.resize(image, (640, 480))` (:123).

`track()` is the PIPELINED path bench.py times as `host_path`: `inflight` detector handles that share one weight copy
(fdt_model_clone), host frames copied into a pinned landing buffer and uploaded in front of their forward, the Detect
record handed to the tracker on the device, no host wait per frame.  Two engines: "pipeline" (default; one
`fdt_pipeline_step_host` call per batch, the library owns the slots) and "async" (the ticket interface
`fdt_model_forward_async` / `fdt_model_async_record` / `fdt_model_release`, two tickets per handle, stepped from here).  The tracks are bit-identical to the synchronous one-handle path
(`pipelined=False`) and to the oracle tracker (tests/test_gpu_entry.py).

    python -m face-detection-and-tracking_amd.iouTracke_cal   (needs frames in `video_file + '.frames.npy'`)
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .layers import PriorBoxLayer
from .pyramid import build_sfd
from .pyramid_mb2_try3 import build_sfd_mobile as build_sfd_mobile_try3
from .tracker import IouTracker
from .utils.calc_performance import calculate_distance, calculate_iou  # noqa: F401  (re-exported like the reference, :18)

# <<<<<<<<<<<<<<<<<<<<<<<<<<parameter configer>>>>>>>>>>>>>>>>>>>>>>>>>>>>   (reference :22-31)
use_iou = True
sigma_iou = 0.4
sigma_dis = 8
sigma_h = 0.6
t_min = 5
display_result = False
use_net = 'repo'
video_file = './image_and_anno/video/video8'
#  <<<<<<<<<<<<<<<<<<<<<<<end of config parameter>>>>>>>>>>>>>>>>>>>>>>>>>>
DEFAULT_ENGINE = "pipeline"      # track(engine=None): see track()

net = None
_clones = {}        # id(net) -> (net, [fdt_model_clone handles]) for the frames in flight of track()


def load_net(weights, width=640, height=480, which=None):
    """reference :93-107: build, load weights, override the priorbox with the frame size."""
    global net
    which = which or use_net
    if which == 'repo':
        net = build_sfd('test', 640, 2)
        net.priorbox = PriorBoxLayer(width, height)
    elif which == 'try3':
        net = build_sfd_mobile_try3('test', 640, 2)
        net.priorbox = PriorBoxLayer(width, height, stride=[4, 8, 16, 32, 64], box=(16, 32, 64, 128, 256))
    else:
        raise ValueError("unknown use_net %r" % (which,))     # the reference does a bare raise() (:105)
    if isinstance(weights, str):
        weights = torch.load(weights, map_location='cpu', weights_only=True)
    net.load_state_dict(weights)
    net.cuda()
    net.eval()
    return net


def _cv_round(v):
    """cvRound: round half to even (what saturate_cast<int>(double) does for the dsize of cv2.resize(fx=, fy=))."""
    return int(np.rint(v))


def detect_face(x, shrink=1):
    """BGR uint8 HWC frame -> ndarray [n,5] (x1,y1,x2,y2,score) in pixels; reference :36-84.  `shrink != 1` resizes the
    frame first (:37-38, cv2.resize(x, None, None, fx=shrink, fy=shrink, INTER_LINEAR)) -- here on the GPU, fused with the
    mean subtraction (fdt_model_forward_resized), to cv2's output size (cvRound(w * shrink), cvRound(h * shrink)); the
    sampling scale is source / output size, which equals cv2's 1 / shrink whenever the scaled size is integral."""
    x = np.ascontiguousarray(x, dtype=np.uint8)
    if shrink != 1:
        h0, w0, _ = x.shape
        width, height = _cv_round(w0 * shrink), _cv_round(h0 * shrink)
        y = net.forward_resized(x, (width, height))
    else:
        height, width, _ = x.shape
        y = net(x)                                       # mean subtraction happens on the GPU
    detections = y.numpy()
    scale = np.array([width, height, width, height], dtype=np.float32)
    boxes, scores = [], []
    for i in range(detections.shape[1]):
        j = 0
        while detections[0, i, j, 0] >= np.float32(0.4):
            scores.append(detections[0, i, j, 0])
            boxes.append(detections[0, i, j, 1:] * scale)
            j += 1
            if j >= detections.shape[2]:
                break
    if len(boxes) == 0:
        return np.array([[0, 0, 0, 0, 0.4]])
    b = np.array(boxes, dtype=np.float32) / shrink
    det = np.column_stack((b[:, 0], b[:, 1], b[:, 2], b[:, 3], np.array(scores, dtype=np.float32)))
    return det[np.where(det[:, 4] >= 0)[0], :]


class _DistanceTracker:
    """The association loop of reference :126-156 with `use_iou = False` (:136-138): the measure is
    calculate_distance (utils/calc_performance.py:34-51, `fdt_pairwise_distance` on the GPU), the best match is the
    argmin and it counts when `dis < sigma_dis`.  One [detections x tracks] distance matrix per frame instead of one
    call per track: a track's column does not change when other detections are deleted, only the candidate rows do."""

    def __init__(self, sigma_dis, sigma_h, t_min):
        self.sigma_dis, self.sigma_h, self.t_min = sigma_dis, sigma_h, t_min
        self.frame_num = 0
        self.tracks_active, self.tracks_finished = [], []

    def step(self, det0):
        self.frame_num += 1
        dets = np.asarray(det0).tolist()
        updated = []
        if self.tracks_active and dets:
            last = np.array([t['bboxes'][-1] for t in self.tracks_active], dtype=np.float64)
            dis = calculate_distance(np.array(dets)[:, :4], last)        # [n dets, n tracks], f64 like the reference
        alive = list(range(len(dets)))            # original row index of each detection still in `dets`
        for ti, track in enumerate(self.tracks_active):
            if len(dets) > 0:
                col = dis[alive, ti]
                best_match = int(col.argmin())                           # :137 (first minimum; NaN wins, like numpy)
                if col[best_match] < self.sigma_dis:                     # :138
                    track['bboxes'].append(dets[best_match][:4])
                    track['max_score'] = max(track['max_score'], dets[best_match][4])
                    updated.append(track)
                    del dets[best_match]
                    del alive[best_match]
                elif track['max_score'] > self.sigma_h and len(track['bboxes']) > self.t_min:
                    self.tracks_finished.append(track)
        new = [{'bboxes': [d[:4]], 'max_score': d[4], 'start_frame': self.frame_num} for d in dets]
        self.tracks_active = updated + new

    def finish(self):
        self.tracks_finished += [t for t in self.tracks_active
                                 if t['max_score'] > self.sigma_h and len(t['bboxes']) >= self.t_min]
        return [{'bboxes': [list(map(float, b)) for b in t['bboxes']], 'max_score': float(t['max_score']),
                 'start_frame': t['start_frame']} for t in self.tracks_finished]


def _handles(n):
    """`n` detector handles for the frames in flight: the module's net + cached fdt_model_clone handles (shared weights)."""
    ent = _clones.get(id(net))
    if ent is None or ent[0] is not net:
        for _, cl in _clones.values():
            for c in cl:
                c.close()
        _clones.clear()
        ent = _clones[id(net)] = (net, [])
    while len(ent[1]) < n - 1:
        ent[1].append(net.clone())
    return [net] + ent[1][:n - 1]


def track(frames, device_resident=True, pipelined=True, inflight=None, size=None, batch=1, engine=None):
    """Run detect + association over an iterable of uint8 BGR HWC frames; returns `tracks_finished`
    (reference :113-156 + :174-175).

    size=(W, H): the frames are raw source frames, resized on the GPU to the network input like
                 `cv2.resize(image, (640, 480))` (:123); None: frames are already at network resolution.
    device_resident (default): the Detect output never leaves the GPU between detection and association (the host
                 unpack of :53-84 runs inside the tracker kernel).  False: the reference's own host flow,
                 `detect_face(image, 1)` + a host-stepped association.
    pipelined (default, needs device_resident): `inflight` handles in flight (None: 4 / 3 by engine), no host wait per frame;
                 `batch` consecutive frames per forward (the association still sees them one by one, in order).
                 False: one handle, one stream, one frame at a time.
    engine (pipelined only): "async" -- the ticket interface, fdt_model_forward_async / _async_record / _release, stepped from
                 here; "pipeline" -- one fdt_pipeline_step_host call per batch, the library owns slots, streams, the pinned
                 landing buffers and the tracker (kept between calls for the same geometry).  Same tracks, bit for bit.
    `use_iou = False` selects the distance measure of :136-138 (host-stepped; the detection still runs on the GPU)."""
    if not use_iou:
        tr = _DistanceTracker(sigma_dis, sigma_h, t_min)
        for image in frames:
            if size is not None:
                raise ValueError("track(size=...) needs use_iou = True (the device-resident path)")
            tr.step(detect_face(image, 1))
        return tr.finish()
    if not device_resident:
        tracker = IouTracker(sigma_iou, sigma_h, t_min)
        for image in frames:
            tracker.step(detect_face(image, 1))
        out = tracker.finish()
        tracker.close()
        return out
    if pipelined:
        engine = engine or DEFAULT_ENGINE
        if inflight is None:
            inflight = 4 if engine == "pipeline" else 3        # measured optima (bench.py host_path, docs/EXPERIMENTS.md R4-11)
        if engine == "pipeline":
            return _track_cabi(frames, max(1, int(inflight)), size, max(1, int(batch)))
        if engine != "async":
            raise ValueError("track(engine=...): 'async' or 'pipeline'")
        return _track_pipelined(frames, max(1, int(inflight)), size, max(1, int(batch)))

    tracker = IouTracker(sigma_iou, sigma_h, t_min)
    L = _lib.lib()
    out_dev = None
    for image in frames:
        image = np.ascontiguousarray(image, dtype=np.uint8)
        SH, SW, _ = image.shape
        W, H = (int(size[0]), int(size[1])) if size is not None else (SW, SH)
        net._sync_attributes(H, W)
        top_k = net.detect.top_k
        if out_dev is None:
            out_dev = torch.empty((2, top_k, 5), dtype=torch.float32, device="cuda")
            frame_dev = torch.empty(image.shape, dtype=torch.uint8, device="cuda")
            stream = torch.cuda.Stream()
            sp = ctypes.c_void_p(stream.cuda_stream)
        with torch.cuda.stream(stream):
            frame_dev.copy_(torch.from_numpy(image), non_blocking=False)
            if size is not None:
                _lib.check(L.fdt_model_forward_resized(net._h, ctypes.c_void_p(frame_dev.data_ptr()), 1, 1, SH, SW, H, W,
                                                       ctypes.c_void_p(out_dev.data_ptr()), None, sp))
            else:
                _lib.check(L.fdt_model_forward_dev(net._h, ctypes.c_void_p(frame_dev.data_ptr()),
                                                   _lib.FRAME_U8_HWC_BGR, 1, H, W,
                                                   ctypes.c_void_p(out_dev.data_ptr()), None, sp))
            tracker.step_dev(ctypes.c_void_p(out_dev.data_ptr()), 2, top_k, W, H, 0.4, sp)
    out = tracker.finish()
    tracker.close()
    return out


def _track_pipelined(frames, inflight, size, batch):
    L = _lib.lib()
    nets = None
    tracker = None
    stream = torch.cuda.Stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    pending = []                       # (handle index, ticket, frames in the ticket) in frame order
    shape = None
    state = {"i": 0}

    def retire():
        k, t, nfr = pending.pop(0)
        rec = ctypes.c_void_p(0)
        _lib.check(L.fdt_model_async_record(nets[k]._h, t, ctypes.byref(rec), sp))    # the tracker stream waits on the device
        if nfr == 1:
            tracker.step_dev(rec, 2, top_k, W, H, 0.4, sp)
        else:
            tracker.step_dev_multi(rec, nfr, 2 * top_k * 5, 2, top_k, W, H, 0.4, sp)
        _lib.check(L.fdt_model_release(nets[k]._h, t, sp))     # slot reusable once the tracker has read it: no host wait

    def issue(block, nfr):
        k = state["i"] % inflight
        state["i"] += 1
        if len(pending) >= 2 * inflight:
            retire()
        t = ctypes.c_int(0)
        _lib.check(L.fdt_model_forward_async(nets[k]._h, _lib.ptr(block), _lib.FRAME_U8_HWC_BGR, block.shape[0], H, W,
                                             SH if size is not None else 0, SW if size is not None else 0,
                                             ctypes.byref(t)))
        pending.append((k, t.value, nfr))

    group = []
    try:
        for image in frames:
            image = np.ascontiguousarray(image, dtype=np.uint8)
            if shape is None:
                shape = image.shape
                SH, SW, _ = shape
                W, H = (int(size[0]), int(size[1])) if size is not None else (SW, SH)
                nets = _handles(inflight)
                plan = net.tuned_plan_text(H, W, batch)
                for n in nets:
                    if getattr(n, "_plan_key", None) != (H, W, batch) and plan:
                        n.import_plan(plan)
                    n._plan_key = (H, W, batch)
                    if n is not net:
                        n.priorbox, n.detect = net.priorbox, net.detect
                        n.firstTime = True
                    n._sync_attributes(H, W)
                top_k = net.detect.top_k
                tracker = IouTracker(sigma_iou, sigma_h, t_min, max_dets=2 * top_k, log_frames=max(256, batch))
            elif image.shape != shape:
                raise ValueError("track(): frame shape changed from %s to %s" % (shape, image.shape))
            if batch == 1:
                issue(image[None], 1)
                continue
            group.append(image)
            if len(group) == batch:
                issue(np.stack(group), batch)
                group = []
        if group:
            # a last, partial batch keeps the batch-`batch` plan of the handles: pad with copies of its last frame and hand
            # only the real frames' records to the tracker
            nfr = len(group)
            issue(np.stack(group + [group[-1]] * (batch - nfr)), nfr)
        while pending:
            retire()
        if tracker is None:
            return []
        torch.cuda.synchronize()
        return tracker.finish()
    finally:
        for k, t, _ in pending:          # an exception mid-sequence: retire what is in flight so the handles stay usable
            L.fdt_model_wait(nets[k]._h, t, None, None, None)
        if tracker is not None:
            torch.cuda.synchronize()
            tracker.close()


_pipes = {}       # (id(net), H, W, SH, SW, inflight, batch) -> CabiPipeline, kept between track() calls


def _track_cabi(frames, inflight, size, batch):
    """track() on fdt_pipeline_step_host: the per-frame host work is one C call (memcpy into the slot's pinned buffer, H2D,
    forward graph, tracker launch, all enqueued)."""
    from .pipeline import CabiPipeline
    pipe = None
    shape = None
    group = []
    i = 0
    try:
        for image in frames:
            image = np.ascontiguousarray(image, dtype=np.uint8)
            if shape is None:
                shape = image.shape
                SH, SW, _ = shape
                W, H = (int(size[0]), int(size[1])) if size is not None else (SW, SH)
                key = (id(net), H, W, SH, SW, inflight, batch, sigma_iou, sigma_h, t_min)
                ent = _pipes.get(key)
                if ent is None or ent[0] is not net:
                    for _, old in _pipes.values():
                        old.close()
                    _pipes.clear()
                    pipe = CabiPipeline(net, H, W, torch.cuda.current_device(), inflight, batch,
                                        source_hw=(SH, SW) if size is not None else None,
                                        sigma_iou=sigma_iou, sigma_h=sigma_h, t_min=t_min, log_frames=max(256, batch),
                                        plan_text=net.tuned_plan_text(H, W, batch))
                    _pipes[key] = (net, pipe)
                else:
                    pipe = ent[1]
                    pipe._trk.reset()
            elif image.shape != shape:
                raise ValueError("track(): frame shape changed from %s to %s" % (shape, image.shape))
            if batch == 1:
                pipe.step_host(i, image)
                i += 1
                continue
            group.append(image)
            if len(group) == batch:
                pipe.step_host(i, np.stack(group))
                i += 1
                group = []
        if group:
            nfr = len(group)      # a last, partial batch: padded with copies of its last frame, only the real ones are tracked
            pipe.step_host(i, np.stack(group + [group[-1]] * (batch - nfr)), nfr)
        if pipe is None:
            return []
        return pipe.finish()
    except BaseException:
        if pipe is not None:
            pipe.sync()
        raise


def save_tracks(tracks, path):
    """reference :177: np.save(video_file + '.npy', np.array(tracks_finished)) -- an object array of
    dicts, which iouTracke_display.py:29 reads back with np.load(...).tolist()."""
    arr = np.empty(len(tracks), dtype=object)
    for i, t in enumerate(tracks):
        arr[i] = t
    np.save(path, arr)
    print("file saved to: " + path)


if __name__ == '__main__':
    frames = np.load(video_file + '.frames.npy')
    print('Loading model..')
    load_net('./net_weight/Res50_pyramid.pth' if use_net == 'repo' else 'net_weight/Mobile_pyramid_try3.pth', 640, 480)
    print('Finished loading model!')
    save_tracks(track(frames, size=(640, 480)), video_file + '.npy')
