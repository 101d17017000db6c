"""The per-GPU detect+track pipeline exactly as it is timed by bench.py and pinned by tests/test_gpu_pipeline.py.

`inflight` frames in flight per GPU: slot k = step % inflight owns a model handle (fdt_model_clone: shared weights,
own activations + captured HIP graph), a HIP stream, its Detect record and candidate-count buffer.  The exchange
(N > 1: one all-gather of the fixed-size records) and the strictly sequential IoU association (reference
iouTracke_cal.py:117-156) run on ONE more stream; HIP events order
    detect(step i, slot k)  ->  exchange + track(step i)  ->  detect(step i + inflight, slot k)
so consecutive frames overlap (the latency-bound small layers and the one-workgroup tracker of one frame run beside
the MFMA-bound layers of the next) while every forward is still batch `batch` and the tracker sees frames in order.
The frames of a step (world x batch records in rank order == frame order) are associated in one launch
(fdt_tracker_step_dev_multi).  Nothing here synchronises with the host; `finish()` does.
"""
import ctypes
import os

import torch

from . import _lib
from .parallel import FrameParallel
from .tracker import IouTracker


class DetectTrackPipeline:
    def __init__(self, net, H, W, device, inflight=3, batch=1, exchange_factory=None, world=1, rank=0,
                 source_hw=None, score_thresh=0.4, sigma_iou=0.4, sigma_h=0.6, t_min=5, log_frames=256,
                 plan_text=None, multi_step=True):
        self.H, self.W, self.B, self.NF = H, W, max(1, batch), max(1, inflight)
        self.world, self.rank, self.dev = world, rank, device
        self.source_hw = source_hw
        self.score_thresh = score_thresh
        self.multi_step = multi_step
        self.nets = [net] + [net.clone() for _ in range(self.NF - 1)]
        for n in self.nets:
            n.firstTime = True
            n._sync_attributes(H, W)
            if plan_text:
                n.import_plan(plan_text)
        self.top_k = net.detect.top_k
        self.REC = 2 * self.top_k * 5                      # one frame's Detect record [2, top_k, 5]
        mk = exchange_factory or (lambda rec: FrameParallel(rank, world, rec, device))
        self.fps = [mk(self.B * self.REC) for _ in range(self.NF)]
        self.counts = [torch.zeros(2 * self.B, dtype=torch.int32, device=device) for _ in range(self.NF)]
        self.tracker = IouTracker(sigma_iou, sigma_h, t_min, max_dets=2 * self.top_k,
                                  log_frames=max(log_frames, world * self.B))
        # non-default torch streams: their handles go through the C ABI, so torch.cuda.Event and the collective are
        # ordered with the library's launches
        # experiment hooks (tools/experiments/*.sh) act only with FDT_EXPERIMENTS=1 in the environment
        exp = (lambda k, d: os.environ.get(k, d)) if os.environ.get("FDT_EXPERIMENTS") == "1" else (lambda k, d: d)
        self._exp_no_track = exp("FDT_EXP_NO_TRACK", "0") == "1"
        parts = int(exp("FDT_CU_PARTS", "1"))      # experiment: detector streams on CU partitions (fdt.h)
        if parts > 1:
            self._raw_streams = []
            for k in range(self.NF):
                sp = ctypes.c_void_p()
                _lib.check(_lib.lib().fdt_stream_create_partition(k % parts, parts, ctypes.byref(sp)))
                self._raw_streams.append(sp)
            self.det_streams = [torch.cuda.ExternalStream(sp.value, device=device) for sp in self._raw_streams]
        else:
            alt = int(exp("FDT_DET_PRIO_ALT", "0"))     # every alt-th detector stream at high priority
            self.det_streams = [torch.cuda.Stream(device=device, priority=(-1 if alt and k % alt == 0 else 0))
                                for k in range(self.NF)]
        # experiment hooks (docs/EXPERIMENTS.md R3-10): a priority of its own moves a stream to another pool of hardware queues
        self.trk_stream = torch.cuda.Stream(device=device, priority=int(exp("FDT_TRK_PRIO", "0")))
        self.sp_det = [ctypes.c_void_p(s.cuda_stream) for s in self.det_streams]
        self.sp_trk = ctypes.c_void_p(self.trk_stream.cuda_stream)
        assert all(p.value for p in self.sp_det) and self.sp_trk.value, "need real stream handles"
        self.det_done = [torch.cuda.Event() for _ in range(self.NF)]
        self.trk_done = [torch.cuda.Event() for _ in range(self.NF)]
        self._L = _lib.lib()

    def _forward(self, k, frames_dev):
        """The forward of slot k as step() enqueues it (same handle, record, counts and stream: the HIP graph a handle
        captures is keyed by exactly these)."""
        L, fp, net = self._L, self.fps[k], self.nets[k]
        if self.source_hw:
            _lib.check(L.fdt_model_forward_resized(net._h, ctypes.c_void_p(frames_dev.data_ptr()), 1, self.B,
                                                   self.source_hw[0], self.source_hw[1], self.H, self.W,
                                                   ctypes.c_void_p(fp.mine.data_ptr()),
                                                   ctypes.c_void_p(self.counts[k].data_ptr()), self.sp_det[k]))
        else:
            _lib.check(L.fdt_model_forward_dev(net._h, ctypes.c_void_p(frames_dev.data_ptr()),
                                               _lib.FRAME_U8_HWC_BGR, self.B, self.H, self.W,
                                               ctypes.c_void_p(fp.mine.data_ptr()),
                                               ctypes.c_void_p(self.counts[k].data_ptr()), self.sp_det[k]))

    def prime(self, frames_dev):
        """One-time initialisation, the equivalent of a compile step: every slot runs its forward twice on `frames_dev`
        (the first builds the plan and uploads the tiled weights, the second is captured into the slot's HIP graph), so that
        no step() -- warm-up or timed -- pays for plan construction or graph capture.  The tracker is not fed."""
        for k in range(self.NF):
            with torch.cuda.stream(self.det_streams[k]):
                self._forward(k, frames_dev)
                self._forward(k, frames_dev)
        torch.cuda.synchronize(self.dev)

    def step(self, i, frames_dev):
        """Enqueue step i: `frames_dev` = torch uint8 tensor [B, h, w, 3] on the device (raw source frames when
        `source_hw` is set: they are resized on the GPU inside the step, iouTracke_cal.py:123)."""
        L, k = self._L, i % self.NF
        fp, net = self.fps[k], self.nets[k]
        st = self.det_streams[k]
        with torch.cuda.stream(st):
            st.wait_event(self.trk_done[k])             # slot k's record was consumed (step i - inflight)
            self._forward(k, frames_dev)
            self.det_done[k].record(st)
        with torch.cuda.stream(self.trk_stream):
            self.trk_stream.wait_event(self.det_done[k])
            # the one exchange step of the path: fixed-size per-frame box lists, rank order == frame order
            g = fp.exchange(self.sp_trk)
            n = self.world * self.B
            if self._exp_no_track:       # experiment: what does the association cost the step?
                pass
            elif self.multi_step:
                self.tracker.step_dev_multi(ctypes.c_void_p(g.data_ptr()), n, self.REC, 2, self.top_k, self.W, self.H,
                                            self.score_thresh, self.sp_trk)
            else:
                for f in range(n):
                    self.tracker.step_dev(ctypes.c_void_p(g.data_ptr() + 4 * f * self.REC), 2, self.top_k, self.W,
                                          self.H, self.score_thresh, self.sp_trk)
            self.trk_done[k].record(self.trk_stream)

    # ---- frames handed over ONE AT A TIME, executed `batch` at a time (cross-frame grouped launches) ----------------------
    # A video source delivers single frames (iouTracke_cal.py:119-124), but at 640x480 a batch-1 forward is a chain of ~110
    # launches of a few hundred workgroups each, and the chip is paid per launch: `batch` consecutive frames of one rank
    # share ONE launch per layer (the handle's plan is the batch-`batch` plan), detection results and tracks are what the
    # one-frame-at-a-time path gives (frames are independent until the tracker, which still sees them in frame order).
    def step_frame(self, i, frame_dev):
        """Frame i of this rank (uint8 [1, h, w, 3] or [h, w, 3] on the device).  The frame is copied into slot
        (i // batch) % inflight's staging batch on that slot's stream; the batch's last frame launches forward + exchange +
        association exactly like step().  flush() runs a partly filled batch (end of the video)."""
        G = self.B
        k, j = (i // G) % self.NF, i % G
        pend = getattr(self, "_pending", None)
        if (pend is not None) if j == 0 else (pend is None or pend[0] != i // G or pend[2] != j):
            # same rule as fdt_pipeline_step_frame: in order from the group's first frame; a flushed group is closed
            raise _lib.FdtError(_lib.FDT_ERR_STATE, "step_frame: frame %d is entry %d of group %d, which is not the open group "
                                                    "at that entry (or another group is still open)" % (i, j, i // G))
        st = self.det_streams[k]
        if getattr(self, "_stage", None) is None:
            h, w = self.source_hw if self.source_hw else (self.H, self.W)
            self._stage = torch.empty((self.NF, G, h, w, 3), dtype=torch.uint8, device=self.dev)
        with torch.cuda.stream(st):
            if j == 0:
                st.wait_event(self.trk_done[k])         # the slot's previous batch was consumed
            self._stage[k, j].copy_(frame_dev.reshape(self._stage.shape[2:]), non_blocking=True)
        self._pending = (i // G, k, j + 1)
        if j == G - 1:
            self._launch_group(i // G, k, G)

    def flush(self):
        """Run the partly filled batch, if any: the unused entries of the staging batch hold older frames whose records
        the tracker is not shown."""
        pend = getattr(self, "_pending", None)
        if pend:
            self._launch_group(*pend)
        self._pending = None

    def _launch_group(self, step, k, n_valid):
        fp = self.fps[k]
        st = self.det_streams[k]
        self._pending = None
        with torch.cuda.stream(st):
            self._forward(k, self._stage[k])
            self.det_done[k].record(st)
        with torch.cuda.stream(self.trk_stream):
            self.trk_stream.wait_event(self.det_done[k])
            g = fp.exchange(self.sp_trk)
            # gathered records are rank-major [world][batch][REC]; frame order is batch-major: frame (j, r) = j * world + r
            for j in range(n_valid):
                if self._exp_no_track:
                    break
                self.tracker.step_dev_multi(ctypes.c_void_p(g.data_ptr() + 4 * j * self.REC), self.world, self.B * self.REC, 2,
                                            self.top_k, self.W, self.H, self.score_thresh, self.sp_trk)
            self.trk_done[k].record(self.trk_stream)

    def record_of_slot(self, k):
        """Host copy of slot k's gathered records [world*B, 2, top_k, 5] (synchronises)."""
        torch.cuda.synchronize(self.dev)
        return self.fps[k].gathered.detach().cpu().numpy().reshape(self.world * self.B, 2, self.top_k, 5)

    def finish(self):
        self.flush()
        torch.cuda.synchronize(self.dev)
        return self.tracker.finish()

    def close(self):
        torch.cuda.synchronize(self.dev)
        for n in self.nets[1:]:
            n.close()
        for fp in self.fps:
            if hasattr(fp, "close"):
                fp.close()
        self.tracker.close()
        for sp in getattr(self, "_raw_streams", []):   # CU-partitioned detector streams (FDT_CU_PARTS): nothing is in flight any more
            _lib.check(_lib.lib().fdt_stream_destroy(sp))
        self._raw_streams = []


class CabiPipeline:
    """The same pipeline behind the C ABI (fdt_pipeline_*, include/fdt.h): libfdt_hip.so owns the in-flight handles, the HIP
    streams and events, the per-slot Detect records and the device tracker -- nothing here but pointers, so a caller without
    torch (tests/test_gpu_cabi_pipeline.py drives it from plain C) runs the timed path too.  Frames are device pointers
    (a torch tensor's data_ptr(), or fdt_dev_malloc memory).  `comm`: an fdt_comm_init_rank handle for world > 1."""

    def __init__(self, net, H, W, device_index=0, inflight=3, batch=1, comm=None, world=1, rank=0, source_hw=None,
                 score_thresh=0.4, sigma_iou=0.4, sigma_h=0.6, t_min=5, log_frames=256, plan_text=None):
        L = _lib.lib()
        net.firstTime = True
        net._sync_attributes(H, W)                         # PriorBox / Detect settings onto the handle (clones copy them)
        self.H, self.W, self.B, self.NF, self.world, self.rank = H, W, max(1, batch), max(1, inflight), world, rank
        self.top_k = net.detect.top_k
        self.REC = 2 * self.top_k * 5
        sh, sw = source_hw if source_hw else (0, 0)
        self._p = L.fdt_pipeline_create(net._h, int(device_index), H, W, self.NF, self.B,
                                        plan_text.encode() if plan_text else None, comm, rank, world, sh, sw,
                                        float(score_thresh), float(sigma_iou), float(sigma_h), int(t_min), int(log_frames))
        if not self._p:
            raise _lib.FdtError(_lib.FDT_ERR_HIP, (L.fdt_last_error() or b"").decode())
        self._L = L
        self._trk = IouTracker.borrowed(L.fdt_pipeline_tracker(self._p), sigma_iou, sigma_h, t_min)

    @staticmethod
    def _ptr(frames):
        return ctypes.c_void_p(frames.data_ptr() if hasattr(frames, "data_ptr") else int(frames))

    def prime(self, frames_dev):
        _lib.check(self._L.fdt_pipeline_prime(self._p, self._ptr(frames_dev)))

    def step(self, i, frames_dev):
        _lib.check(self._L.fdt_pipeline_step(self._p, int(i), self._ptr(frames_dev)))

    def step_host(self, i, frames_host, n_valid=None):
        """Step i from host frames: a C-contiguous uint8 array [batch, h, w, 3] (free for reuse on return)."""
        _lib.check(self._L.fdt_pipeline_step_host(self._p, int(i), _lib.ptr(frames_host), int(self.B if n_valid is None else n_valid)))

    def step_frame(self, i, frame_dev):
        _lib.check(self._L.fdt_pipeline_step_frame(self._p, int(i), self._ptr(frame_dev)))

    def flush(self):
        _lib.check(self._L.fdt_pipeline_flush(self._p))

    def sync(self):
        _lib.check(self._L.fdt_pipeline_sync(self._p))

    def mark(self, which):
        _lib.check(self._L.fdt_pipeline_mark(self._p, int(which)))

    def elapsed_ms(self):
        ms = ctypes.c_float(0)
        _lib.check(self._L.fdt_pipeline_elapsed_ms(self._p, ctypes.byref(ms)))
        return float(ms.value)

    def stamps_enable(self, n):
        """Record a timing event behind the association of each of the next n groups (latency measurements); 0 = off."""
        _lib.check(self._L.fdt_pipeline_stamps_enable(self._p, int(n)))

    def stamps_read(self, n):
        """ms from mark(0) to the completion of every group stamped so far (numpy f32 array)."""
        import numpy as np
        out = np.zeros(int(n), np.float32)
        cnt = ctypes.c_int(0)
        _lib.check(self._L.fdt_pipeline_stamps_read(self._p, _lib.ptr(out), int(n), ctypes.byref(cnt)))
        return out[:cnt.value]

    def slot(self, k):
        """(model handle, stream, record ptr, gathered-records ptr, counts ptr) of slot k, as c_void_p."""
        out = [ctypes.c_void_p() for _ in range(5)]
        _lib.check(self._L.fdt_pipeline_slot(self._p, int(k), *[ctypes.byref(o) for o in out]))
        return tuple(out)

    def record_of_slot(self, k):
        """Host copy of slot k's gathered records [world*B, 2, top_k, 5] (synchronises)."""
        import numpy as np
        self.sync()
        g = self.slot(k)[3]
        out = np.empty((self.world * self.B, 2, self.top_k, 5), np.float32)
        _lib.check(self._L.fdt_dev_download(_lib.ptr(out), g, out.nbytes))
        return out

    def counts_of_slot(self, k):
        import numpy as np
        self.sync()
        out = np.empty((self.B, 2), np.int32)
        _lib.check(self._L.fdt_dev_download(_lib.ptr(out), self.slot(k)[4], out.nbytes))
        return out

    def finish(self):
        self.flush()
        self.sync()
        return self._trk.finish()

    def close(self):
        if getattr(self, "_p", None):
            self._L.fdt_pipeline_destroy(self._p)
            self._p = None
            self._trk.close()

    __del__ = close
