"""Host handle of the device-resident IoU tracker (`fdt_tracker_*`), i.e. the inline tracker of
reference iouTracke_cal.py:113-156,174-177 as an object.  Tracks come back with the reference's
schema: {'bboxes': [[x1,y1,x2,y2],...], 'max_score': float, 'start_frame': int}
(reference iouTracke_cal.py:150-154; consumed by iouTracke_display.py:29,44-63)."""
import ctypes as C

import numpy as np

from . import _lib


class IouTracker:
    def __init__(self, sigma_iou=0.4, sigma_h=0.6, t_min=5, max_dets=1500, log_frames=64):
        self._h = _lib.lib().fdt_tracker_create(float(sigma_iou), float(sigma_h), int(t_min),
                                                int(max_dets), int(log_frames))
        if not self._h:
            raise _lib.FdtError(_lib.FDT_ERR_HIP, (_lib.lib().fdt_last_error() or b"").decode())
        self.sigma_iou, self.sigma_h, self.t_min = sigma_iou, sigma_h, t_min

    @classmethod
    def borrowed(cls, handle, sigma_iou=0.4, sigma_h=0.6, t_min=5):
        """A view of a tracker another object owns (fdt_pipeline_tracker): finish() / tracks work, close() does not destroy."""
        t = object.__new__(cls)
        t._h, t._borrowed = handle, True
        t.sigma_iou, t.sigma_h, t.t_min = sigma_iou, sigma_h, t_min
        return t

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                _lib.lib().fdt_tracker_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        _lib.check(_lib.lib().fdt_tracker_reset(self._h))

    def step(self, det0):
        """det0: [n,5] rows (x1,y1,x2,y2,score) -- what detect_face() returns."""
        d = np.ascontiguousarray(det0, dtype=np.float64).reshape(-1, 5)
        _lib.check(_lib.lib().fdt_tracker_step(self._h, _lib.ptr(d), d.shape[0]))

    def step_dev(self, det_out_ptr, num_classes, top_k, width, height, score_thresh=0.4, stream=None):
        """Consume the device-resident Detect output of one image (no host round trip)."""
        _lib.check(_lib.lib().fdt_tracker_step_dev(self._h, det_out_ptr, num_classes, top_k,
                                                   int(width), int(height), float(score_thresh),
                                                   stream))

    def step_dev_multi(self, det_out_ptr, n_frames, stride_floats, num_classes, top_k, width, height, score_thresh=0.4,
                       stream=None):
        """The n_frames records of one frame-parallel step (rank order == frame order) in ONE launch; bit-identical
        to n_frames step_dev calls."""
        _lib.check(_lib.lib().fdt_tracker_step_dev_multi(self._h, det_out_ptr, int(n_frames), int(stride_floats),
                                                         num_classes, top_k, int(width), int(height),
                                                         float(score_thresh), stream))

    def stats(self):
        """Which association form the frames so far ran (fdt_tracker_stats): {'frames', 'candidate', 'exact_nan',
        'exact_overflow', 'exact_sigma'} -- both forms reproduce iouTracke_cal.py:129-148 bit for bit."""
        fr = C.c_longlong(0)
        form = (C.c_longlong * 4)()
        _lib.check(_lib.lib().fdt_tracker_stats(self._h, C.byref(fr), form))
        return {"frames": int(fr.value), "candidate": int(form[0]), "exact_nan": int(form[1]),
                "exact_overflow": int(form[2]), "exact_sigma": int(form[3])}

    def finish(self):
        L = _lib.lib()
        _lib.check(L.fdt_tracker_finish(self._h))
        n = C.c_int(0)
        _lib.check(L.fdt_tracker_num_tracks(self._h, C.byref(n)))
        tracks = []
        for i in range(n.value):
            nb, ms, sf = C.c_int(0), C.c_double(0), C.c_int(0)
            _lib.check(L.fdt_tracker_track_info(self._h, i, C.byref(nb), C.byref(ms), C.byref(sf)))
            boxes = np.empty((nb.value, 4), dtype=np.float64)
            _lib.check(L.fdt_tracker_track_boxes(self._h, i, _lib.ptr(boxes)))
            tracks.append({'bboxes': boxes.tolist(), 'max_score': ms.value, 'start_frame': sf.value})
        return tracks
