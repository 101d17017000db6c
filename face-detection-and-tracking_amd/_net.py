"""Shared host-side detector object behind `pyramid.SFD` / `pyramid_mb2_try3.SFD_mobile`.

Mirrors what callers of the reference modules rely on: `load_state_dict`,
`load_weights`, mutable `.priorbox` / `.firstTime` / `.detect` attributes, `.priors` after a
forward, `.cuda()` / `.eval()` no-ops, and `net(x) -> Tensor[B,2,top_k,5]`.  All compute happens
in libfdt_hip.so; torch is only the container for weights in and detections out.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .layers import Detect, PriorBoxLayer


class DetectorNet:
    _arch = None
    _default_priorbox = None     # () -> PriorBoxLayer
    _default_detect = None       # () -> Detect

    def __init__(self, phase, num_classes, size, device=0):
        if phase != 'test':
            raise NotImplementedError("only phase='test' (inference) is implemented on this path")
        self.phase = phase
        self.num_classes = num_classes
        self.size = size
        self.firstTime = True
        self.priorbox = type(self)._default_priorbox(size)
        self.priors = None
        self.detect = type(self)._default_detect(num_classes)
        self.training = False
        self._device = device
        self._h = _lib.lib().fdt_model_create(self._arch, device)
        if not self._h:
            raise _lib.FdtError(_lib.FDT_ERR_HIP, (_lib.lib().fdt_last_error() or b"").decode())
        self._loaded = False
        self._sd_cache = None
        self._prior_shape = None

    def clone(self):
        """A second handle for another frame in flight: own stream / activations / plan, SHARED device weights
        (fdt_model_clone).  Attribute objects (.priorbox, .detect) are shared references like a shallow module copy."""
        if not self._loaded:
            raise RuntimeError("weights not loaded: call load_state_dict / load_weights first")
        c = object.__new__(type(self))
        c.__dict__.update(self.__dict__)
        c._h = _lib.lib().fdt_model_clone(self._h)
        if not c._h:
            raise _lib.FdtError(_lib.FDT_ERR_HIP, (_lib.lib().fdt_last_error() or b"").decode())
        c.firstTime = True
        c._prior_shape = None
        c._dev_out = {}
        c.__dict__.pop("_plan_key", None)
        return c

    def enable_graph(self, on=True):
        """Replay the forward as a captured HIP graph (default on)."""
        _lib.check(_lib.lib().fdt_model_enable_graph(self._h, 1 if on else 0))

    # ---- nn.Module surface the reference's callers touch ------------------------------------
    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        self.training = False
        return self

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().fdt_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def state_dict(self):
        return self._sd_cache

    def load_state_dict(self, state_dict, strict=True):
        L = _lib.lib()
        unexpected = []
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            dims = (C.c_longlong * max(a.ndim, 1))(*a.shape)
            rc = L.fdt_model_set_tensor(self._h, k.encode(), _lib.ptr(a), a.ndim, dims)
            if rc == _lib.FDT_ERR_NAME:
                unexpected.append(k)
            else:
                _lib.check(rc)
        n = C.c_int(0)
        _lib.check(L.fdt_model_missing(self._h, C.byref(n)))
        missing = []
        for i in range(n.value):
            buf = C.create_string_buffer(256)
            _lib.check(L.fdt_model_missing_name(self._h, i, buf, 256))
            missing.append(buf.value.decode())
        if missing or (strict and unexpected):
            msg = []
            if missing:
                msg.append("Missing key(s) in state_dict: " + ", ".join('"%s"' % k for k in missing))
            if strict and unexpected:
                msg.append("Unexpected key(s) in state_dict: " + ", ".join('"%s"' % k for k in unexpected))
            raise RuntimeError("Error(s) in loading state_dict for %s:\n\t%s"
                               % (type(self).__name__, "\n\t".join(msg)))
        _lib.check(L.fdt_model_finalize(self._h))
        self._sd_cache = state_dict
        self._loaded = True
        return self

    def load_weights(self, base_file):
        """reference pyramid.py:353-364: `model_dict.update({k: v for k in pretrained if k in model_dict})` -- unknown
        keys are dropped silently.  The reference also keeps keys the file lacks at their random initial values; a
        detector with random layers is never what an inference caller wants, so here missing keys still raise (the
        message lists them) instead of silently producing garbage."""
        print('Loading weights into state dict...')
        sd = torch.load(base_file, map_location='cpu', weights_only=True)
        self.load_state_dict(sd, strict=False)
        print('Finished!')

    # ---- forward ---------------------------------------------------------------------------------
    def _sync_attributes(self, H, W):
        L = _lib.lib()
        if self.firstTime:
            pb = self.priorbox
            nlev = len(pb.stride)
            if any(int(s) != 1 for s in pb.scales[:nlev]) or any(len(a) for a in pb.aspect_ratios[:nlev]):
                raise NotImplementedError("the fused path supports scale=1 / no aspect ratios "
                                          "(every configuration the reference uses)")
            st = (C.c_int * nlev)(*[int(s) for s in pb.stride])
            bx = (C.c_int * nlev)(*[int(b) for b in pb.box[:nlev]])
            _lib.check(L.fdt_model_set_priorbox(self._h, int(pb.width), int(pb.height), nlev, st, bx))
            self.firstTime = False
            self._prior_shape = (H, W)
        elif self._prior_shape != (H, W):
            raise RuntimeError("priors were generated for input %s, got %s; set net.firstTime = True "
                               "(the reference fails with a size mismatch here)" % (self._prior_shape, (H, W)))
        d = self.detect
        _lib.check(L.fdt_model_set_detect(self._h, int(d.top_k), float(d.conf_thresh),
                                          float(d.nms_thresh), int(d.nms_top_k)))

    def _prepare(self, x):
        if isinstance(x, torch.Tensor):
            x = x.detach().cpu().numpy()
        x = np.asarray(x)
        if x.dtype == np.uint8:
            if x.ndim == 3:
                x = x[None]
            B, H, W, _ = x.shape
            return np.ascontiguousarray(x), _lib.FRAME_U8_HWC_BGR, B, H, W
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, _, H, W = x.shape
        return x, _lib.FRAME_F32_NCHW, B, H, W

    def __call__(self, x):
        """x: [B,3,H,W] f32 (mean-subtracted BGR, what the reference passes) or uint8 [B,H,W,3]
        raw BGR frames (mean subtraction then happens on the GPU)."""
        if not self._loaded:
            raise RuntimeError("weights not loaded: call load_state_dict / load_weights first")
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return self._call_device(x)
        x, fmt, B, H, W = self._prepare(x)
        self._sync_attributes(H, W)
        out = np.empty((B, 2, self.detect.top_k, 5), dtype=np.float32)
        counts = np.zeros((B, 2), dtype=np.int32)
        _lib.check(_lib.lib().fdt_model_forward(self._h, _lib.ptr(x), fmt, B, H, W, _lib.ptr(out),
                                                _lib.ptr(counts)))
        self.last_counts = counts
        self.priors = torch.from_numpy(self.get_tensor("priors")[0])
        return torch.from_numpy(out)

    def _call_device(self, x):
        """`net(x.cuda())` as the reference writes it (iouTracke_cal.py:49-50, My_test.py:37-39): the frames already live in
        HBM, so they are consumed there (fdt_model_forward_dev on the handle's stream) instead of bouncing through the host.
        Returns the same CPU tensor as the host path (detection.py:48 creates the output on the default device)."""
        x = x.detach()
        if x.dtype == torch.uint8:
            if x.dim() == 3:
                x = x[None]
            x = x.contiguous()
            fmt, (B, H, W, _) = _lib.FRAME_U8_HWC_BGR, x.shape
        else:
            x = x.to(torch.float32).contiguous()
            fmt, (B, _, H, W) = _lib.FRAME_F32_NCHW, x.shape
        if x.device.index not in (None, self._device):
            raise ValueError("input is on cuda:%s but the model was built for cuda:%d" % (x.device.index, self._device))
        self._sync_attributes(H, W)
        # persistent output buffers per (B, top_k): the handle's captured HIP graph is keyed by these addresses, so a fresh
        # tensor per call would pay a capture + instantiate every time instead of a replay
        key = (B, int(self.detect.top_k), x.device.index)
        bufs = self.__dict__.setdefault("_dev_out", {})
        if key not in bufs:
            bufs.clear()
            bufs[key] = (torch.empty((B, 2, key[1], 5), dtype=torch.float32, device=x.device),
                         torch.zeros((B, 2), dtype=torch.int32, device=x.device))
        out, counts = bufs[key]
        torch.cuda.current_stream(x.device).synchronize()       # x is complete; the forward runs on the handle's own stream
        _lib.check(_lib.lib().fdt_model_forward_dev(self._h, C.c_void_p(x.data_ptr()), fmt, B, H, W,
                                                    C.c_void_p(out.data_ptr()), C.c_void_p(counts.data_ptr()), None))
        torch.cuda.synchronize(x.device)
        self.last_counts = counts.cpu().numpy()
        self.priors = torch.from_numpy(self.get_tensor("priors")[0])
        return out.cpu()

    forward = __call__

    def forward_resized(self, frames, size):
        """frames: uint8 [B,SH,SW,3] (or [SH,SW,3]) raw BGR frames; size = (W, H) network input.  The
        cv2.resize(image, (W, H)) of reference iouTracke_cal.py:123 and the mean subtraction run on the
        GPU (parity with cv2 itself is unpinned: cv2 is not available to the build)."""
        x = np.ascontiguousarray(frames, dtype=np.uint8)
        if x.ndim == 3:
            x = x[None]
        B, SH, SW, _ = x.shape
        W, H = int(size[0]), int(size[1])
        self._sync_attributes(H, W)
        out = np.empty((B, 2, self.detect.top_k, 5), dtype=np.float32)
        counts = np.zeros((B, 2), dtype=np.int32)
        _lib.check(_lib.lib().fdt_model_forward_resized(self._h, _lib.ptr(x), 0, B, SH, SW, H, W, _lib.ptr(out),
                                                        _lib.ptr(counts), None))
        self.last_counts = counts
        return torch.from_numpy(out)

    # ---- debugging / parity helpers ------------------------------------------------------------
    def forward_raw(self, x):
        """(loc [B,P,4], softmaxed conf [B,P,2]) without Detect."""
        x, fmt, B, H, W = self._prepare(x)
        self._sync_attributes(H, W)
        # a first call is needed to know P: run once with small outputs via get_tensor
        L = _lib.lib()
        P = self._num_priors_for(x, fmt, B, H, W)
        loc = np.empty((B, P, 4), np.float32)
        conf = np.empty((B, P, 2), np.float32)
        _lib.check(L.fdt_model_forward_raw(self._h, _lib.ptr(x), fmt, B, H, W, _lib.ptr(loc), _lib.ptr(conf)))
        return loc, conf

    def _num_priors_for(self, x, fmt, B, H, W):
        def half(n):
            return (n - 1) // 2 + 1
        h, w = half(half(H)), half(half(W))
        P = 0
        for _ in range(self._n_sources):
            P += h * w
            h, w = half(h), half(w)
        return P

    def get_tensor(self, name):
        L = _lib.lib()
        dims = (C.c_longlong * 4)()
        _lib.check(L.fdt_model_get_tensor(self._h, name.encode(), None, 0, dims))
        shape = tuple(int(d) for d in dims)
        out = np.empty(shape, np.float32)
        _lib.check(L.fdt_model_get_tensor(self._h, name.encode(), _lib.ptr(out), out.size, dims))
        if name in ("loc", "conf", "conf_logits", "priors"):
            out = out.reshape(shape[:3])
        return out

    def autotune(self, iters=3):
        """Measure every (tile, split-K) variant per conv layer at the last forward's shape, keep the best."""
        _lib.check(_lib.lib().fdt_model_autotune(self._h, int(iters)))

    def export_plan(self):
        """Text form of the current per-layer kernel plan (see fdt_model_export_plan)."""
        L = _lib.lib()
        need = C.c_int(0)
        _lib.check(L.fdt_model_export_plan(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _lib.check(L.fdt_model_export_plan(self._h, buf, need.value, C.byref(need)))
        return buf.value.decode()

    def import_plan(self, text):
        _lib.check(_lib.lib().fdt_model_import_plan(self._h, text.encode()))

    _ARCH_NAMES = {_lib.ARCH_RES50: "res50", _lib.ARCH_TRY3: "try3", _lib.ARCH_FACEBOX: "facebox", _lib.ARCH_TRY4: "try4",
                   _lib.ARCH_TRY5: "try5", _lib.ARCH_TRY1: "try1", _lib.ARCH_TRY2: "try2"}

    def tuned_plan_text(self, H, W, B=1):
        """The committed autotuner result for (arch, W x H, batch) under tuned/, or None: per-layer (kernel class, tile,
        split-K, workgroup map) measured on MI355X.  What bench.py imports; the entry points use it so that they run the
        same kernels (an un-planned shape falls back to the analytic choice of `Builder::choose`)."""
        import os
        f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned",
                         "%s_%dx%d_b%d.plan" % (self._ARCH_NAMES[self._arch], W, H, B))
        return open(f).read() if os.path.exists(f) else None

    def flops_per_frame(self):
        f = C.c_double(0)
        _lib.check(_lib.lib().fdt_model_flops(self._h, C.byref(f)))
        return f.value

    def traffic(self):
        """(activation bytes, weight bytes, per-op bytes in profile_read() order) of one forward of the current plan:
        the algorithmic un-fused lower bound the HBM rooflines are computed from (fdt_model_traffic)."""
        act, wts, n = C.c_double(0), C.c_double(0), C.c_int(0)
        per = np.zeros(512, np.float64)
        _lib.check(_lib.lib().fdt_model_traffic(self._h, C.byref(act), C.byref(wts), 512, _lib.ptr(per), C.byref(n)))
        return act.value, wts.value, per[:n.value].copy()

    def profile(self, on=True):
        _lib.check(_lib.lib().fdt_model_profile_enable(self._h, 1 if on else 0))

    def profile_segment(self, first_op, last_op):
        """One event pair around ops [first_op, last_op] of the next forwards (fdt_model_profile_segment); first_op < 0: off."""
        _lib.check(_lib.lib().fdt_model_profile_segment(self._h, int(first_op), int(last_op)))

    def profile_segment_ms(self):
        ms = C.c_float(0)
        _lib.check(_lib.lib().fdt_model_profile_segment_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def profile_read(self):
        L = _lib.lib()
        n = C.c_int(0)
        cap = 512
        names = C.create_string_buffer(cap * 48)
        ms = np.zeros(cap, np.float32)
        fl = np.zeros(cap, np.float64)
        _lib.check(L.fdt_model_profile_read(self._h, cap, names, _lib.ptr(ms), _lib.ptr(fl), C.byref(n)))
        out = []
        for i in range(min(n.value, cap)):
            nm = names.raw[i * 48:(i + 1) * 48].split(b"\0")[0].decode()
            out.append((nm, float(ms[i]), float(fl[i])))
        return out
