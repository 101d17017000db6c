"""CPU oracle (test infrastructure only): cv2.resize(src, (W, H), INTER_LINEAR) on 8-bit 3-channel
frames, restated from OpenCV's generic 8-bit path (imgproc/resize.cpp: HResizeLinear + the
VResizeLinear<uchar,int,short> fixed-point formula).  PARITY UNPINNED: cv2 is not installed in the build
image and the reference holds no resized fixture, so this restatement is checked against nothing but
itself and the HIP kernel; reference call sites: iouTracke_cal.py:123, FACEBOX/My_test_facebox.py:13."""
import numpy as np


def _coef(dsize, ssize):
    scale = ssize / dsize
    d = np.arange(dsize)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0; s[lo] = 0
    hi = s >= ssize - 1
    f[hi] = 0; s[hi] = ssize - 1
    s1 = np.minimum(s + 1, ssize - 1)
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, s1, a0, a1


def resize_linear_u8(src, width, height):
    src = np.asarray(src, dtype=np.uint8)
    SH, SW, _ = src.shape
    sx0, sx1, ax0, ax1 = _coef(width, SW)
    sy0, sy1, by0, by1 = _coef(height, SH)
    s = src.astype(np.int64)
    h = s[:, sx0, :] * ax0[None, :, None] + s[:, sx1, :] * ax1[None, :, None]      # [SH, W, 3]
    v = (((by0[:, None, None] * (h[sy0] >> 4)) >> 16) + ((by1[:, None, None] * (h[sy1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)
