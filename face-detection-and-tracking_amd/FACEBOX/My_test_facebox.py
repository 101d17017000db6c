"""`detect(im)` of the reference FaceBoxes driver (reference FACEBOX/My_test_facebox.py:12-36).

The reference resizes any input to 1024x1024 with cv2 first (:13); cv2 is outside the path, so `im`
must already be a 1024x1024 uint8 BGR frame.  Everything after the resize -- /255, FaceBox forward,
softmax, decode_np, nms_np -- runs on the GPU in one call."""
import numpy as np

net = None            # set by the caller, like the reference's module-level `net` (:40-44)
data_encoder = None   # kept for API symmetry; decode runs inside net.detect_frames()


def detect(im):
    im = np.ascontiguousarray(im, dtype=np.uint8)
    if im.shape[:2] != (1024, 1024):
        raise ValueError("resize the frame to 1024x1024 first (reference My_test_facebox.py:13 uses cv2.resize)")
    (boxes_, probs_), = net.detect_frames(im[None])
    return boxes_, probs_
