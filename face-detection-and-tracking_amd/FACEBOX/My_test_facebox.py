"""`detect(im)` of the reference FaceBoxes driver (reference FACEBOX/My_test_facebox.py:12-36).

The reference resizes any input to 1024x1024 with cv2 first (:13); here that resize (8-bit INTER_LINEAR restatement;
parity with cv2 itself is unpinned, cv2 is not available to the build) and everything after it -- /255, FaceBox
forward, softmax, decode_np, nms_np -- run on the GPU in one call."""
import numpy as np

net = None            # set by the caller, like the reference's module-level `net` (:40-44)
data_encoder = None   # kept for API symmetry; decode runs inside net.detect_frames()


def detect(im):
    im = np.ascontiguousarray(im, dtype=np.uint8)
    (boxes_, probs_), = net.detect_frames(im[None])
    return boxes_, probs_
