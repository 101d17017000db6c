"""Offline multi-face IoU tracker entry point, mirroring reference iouTracke_cal.py.

Same module-level configuration names (reference iouTracke_cal.py:22-31), same `detect_face(x, shrink)`
contract (:36-84), same per-frame association and finalisation (:126-156, :174-177) and the same
`.npy` track schema (:150-154,:177) -- but detection, the host unpack and the association run on the
MI355X (`fdt_model_forward*`, `fdt_tracker_step*`).  Video decoding / `cv2.resize` / display are not
part of the path (cv2 is not a dependency): frames come from any iterable of uint8 BGR HWC arrays
already at network resolution (the reference resizes to 640x480 at :123).

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
from .utils.calc_performance import calculate_iou  # noqa: F401  (re-exported like the reference)

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

net = None


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


def detect_face(x, shrink=1):
    """BGR uint8 HWC frame -> ndarray [n,5] (x1,y1,x2,y2,score) in pixels; reference :36-84."""
    if shrink != 1:
        raise NotImplementedError("shrink != 1 needs cv2.resize, which is outside the path "
                                  "(the reference always calls detect_face(image, 1), :124)")
    height, width, _ = x.shape
    y = net(np.ascontiguousarray(x, dtype=np.uint8))     # mean subtraction happens on the GPU
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


def track(frames, device_resident=True):
    """Run detect + IoU association over an iterable of frames; returns `tracks_finished`
    (reference :113-156 + :174-175).  With `device_resident` the Detect output never leaves the GPU
    between detection and association (the host unpack of :53-84 runs inside the tracker kernel)."""
    if not use_iou:
        raise NotImplementedError("use_iou=False (calculate_distance, reference :136-138) is dead code "
                                  "under the reference's own configuration")
    tracker = IouTracker(sigma_iou, sigma_h, t_min)
    L = _lib.lib()
    out_dev = None
    for image in frames:
        if not device_resident:
            tracker.step(detect_face(image, 1))
            continue
        image = np.ascontiguousarray(image, dtype=np.uint8)
        H, W, _ = image.shape
        net._sync_attributes(H, W)
        top_k = net.detect.top_k
        if out_dev is None:
            out_dev = torch.empty((2, top_k, 5), dtype=torch.float32, device="cuda")
            frame_dev = torch.empty(image.shape, dtype=torch.uint8, device="cuda")
            stream = torch.cuda.Stream()
            sp = ctypes.c_void_p(stream.cuda_stream)
        with torch.cuda.stream(stream):
            frame_dev.copy_(torch.from_numpy(image), non_blocking=False)
            _lib.check(L.fdt_model_forward_dev(net._h, ctypes.c_void_p(frame_dev.data_ptr()),
                                               _lib.FRAME_U8_HWC_BGR, 1, H, W,
                                               ctypes.c_void_p(out_dev.data_ptr()), None, sp))
            tracker.step_dev(ctypes.c_void_p(out_dev.data_ptr()), 2, top_k, W, H, 0.4, sp)
    return tracker.finish()


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
    load_net('./net_weight/Res50_pyramid.pth' if use_net == 'repo' else 'net_weight/Mobile_pyramid_try3.pth',
             frames.shape[2], frames.shape[1])
    print('Finished loading model!')
    save_tracks(track(frames), video_file + '.npy')
