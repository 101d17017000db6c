import sys, ctypes, importlib, numpy as np
sys.path.insert(0, "/root/repo")
M = lambda n: importlib.import_module("face-detection-and-tracking_amd." + n)
L = M("_lib"); synth = M("synth")
def free():
    f, t = ctypes.c_longlong(0), ctypes.c_longlong(0)
    L.check(L.lib().fdt_device_mem_info(ctypes.byref(f), ctypes.byref(t))); return f.value
sd = synth.make_state_dict("res50", seed=0)
frame = synth.make_frames(1, 128, 160, seed=1)[0]
base = free()
for i in range(4):
    net = M("pyramid").build_sfd('test', 640, 2); net.load_state_dict(sd)
    net.priorbox = M("layers").PriorBoxLayer(160, 128); net.detect = M("layers").Detect(2, 0, 750, 0.05, 0.35)
    y = net(frame); c = net.clone(); y2 = c(frame)
    net.autotune(1) if i == 0 else None
    tr = M("tracker").IouTracker(0.4, 0.6, 5); tr.step(np.zeros((1, 5), np.float32)); tr.finish(); tr.close()
    net.close(); y3 = c(frame); assert np.array_equal(y2.numpy(), y3.numpy()); c.close()
    print("cycle", i, "free delta MB", (base - free()) / 1e6, flush=True)
fb = M("FACEBOX.networks").FaceBox()
z = np.load("/root/repo/tests/golden/faceboxes_weights.npz"); fb.load_state_dict({k: z[k] for k in z.files})
g = np.load("/root/repo/tests/golden/facebox.npz")
for i in range(3):
    fb.detect_frames(np.stack([g["img0_frame"]] * 2))
    print("facebox", i, "free delta MB", (base - free()) / 1e6, flush=True)
fb.close()
print("after facebox close", (base - free()) / 1e6)
