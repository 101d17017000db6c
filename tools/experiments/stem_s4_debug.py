import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import test_gpu_conv as t
rng = np.random.default_rng(1)
for (B, H, W, Cout) in ((1, 64, 128, 24), (1, 45, 77, 24)):
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
    rc, got = t.run_conv(x, w, None, 7, 4, 3, 1, tile=36)
    exp = t.reference(x, w, None, 7, 4, 3, 1)
    err = np.abs(got - exp)
    print((H, W), "rc", rc, "max err", err.max(), "bad frac", (err > 1e-4).mean())
    bad = np.argwhere(err > 1e-4)
    print("bad cout set", sorted(set(bad[:, 1].tolist()))[:40], "rows", sorted(set(bad[:, 2].tolist()))[:40], "cols", sorted(set(bad[:, 3].tolist()))[:40])
    # which single tap explains it?  impulse test
    for c, ky, kx in ((0, 0, 0), (0, 0, 1), (0, 0, 6), (1, 3, 3), (2, 6, 5)):
        w1 = np.zeros_like(w); w1[:, c, ky, kx] = 1.0
        rc, g1 = t.run_conv(x, w1, None, 7, 4, 3, 1, tile=36)
        e1 = t.reference(x, w1, None, 7, 4, 3, 1)
        print("   impulse", (c, ky, kx), "max err", np.abs(g1 - e1).max())
