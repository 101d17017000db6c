import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_conv as T
case = tuple(int(v) for v in sys.argv[1:12])
k, s, p, d, tile, split, B, Cin, H, W, Cout = case
rng = np.random.default_rng(1)
x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
rc, a = T.run_conv(x, w, None, k, s, p, d, tile=tile, split=split)
for i in range(3):
    rc, f = T.run_conv(x, w, None, k, s, p, d, tile=tile, split=split | 0x1000)
    bad = np.argwhere(f != a)
    print(rc, "mismatches", len(bad), "of", a.size)
    if len(bad):
        print("channels", np.unique(bad[:, 1])[:20], "rows", np.unique(bad[:, 2])[:40], "cols", np.unique(bad[:, 3])[:70])
        b0 = tuple(bad[0]); print(b0, f[b0], a[b0])
