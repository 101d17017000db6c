import sys, os
sys.path.insert(0, "tools")
import conv_bench as cb
for (cin, h, w, cout) in ((256, 256, 256, 256), (512, 128, 128, 512), (1024, 64, 64, 1024)):
    gf = 2.0 * h * w * cout * cin / 1e9
    for up in (0, 1):
        best = None
        for kind in (0, 10, 11):
            for t in range(len(cb.TILE)):
                for sp in (1, 2, 4):
                    ms = cb.bench(kind, t, sp, cin, h, w, cout, 0, up, iters=10)
                    if ms and (best is None or ms < best[0]):
                        best = (ms, cb.KIND[kind], cb.TILE[t], sp)
        print("1x1 %d->%d @%dx%d up=%d: best %s %s /%d %.1f us %.1f TF/s" % (cin, cout, h, w, up, best[1], best[2], best[3], best[0] * 1e3, gf / best[0]), flush=True)
