import sys
sys.path.insert(0, "/root/repo/tools")
import conv_bench as cb
for (cin, h, w, cout, res) in ((1024, 64, 64, 256, 0), (256, 64, 64, 1024, 1), (512, 128, 128, 128, 0), (128, 128, 128, 512, 1), (2048, 32, 32, 512, 0), (512, 32, 32, 2048, 1)):
    gf = 2.0 * h * w * cout * cin / 1e9
    rows = []
    for kind in (0, 10, 11):
        for t in range(len(cb.TILE)):
            for sp in (1, 2, 4):
                ms = cb.bench(kind, t, sp, cin, h, w, cout, res, 0, iters=10)
                if ms: rows.append((ms, cb.KIND[kind], cb.TILE[t], sp))
    rows.sort()
    b1 = min(r for r in rows if r[3] == 1); b2 = min(r for r in rows if r[3] == 2); b4 = min((r for r in rows if r[3] == 4), default=None)
    print("1x1 %d->%d @%dx%d res %d (%.2f GF): /1 %.1f us (%s %s)  /2 %.1f us (%s %s)  /4 %s" % (cin, cout, h, w, res, gf, b1[0]*1e3, b1[1], b1[2], b2[0]*1e3, b2[1], b2[2], "%.1f us" % (b4[0]*1e3) if b4 else "-"), flush=True)
