#!/bin/bash
# conv_b3.h (split-bf16 1x1) against the f32 1x1 classes on the 1x1 shapes of Res50 @1024^2, four frames per launch
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python - <<'PY'
import sys, os
sys.path.insert(0, "tools")
import conv_bench as cb
B = 4
shapes = [("layer1.0.conv3 64->256 @256 +res", 64, 256, 256, 256, 1), ("layer1.1.conv1 256->64 @256", 256, 256, 256, 64, 0),
          ("layer2.0.conv1 256->128 @256", 256, 256, 256, 128, 0), ("layer2.1.conv3 128->512 @128 +res", 128, 128, 128, 512, 1),
          ("layer2.1.conv1 512->128 @128", 512, 128, 128, 128, 0), ("layer3.1.conv3 256->1024 @64 +res", 256, 64, 64, 1024, 1),
          ("layer3.1.conv1 1024->256 @64", 1024, 64, 64, 256, 0), ("layer4.1.conv3 512->2048 @32 +res", 512, 32, 32, 2048, 1),
          ("layer4.1.conv1 2048->512 @32", 2048, 32, 32, 512, 0), ("latlayer_fc 2048->2048 @32", 2048, 32, 32, 2048, 0),
          ("conv4_ct.main 1024->1024 @64", 1024, 64, 64, 1024, 0), ("conv3_ct.main 512->512 @128", 512, 128, 128, 512, 0)]
print("%-38s %28s   %28s" % ("layer (batch 4)", "best f32 class", "split-bf16 (class 21)"))
for name, cin, h, w, cout, res in shapes:
    gf = 2.0 * B * h * w * cout * cin / 1e9
    best = None
    for kind in (0, 10, 11, 16, 17):
        for t in range(len(cb.TILE)):
            for sp in (1, 2, 4):
                ms = cb.bench(kind, t, sp, cin, h, w, cout, res, 0, 10, B)
                if ms and (best is None or ms < best[0]):
                    best = (ms, kind, t, sp)
    b3 = None
    for t in (7, 8, 11, 12):
        for sp in (1, 2, 4):
            ms = cb.bench(21, t, sp, cin, h, w, cout, res, 0, 10, B)
            if ms and (b3 is None or ms < b3[0]):
                b3 = (ms, 21, t, sp)
    f = lambda r: "%7.1f us %6.1f TF/s %s/%d" % (r[0] * 1e3, gf / r[0], cb.KIND[r[1]] + ":" + cb.TILE[r[2]], r[3])
    print("%-38s %28s   %28s   x%.2f" % (name, f(best), f(b3), best[0] / b3[0]))
PY
