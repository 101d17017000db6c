#!/usr/bin/env python3
"""profiles/<round>/conv_hbm_traffic.json, per kernel class and per layer, from rocprofv3 --pmc per-dispatch CSVs of one
forward sequence of the committed plan (tools/experiments/traffic_r4.sh):
    python tools/traffic_by_class.py RD_dispatches.csv WR_dispatches.csv OPS.json OUT.json [CAL_DIR|-] [FRAMES_PER_FORWARD]
("per frame" below = per FORWARD of FRAMES_PER_FORWARD frames: the grouped default of bench.py runs four frames per launch)
Read bytes = 32 * TCC_EA0_RDREQ_32B + 64 * _64B + 128 * _128B (the TCC's memory-side read requests BY SIZE CLASS: exact on a
1 GiB float4 copy, where FETCH_SIZE reports half -- it tallies every non-32-byte request at 64 bytes); written bytes =
64 * TCC_EA0_WRREQ_64B + 32 * the rest.  The conv dispatches of a forward are in op order (one conv-kernel dispatch per conv op),
so dispatch j of a forward is op j of tools/dump_ops.py: measured bytes sit next to the op's algorithmic bytes (input + output
(+ residual, upsample source) + weights, each once)."""
import collections
import csv
import json
import os
import sys

KIND = ["1x1s1", "1x1s2", "3x3s1", "3x3d2", "3x3s2", "7x7s2", "7x7s4", "5x5s2", "3x3s1_wino", "3x3d2_wino", "1x1s1_k32", "1x1s1_k64",
        "7x7s2p1", "3x3s1_n8", "3x3s1_wino44", "3x3d2_wino44", "1x1s1_p16", "1x1s1_p32", "7x7s2_u8", "7x7s4_u8", "7x7s4_k168", "1x1s1_b3", "7x7s4_b3", "1x1s2_b3", "7x7s2_u8b", "7x7s4_u8b", "1x1s1_pb3", "3x3s2_b3"]
TILE = ["128x128", "128x64", "128x32", "64x64", "64x128", "128x128W", "128x64W", "128x128R3", "128x64R3", "64x64R3", "64x128R3",
        "128x128WR3", "128x64WR3", "128x32R3", "w64x64", "w64x64R3", "w128x32", "w128x32R3", "w32x128", "w32x128R3", "w64x64W",
        "w8_64x64", "w8_64x64R3", "w8_128x32R3", "w8_64x64W", "128x128R4", "128x64R4", "64x64R4", "64x128R4", "w4_64x64R3",
        "w4_64x64W", "n8_32x64", "w44_32x64", "w44b_32x64", "p128x64", "p128x128", "128x32W", "r2_128x128", "r2_128x64", "r1_128x128", "r1_128x64"]


def is_conv(n):
    return any(k in n for k in ("conv_kernel", "conv_wino", "conv1x1p_kernel", "conv_stem_u8_kernel", "conv_stem_s4_kernel", "conv_n8_kernel", "conv_b3_kernel", "conv_stem_s4_b3_kernel", "conv_stem_u8b_kernel", "conv1x1p_b3_kernel")) and "reduce" not in n


def per_dispatch(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        e = d.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(d.values())


def rd_bytes(e):
    return 32 * e.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * e.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * e.get("TCC_EA0_RDREQ_128B_sum", 0)


def wr_bytes(e):
    w64 = e.get("TCC_EA0_WRREQ_64B_sum", 0)
    return 64 * w64 + 32 * (e.get("TCC_EA0_WRREQ_sum", 0) - w64)


rd = per_dispatch(sys.argv[1])
wr = per_dispatch(sys.argv[2])
ops = json.load(open(sys.argv[3]))
n_ops = len(ops)
rdc = [e for e in rd if is_conv(e["name"])]
wrc = [e for e in wr if is_conv(e["name"])]
assert len(rdc) % n_ops == 0 and len(wrc) == len(rdc), (len(rdc), len(wrc), n_ops)
F = len(rdc) // n_ops
skip = max(F - 8, 0)                                   # the last forwards: steady state (plans built, weights tiled)
use = range(skip, F)
per_op = []
for j, op in enumerate(ops):
    r = sum(rd_bytes(rdc[f * n_ops + j]) for f in use) / len(use)
    w = sum(wr_bytes(wrc[f * n_ops + j]) for f in use) / len(use)
    per_op.append({"op": op["op"], "class": "%s %s" % (KIND[op["kind"]], TILE[op["tile"]]), "split": op["split"],
                   "algorithmic_bytes": op["algorithmic_bytes"], "read_bytes": r, "written_bytes": w,
                   "ratio": (r + w) / op["algorithmic_bytes"], "excess_bytes": r + w - op["algorithmic_bytes"]})
classes = collections.OrderedDict()
for o in per_op:
    c = classes.setdefault(o["class"], {"launches_per_frame": 0, "algorithmic_bytes": 0.0, "read_bytes": 0.0, "written_bytes": 0.0})
    c["launches_per_frame"] += 1
    for k in ("algorithmic_bytes", "read_bytes", "written_bytes"):
        c[k] += o[k]
for c in classes.values():
    c["ratio"] = (c["read_bytes"] + c["written_bytes"]) / c["algorithmic_bytes"]
    c["bytes_per_launch"] = (c["read_bytes"] + c["written_bytes"]) / c["launches_per_frame"]
    c["algorithmic_bytes_per_launch"] = c["algorithmic_bytes"] / c["launches_per_frame"]
tot_a = sum(o["algorithmic_bytes"] for o in per_op)
tot_m = sum(o["read_bytes"] + o["written_bytes"] for o in per_op)
others = collections.defaultdict(lambda: [0, 0.0])
for lst, fn in ((rd, rd_bytes), (wr, wr_bytes)):
    for e in lst:
        if not is_conv(e["name"]) and ("fdt::" in e["name"]):
            k = e["name"].split("(")[0].split("::")[-1][:48]
            others[k][0] += 1
            others[k][1] += fn(e)
FPF = int(sys.argv[6]) if len(sys.argv) > 6 else 1
cal = {}
if len(sys.argv) > 5 and sys.argv[5] != "-":
    def mean_of(path, names):
        agg = collections.defaultdict(float)
        n = 0
        for r in csv.DictReader(open(path)):
            if is_conv(r["Kernel_Name"]) or "copy4" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] = float(r["Mean"])
        return agg
    known = {"copy": (1 << 30, 1 << 30, "1 GiB float4 copy"),
             "t8x16_64Brows": (256 * 65536 * 4 + 256 * 128 * 4, 128 * 65536 * 4, "1x1 256->128 @256^2, Tile<8,16,128>: 64-byte row segments"),
             "t4x32_128Brows": (256 * 65536 * 4 + 256 * 64 * 4, 64 * 65536 * 4, "1x1 256->64 @256^2, Tile<4,32,64>: 128-byte rows"),
             "k32_t4x32": (256 * 65536 * 4 + 256 * 64 * 4, 64 * 65536 * 4, "the same layer, 32 channels per stage"),
             "persistent_p16": (256 * 65536 * 4 + 256 * 64 * 4, 64 * 65536 * 4, "the same layer, persistent-tile kernel"),
             "wino44_16Bpieces": (256 * 65536 * 4 + 36 * 256 * 64 * 4, 64 * 65536 * 4, "3x3 256->64 @256^2, F(4x4): 16-byte pieces of 160-byte row segments, 18 rows per 16"),
             "wino22_w4": (256 * 65536 * 4 + 16 * 256 * 64 * 4, 64 * 65536 * 4, "3x3 256->64 @256^2, quarter-split F(2x2), 8x32-pixel tiles")}
    for n, (kb_r, kb_w, what) in known.items():
        try:
            a = mean_of(os.path.join(sys.argv[5], "cal_%s_rd.csv" % n), None)
            f = mean_of(os.path.join(sys.argv[5], "cal_%s_fetch.csv" % n), None)
            w = mean_of(os.path.join(sys.argv[5], "cal_%s_wr.csv" % n), None)
        except OSError:
            continue
        cal[n] = {"what": what, "single_read_bytes": kb_r, "read_bytes_by_size_class": rd_bytes(a), "FETCH_SIZE_bytes": f.get("FETCH_SIZE", 0) * 1024,
                  "read_over_single_read": rd_bytes(a) / kb_r, "output_bytes": kb_w, "written_bytes": wr_bytes(w)}
out = {
    "method": __doc__.split("\n\n")[0].split("\n", 3)[-1] if False else
              "read bytes = 32*TCC_EA0_RDREQ_32B + 64*_64B + 128*_128B, written = 64*TCC_EA0_WRREQ_64B + 32*(WRREQ - _64B); separate "
              "rocprofv3 --pmc passes of `bench.py --steps 8 --warmup 2 --cpu-frames 0 --host-frames 0 --inflight 1 --profile-frames 1 "
              "--graph 0`; mean over the last %d of %d forwards; per dispatch, joined with the op list by launch order" % (len(use), F),
    "forwards": F, "frames_per_forward": FPF, "conv_launches_per_frame": n_ops,
    "hbm_bytes_per_frame": tot_m, "algorithmic_bytes_per_frame": tot_a, "ratio": tot_m / tot_a,
    "hbm_bytes_per_launch": tot_m / n_ops, "algorithmic_bytes_per_launch": tot_a / n_ops,
    "calibration": cal,
    "by_class": [dict(cls=k, **{kk: (round(v, 4) if isinstance(v, float) and kk == "ratio" else round(v) if isinstance(v, float) else v)
                                for kk, v in c.items()}) for k, c in sorted(classes.items(), key=lambda kv: -kv[1]["read_bytes"] - kv[1]["written_bytes"])],
    "top_over_fetchers": [dict(op=o["op"], cls=o["class"], split=o["split"], ratio=round(o["ratio"], 3), excess_MB=round(o["excess_bytes"] / 1e6, 1),
                               algorithmic_MB=round(o["algorithmic_bytes"] / 1e6, 1), read_MB=round(o["read_bytes"] / 1e6, 1),
                               written_MB=round(o["written_bytes"] / 1e6, 1))
                          for o in sorted(per_op, key=lambda o: -o["excess_bytes"])[:12]],
    "other_kernels_total_bytes_over_all_forwards": {k: {"dispatches": v[0] // 2, "bytes": round(v[1])} for k, v in sorted(others.items(), key=lambda kv: -kv[1][1])[:8]},
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("overall: %.1f MB per forward of %d frame(s) measured vs %.1f MB algorithmic = %.2fx (%.1f vs %.1f MB per launch)" % (
    tot_m / 1e6, FPF, tot_a / 1e6, tot_m / tot_a, tot_m / n_ops / 1e6, tot_a / n_ops / 1e6))
for c in out["by_class"][:10]:
    print("  %-28s x%-3d %8.1f MB alg %8.1f MB read %8.1f MB written  ratio %.2f" % (
        c["cls"], c["launches_per_frame"], c["algorithmic_bytes"] / 1e6, c["read_bytes"] / 1e6, c["written_bytes"] / 1e6, c["ratio"]))
for o in out["top_over_fetchers"][:8]:
    print("  over-fetch: %-26s %-24s /%-2d ratio %.2f (+%.0f MB)" % (o["op"], o["cls"], o["split"], o["ratio"], o["excess_MB"]))
