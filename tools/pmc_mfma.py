#!/usr/bin/env python3
"""MFMA pipe utilisation per kernel from one rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline
    python tools/pmc_mfma.py DIR out.json

SQ_VALU_MFMA_BUSY_CYCLES sums the cycles every SIMD's matrix pipe was busy (16 per v_mfma_f32_16x16x32_bf16: checked against the
instruction count of the fc1 launch), GRBM_GUI_ACTIVE the active cycles of the 8 XCDs: busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 x SIMDs)."""
import collections
import csv
import glob
import json
import sys

src, out = sys.argv[1], sys.argv[2]
simds = 256 * 4
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for pre in ("void (anonymous namespace)::", "(anonymous namespace)::"):
            if k.startswith(pre):
                k = k[len(pre):]
        k = k.split("(")[0].replace("unsigned short", "bf16")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
res = {"_note": "per-launch averages; mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    m, g = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if m <= 0 or g <= 0 or not n[k]:
        continue
    res[k] = {"launches": n[k], "mfma_busy_cycles": m / n[k], "gui_active_cycles_per_xcd": g / n[k] / 8,
              "mfma_busy_fraction": round(m / (g / 8 * simds), 4)}
json.dump(res, open(out, "w"), indent=1)
print(f"wrote {out} ({len(res) - 1} kernels)")
