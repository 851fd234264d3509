#!/usr/bin/env python3
"""Aggregate two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline
    python tools/pmc_aggregate.py OUT profiles/rNN_pmc_traffic.json

Corrections follow MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16-byte streaming stores.
"""
import collections
import csv
import glob
import json
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return n.replace("unsigned short", "bf16")


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0, 0.0, []])
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if len(k) > 100 or k.startswith("at::"):
            continue
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
        d[k][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[k][3].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


def main(src, dst, frames_per_step=4096):
    f = agg(glob.glob(f"{src}/fetch/*/*_counter_collection.csv")[0])
    w = agg(glob.glob(f"{src}/write/*/*_counter_collection.csv")[0])
    out = {"_note": "per-launch averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction)",
           "_frames_per_step": frames_per_step}
    for k in sorted(f):
        if k.startswith("_"):
            continue
        n = f[k][0]
        fk = f[k][1] / n
        wk = w[k][1] / max(1, w[k][0]) if k in w else 0.0
        out[k] = {"launches": n, "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                  "hbm_bytes": int((2 * fk + wk) * 1024), "avg_ns_profiled": int(f[k][2] / n),
                  # a kernel's first launch can be cold by two orders of magnitude (code object upload): the median is the figure to quote
                  "median_ns_profiled": int(sorted(f[k][3])[n // 2])}
    json.dump(out, open(dst, "w"), indent=1)
    print(f"wrote {dst} ({len(out) - 1} kernels)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4096)
