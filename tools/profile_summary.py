#!/usr/bin/env python3
"""Condenses the CSVs that tools/profile.sh collected into summary.json (per-dispatch averages of every
counter for the dominant kernel of each output kind, plus the kernel-trace statistics of bench.py) and
traffic.json (fabric-side bytes per launch, which bench.py reports as roofline.traffic).

Byte counts: FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B as rocprofv3 prints them for this
chip; on gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes for wide streaming reads, so reads
are reported both raw and doubled and the doubled figure is the one used (it is an upper bound for the
4-byte-per-lane sample loads of this kernel)."""
import collections
import csv
import glob
import json
import os
import sys


def per_dispatch(path, needle):
    acc = collections.defaultdict(float)
    disp = set()
    for r in csv.DictReader(open(path)):
        if needle not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        disp.add(r["Dispatch_Id"])
    n = max(1, len(disp))
    return {k: v / n for k, v in acc.items()}, len(disp)


def main():
    out, seconds = sys.argv[1], int(sys.argv[2])
    samples = seconds * 384000
    summary = {"seconds": seconds, "samples": samples, "kinds": {}}
    traffic = {}
    for kind in ("rgba", "u8"):
        c = {}
        nd = 0
        for f in sorted(glob.glob(os.path.join(out, kind + "_*.csv"))):
            d, n = per_dispatch(f, "stft1024")
            c.update(d)
            nd = max(nd, n)
        if not c:
            continue
        summary["kinds"][kind] = {"dispatches_per_pass": nd, "per_dispatch": {k: round(v, 1) for k, v in sorted(c.items())}}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd_raw = c["FETCH_SIZE"] * 1024.0
            wr = c["WRITE_SIZE"] * 1024.0
            traffic[kind] = {"samples": samples, "read_bytes_raw": int(rd_raw), "read_bytes_corrected": int(2 * rd_raw),
                             "write_bytes": int(wr), "bytes_per_launch": int(2 * rd_raw + wr)}
    ks = os.path.join(out, "bench_kernel_stats.csv")
    if os.path.isfile(ks):
        summary["bench_kernel_stats"] = [r for r in csv.DictReader(open(ks))][:8]
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
