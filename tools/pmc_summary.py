#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> profiles/r02_pmc_traffic.json (HBM bytes per launch of every kernel).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv

Kernel names are folded to the labels bench.py's in-library profiler uses, so the bench can look its dominant
kernel up.  FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section) -- both the raw and the x2-corrected totals are kept.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def fold(name):
    m = re.search(r"conv_fwd_bf16p_kernel<(\d+), (\d+), \d+, \d+, \d+, (true|false), (\d+), \d+(?:, [^>]*)?>", name)
    if m:
        return f"conv_fwd_bf16p_kernel<KS={m[1]},BM={m[2]},up2={int(m[3] == 'true')},NS={m[4]}>"
    m = re.search(r"conv_fwd_bf16p2_kernel<(\d+), (\d+), (true|false)(?:, [^>]*)?>", name)
    if m:
        return f"conv_fwd_bf16p2_kernel<LOG2W={m[1]},BM={m[2]},up2={int(m[3] == 'true')},NS=2>"
    m = re.search(r"conv_fwd_bf16p3_kernel<(\d+), (\d+), (true|false), \d+(?:, [^>]*)?>", name)
    if m:
        return f"conv_fwd_bf16p3_kernel<LOG2W={m[1]},BM={m[2]},up2={int(m[3] == 'true')},NS=2>"
    m = re.search(r"conv_wgrad_bf16p_kernel<(\d+), (true|false), (\d+)(?:, \d+)*>", name)
    if m:
        return f"conv_wgrad_bf16p_kernel<LOG2W={m[1]},BM={m[3]},up2={int(m[2] == 'true')},NS=2>"
    m = re.search(r"itcv::(\w+)(<[^(]*>)?\(", name)
    return (m[1] + (m[2] or "")) if m else name[:80]


def collect(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[fold(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_digest
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    math = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
    out = {"csrc_sha256": csrc_digest(),
           "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                   "`bench.py --steps 2 --warmup 1 --no-graph --no-modes --no-cpu-baseline`, c2 workload. KiB per launch averaged "
                   "over all launches of the kernel; fetch_x2 applies the gfx950 FETCH_SIZE correction for wide "
                   "coalesced reads (16 B/lane LDS-DMA / dwordx4 loads, which is what the planes kernels issue).",
           "math": math, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(k, [0, 0.0])
        nw, sw = write.get(k, [0, 0.0])
        fk, wk = (sf / nf if nf else 0.0), (sw / nw if nw else 0.0)
        out["kernels"][k] = {"launches": max(nf, nw), "fetch_kib_per_launch": round(fk, 1),
                             "write_kib_per_launch": round(wk, 1),
                             "hbm_bytes_per_launch_raw": int((fk + wk) * 1024),
                             "hbm_bytes_per_launch_fetch_x2": int((2 * fk + wk) * 1024)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
