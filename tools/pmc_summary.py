#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> profiles/r03_pmc_traffic.json (HBM bytes per launch of every kernel, per workload).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ev/pmc_fetch_c2 -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/ev/pmc_write_c2 -o w -- python3 bench.py ...
    python tools/pmc_summary.py MATH c2 F.csv W.csv [c3 F.csv W.csv ...]

Kernel names are folded to the labels bench.py's in-library profiler uses (the plane format NS comes from the template
arguments), so the bench can look its dominant kernel up.  FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section) -- both the raw and the
x2-corrected totals are kept.  `max_launch`: the launch of the kernel with the most traffic (the two passes run the same
program, so the k-th launch of a kernel in one is the k-th in the other) -- for kernels whose name carries no shape
(the BatchNorm passes) that is the largest tensor's launch.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def _ns(flag_f16, ns="2"):
    return "4" if flag_f16 == "true" else ns


def fold(name):
    m = re.search(r"conv_fwd_bf16p_kernel<(\d+), (\d+), \d+, \d+, \d+, (true|false), (\d+), \d+, (?:true|false), (true|false)>", name)
    if m:
        return f"conv_fwd_bf16p_kernel<KS={m[1]},BM={m[2]},up2={int(m[3] == 'true')},NS={_ns(m[5], m[4])}>"
    m = re.search(r"conv_fwd_bf16p2_kernel<(\d+), (\d+), (true|false), (?:true|false), (true|false)>", name)
    if m:
        return f"conv_fwd_bf16p2_kernel<LOG2W={m[1]},BM={m[2]},up2={int(m[3] == 'true')},NS={_ns(m[4])}>"
    m = re.search(r"conv_fwd_bf16p3_kernel<(\d+), (\d+), (true|false), \d+, (?:true|false), (?:true|false), (true|false)>", name)
    if m:
        return f"conv_fwd_bf16p3_kernel<LOG2W={m[1]},BM={m[2]},up2={int(m[3] == 'true')},NS={_ns(m[4])}>"
    m = re.search(r"conv_wgrad_bf16p_kernel<(\d+), (true|false), (\d+), \d+, \d+, \d+, \d+, (true|false)>", name)
    if m:
        return f"conv_wgrad_bf16p_kernel<LOG2W={m[1]},BM={m[3]},up2={int(m[2] == 'true')},NS={_ns(m[4])}>"
    m = re.search(r"itcv::(bn_act_fwd_planes_kernel|bn_bwd_apply_planes)<", name)
    if m:
        return m[1]                      # every instantiation together: the bench pairs it with its largest bucket
    m = re.search(r"itcv::(\w+)(<[^(]*>)?\(", name)
    return (m[1] + (m[2] or "")) if m else name[:80]


def collect(path, counter):
    """label -> list of per-launch counter values (in launch order)."""
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[fold(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def summarise(fpath, wpath, math, cfg):
    fetch, write = collect(fpath, "FETCH_SIZE"), collect(wpath, "WRITE_SIZE")
    out = {"math": math, "command": f"bench.py --config {cfg} --math {math} --steps 2 --warmup 1 --no-graph --no-modes --no-cpu-baseline",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(k, []), write.get(k, [])
        fk, wk = (sum(fv) / len(fv) if fv else 0.0), (sum(wv) / len(wv) if wv else 0.0)
        rec = {"launches": max(len(fv), len(wv)), "fetch_kib_per_launch": round(fk, 1), "write_kib_per_launch": round(wk, 1),
               "hbm_bytes_per_launch_raw": int((fk + wk) * 1024), "hbm_bytes_per_launch_fetch_x2": int((2 * fk + wk) * 1024)}
        if fv and len(fv) == len(wv):
            i = max(range(len(fv)), key=lambda j: 2 * fv[j] + wv[j])
            rec["max_launch"] = {"fetch_kib": round(fv[i], 1), "write_kib": round(wv[i], 1),
                                 "hbm_bytes_fetch_x2": int((2 * fv[i] + wv[i]) * 1024)}
        out["kernels"][k] = rec
    return out


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_digest
    math, rest = sys.argv[1], sys.argv[2:]
    out = {"csrc_sha256": csrc_digest(),
           "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains beside the "
                   "kernel trace) over two eager steps of each workload.  KiB per launch averaged over all launches of the "
                   "kernel; fetch_x2 applies the gfx950 FETCH_SIZE correction for wide coalesced reads (16 B/lane LDS-DMA / "
                   "dwordx4 loads, which is what the planes and BatchNorm kernels issue).",
           "configs": {}}
    for i in range(0, len(rest), 3):
        out["configs"][rest[i]] = summarise(rest[i + 1], rest[i + 2], math, rest[i])
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
