#!/usr/bin/env python3
"""Fold the two rocprofv3 PMC passes of tools/profile_round.sh into profiles/<tag>_traffic.json.

usage: python tools/traffic_json.py gpurun_out/prof_<tag> profiles/<tag>_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes
of wide coalesced 16-B/lane reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B/lane stores.
"""
import collections
import csv
import json
import sys

KEYS = {"gemm_bf3_kernel<0": "gemm_bf3_kernel", "gemm_bf3_kernel<1": "gemm_bf3_kernel<1>", "attn_bf3_kernel": "attn_bf3_kernel",
        "gemm_kernel<0": "gemm_kernel<0>", "gemm_kernel<1": "gemm_kernel<1>", "attn_kernel": "attn_kernel",
        "align_main_kernel": "align_main_kernel", "layernorm_kernel": "layernorm_kernel", "layernorm_pair_kernel": "layernorm_kernel",
        "layernorm_fh2_kernel": "layernorm_fh2_kernel", "attn_fh2_kernel": "attn_fh2_kernel", "attn_fh2_v2_kernel": "attn_fh2_kernel", "upsample2x_kernel": "upsample2x_kernel",
        "head_final_kernel": "head_final_kernel", "head_final128_kernel": "head_final_kernel"}


def classify(name):
    """kernel name of the trace -> key of the json; the fh2 GEMM's LAST template argument says linear (0) or implicit conv (1)."""
    n = name.replace("a3r::", "").replace("void ", "")
    if "gemm_fh2_kernel<" in n:
        args = n[n.index("<") + 1:n.index(">")].split(",")
        # template <BM, BN, WM, WN, NS, FULL, AMODE[, PASSES]>: AMODE 1 is the implicit 3x3 convolution
        amode = args[6].strip() if len(args) > 6 else "0"
        return "gemm_fh2_kernel<1>" if amode == "1" else "gemm_fh2_kernel"
    for k, v in KEYS.items():
        if k in n:
            return v
    return None


def collect(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = classify(r["Kernel_Name"])
        if key:
            per[key].append(float(r["Counter_Value"]))
    return per


def main(src, dst):
    f = collect(f"{src}/fetch/f_counter_collection.csv", "FETCH_SIZE")
    w = collect(f"{src}/write/w_counter_collection.csv", "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (each with --kernel-trace only) over "
                   "`python bench.py --no-cpu-baseline --steps 2 --warmup 1 --align-iters 10 --no-cache-run` (the bench's default 42 pairs/step, 512x384). "
                   "Counters are KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): read bytes = 2 * FETCH_SIZE * 1024 for wide "
                   "coalesced reads; WRITE_SIZE exact. Infinity-Cache hits are counted: fabric traffic, an upper bound on HBM traffic.",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fa = sum(f[k]) / len(f[k]) if f.get(k) else 0.0
        wa = sum(w[k]) / len(w[k]) if w.get(k) else 0.0
        out["kernels"][k] = {"launches_profiled": len(f.get(k, [])), "FETCH_SIZE_KiB_per_launch": round(fa, 1),
                             "WRITE_SIZE_KiB_per_launch": round(wa, 1), "traffic_bytes_per_launch": int(2 * fa * 1024 + wa * 1024)}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
