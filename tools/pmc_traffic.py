#!/usr/bin/env python3
"""HBM traffic per launch of each kernel class from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d F -- python3 tools/prof_forward.py --config 2
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d W -- python3 tools/prof_forward.py --config 2
    python tools/pmc_traffic.py F W profiles/r01_pmc_traffic_config2.json "config 2"

Corrections for gfx950: both counters are in KiB; FETCH_SIZE counts 64 B per 128-B request for wide
(16 B/lane) coalesced reads, which is what every kernel here issues, so the read side is doubled.
Only the LAST forward of the run is used (prof_forward runs a warm-up forward first).
"""
import collections
import csv
import glob
import json
import re
import sys

fdir, wdir, out, label = sys.argv[1:5]
order_file = sys.argv[5] if len(sys.argv) > 5 else None  # launch order written by prof_forward.py --dump-order


def load(d, counter):
    path = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def klass(name):
    """Kernel class of a dispatch; the conv3x3 kernels (generic ring, 256x256 ping-pong tile, fp32 form that
    normalises its operand) are one class, as are the dense GEMM kernels."""
    m = re.search(r"(conv3x3_igemm_kernel|conv3x3_wide_kernel|conv3x3_gn_kernel|conv3x3_pair_kernel|conv3x3_chunk_kernel|"
                  r"conv3x3_chunk_pair_kernel|conv3x3_halo_kernel|gemm_dma_kernel|"
                  r"gemm_wide_kernel|gemm_gn_kernel|gemm_pair_kernel|gemm_panel_kernel|ffn_geglu_kernel|attn_kernel|"
                  r"gn_stats_kernel|gn_apply_kernel|layernorm_kernel|splitk_reduce_kernel|linear_rows_kernel)", name)
    if not m:
        return None
    return {"conv3x3_igemm_kernel": "conv3x3", "conv3x3_wide_kernel": "conv3x3", "conv3x3_gn_kernel": "conv3x3",
            "conv3x3_pair_kernel": "conv3x3", "conv3x3_chunk_kernel": "conv3x3", "conv3x3_chunk_pair_kernel": "conv3x3", "conv3x3_halo_kernel": "conv3x3",
            "gemm_dma_kernel": "gemm", "gemm_wide_kernel": "gemm", "gemm_gn_kernel": "gemm",
            "gemm_pair_kernel": "gemm", "gemm_panel_kernel": "gemm", "ffn_geglu_kernel": "gemm"}.get(m.group(1), m.group(1))


def last_forward(rows):
    """Dispatches of the final forward: everything after the last nchw_to_rows launch (first kernel of a forward's U-Net)."""
    idx = max(i for i, r in enumerate(rows) if "linear_rows_kernel" in r["Kernel_Name"])
    # a forward starts with 3 linear_rows launches; step back to the first of that run
    while idx > 0 and "linear_rows_kernel" in rows[idx - 1]["Kernel_Name"]:
        idx -= 1
    return rows[idx:]


res = collections.defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
for r in last_forward(load(fdir, "FETCH_SIZE")):
    k = klass(r["Kernel_Name"])
    if k:
        res[k]["launches"] += 1
        res[k]["fetch_bytes"] += float(r["Counter_Value"]) * 1024 * 2
for r in last_forward(load(wdir, "WRITE_SIZE")):
    k = klass(r["Kernel_Name"])
    if k:
        res[k]["write_bytes"] += float(r["Counter_Value"]) * 1024
summary = {"label": label, "unit": "bytes", "corrections": "KiB->bytes; FETCH_SIZE x2 (gfx950 halves wide coalesced reads)",
           "per_forward": {}, "per_launch": {}}
for k, v in res.items():
    tot = v["fetch_bytes"] + v["write_bytes"]
    summary["per_forward"][k] = {"launches": v["launches"], "fetch": v["fetch_bytes"], "write": v["write_bytes"], "total": tot}
    summary["per_launch"][k] = tot / max(v["launches"], 1)
if order_file:
    # conv3x3 traffic by layer shape: the i-th conv dispatch of the forward is the i-th conv launch the engine recorded
    tags = json.load(open(order_file)).get("conv3x3", [])
    fe = [float(r["Counter_Value"]) * 2048 for r in last_forward(load(fdir, "FETCH_SIZE")) if klass(r["Kernel_Name"]) == "conv3x3"]
    wr = [float(r["Counter_Value"]) * 1024 for r in last_forward(load(wdir, "WRITE_SIZE")) if klass(r["Kernel_Name"]) == "conv3x3"]
    if len(tags) == len(fe) == len(wr):
        esz = 2 if "bf16" in label else 4
        by = collections.OrderedDict()
        for tag, f_, w_ in zip(tags, fe, wr):
            kv = dict(t.split("=") for t in tag.split())
            m, n, k = int(kv["m"]), int(kv["n"]), int(kv["k"])
            rows_in = m // 4 if kv.get("up") == "1" else m * (4 if kv.get("s") == "2" else 1)
            alg = (rows_in * (k // 9) + n * k + m * n) * esz  # input once + weights once + output once (folded 1x1 terms ignored)
            key = " ".join(f"{a}={kv[a]}" for a in ("m", "n", "k", "s", "up"))
            d = by.setdefault(key, {"launches": 0, "fetch": 0.0, "write": 0.0, "algorithmic": 0.0})
            d["launches"] += 1
            d["fetch"] += f_
            d["write"] += w_
            d["algorithmic"] += alg
        summary["conv3x3_by_shape"] = by
        print("conv3x3 by shape (MB per launch): fetch, write, algorithmic, ratio")
        for key, d in sorted(by.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
            L = d["launches"]
            print(f"  {key:44s} x{L}  {d['fetch']/L/1e6:8.1f} {d['write']/L/1e6:7.1f} {d['algorithmic']/L/1e6:8.1f}  "
                  f"{(d['fetch'] + d['write']) / d['algorithmic']:5.2f}")
    else:
        print(f"launch order has {len(tags)} convs, the counter passes {len(fe)} / {len(wr)}: no per-shape table")
json.dump(summary, open(out, "w"), indent=1)
for k, v in sorted(summary["per_forward"].items(), key=lambda kv: -kv[1]["total"]):
    print(f"{k:24s} x{v['launches']:3d}  fetch {v['fetch']/1e6:9.1f} MB  write {v['write']/1e6:9.1f} MB  per launch {summary['per_launch'][k]/1e6:8.2f} MB")
