#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps of ONE graph-replayed denoising step, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --ddpm-steps 4 --steps 1 --warmup 1 --no-cpu-baseline
    python tools/trace_forward.py gpurun_out/trace [--top 40]

The step taken is the span between the last two `ddpm_step_kernel` dispatches (a hipGraph replay: no host
launch latency in the gaps).
"""
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
path = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "ddpm_step_kernel" in r["Kernel_Name"]]
if len(idx) < 2:
    sys.exit("need at least two ddpm_step_kernel dispatches in the trace")
step = rows[idx[-2] + 1: idx[-1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"one step: {len(step)} dispatches, wall {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, "
      f"gaps {(t1 - t0 - busy) / 1e6:.3f} ms")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("unsigned short", "bf16")


agg = collections.OrderedDict()
classes = collections.defaultdict(lambda: [0, 0.0])
prev_end = None
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    if prev_end is not None:
        a[2] += max(0, s - prev_end) / 1e3
    prev_end = e
    c = classes[key[0].split("<")[0]]
    c[0] += 1
    c[1] += (e - s) / 1e3
print("\nby kernel family:")
for k, (n, us) in sorted(classes.items(), key=lambda kv: -kv[1][1]):
    print(f"  {us / 1e3:8.3f} ms  x{n:<4d} {us / n:8.2f} us avg  {k}")
print(f"\ntop {top} (kernel, blocks, grid.y): total us, count, avg us, avg gap before (us)")
for key, (n, us, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {us:9.1f}  x{n:<3d} {us / n:8.2f}  gap {gap / n:5.2f}  {key[0]} blocks={key[1]} y={key[2]}")
