#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run into profiles/<tag>_kernel_stats.{csv,md}.

    python tools/summarize_rocprof.py gpurun_out/prof_bench_f32 r01_bench_config2_f32 "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
"""
import collections
import csv
import glob
import os
import re
import shutil
import sys

src, tag, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
shutil.copy(stats, os.path.join(root, "profiles", tag + "_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    m = re.match(r"(?:void )?([\w:]+)", short)
    agg[m.group(1)][0] += int(r["Calls"])
    agg[m.group(1)][1] += float(r["TotalDurationNs"])
with open(os.path.join(root, "profiles", tag + "_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats: {tag}\n\ncommand: `{cmd}`\n\n")
    f.write(f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n\n")
    f.write("| kernel (all template instantiations) | calls | total ms | avg us | share |\n|---|---|---|---|---|\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
        f.write(f"| {k} | {v[0]} | {v[1]/1e6:.2f} | {v[1]/v[0]/1e3:.2f} | {100*v[1]/tot:.1f}% |\n")
    f.write("\n## top instantiations\n\n| kernel | calls | avg us | share |\n|---|---|---|---|\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\(GemmArgs\)|\(AttnArgs\)", "", name)[:110]
        f.write(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.1f}% |\n")
print(open(os.path.join(root, "profiles", tag + "_kernel_stats.md")).read()[:1500])
