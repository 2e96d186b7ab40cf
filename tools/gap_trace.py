"""Timeline between two consecutive launches of the fused flow step, from a rocprofv3 kernel trace (csv): what runs in the
gap and how long the GPU idles there.  usage: gap_trace.py <bench_kernel_trace.csv> [kernel-name-prefix]"""
import csv
import sys
from collections import Counter, defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "void k_pcn_flow_fused"
idx = [i for i, r in enumerate(rows) if r[2].startswith(key)]
gaps, inside, pat = [], defaultdict(list), Counter()
timelines = []
bound = []  # temperature boundaries: (wall between the two fused launches, GPU busy inside, launches inside)
for a, b in zip(idx[:-1], idx[1:]):
    between = rows[a + 1:b]
    if len(between) > 6:  # a temperature boundary, not a step boundary
        bound.append((rows[b][0] - rows[a][1], sum(e - s for s, e, _ in between), len(between)))
        timelines.append((rows[b][0] - rows[a][1], rows[a][1], between + [rows[b]]))
        continue
    gap = rows[b][0] - rows[a][1]
    busy = sum(e - s for s, e, _ in between)
    gaps.append((gap, busy))
    pat[tuple(n.split("(")[0][:40] for _, _, n in between)] += 1
    for s, e, n in between:
        inside[n.split("(")[0][:40]].append(e - s)
gaps.sort()
m = len(gaps)
print(f"{m} step boundaries; gap fused->fused median {gaps[m // 2][0] / 1e3:.1f} us, mean {sum(g for g, _ in gaps) / m / 1e3:.1f} us; "
      f"busy inside mean {sum(b for _, b in gaps) / m / 1e3:.1f} us")
for p, c in pat.most_common(4):
    print(c, p)
for n, v in inside.items():
    print(f"  {n}: {len(v)} x {sum(v) / len(v) / 1e3:.1f} us")

bound = [b for b in bound if b[0] < 20e6]  # drop the gaps between runs (host-side set-up)
if bound:
    bound.sort()
    m = len(bound)
    print(f"{m} temperature boundaries (importance step + reference fit, between two fused launches): median wall "
          f"{bound[m // 2][0] / 1e3:.0f} us, GPU busy {bound[m // 2][1] / 1e3:.0f} us, {bound[m // 2][2]} launches; mean wall "
          f"{sum(b[0] for b in bound) / m / 1e3:.0f} us, busy {sum(b[1] for b in bound) / m / 1e3:.0f} us")

import os
if os.environ.get("ALL") and bound:  # the distribution: wall us / busy us / launches of every boundary, sorted by wall
    print("all boundaries (wall/busy/launches):", " ".join(f"{b[0] / 1e3:.0f}/{b[1] / 1e3:.0f}/{b[2]}" for b in bound))
if os.environ.get("TIMELINE") and timelines:
    timelines = [t for t in timelines if t[0] < 20e6]
    timelines.sort(key=lambda t: t[0])
    which = len(timelines) * 9 // 10 if os.environ.get("TIMELINE") == "p90" else len(timelines) // 2
    wall, t0, ks = timelines[which]
    print(f"-- the {'90th-percentile' if which != len(timelines) // 2 else 'median'} boundary ({wall / 1e3:.0f} us): start offset, duration, idle before, kernel")
    prev = t0
    for s_, e_, n_ in ks:
        print(f"{(s_ - t0) / 1e3:8.1f} {(e_ - s_) / 1e3:7.1f} {(s_ - prev) / 1e3:7.1f}  {n_.split('(')[0][:70]}")
        prev = e_
