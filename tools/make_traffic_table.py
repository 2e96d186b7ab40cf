#!/usr/bin/env python
"""Build profiles/<tag>_pmc_traffic.md and refresh profiles/traffic_per_launch.json from the rocprofv3 PMC
summaries written by tools/prof_round.sh (FETCH_SIZE and WRITE_SIZE collected in separate passes; raw unit KB).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half of the bytes of wide coalesced streaming
reads, so the fetch column is shown raw and doubled."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01c"
src = os.path.join(ROOT, "gpurun_out", tag)


def parse(path):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        else:
            m = re.search(r"mean=([0-9.e+]+)", line)
            if m and name:
                out[name] = float(m.group(1))
    return out


n, d = 1_000_000, 32
alg = {"k_bis_sums": 24 * n, "k_weights_max<1>": 24 * n, "k_weights_sums<1>": 24 * n, "k_weights_m2_lse": 24 * n,
       "k_weights_map<1>": 32 * n, "k_tile_sum": 8 * n, "k_exact_tile_td_launch": 8 * n, "k_exact_tile_write": 16 * n,
       "k_divide_dev": 16 * n, "k_pcg64_uniforms": 8 * n, "k_search": 24 * n, "k_gather16": (2 * (d * 8 + 24) + 8) * n, "k_search_guided": 24 * n, "k_guide_build": 8 * n,
       "k_is_weights": 32 * n, "k_exact_tile_td_scan": 8 * n, "k_search_pcg": 16 * n, "k_pcn_flow_fused": (2 * d * 8 + 48) * n,  # 2 d s + three log-density arrays read and (accepted rows) written
       "k_pcn_reg<double, 32, 1, 1>": (2 * d * 8 + 16) * n, "k_pcn_reg_flow<double, 32, 1, 0>": (2 * d * 8 + 16) * n,
       "k_pcn_reg_flow<double, 32, 1, 1>": (2 * d * 8 + 48) * n, "k_coupling_logprob<16, 64, double, 512, 2>": (d * 8 + 8) * n}
rows = []
for kind in ("bench", "kbench"):
    f = parse(os.path.join(src, f"pmc_{kind}_FETCH_SIZE.txt"))
    w = parse(os.path.join(src, f"pmc_{kind}_WRITE_SIZE.txt"))
    for name in f:
        if not (name.startswith("k_") or name.startswith("void k_")):
            continue
        short = name.replace("void ", "")
        key = next((k for k in alg if short.startswith(k)), None)
        rows.append((kind, short[:60], f[name] * 1e3, w.get(name, 0.0) * 1e3, alg.get(key)))
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.md"), "w") as fh:
    fh.write(f"# HBM traffic per launch from rocprofv3 PMC ({tag})\n\n"
             "Commands (separate passes per counter, as MI355X_MICROARCH.md §HBM prescribes), see tools/prof_round.sh:\n"
             "`rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1` (bench)\n"
             "and `... -- python3 tools/kbench.py gather pcn cdf weights flow` (kbench); the same with `--pmc WRITE_SIZE`.\n"
             "N = 1,000,000 particles, d = 32, fp64 x. Raw counters are KB, per-launch means. gfx950 correction: FETCH_SIZE\n"
             "reports half of the bytes of a wide coalesced streaming read, hence the `fetch x2` column; for the random row\n"
             "gather and the binary search the raw value is the one consistent with the bytes they can touch (x2 would\n"
             "exceed N x row bytes), so `traffic_per_launch.json` takes those two raw. `bench` rows average over every\n"
             "launch of a kernel in the run (including the small warm-up populations); `kbench` rows are pure 1M launches.\n\n"
             "| run | kernel | FETCH raw MB | fetch x2 MB | WRITE MB | algorithmic MB |\n|---|---|---|---|---|---|\n")
    for kind, name, fb, wb, ab in rows:
        fh.write(f"| {kind} | `{name}` | {fb/1e6:.1f} | {2*fb/1e6:.1f} | {wb/1e6:.1f} | {'' if ab is None else f'{ab/1e6:.0f}'} |\n")
tr = {"_comment": f"HBM bytes per launch (FETCH x2 + WRITE) from profiles/{tag}_pmc_traffic.md; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"}
for kind, name, fb, wb, ab in rows:
    if kind != "bench":
        continue
    base = name.split("(")[0]
    base = re.sub(r"_pow2<\d+>$", "", base)  # bench.py labels every gather variant k_gather16
    base = re.sub(r"<\d+>$", "<KT>", base) if base.startswith("k_weights_max") or base.startswith("k_weights_sums") else base
    # random row gather / binary search: the raw counter already matches the bytes those kernels can physically
    # touch (x2 would exceed them), so they are taken raw; streaming kernels are doubled
    mult = 1 if base.startswith(("k_gather16", "k_search")) else 2
    tr[f"{base}|n={n}|d={d}|f64"] = round(mult * fb + wb, -5)
    if base.startswith(("k_pcn_flow_fused<", "k_is_weights<")):  # bench.py looks these up without the template arguments
        tr.setdefault(f"{base.split('<')[0]}|n={n}|d={d}|f64", round(mult * fb + wb, -5))  # (the first row: the headline instantiation)
json.dump(tr, open(os.path.join(ROOT, "profiles", "traffic_per_launch.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.md")).read())
