#!/usr/bin/env python
"""Stamp profiles/traffic_per_launch.json and profiles/sq_counters.json with the hash of the dominant kernel's sources
(bench.kernel_source_hash): run right after the counters were re-collected on the current tree.  bench.py nulls
`roofline.traffic` / `valu_active` / `mfma_busy` when the stamp does not match the tree it runs on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

h = kernel_source_hash()
for name in sys.argv[1:] or ("traffic_per_launch.json", "sq_counters.json"):
    p = os.path.join(ROOT, "profiles", name)
    d = json.load(open(p))
    d["_kernel_source_hash"] = h
    json.dump(d, open(p, "w"), indent=1)
    print(name, "stamped", h)
