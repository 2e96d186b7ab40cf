"""Print the headline numbers of a bench.py JSON line (helper for gpurun one-liners)."""
import json
import sys

b = json.loads(open(sys.argv[1]).read())
print("value %.4g  ms/run %.2f  ms/mutation-step %.4f  rms z %.2f  accept %.3f" % (
    b["value"], b["ms_per_step"], b["ms_per_mutation_step"], b["log_evidence"]["rms_z"], b["mean_accept"]))
r = b["roofline"]
print({k: r[k] for k in r if k not in ("per_kernel",)})
for k, v in list(r["per_kernel"].items())[:10]:
    print(k, v)
if "extra" in b:
    e = b["extra"]
    for k, v in e.items():
        if k == "is_only_step":
            print("is_only", {kk: vv for kk, vv in v.items() if kk not in ("per_kernel_us", "scalars", "workload")})
            print("  ", v["per_kernel_us"])
        else:
            print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()})
if "cpu_baseline" in b:
    c = b["cpu_baseline"]
    print("cpu", c["value"], c["cores"], c["single_thread"]["value"])
