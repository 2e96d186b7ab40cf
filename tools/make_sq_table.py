#!/usr/bin/env python
"""profiles/sq_counters.json from the rocprofv3 --pmc summaries of tools/prof_r05.sh (tools/pmc_summary.py format): fractions of
SIMD time = counter / (SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs); SQ_ACTIVE_INST_* and SQ_WAIT_* count quad-cycles.
usage: tools/make_sq_table.py <tag>   (reads gpurun_out/pmc_flowstep_<tag>_{hi,mid}/summary.txt and gpurun_out/<tag>/pmc_*.txt)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def parse(path, kernel_prefix):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        elif name and name.replace("void ", "").startswith(kernel_prefix):
            m = re.match(r"\s+(\S+)\s+n=\s*\d+ mean=([0-9.e+]+)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
    return out


def entry(c, units, source, unit_name):
    simd = c["SQ_BUSY_CYCLES"] / 32 * 1024
    e = {"valu_active": round(c["SQ_ACTIVE_INST_VALU"] * 4 / simd, 3), "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 3),
         f"valu_insts_per_{unit_name}": round(c["SQ_INSTS_VALU"] / units), f"mfma_insts_per_{unit_name}": round(c["SQ_INSTS_MFMA"] / units),
         f"lds_insts_per_{unit_name}": round(c["SQ_INSTS_LDS"] / units), f"salu_insts_per_{unit_name}": round(c["SQ_INSTS_SALU"] / units),
         "wait_inst_any_of_wave_cycles": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
         "wait_any_of_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
         f"simd_cycles_per_{unit_name}": round(simd / units), "source": source}
    if "FETCH_SIZE" in c:
        e["hbm_bytes_per_launch"] = round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1e3)  # KB; FETCH_SIZE reports half of a wide streaming read
    return e


out = {"_comment": "SQ counters of the step kernels, rocprofv3 --pmc in separate passes (tools/prof_r05.sh); fractions of SIMD time = counter / "
                   "(SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs); SQ_ACTIVE_INST_* count quad-cycles; 'hi': RHO=0.02 ADAPT=0 (the "
                   "headline's 98 % acceptance), 'accept47': adapted step size"}
n = 1_000_000
for mode, suffix in (("hi", ""), ("mid", "|accept47")):
    src = f"gpurun_out/pmc_flowstep_{tag}_{mode}/summary.txt"
    c = parse(os.path.join(ROOT, src), "k_pcn_flow_fused<double, 64, 0, true, 0, false>")
    e = entry(c, (n + 63) // 64, f"profiles/{tag}_pmc_fused_step_{mode}.txt", "tile")
    # (the key names bench.py reads)
    e["valu_insts_per_tile"], e["mfma_insts_per_tile"] = e["valu_insts_per_tile"], e["mfma_insts_per_tile"]
    out[f"k_pcn_flow_fused|n={n}|d=32|f64|f64{suffix}"] = e
c = parse(os.path.join(ROOT, f"gpurun_out/{tag}/pmc_flow16_d64.txt"), "k_pcn_flow16<double, 64, 64, 0, 0, false")
out[f"k_pcn_flow16|n={n}|d=64|f64|f64|coupling"] = entry(c, (n + 15) // 16, f"profiles/{tag}_pmc_flow16_d64.txt", "16_particle_group")
c = parse(os.path.join(ROOT, f"gpurun_out/{tag}/pmc_config5_step.txt"), "k_pcn_mm<double, 128, 0, 3>")
out[f"k_tpcn_mm_step|n={n}|d=128|f64|f64"] = entry(c, (n + 15) // 16, f"profiles/{tag}_pmc_config5_step.txt", "16_particle_group")
p_maf = os.path.join(ROOT, f"gpurun_out/{tag}/pmc_flow16_maf_d128.txt")  # (round 6: the reference's default flow class at configs[4]'s dimension)
if os.path.exists(p_maf):
    c = parse(p_maf, "k_pcn_flow16<double, 128, 64, 1, 0, false")
    out[f"k_pcn_flow16|n={n}|d=128|f64|f64|maf"] = entry(c, (n + 15) // 16, f"profiles/{tag}_pmc_flow16_maf_d128.txt", "16_particle_group")
json.dump(out, open(os.path.join(ROOT, "profiles", "sq_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
