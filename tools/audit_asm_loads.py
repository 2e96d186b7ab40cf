"""Static audit of the hand-issued LDS reads in k_pcn_flow_fused.  hipcc counts an inline-asm load's
destination as written at the end of the asm statement, so any instruction that touches the destination between the
`ds_read_b128` and the asm `s_waitcnt` of its batch (a copy, a spill, a reuse by the register allocator) would read or
clobber data still in flight.  The batches are waited for in the order they are issued, so the check walks every
instantiation's assembly with a FIFO: a read statement pushes its four destinations, a wait statement releases the oldest
batch, and nothing outside those statements may name a register that is still in the FIFO.
usage: audit_asm_loads.py [extra hipcc flags]  |  audit_asm_loads.py --asm <device assembly of asmc_pcn_fused.hip>
The library build runs the second form on the assembly of the very compilation that produced the object (csrc/Makefile), so
every flag combination that builds is audited, and a violation fails the build."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "aspire_amd", "csrc", "asmc_pcn_fused.hip")
out = "/tmp/audit_fused.s"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
       "-Wno-unused-function", "-S", "--cuda-device-only", src, "-o", out] + sys.argv[1:]
if len(sys.argv) >= 3 and sys.argv[1] == "--asm":  # audit an assembly file the build already produced (csrc/Makefile: -save-temps)
    out = sys.argv[2]
else:
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


kernel, in_asm, fifo, cur, bad, n_batches, n_kern = None, False, [], None, 0, 0, 0
for ln, line in enumerate(open(out), 1):
    t = line.strip()
    if re.match(r"^_Z\d+k_pcn_flow_fused.*:", t):
        if fifo:
            print(f"{kernel}: {len(fifo)} batches never waited for")
            bad += 1
        kernel, fifo = t.split(":")[0][:70], []
        n_kern += 1
        continue
    if kernel is None or not t:
        continue
    if "#ASMSTART" in t:
        in_asm, cur = True, set()
        continue
    if "#ASMEND" in t:
        in_asm = False
        if cur:
            fifo.append(cur)
            n_batches += 1
        cur = None
        continue
    if t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    if t.startswith("s_endpgm"):
        if fifo:
            print(f"{kernel}: {len(fifo)} batches never waited for")
            bad += 1
        kernel, fifo = None, []
        continue
    toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
    used = set().union(*[regs(x) for x in toks]) if toks else set()
    inflight = set().union(*fifo) if fifo else set()
    if in_asm and t.startswith("ds_read_b128"):
        dest = regs(toks[0])
        if dest & inflight or dest & cur:
            print(f"line {ln} {kernel}: read overwrites a destination still in flight: {t}")
            bad += 1
        cur |= dest
        continue
    if in_asm and t.startswith("s_waitcnt") and fifo:
        fifo.pop(0)
        continue
    if used & inflight:
        print(f"line {ln} {kernel}: touches a destination before its wait: {t}")
        bad += 1
print(f"{n_kern} instantiations, {n_batches} read batches, {bad} violations")
sys.exit(1 if bad else 0)
