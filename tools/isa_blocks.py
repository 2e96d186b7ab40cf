#!/usr/bin/env python
"""Per-basic-block instruction histogram of ONE kernel in a hipcc -S listing - the reading that found round 6's kernel changes
(per-quad dimension tests, a hoisted bias dword, address adds for ds_read2_b64, scalar loads parked in vector lanes).

  hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -S --cuda-device-only asmc_pcn_fused.hip -o /tmp/fused.s
  tools/isa_blocks.py /tmp/fused.s _Z16k_pcn_flow_fusedIdLi64ELi0ELb1ELi0ELb0E            # one line per block of >= 40 instructions
  tools/isa_blocks.py /tmp/fused.s _Z16k_pcn_flow_fusedIdLi64ELi0ELb1ELi0ELb0E .LBB0_177  # opcode histogram of that block
  tools/isa_blocks.py /tmp/fused.s _Z16k_pcn_flow_fusedIdLi64ELi0ELb1ELi0ELb0E .LBB0_177! # its text
  tools/isa_blocks.py /tmp/pcn.s --spills                                                  # every kernel: lane moves, scratch, registers

Columns: n instructions, valu (vector, without MFMA), mov (v_mov / v_readlane / v_writelane), mfma, ds (LDS), scr (scratch), wait
(s_waitcnt), nop, bar (s_barrier), the loop the block belongs to, its branches.  A block's fall-through code behind a conditional
branch is counted with the block (labels delimit blocks): check the text before reading a count as 'executed'."""
import collections
import re
import subprocess
import sys


def split_blocks(body):
    blocks, cur = [], ["entry", [], ""]
    for line in body.splitlines():
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", line)
        if m:
            blocks.append(cur)
            cur = [m.group(1), [], m.group(2).strip()]
        else:
            t = line.strip()
            if t and not t.startswith(";") and not t.startswith("."):
                cur[1].append(t)
    blocks.append(cur)
    return blocks


def kernel_body(src, key):
    a = src.index(key)
    a = src.index("\n", a)
    b = src.index(".Lfunc_end", a)
    return src[a:b], src[b:b + 2500]


def count(ins):
    c = collections.Counter(x.split()[0] for x in ins)
    pre = lambda *p: sum(n for k, n in c.items() if k.startswith(p))  # noqa: E731
    return c, dict(valu=pre("v_") - pre("v_mfma"), mov=pre("v_mov", "v_readlane", "v_writelane"), mfma=pre("v_mfma"), ds=pre("ds_"),
                   scr=pre("scratch"), wait=c.get("s_waitcnt", 0), nop=c.get("s_nop", 0), bar=c.get("s_barrier", 0))


def main():
    src = open(sys.argv[1]).read()
    if sys.argv[2] == "--spills":
        for m in re.finditer(r"^(_Z\S+):\s*; @", src, re.M):
            body, tail = kernel_body(src, m.group(1) + ":")
            c, k = count([ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((";", "."))])
            meta = dict(re.findall(r"; (NumVgprs|ScratchSize|Occupancy): (\d+)", tail)[:3])
            lanes = c.get("v_readlane_b32", 0) + c.get("v_writelane_b32", 0)
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:100]
            print(f"{name:100s} valu={k['valu']:5d} lane-moves={lanes:4d} scratch={meta.get('ScratchSize')} vgpr={meta.get('NumVgprs')} occ={meta.get('Occupancy')}")
        return
    body, _ = kernel_body(src, sys.argv[2])
    blocks = split_blocks(body)
    if len(sys.argv) > 3:
        byname = {b[0]: b for b in blocks}
        for want in sys.argv[3:]:
            text = want.endswith("!")
            ins = byname[want.rstrip("!")][1]
            if text:
                print("\n".join(ins))
            else:
                print(want, len(ins), sorted(count(ins)[0].items(), key=lambda kv: -kv[1]))
        return
    for name, ins, comment in blocks:
        if len(ins) < 40:
            continue
        _, k = count(ins)
        br = [x for x in ins if x.startswith(("s_cbranch", "s_branch"))]
        print(f"{name:11s} n={len(ins):4d} valu={k['valu']:4d} mov={k['mov']:3d} mfma={k['mfma']:3d} ds={k['ds']:3d} scr={k['scr']:2d} wait={k['wait']:3d} "
              f"nop={k['nop']:3d} bar={k['bar']} {comment[2:30]:28s} {' '.join(b.split()[0][2:] + '->' + b.split()[-1] for b in br)}")


if __name__ == "__main__":
    main()
