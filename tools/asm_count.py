"""tools/asm_count.py FILE.s SUBSTRING: static instruction mix of the kernel whose mangled name contains SUBSTRING (whole
kernel body: prologue + loops + epilogue, every instruction counted once)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
names = [l for l in re.findall(r"^(_Z\S+):", txt, re.M) if sys.argv[2] in l]
for name in names[: int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    body = txt[txt.index(name + ":"):]
    body = body[: body.index("s_endpgm")]
    ins = []
    for l in body.splitlines():
        t = l.strip()
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        ins.append(t.split()[0])
    c = collections.Counter()
    for i in ins:
        if "mfma_f64" in i:
            k = "mfma_f64"
        elif "mfma" in i:
            k = "mfma_f16"
        elif i.startswith("v_"):
            k = "valu"
        elif i.startswith(("ds_", "global_", "buffer_", "scratch_")):
            k = "_".join(i.split("_")[:2])
        elif i.startswith("s_") and not i.startswith(("s_waitcnt", "s_nop", "s_barrier")):
            k = "salu"
        else:
            k = i
        c[k] += 1
    print(name[:90])
    print(" total", sum(c.values()), dict(c.most_common(16)))
    v = collections.Counter(i for i in ins if i.startswith("v_") and "mfma" not in i)
    print(" valu:", dict(v.most_common(28)))
