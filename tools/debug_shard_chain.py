"""Debug aid: host emulation of k_exact_chain_pk over the packed tile records (prints the first tile that fails)."""
import math
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_parity import _weights  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402

TWO53 = 1 << 53


def binade(s):
    return math.frexp(s)[1] - 1 if s > 0 else -1022


def main():
    n, bounds, kind = int(sys.argv[1]), [int(v) for v in sys.argv[2].split(",")], sys.argv[3]
    eng = HipEngine(0, n_max=1 << 21, d_max=4)
    w = _weights(n, 23 + n % 1000, kind)
    cuts = [0] + bounds + [n]
    recs = []
    for r in range(len(cuts) - 1):
        ws = eng.asarray(w[cuts[r]:cuts[r + 1]])
        cdf, rec = eng.cdf_shard_records(ws, float(np.sum(w[:cuts[r]])), r == 0)
        recs.append(rec.cpu().numpy())
    pk = np.concatenate(recs)
    ref = np.cumsum(w)
    # global element index where each tile starts
    starts = []
    for r in range(len(cuts) - 1):
        nt = recs[r].shape[0]
        starts += [cuts[r] + 2048 * t for t in range(nt)]
    s, t = 0.0, 0
    if pk[0, 3] == 3:
        s = pk[0, 8:9].view(np.float64)[0]
        t = 1
    while t < len(pk):
        a0, a1, e, flag, b0, b1, c, nf_c = (int(v) for v in pk[t, :8])
        wc = pk[t, 8:9].view(np.float64)[0]
        exact_in = ref[starts[t] - 1]
        if s != exact_in:
            print(f"tile {t}: chain state {s!r} != exact incoming sum {exact_in!r}")
            return
        ec = binade(s)
        S = int(math.ldexp(s, 52 - ec))
        if flag == 1 and e == ec:
            S_out = S + (a1 if S & 1 else a0)
            if S_out >= TWO53:
                print(f"tile {t} (start {starts[t]}): flag 1 but overflows: e={e} S={S} a0={a0} a1={a1}")
                return
            s = math.ldexp(S_out, ec - 52)
            t += 1
            continue
        if flag == 2 and e == ec:
            S_A = S + (a1 if S & 1 else a0)
            ok = S_A < TWO53 and S_A + nf_c >= TWO53
            if ok:
                s_new = math.ldexp(S_A, e - 52) + wc
                if binade(s_new) == e + 1:
                    S2 = int(math.ldexp(s_new, 52 - (e + 1)))
                    S_B = S2 + (b1 if S2 & 1 else b0)
                    if S_B < TWO53:
                        s = math.ldexp(S_B, e + 1 - 52)
                        t += 1
                        continue
            print(f"tile {t} (start {starts[t]}): flag 2 verification failed: e={e} ec={ec} S={S} S_A={S_A} nf_c={nf_c} wc={wc} c={c}")
            return
        print(f"tile {t} (start {starts[t]}): flag={flag} e={e} but chain binade {ec}, s={s!r} (rank boundary tiles: "
              f"{[int(np.searchsorted(starts, cu)) for cu in cuts[1:-1]]})")
        return
    print("chain ok, total", s, "ref", ref[-1])


main()
