import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from conftest import synth
from aspire_amd.engine import HipEngine
eng = HipEngine(0, n_max=1 << 20, d_max=32)
for n in (10, 2000, 65536, 300000):
    x, ll, lp, lq = synth(n, 4, 1)
    t = [eng.asarray(a) for a in (ll, lp, lq)]
    print(n, eng.find_beta(*t, 0.0, 0.5, 1e-6))
