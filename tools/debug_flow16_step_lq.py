import os, sys
import numpy as np, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from conftest import random_coupling_flow, random_maf_flow
from aspire_amd.engine import HipEngine
eng=HipEngine(0, n_max=1<<16, d_max=128)
for kind,d in (("coupling",64),("maf",64),("coupling",128),("maf",128)):
    n=4000
    flow = random_coupling_flow(d,4,64,seed=6) if kind=="coupling" else random_maf_flow(d,3,64,seed=6)
    dev=flow.device_coupling(eng)
    g=np.random.default_rng(1)
    x=eng.asarray(0.9*g.normal(size=(n,d)))
    t=eng.make_mixture([0.0], np.zeros((1,d)), np.ones((1,d)))
    mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
    ll, lp, lq = eng.mixture_logpdf(x,t), eng.mixture_logpdf(x,t), eng.coupling_logprob(x,dev)
    x0=x.clone()
    acc,_,_=eng.pcn_mutate_flow(x, ll, lp, lq, 0.4, mu, eye, eye, t, t, dev, 7, 0, 0.05, 1, 0, 0.234, False, "f64", 0.0)
    moved=(x!=x0).any(dim=1)
    lq2=eng.coupling_logprob(x,dev)
    err=(lq-lq2).abs()[moved]
    print(kind,d,"accepted",int(acc[0]),"moved",int(moved.sum()),"max |lq_step - lq_kernel| on moved rows", float(err.max()) if err.numel() else None, "n>1e-3:", int((err>1e-3).sum()))
