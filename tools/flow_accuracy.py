"""Accuracy of the flow log-density on the GPU (split-fp16 MFMA default, ASMC_FLOW_MATH=f32 fp32 MFMA chain) and of torch fp32
on the CPU, each against the same flow evaluated in fp64: max / rms absolute error and max relative error of log q."""
import os, sys
import numpy as np, torch
ROOT=os.getcwd(); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from conftest import random_coupling_flow
from aspire_amd.engine import HipEngine
eng=HipEngine(0,n_max=1<<17,d_max=32)
for hidden,scale_x in [(64,1.0),(32,3.0),(128,0.2)]:
    d,n=32,1<<16
    nl=4 if hidden<128 else 1
    flow=random_coupling_flow(d,nl,hidden,seed=11)
    f64=random_coupling_flow(d,nl,hidden,seed=11,dtype=torch.float64)
    f64.layers.load_state_dict(flow.layers.state_dict()); f64.loc,f64.scale=flow.loc.double(),flow.scale.double()
    g=np.random.default_rng(2); x=scale_x*g.normal(size=(n,d)); x[:64]*=4.0
    with torch.no_grad(): ref=f64.log_prob(torch.as_tensor(x)).numpy()
        
    dev=flow.device_coupling(eng); xd=eng.asarray(x)
    os.environ["ASMC_FLOW_MATH"]="f32"; a=eng.coupling_logprob(xd,dev).cpu().numpy()
    del os.environ["ASMC_FLOW_MATH"]; b=eng.coupling_logprob(xd,dev).cpu().numpy()
    with torch.no_grad(): t32=flow.log_prob(torch.as_tensor(x,dtype=torch.float32)).double().numpy()
    for nm,v in (("f32mfma",a),("split",b),("torch32",t32)):
        e=np.abs(v-ref); print(hidden,scale_x,nm,"max %.3e rms %.3e rms(typical rows) %.3e max rel %.3e"%(e.max(),np.sqrt(np.mean(e**2)),np.sqrt(np.mean(e[64:]**2)),np.max(e/np.maximum(np.abs(ref),1))))
