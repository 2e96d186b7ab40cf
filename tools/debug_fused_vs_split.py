import os, sys
import numpy as np, torch
ROOT=os.getcwd(); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from conftest import random_coupling_flow
from aspire_amd.engine import HipEngine
eng=HipEngine(0,n_max=1<<17,d_max=32)
d,n,beta,rho=32,5000,0.35,0.4
flow=random_coupling_flow(d,4,64); dev=flow.device_coupling(eng)
g=torch.Generator(eng.device).manual_seed(3)
x0=torch.randn((n,d),device=eng.device,dtype=torch.float64,generator=g)
t_ll=eng.make_mixture([0.3],np.full((1,d),0.25),np.ones((1,d))*1.5)
t_lp=eng.make_mixture([-0.5*d*np.log(2*np.pi)],np.zeros((1,d)),np.ones((1,d)))
mu=eng.asarray(0.1*np.arange(d)/d)
A=np.eye(d)+0.05*np.tril(np.random.default_rng(2).normal(size=(d,d)),-1)
L,Linv=eng.asarray(A),eng.asarray(np.linalg.inv(A))
for n_steps in (1,5):
    x=x0.clone(); ll,lp,lq=eng.mixture_logpdf(x,t_ll),eng.mixture_logpdf(x,t_lp),eng.coupling_logprob(x,dev)
    n_acc,_,_=eng.pcn_mutate_flow(x,ll,lp,lq,beta,mu,L,Linv,t_ll,t_lp,dev,77,1000,rho,n_steps,5,0.234,False,"f64",0.0)
    xb=x0.clone(); llb,lpb,lqb=eng.mixture_logpdf(xb,t_ll),eng.mixture_logpdf(xb,t_lp),eng.coupling_logprob(xb,dev)
    acc=[]
    for t in range(n_steps):
        xp,q0,q1=eng.pcn_propose(xb,mu,L,Linv,rho,77,1000,5+t,nu=0.0)
        lqn=eng.coupling_logprob(xp,dev)
        acc.append(eng.pcn_accept(xb,xp,llb,lpb,lqb,eng.mixture_logpdf(xp,t_ll),eng.mixture_logpdf(xp,t_lp),lqn,q0,q1,beta,77,1000,5+t))
    close=((x-xb).abs()<=1e-9*(1+xb.abs())).all(dim=1)
    moved=(x!=x0).any(dim=1)
    print("steps",n_steps,"mismatch",int((~close).sum()),"acc fused",n_acc,"acc split",acc)
    e=(lq-eng.coupling_logprob(x,dev)).abs()
    print("   carried lq vs standalone at final x: max",float(e.max()),"on moved rows", float(e[moved].max()) if moved.any() else None, "n>1e-4:",int((e>1e-4).sum()))
