import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aspire_amd import Aspire, Samples, DiagGaussianMixture

d = 32
lik = DiagGaussianMixture.isotropic(d, normalized=False)      # evaluated inside the fused pCN kernel
aspire = Aspire(log_likelihood=lik, log_prior=lik, dims=d, flow_backend="gaussian")
aspire.fit(Samples(1.5 * np.random.default_rng(0).normal(size=(5000, d))))
post, hist = aspire.sample_posterior(sampler="smc", n_samples=1_000_000, sampler_kwargs=dict(n_steps=32, noise="f32"),
                                     store_sample_history=False, return_history=True)
print(post.log_evidence, post.log_evidence_error, hist.beta)   # analytic: (d/2) log(pi) = 18.3157
print(hist.mcmc_nu)
