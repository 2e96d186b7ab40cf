"""Host-side profile of the headline run (configs[2]: trained coupling flow + 32 pCN steps per temperature): wall vs the
library's GPU-busy time per run, and cProfile of one run."""
import cProfile, pstats, os, sys, time, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_coupling_flow  # noqa
from aspire_amd.engine import HipEngine
from aspire_amd.flows import CouplingFlow
from aspire_amd.samplers.smc import HipSMC
from aspire_amd.targets import DiagGaussianMixture

n, d = int(os.environ.get("N", 1 << 20)), 32
eng = HipEngine(0, n_max=n, d_max=32)
comm = None
if os.environ.get("SHARDED"):  # the sharded code path over a one-rank RCCL group (tools/rig1.sh)
    import torch.distributed as dist
    from aspire_amd.comm import TorchDistComm

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    comm = TorchDistComm(eng.device)
    comm.force_sharded = True
lik = DiagGaussianMixture.isotropic(d, normalized=False)
flow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
flow.fit(1.5 * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)


def run(seed):
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(seed),
                dtype="float64", **({"comm": comm} if comm is not None else {}))
    out = sp.sample(n, sampler_kwargs=dict(n_steps=32, noise="f64", step_fn="pcn"), store_sample_history=False)
    return sp, out


run(1)
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    sp, out = run(2 + k)
    torch.cuda.synchronize()
    print(f"run {k}: wall {1e3 * (time.perf_counter() - t0):.1f} ms, {len(sp.history.beta)} temperatures, logZ {float(out.log_evidence):.3f}")
eng.profile(True)
run(9)
rep = eng.profile_report()
eng.profile(False)
print("GPU busy (library kernels) per run: %.1f ms" % sum(c * ms for c, ms in rep.values()))
pr = cProfile.Profile()
pr.enable()
run(10)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats("aspire_amd", 40)
print(s.getvalue()[:9000])
