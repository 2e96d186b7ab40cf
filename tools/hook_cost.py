#!/usr/bin/env python
"""What the sharded form of the fused flow step costs per step on ONE GPU (one-rank RCCL group): the same 32-step mutation with
no count exchange (single-rank form: the step's last block adapts), with the library's own ncclAllReduce between the steps and
with the Python callback (torch.distributed) - HIP-event time of k_pcn_flow_fused per launch and wall time per call."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29573")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from conftest import random_coupling_flow, synth

    from aspire_amd.comm import TorchDistComm
    from aspire_amd.engine import HipEngine

    n, d, steps = 1_000_000, 32, 32
    eng = HipEngine(0, n_max=n, d_max=d)
    x, _, _, _ = synth(n, d, 3)
    xd = eng.asarray(x)
    tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    flow = random_coupling_flow(d, 4, 64)
    dev = flow.device_coupling(eng)
    mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))

    comms = {}
    for kind in ("rccl", "python"):  # one communicator per form, made (and self-tested) once
        hc = TorchDistComm(eng.device)
        hc.force_sharded = True
        if kind == "python":
            hc._rccl = None
        else:
            assert hc.rccl_direct() is not None
        comms[kind] = hc
    adapt = os.environ.get("ADAPT", "0") == "1"

    def run(kind, profile):
        xx = xd.clone()
        a_, b_, c_ = eng.mixture_logpdf(xx, tgt), eng.mixture_logpdf(xx, tgt), eng.coupling_logprob(xx, dev)
        if kind:
            hc = comms[kind]
            eng.set_count_hook(hc, n)
        eng.profile(profile)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            eng.pcn_mutate_flow(xx, a_, b_, c_, 0.5, mu, eye, eye, tgt, tgt, dev, 11, 0, 0.02, steps, 3, 0.234, adapt, "f64", 0.0)
        finally:
            eng.set_count_hook(None, None)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        rep = eng.profile_report() if profile else {}
        eng.profile(False)
        return wall, rep.get("k_pcn_flow_fused", (0, 0.0))

    for kind in (None, "rccl", "python", None, "rccl", "python"):
        run(kind, False)
        wall, _ = run(kind, False)
        _, (cnt, us) = run(kind, True)
        print(f"{str(kind):7s} wall {wall:7.3f} ms per {steps}-step call = {wall / steps * 1e3:6.1f} us/step; k_pcn_flow_fused {cnt} x {us:7.2f} us (events)")


if __name__ == "__main__":
    main()
