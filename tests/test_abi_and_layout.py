"""C-ABI library loads and exports exactly what include/asmc.h declares (no compute: no GPU here);
the product never touches oracle/ and refuses to run without a HIP device."""
import ast
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "asmc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asmc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from aspire_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 25
    assert sorted(_lib.SIGNATURES) == declared  # the ctypes table binds exactly the header's functions
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.asmc_abi_version() == _lib.ASMC_ABI_VERSION


def test_struct_layouts_match_header():
    import ctypes

    from aspire_amd._lib import AsmcMixture, AsmcPcnParams

    assert ctypes.sizeof(AsmcMixture) == 32
    assert AsmcPcnParams.log_likelihood.offset == 40 and AsmcPcnParams.seed.offset == 40 + 3 * 32
    assert ctypes.sizeof(AsmcPcnParams) == 40 + 96 + 8 + 8 + 8 + 8 + 8 and AsmcPcnParams.nu.offset == 168


def test_error_reporting_without_gpu():
    import ctypes

    from aspire_amd import _lib

    lib = _lib.load()
    ctx = ctypes.c_void_p()
    rc = lib.asmc_ctx_create(ctypes.byref(ctx), 0, 0, 4)  # n_max = 0 is an argument error before any HIP call
    assert rc == -1 and b"n_max" in lib.asmc_last_error()


def test_product_has_no_cpu_fallback():
    import torch

    from aspire_amd import _lib
    from aspire_amd.engine import HipEngine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AsmcError, match="no CPU fallback"):
        HipEngine(0)


def test_product_never_imports_oracle_or_reference():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "aspire_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            tree = ast.parse(src)
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                for n in names:
                    assert not n.split(".")[0] in ("oracle", "ref_shim", "oracle_engine", "make_golden"), (f, n)
            assert "/root/reference" not in src, f
    for dirpath, _, files in os.walk(os.path.join(pkg, "csrc")):
        for f in files:
            if f.endswith((".hip", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower().replace("oracle-backed", ""), f


def test_oracle_vs_reference_live_if_present(oracle):
    """In the build container the oracle is additionally checked against the imported reference."""
    import numpy as np

    import ref_shim

    if not ref_shim.reference_available():
        pytest.skip("reference tree not present (GPU box)")
    rs, smc, mcmc, ut = ref_shim.import_reference()
    g = np.random.default_rng(123)
    x = g.normal(size=(777, 3))
    ll, lp, lq = g.normal(size=777) * 3, g.normal(size=777), g.normal(size=777)
    s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=0.1)
    assert np.array_equal(np.asarray(s.log_weights(0.45)), oracle.log_weights(ll, lp, lq, 0.1, 0.45))
    assert float(s.log_evidence_ratio(0.45)) == oracle.log_evidence_ratio(ll, lp, lq, 0.1, 0.45)
    out = s.resample(0.45, rng=np.random.default_rng(5))
    idx = oracle.resample_indices(ll, lp, lq, 0.1, 0.45, np.random.default_rng(5).random(777))
    assert np.array_equal(np.asarray(out.x), x[idx])


def test_hand_issued_lds_reads_are_not_touched_before_their_waits():
    """k_pcn_flow_fused issues the mat-vec's coefficient reads from inline asm, several batches ahead of the waits that cover
    them; hipcc must not copy, spill or reuse a destination register in between (it believes the data is there when the asm
    statement ends).  tools/audit_asm_loads.py compiles the kernel to assembly (no GPU needed) and walks every instantiation."""
    import shutil
    import subprocess
    import sys

    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_asm_loads.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # (round 4: the default build reads the mat-vec's coefficients as MFMA operands - no hand-issued batches are left in it; the
    # audit still guards the -DFUSED_MVMFMA=0 / -DFUSED_INPLACE=0 diagnostic builds, which the Makefile rule audits when built)
    import re

    m = re.search(r"(\d+) instantiations, (\d+) read batches, 0 violations", r.stdout)
    assert m and int(m.group(1)) >= 24, r.stdout


def test_flow_packing_places_every_weight_once():
    """asmc_coupling_pack / asmc_maf_pack are host-only (no GPU): the MFMA operand image is a permutation of the layers' weights and
    biases padded with zeros - coupling flow at a padded dimension, autoregressive flow at an odd one (its MADE masks leave zeros in
    the matrices) - and a weight outside the fp16 operand range of the split-fp16 kernels is refused, not silently clipped."""
    from aspire_amd import _lib
    from aspire_amd.engine import pack_coupling, pack_maf

    lib = _lib.load()
    g = np.random.default_rng(0)

    def layers(n, w_in, hidden, w_out, mask_p=0.0):
        ws, bs = [], []
        for _ in range(n):
            for shape in ((hidden, w_in), (hidden, hidden), (w_out, hidden)):
                w = g.normal(size=shape).astype(np.float32)
                if mask_p:
                    w[g.random(size=shape) < mask_p] = 0.0
                ws.append(w)
                bs.append(g.normal(size=shape[0]).astype(np.float32))
        return ws, bs

    d, hidden, n = 20, 32, 2
    ws, bs = layers(n, d // 2, hidden, d)
    packed = pack_coupling(lib, d, hidden, ws, bs)
    assert packed.size == lib.asmc_coupling_pack_floats(d, n, hidden)
    want = np.sort(np.concatenate([a.ravel() for a in ws + bs]))
    assert np.array_equal(np.sort(packed[packed != 0.0]), want)

    d, hidden, n = 7, 64, 3
    ws, bs = layers(n, d, hidden, 2 * d, mask_p=0.4)
    packed = pack_maf(lib, d, hidden, ws, bs)
    assert packed.size == lib.asmc_maf_pack_floats(d, n, hidden)
    want = np.sort(np.concatenate([a.ravel() for a in ws + bs]))
    assert np.array_equal(np.sort(packed[packed != 0.0]), want[want != 0.0])

    ws[1][3, 5] = 1e5  # beyond fp16: the split products would see inf
    with pytest.raises(_lib.AsmcError, match="fp16 operand range"):
        pack_maf(lib, d, hidden, ws, bs)
    assert lib.asmc_maf_pack_floats(129, 3, 64) < 0 and lib.asmc_coupling_pack_floats(21, 2, 64) < 0  # shapes without kernels
    assert lib.asmc_coupling_pack_floats(130, 2, 64) < 0 and lib.asmc_maf_pack_floats(64, 3, 48) < 0

    # more than 32 dimensions (round 5; packed layout 1, csrc/asmc_flow16.hip): biases in fp32, every weight ONCE as an fp16
    # (hi, lo) pair - hi + lo is the weight to fp32 accuracy - whatever the padding (d = 48 -> 64, 100 -> 128) and the kind
    assert lib.asmc_flow_layout(_lib.ASMC_FLOW_COUPLING, 32, 64) == 0 and lib.asmc_flow_layout(_lib.ASMC_FLOW_MAF, 33, 64) == 1
    assert lib.asmc_flow_layout(_lib.ASMC_FLOW_COUPLING, 48, 64) == 1 and lib.asmc_flow_layout(_lib.ASMC_FLOW_COUPLING, 47, 64) < 0
    for kind, d, hidden, n in (("coupling", 48, 64, 2), ("coupling", 128, 32, 1), ("maf", 33, 64, 2), ("maf", 100, 64, 1), ("maf", 64, 128, 1)):
        if kind == "coupling":
            ws, bs = layers(n, d // 2, hidden, d)
            packed = pack_coupling(lib, d, hidden, ws, bs)
            assert packed.size == lib.asmc_coupling_pack_floats(d, n, hidden)
        else:
            ws, bs = layers(n, d, hidden, 2 * d, mask_p=0.3)
            packed = pack_maf(lib, d, hidden, ws, bs)
            assert packed.size == lib.asmc_maf_pack_floats(d, n, hidden)
        D = 64 if d <= 64 else 128
        n_out = (D if kind == "coupling" else 2 * D)
        n_bias = n * (2 * hidden + n_out)
        bias = packed[:n_bias]
        want_b = np.sort(np.concatenate([b.ravel() for b in bs]))
        assert np.array_equal(np.sort(bias[bias != 0.0]), want_b[want_b != 0.0])
        halves = packed[n_bias:].view(np.float16).reshape(-1, 2, 64, 8).astype(np.float32)  # [block x K step][hi | lo][lane][8]
        vals = (halves[:, 0] + halves[:, 1]).ravel()
        got = np.sort(vals[vals != 0.0])
        want = np.sort(np.concatenate([a.ravel() for a in ws]))
        want = want[want != 0.0]
        assert got.size == want.size
        np.testing.assert_allclose(got, want, rtol=3e-7, atol=4e-8)  # (the lo half of a small weight is an fp16 subnormal: 6e-8 apart)


def test_bench_launcher_refuses_without_devices_and_fails_with_its_ranks():
    """bench.py --gpus N (N > 1) without a launcher becomes the launcher before any GPU call.  Here (no GPU):
    it refuses when fewer devices are visible than ranks asked for, and when its ranks fail it fails with them - never a line
    for another job than the one asked for."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU host: the refusal path needs fewer devices than ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ASMC_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout.strip() == "" and "refusing" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300, env=dict(env, ASMC_BENCH_DEVICE="0", ASMC_BENCH_BACKEND="gloo"))
        assert r.returncode != 0 and r.stdout.strip() == "" and "rank exit codes" in r.stderr
