"""Container-only shim that makes the *real* reference importable for golden generation.

TEST INFRASTRUCTURE — never imported by the product (`aspire_amd/`), never run on the GPU box
(`/root/reference` does not exist there).  Used only by `oracle/make_golden.py`, `tests/tools/ref_ratio.py`,
`tests/test_reference_seam.py` and `tests/test_abi_and_layout.py::test_oracle_vs_reference_live_if_present` (which skip when
`/root/reference` is absent).

The reference (mj-will/aspire, `/root/reference/src/aspire`) declares python>=3.11 and depends on
five packages missing from this image.  Substitutions:
  array_api_compat  <- sklearn.externals.array_api_compat (vendored v1.12)
  array_api_extra   <- sklearn.externals.array_api_extra (+ a `default_dtype` helper)
  wrapt             <- a minimal `wrapt.decorator`
  h5py              <- stub whose File is oracle/fake_h5.FakeFile (in-memory groups / datasets with h5py's protocol)
  orng              <- ArrayRNG delegating to numpy.random.default_rng
"""
from __future__ import annotations

import functools
import os
import sys
import types

import numpy as np

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "aspire"))


def install() -> None:
    """Install the substitute modules and put the reference on sys.path (idempotent)."""
    if "aspire" in sys.modules and getattr(sys.modules["aspire"], "__file__", "").startswith(REFERENCE_SRC):
        return
    if not reference_available():
        raise RuntimeError("reference tree not present; the shim only works in the build container")
    import sklearn.externals.array_api_compat as aac
    import sklearn.externals.array_api_compat.common as aaccommon
    import sklearn.externals.array_api_compat.common._typing as aactyp
    import sklearn.externals.array_api_compat.numpy as aacnp
    import sklearn.externals.array_api_compat.torch as aact
    import sklearn.externals.array_api_extra as xpx

    sys.modules.update(
        {
            "array_api_compat": aac,
            "array_api_compat.numpy": aacnp,
            "array_api_compat.torch": aact,
            "array_api_compat.common": aaccommon,
            "array_api_compat.common._typing": aactyp,
        }
    )
    if not hasattr(xpx, "default_dtype"):

        def default_dtype(xp, kind="real floating", *, device=None):
            if aac.is_torch_namespace(xp):
                import torch

                return torch.get_default_dtype()
            return xp.asarray(1.0).dtype

        xpx.default_dtype = default_dtype
    sys.modules["array_api_extra"] = xpx

    wrapt = types.ModuleType("wrapt")

    def decorator(wrapper):
        def deco(wrapped):
            class _Desc:
                def __init__(self, f):
                    self.f = f
                    functools.update_wrapper(self, f)

                def __get__(self, inst, owner):
                    if inst is None:
                        return self
                    bound = self.f.__get__(inst, owner)

                    @functools.wraps(self.f)
                    def call(*a, **k):
                        return wrapper(bound, inst, a, k)

                    call.__func__ = self.f
                    return call

                def __call__(self, *a, **k):
                    return wrapper(self.f, None, a, k)

            return _Desc(wrapped)

        return deco

    wrapt.decorator = decorator
    sys.modules["wrapt"] = wrapt

    h5py = types.ModuleType("h5py")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_h5 import FakeFile  # in-memory stand-in: the reference's AspireFile subclasses h5py.File (utils.py:910)

    h5py.File = FakeFile
    h5py.string_dtype = lambda **k: __import__("numpy").dtype("O", metadata={"vlen": str})  # what the real one returns
    sys.modules["h5py"] = h5py

    orng = types.ModuleType("orng")

    class ArrayRNG:
        def __init__(self, backend="numpy", seed=None, **k):
            self._g = np.random.default_rng(seed)
            self.bit_generator = self._g.bit_generator

        def __getattr__(self, n):
            return getattr(self._g, n)

    orng.ArrayRNG = ArrayRNG
    sys.modules["orng"] = orng

    os.environ["SCIPY_ARRAY_API"] = "1"
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)


def import_reference():
    """Return the reference's (samples module, smc base module, mcmc module, utils module)."""
    install()
    import aspire.samplers.mcmc as ref_mcmc
    import aspire.samplers.smc.base as ref_smc
    import aspire.samples as ref_samples
    import aspire.utils as ref_utils

    return ref_samples, ref_smc, ref_mcmc, ref_utils
