"""pytest plugin (container only; TEST INFRASTRUCTURE): runs the REFERENCE's own unit-test files against THIS package.

`python -m pytest /root/reference/tests/test_samples.py -p ref_alias_plugin ...` imports the reference's test module unchanged; the
names it imports from `aspire` (`aspire.samples`, `aspire.history`, `aspire.utils`, ...) resolve to `aspire_amd`'s modules, h5py to
the in-memory stand-in of oracle/fake_h5.py, array_api_compat to the copy vendored in scikit-learn (oracle/ref_shim.py's
substitutions, without putting the reference's sources on the path).  Classes of the reference that are outside this repository's
scope (`MCMCSamples`, `PTMCMCSamples`: the emcee / parallel-tempering containers) are placeholders that skip the test that builds one.
Nothing of the reference is copied: its test files are read where they lie.  Driven by tests/test_reference_unit_tests.py.
"""
import os
import sys
import types

# /root/reference is read-only for this repository: pytest's assertion rewriter would otherwise cache bytecode of the reference's
# test modules in a __pycache__ next to them.  Set before any of them is imported (a `-p` plugin loads ahead of collection).
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _install():
    import numpy as np
    import pytest
    import sklearn.externals.array_api_compat as aac
    import sklearn.externals.array_api_compat.common as aaccommon
    import sklearn.externals.array_api_compat.numpy as aacnp
    import sklearn.externals.array_api_compat.torch as aact
    from fake_h5 import FakeFile as _MemFile

    class FakeFile(_MemFile):
        """The in-memory file, plus an empty file on disk under its name: callers test `Path(...).is_file()` before resuming."""

        def __init__(self, name, mode="r", *a, **k):
            super().__init__(name, mode, *a, **k)
            if mode not in ("r", "r+"):
                open(str(name), "ab").close()

    sys.modules.update({"array_api_compat": aac, "array_api_compat.numpy": aacnp, "array_api_compat.torch": aact,
                        "array_api_compat.common": aaccommon})
    import sklearn.externals.array_api_extra as xpx

    sys.modules["array_api_extra"] = xpx
    h5py = types.ModuleType("h5py")
    h5py.File = FakeFile
    h5py.string_dtype = lambda **k: np.dtype("O", metadata={"vlen": str})
    sys.modules["h5py"] = h5py

    import aspire_amd
    from aspire_amd import history, io, samples, transforms
    from aspire_amd import samples as samples_mod
    from oracle_engine import OracleEngine

    samples_mod._default_engine = OracleEngine()  # (no GPU in the build container: the engine's CPU test double)
    io.h5py_available = lambda: True
    io.open_h5 = lambda path, mode="r": FakeFile(path, mode)

    def out_of_scope(name):
        class _Placeholder:
            def __init__(self, *a, **k):
                pytest.skip(f"{name} is outside this repository's scope (SURVEY.md section 2)")

            @classmethod
            def from_samples(cls, *a, **k):
                pytest.skip(f"{name} is outside this repository's scope (SURVEY.md section 2)")

        _Placeholder.__name__ = name
        return _Placeholder

    pkg = types.ModuleType("aspire")
    pkg.__path__ = []  # a package: `import aspire.samples` resolves through sys.modules
    pkg.Aspire = aspire_amd.Aspire
    m_samples = types.ModuleType("aspire.samples")
    m_samples.__dict__.update({k: v for k, v in vars(samples).items() if not k.startswith("__")})
    for name in ("MCMCSamples", "PTMCMCSamples"):
        if not hasattr(m_samples, name):
            setattr(m_samples, name, out_of_scope(name))
    m_utils = types.ModuleType("aspire.utils")
    m_utils.AspireFile = FakeFile

    def copy_array(x, xp=None):  # (utils.copy_array: a helper of the reference's transform tests)
        return x.clone() if hasattr(x, "clone") else np.array(x, copy=True)

    m_utils.copy_array = copy_array
    for name in ("recursively_save_to_h5_file", "load_from_h5_file", "dump_state", "load_state"):
        setattr(m_utils, name, getattr(io, name))
    sys.modules.update({"aspire": pkg, "aspire.samples": m_samples, "aspire.history": history, "aspire.utils": m_utils,
                        "aspire.transforms": transforms})
    pkg.samples, pkg.history, pkg.utils, pkg.transforms = m_samples, history, m_utils, transforms


_install()
