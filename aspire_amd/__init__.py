"""aspire_amd — MI355X-native SMC particle-batch engine behind aspire's
`Aspire.sample_posterior(sampler="smc", ...)` / `Samples` API (reference: mj-will/aspire).

The package holds only what that hot path needs: the HIP kernels + C ABI (`csrc/`,
`include/asmc.h`), the ctypes binding (`_lib`, `engine`), and the host-side mirror of the
reference's operator interface (`samples`, `samplers`, `aspire`, `flows`, `targets`).
"""
from .aspire import Aspire
from .flows import CouplingFlow, Flow, GaussianFlow
from .history import SMCHistory
from .samples import Samples, SMCSamples
from .targets import DiagGaussianMixture

__all__ = ["Aspire", "Samples", "SMCSamples", "SMCHistory", "Flow", "GaussianFlow", "CouplingFlow",
           "DiagGaussianMixture"]
__version__ = "0.1.0"
