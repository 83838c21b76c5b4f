"""
mogp_emulator_amd -- MI355X (gfx950) native fit + predict backend for mogp_emulator's GPU seam.

  csrc/            hand-written HIP kernels + C ABI (libmogp_hip.so, header: include/mogp_hip.h)
  _capi.py         ctypes prototypes of the C ABI
  libgpgpu.py      drop-in for the reference's pybind11 module `libgpgpu`
  LibGPGPU.py, GaussianProcessGPU.py, MultiOutputGP_GPU.py, fitting.py, Priors.py, Kernel.py
                   host-side mirrors of the reference's GPU-facing Python interface
  HistoryMatching.py, SequentialDesign.py, validation.py
                   consumers of the batched prediction (implausibility, MICE scoring, validation errors)
  dist.py          one-process-per-GPU sharding of emulators + single gather (torch.distributed/RCCL)
"""
from .LibGPGPU import HAVE_LIBGPGPU, gpu_usable            # noqa: F401

if HAVE_LIBGPGPU:
    from .GaussianProcessGPU import GaussianProcessGPU, PredictResult   # noqa: F401
    from .MultiOutputGP_GPU import MultiOutputGP_GPU                      # noqa: F401
    from .fitting import fit_GP_MAP                                         # noqa: F401
    from .Kernel import SquaredExponential, Matern52, ProductMat52, UniformSqExp, UniformMat52   # noqa: F401
    from .Priors import GPPriors, MeanPriors, InvGammaPrior, GammaPrior, LogNormalPrior, WeakPrior   # noqa: F401
    from .HistoryMatching import HistoryMatching                           # noqa: F401
    from .SequentialDesign import MICEFastGP, mice_criterion               # noqa: F401
    from . import validation                                                # noqa: F401

__version__ = "0.1.0"
