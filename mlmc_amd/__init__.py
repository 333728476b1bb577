"""mlmc_amd -- MI355X-native moment estimation and maximum-entropy PDF reconstruction for MLMC.

Drop-in for the post-processing hot path of GeoMop/MLMC (mlmc.moments, mlmc.quantity.quantity_estimate,
mlmc.estimator, mlmc.tool.simple_distribution / distribution): same names and call signatures, arithmetic in
hand-written HIP kernels (libmlmc_hip.so, C ABI in include/mlmc_hip.h).  There is no CPU fallback.
"""
from .moments import Moments, Monomial, Fourier, Legendre, Spline, TransformedMoments  # noqa: F401
from .sampler import DeviceSampler  # noqa: F401

__version__ = "0.1.0"
