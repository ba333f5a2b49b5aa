"""
lsqfitgp_amd -- MI355X (gfx950) native implementation of the dense-GP fit/predict
hot path of lsqfitgp: kernel Gram-matrix build (ExpQuad, Maternp, BART),
regularised blocked Cholesky, triangular solves, posterior mean/covariance and
log marginal likelihood, behind lsqfitgp's own ``GP`` / ``Decomposition`` API
surface (raw outputs).  Python host code -> ctypes -> ``liblsqfitgp_hip.so``
(hand-written HIP kernels, ``include/lsqfitgp_hip.h``).  No CPU fallback.
"""

from ._lib import LibraryError
from ._device import Context, DeviceArray, asdevice
from ._linalg import Decomposition, Chol
from ._kernels import (Kernel, ExpQuad, Maternp, Matern, GammaExp, Cauchy, Wendland, CausalExpQuad, Log, Constant, White, Expon, HoleEffect,
                       Periodic, Cos, Sinc, StationaryFracBrownian, Linear, Wiener, OrnsteinUhlenbeck, BrownianBridge, Decaying,
                       FracBrownian, WienerIntegral, Celerite, Harmonic, Circular, Taylor, Rescaling, Categorical, NNKernel, Gibbs, MA, AR, BagOfWords, Color, Pink, Bessel, Zeta, BART)
from ._GP import GP, Diagonal, diag
from ._fit import empbayes_fit
from ._fastraniter import raniter, sample
from ._uncertain import UVal
from ._dual import Dual
from . import _linalg
from . import copula
from . import bayestree

__version__ = '0.1.0'
