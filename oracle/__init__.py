"""
oracle/ -- CPU restatement of lsqfitgp's dense-GP fit/predict hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / reported CPU baseline.  The product
package ``lsqfitgp_amd`` never imports it and fails loudly when its HIP
library is missing.

What it restates (numpy/scipy, every function cites the reference file:line it
follows; paths relative to the reference checkout):

* ``decomp``  -- ``src/lsqfitgp/_linalg/_decomp.py:245-255,349-361,376-490``
  (``Decomposition``/``Chol``: pow-2 diagonal scaling, Gershgorin jitter,
  Cholesky, solves, quadratic forms, Normal log-density value).
* ``kernels`` -- ``src/lsqfitgp/_Kernel/_isotropic.py:61-81``,
  ``_Kernel/_util.py:74-99``, ``_Kernel/_ops.py:292-326``,
  ``_kernels/_basic.py:61-75``, ``_kernels/_matern.py:29-49``,
  ``_special/_bessel.py:101-110``, ``_kernels/_bart.py`` (BART correlation,
  splits, indices) and ``_jaxext/_fasthash.py:56-102`` (fasthash64/32).
* ``kernels.Kernel.diff`` -- ``src/lsqfitgp/_Kernel/_ops.py:115-211`` for ExpQuad, in closed form.
* ``gp``      -- ``src/lsqfitgp/_GP/_compute.py:45-136,138-305,336-422,430-516``
  and ``src/lsqfitgp/_GP/_elements.py:158-346,408-649,758-826`` (raw outputs only).

PARITY PINNING STATUS: **pinned at the fit level by outputs of the reference
stored in its user guide; otherwise parity unpinned against the reference's
JAX-CPU outputs** (the reference holds no stored posterior means/covariances).  The reference cannot be imported in the build container
(``ModuleNotFoundError: No module named 'jax'`` at
``src/lsqfitgp/_patch_jax.py:22`` -- an ordinary error, jax/gvar are simply not
installed and there is no network), and the arithmetic of the path lives in
the un-vendored dependency jax/jaxlib>=0.4.35 (``pyproject.toml:42-43``) whose
CPU lowering is LAPACK ``potrf``/``trsm`` + BLAS ``gemm``; scipy calls the same
LAPACK family here.  What *is* pinned, in ``tests/test_oracle_*.py``:

* outputs printed by the reference itself: the hyperparameter fits of
  ``docs/userguide/hyper.rst:49-70,124-156,252-257`` (``2.16(17)``,
  ``log(sdev) 0.89(33)``, ``log(scale) 1.051(77)``, ``sdev 2.44(81)``,
  ``scale 2.86(22)``), which put the ExpQuad Gram build, the jittered Cholesky
  and the Normal log-density of this restatement on the line through the
  argmax and curvature of the marginal likelihood
  (``tests/test_fit_host.py``);
* a posterior printed by the reference: the worked example of
  ``docs/userguide/integrals.rst:34-82`` (derivative observations of an ExpQuad
  process, ``-1.9157(27)`` and ``0.0022(27)``), reproduced digit for digit
  (``tests/test_deriv_known_answer.py``);
* the reference's only stored known-answer table on this path, the
  ``fasthash64``/``fasthash32`` vectors of ``tests/test_jax.py:112-186``
  (copied as data into ``tests/golden/fasthash_kat.json``);
* every identity / property test the reference holds for the path
  (``tests/linalg/test_decomp.py:145-225``, ``tests/GP/test_pred.py``,
  ``tests/GP/test_GP.py`` raw-mode subset, ``tests/kernels/test_kernels.py``
  pins for ExpQuad/Maternp, ``tests/kernels/test_bart.py``), re-run against
  this restatement with the reference's own tolerances.
"""

from . import decomp, kernels, gp  # noqa: F401
from .decomp import Decomposition, Chol  # noqa: F401
from .kernels import (ExpQuad, Maternp, Matern, GammaExp, Cauchy, Wendland, CausalExpQuad, Log, Constant, White, Expon, HoleEffect, Periodic, Cos,  # noqa: F401
                      Sinc, StationaryFracBrownian, Linear, Wiener, OrnsteinUhlenbeck, BrownianBridge, Decaying, FracBrownian, WienerIntegral, Celerite, Harmonic, Circular, Taylor, Rescaling, Categorical, NNKernel, Gibbs, MA, AR, BagOfWords, Color, Pink, Bessel, Zeta, BART)
from .gp import GP  # noqa: F401
