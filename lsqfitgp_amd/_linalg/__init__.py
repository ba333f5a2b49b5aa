""" mirror of ``src/lsqfitgp/_linalg/__init__.py:20-25``: the decomposition API """

from ._decomp import Decomposition, Chol, solve_batched, solve_triangular_python
