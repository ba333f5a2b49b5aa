""" mirror of ``src/lsqfitgp/bayestree/__init__.py``: GP versions of Bayesian tree models (``bart``) """

from ._bart import bart  # noqa: F401
