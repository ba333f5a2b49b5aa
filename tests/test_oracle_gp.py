""" the GP identity checks of tests/gp_checks.py against the CPU oracle """

import oracle
from .gp_checks import GPChecks


class TestOracleGP(GPChecks):
    lgp = oracle
    validates = False
