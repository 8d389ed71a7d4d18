"""The CONEX_* ctypes table moved into the package (conex_amd/capi.py); kept importable under its
old name for the tests."""
from conex_amd.capi import *  # noqa: F401,F403
from conex_amd.capi import IterationStats, SolverConfiguration, api, c_double_p, colmajor, default_config, dp  # noqa: F401
