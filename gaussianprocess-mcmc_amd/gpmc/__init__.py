"""gpmc -- MI355X-native GP-likelihood core behind the kcGP / kcMCMC protocol of t-kychen/GaussianProcess-MCMC.

GPU-only: importing is cheap, but every numerical entry point raises if libgpmc.so or the HIP device
is missing (there is no CPU fallback; the CPU oracle lives under ``oracle/`` and is test-only).
"""
from . import _lib  # noqa: F401
from ._lib import GpmcError, build, get_option, set_option  # noqa: F401

__all__ = ['covK', 'likK', 'tools', 'gpK', 'core', 'ops', 'sliceSample', 'framework', 'chains', 'build', 'GpmcError']


def set_return(mode):
    """'torch' (default): the kcGP mirrors (covK.*.getCovMatrix, tools.jitchol, tools.solve_chol, the protocol function
    sliceSample.aux_var_model) return device tensors; 'numpy': fresh host ndarrays, the reference's own return type, so
    NumPy sampler code written against kcGP (kcMCMC/sliceSample.py:165-207) runs on top of them unchanged."""
    from . import ops
    ops.set_return(mode)


def __getattr__(name):
    if name in ('covK', 'likK', 'tools', 'gpK', 'core', 'ops', 'sds', 'sliceSample', 'framework', 'chains'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
