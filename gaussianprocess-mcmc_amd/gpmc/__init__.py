"""gpmc -- MI355X-native GP-likelihood core behind the kcGP / kcMCMC protocol of t-kychen/GaussianProcess-MCMC.

GPU-only: importing is cheap, but every numerical entry point raises if libgpmc.so or the HIP device
is missing (there is no CPU fallback; the CPU oracle lives under ``oracle/`` and is test-only).
"""
from . import _lib  # noqa: F401
from ._lib import GpmcError, build, get_option, set_option  # noqa: F401

__all__ = ['covK', 'likK', 'tools', 'gpK', 'core', 'ops', 'sliceSample', 'framework', 'chains', 'build', 'GpmcError']


def __getattr__(name):
    if name in ('covK', 'likK', 'tools', 'gpK', 'core', 'ops', 'sds', 'sliceSample', 'framework', 'chains'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
