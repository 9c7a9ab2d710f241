"""``kcGP`` drop-in: the package name the reference imports (framework.py:6, kcMCMC/sliceSample.py:13:
``from kcGP import covK, likK, tools``), backed by the MI355X kernels of ``gpmc``."""
from gpmc import covK, gpK, likK, tools  # noqa: F401
