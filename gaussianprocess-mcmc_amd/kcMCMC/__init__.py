"""``kcMCMC`` drop-in: ``from kcMCMC import sliceSample`` (demoRegression.py:13) -> the device sampler."""
from gpmc import sliceSample  # noqa: F401
sdsK = sliceSample   # the name framework.py:10 imports
