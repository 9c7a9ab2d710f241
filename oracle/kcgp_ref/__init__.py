"""NumPy/SciPy restatement of the ``kcGP`` protocol the reference sampler calls.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED: the
reference git-ignores ``kcGP`` (``/root/reference/.gitignore:12``); semantics
are taken from the call sites in ``kcMCMC/sliceSample.py`` and from the
published pyGPs 1.3.4 algorithm (``requirements.txt:10``).
"""
from . import covK, likK, tools  # noqa: F401
