"""CPU oracle for the GP-likelihood hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / reported baseline -- never as the
thing measured or shipped.  The product path (``gaussianprocess-mcmc_amd/``)
must fail loudly when its HIP library is missing; it never falls back here.

Parity status
-------------
* ``sds_ref`` (aux_var_model, log_gamma, the log-marginal expression, the
  surrogate-data slice-sampling loop and the two chain drivers) restates
  ``/root/reference/kcMCMC/sliceSample.py:76-232``, ``framework.py:59-77`` and
  ``demoRegression.py:15-32,116-131``.  It is PINNED: ``oracle/make_golden.py``
  runs the reference's own ``sliceSample.py`` (loaded by file path, unmodified)
  in the build container and the fixtures it wrote under ``tests/golden/`` are
  compared with this restatement by ``tests/test_oracle_golden.py``.
* ``kcgp_ref`` (covK.RBF / tools.jitchol / tools.solve_chol / likK.*) restates
  the *missing* ``kcGP`` package (git-ignored by the reference, a private fork
  of pyGPs 1.3.4, ``requirements.txt:10``) from its call sites and the
  published pyGPs algorithm.  There is no reference source, test or golden
  vector for it: PARITY UNPINNED at that boundary.  The oracle of record for
  the K build and the Cholesky is plain NumPy/SciPy math on the same (X, y).
"""
