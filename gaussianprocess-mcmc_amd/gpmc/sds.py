"""Device-resident auxiliary-variable model of surrogate-data slice sampling.

Mirror of ``aux_var_model`` (kcMCMC/sliceSample.py:165-207) plus the two expressions that consume it
(``ita``, sliceSample.py:108; ``prop_f``, :140; ``curG``/``propG``, :122,147), with every dense step a
libgpmc kernel and nothing but scalars and N-vectors crossing to the host.

Default form ``'inv'`` (4 N^3 / 3 flops per proposal).  With S = s I (the kernel diagonal is constant, so S_ii is one
scalar) and A = K + S = L L^T:   R = K - V^T V = K - K A^-1 K  ==  S - S A^-1 S  =  s I - (s L^-T)(s L^-T)^T.

    workspace A, (2 np_ + 128) x np_ :  rows [0, np_)            K + S   (lower; S_ii fused by the K build)
                                        row  np_                 g       (becomes alpha^T = (L^-1 g)^T)
                                        rows [np_+128, 2np_+128) s I     (becomes P = s L^-T, upper-trapezoidal)
    gpmc_potrf_tri(A, ntri = np_)    -> L, alpha, P in one factorisation; the trapezoidal rows cost N^3/3   (a3, a4, a11)
    workspace R, (np_ + 128) x np_   :  gpmc_syrk_tri -> R + 1e-11 I = (s + 1e-11) I - P P^T, N^3/3           (a5, a7's shift)
    m = g - P alpha   (gpmc_trmv_upper; == R S^-1 g)                                                          (a6)
    row np_ = f - m ; gpmc_potrf(R) -> C = chol(R + 1e-11 I), eta = C^-1 (f - m)                              (a7, a9)
    f' = gpmc_trmv_lower(C, eta) + m                                                                          (a10)

One proposal of the sampler's inner loop (sliceSample.py:136-147) is ``SDSWorkspace.proposal(hyp)``: the whole sequence
above plus f' and the five scalars into pinned host memory, recorded once (stream capture) and replayed, one stream
synchronisation per proposal; the jitter schedules of both factorisations run eagerly when the replayed happy path
reports a failed pivot.

Form ``'syrk'`` is the first implementation (8 N^3 / 3): K itself rides on the factorisation (V^T = K L^-T, N^3) and
R = K - V^T V is a full SYRK (N^3); kept for cross-checks.  Both follow the reference's seeded chains decision for
decision (tests/test_gpu_sds.py); the reference itself spends ~11.3 N^3 flops here (dense S, LU solves, two inverses).
"""
import os

import numpy as np
import torch

from . import _lib, covK, ops

RECORD_MIN_NP = 6144      # SDSWorkspace.proposal records its launch sequence from this many padded columns on
R_JITTER = 1e-11          # sliceSample.py:205


class _NullCtx(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class SDSWorkspace(object):
    def __init__(self, x, kernel='rbf', dtype=torch.float64, form='inv'):
        """``dtype``: float64 is the sampler's arithmetic (the reference's).  A float32 workspace is accepted, but R = S - S
        (K+S)^-1 S has eigenvalues at rounding level, and at float32 resolution jitchol's schedule (at most 1e-5 of the mean
        diagonal) cannot make it positive definite at realistic sizes: ``aux_var_model`` then raises LinAlgError exactly as
        the schedule prescribes.  The fp32 configuration of BASELINE.json is the log-marginal plan (core.LogMarginalPlan)."""
        ops.require_gpu()
        if form not in ('inv', 'syrk'):
            raise ValueError("form must be 'inv' or 'syrk'")
        self.kernel = kernel
        self.dtype = dtype
        self.form = form
        self.X = ops.to_dev(x, dtype)
        if self.X.dim() == 1:
            self.X = self.X.reshape(-1, 1)
        self.n, self.d = self.X.shape
        dev = self.X.device
        n = self.n
        self.np_ = np_ = ops.pad(n)
        self.mpA = 2 * np_ + ops.PAD
        self.mpR = np_ + ops.PAD
        # 'inv': g block right below the matrix, the trapezoidal rows LAST (gpmc_potrf_tri); 'syrk': K rows, then g
        self.g_row = np_ if form == 'inv' else 2 * np_
        self.rows0 = np_ + ops.PAD if form == 'inv' else np_
        self.WA = torch.zeros((self.mpA, np_), dtype=dtype, device=dev)
        self.WR = torch.zeros((self.mpR, np_), dtype=dtype, device=dev)
        self.infoA = torch.zeros(1, dtype=torch.int32, device=dev)
        self.infoR = torch.zeros(1, dtype=torch.int32, device=dev)
        # one scratch per factorisation: each holds the sync words of its tile-DAG launches, zeroed by a memset in front of
        # the launch -- and in a REPLAYED sequence (HIP runtime 7.0) a memset node is not reliably ordered behind the
        # kernel node before it: with one shared scratch the second factorisation's memset cleared the flags of the first
        # one's running kernel from the second replay on (round 3).  With a region of its own every memset
        # only has to precede its own kernel.
        self.scratch = ops.potrf_scratch(np_, dev)
        self.scratchR = ops.potrf_scratch(np_, dev)
        self.out = torch.zeros(4, dtype=torch.float64, device=dev)
        nk = 2 + (1 if kernel == 'rbf' else self.d)
        # [kpS | kp0] in pinned host memory, read by the kernels THROUGH ITS DEVICE MAPPING (same address on ROCm): no copy per
        # proposal -- the copy was a 3 us blit plus ~17 us before the next kernel could start (profiles/r03_sds_timeline_*),
        # 5 % of a proposal at N = 455.  The K build stages the values in LDS once per workgroup; the other readers take one
        # scalar load per wave.  The host rewrites the block only between proposals, every one of which ends synchronised.
        self.kp_host = torch.zeros(2 * nk, dtype=dtype).pin_memory()
        self.kpS, self.kp0 = self.kp_host[:nk], self.kp_host[nk:]
        self.m = torch.zeros(n, dtype=dtype, device=dev)
        self.gs = torch.zeros(n, dtype=dtype, device=dev)
        self.prop_f = torch.zeros(n, dtype=dtype, device=dev)
        self.Rdiag = torch.zeros(n, dtype=dtype, device=dev)
        # second copy of -P P^T, written by the same launch that forms R: chol(R + 1e-11 I) works in place and needs the
        # jitter schedule about one proposal in three at N = 8192, and restoring R is a copy instead of another N^3/3
        self.R0 = torch.zeros((np_, np_), dtype=dtype, device=dev) if form == 'inv' else None
        # inputs of the recorded proposal sequence (fixed addresses) and its results in pinned host memory
        self.f_in = torch.zeros(n, dtype=dtype, device=dev)
        self.g_in = torch.zeros(n, dtype=dtype, device=dev)
        self.ita_in = torch.zeros(n, dtype=dtype, device=dev)
        self.res_host = torch.zeros(8 + n, dtype=torch.float64).pin_memory()     # [G, logdet, quad, infoA, infoR, -, -, -, f'...]
        self._seq = None
        self._seq_stream = None
        self._proposals = 0
        if self.R0 is not None:
            # the jitter schedule's own launches, once on zeros: their code objects load here, not inside the first
            # proposal that needs them (~15 ms on a fresh process)
            float(self.Rdiag.mean())
            self.R0.diagonal()[:n].copy_(self.Rdiag)
            self.WR[:np_].copy_(self.R0)
            self.WR.diagonal()[:n].add_(0.0)                      # (_chol_R's two: the happy path has them in one launch
            torch.sub(self.f_in, self.m, out=self.WR[np_, :n])    #  of its own, so a retry would be their first use)
        self._last_stream = None   # the stream the last sequence went to, when it is not the current one
        self._dirty = False        # a failed factorisation of K + S has run: its riding rows may hold NaN anywhere

    # ------------------------------------------------------------------ pieces
    def _set_hyp(self, hyp):
        kobj, sn = covK.kernel_from_hyp(hyp, self.kernel, self.dtype)
        kp = kobj.kparams()
        self.Sii = covK.s_diag_scalar(kp[0], sn)        # S = Sii * I : the kernel diagonal is the constant sf2
        kp0 = kp.copy()
        kp[1] = self.Sii
        self._kp0_sf2 = float(kp0[0])
        nk = kp.shape[0]
        self.kp_host[:nk] = torch.as_tensor(kp, dtype=self.dtype)
        self.kp_host[nk:] = torch.as_tensor(kp0, dtype=self.dtype)

    def _factor_A(self, jitter):
        """K + S (+ jitter) with g and either s I ('inv') or K ('syrk') riding: L, alpha = L^-1 g and P = s L^-T or
        V^T = K L^-T, then the log-marginal reduction.  No host value enters a launch (the diagonal shift is read from
        kpS[1] on the device), so the same calls can be recorded."""
        n, np_ = self.n, self.np_
        if jitter:
            self.kpS[1] = self.Sii + jitter
        self._sA = self.Sii + jitter                              # what was added to the diagonal of K
        ops.kbuild(self.kernel, self.X, self.kpS, self.WA, np_, np_, ops.KB_LOWER | ops.KB_PAD_IDENTITY)
        rows = self.WA[self.rows0:self.rows0 + np_]
        if self.form == 'inv':
            self._clean()
            ops.reset_tri(rows, np_, self.kpS[1:2])               # last proposal's P -> s I (upper triangle only)
        else:
            ops.kbuild(self.kernel, self.X, self.kp0, rows, np_, np_, ops.KB_FULL)
        ops.veccopy(self.WA[self.g_row], self.g_in, n)           # (a kernel, not a copy node: see _record)
        if self.form == 'inv':
            ops.potrf_tri(self.WA, self.mpA, np_, np_, self.infoA, self.scratch, rhs_rows=1)      # one row in the g block
        else:
            ops.potrf(self.WA, self.mpA, np_, self.infoA, self.scratch)
        ops.logdet_quad(self.WA, n, self.WA[self.g_row], self.infoA, self.out)

    def _clean(self):
        """A factorisation that met a non-positive pivot keeps running on NaN, and 0 * NaN turns the structural zeros of
        the riding rows (left of P's diagonal, the rest of the g block) into NaN; ``gpmc_reset_tri`` only re-arms the
        upper triangle, so after a failure everything below the matrix is cleared once, eagerly."""
        if self._dirty:
            self.WA[self.np_:].zero_()
            self._dirty = False

    def _make_R(self):
        """R = K - V^T V (= S - S A^-1 S) into WR (no 1e-11 shift yet), m = R S^-1 g, and diag(R) saved for the jitter
        schedule (the in-place factorisation destroys it)"""
        n, np_ = self.n, self.np_
        g = self.g_in
        rows = self.WA[self.rows0:self.rows0 + np_]
        if self.form == 'inv':
            ops.syrk_tri(rows, self.WR, n, np_, 0.0, C2=self.R0)  # -P P^T (identity on the padding), kept twice ...
            self.WR.diagonal()[:n].add_(self.kpS[1])              # ... + s_A I, the shift read on the device
            # m = R S^-1 g = (s_A I - P P^T) g / s = (s_A / s)(g - P alpha), alpha = L^-1 g the riding row (P^T g = s_A alpha):
            # one triangular product, R itself is not read.  s_A = s unless K + S needed jitter.
            ops.trmv_upper(rows, n, self.WA[self.g_row], g, -1.0, self.m)
            if self._sA != self.Sii:
                self.m.mul_(self._sA / self.Sii)
        else:
            ops.kbuild(self.kernel, self.X, self.kp0, self.WR, np_, np_, ops.KB_LOWER | ops.KB_PAD_IDENTITY)
            ops.syrk_diff(self.WR, rows, np_, self.WR, n, np_, 0.0)
            torch.div(g, self.Sii, out=self.gs)                  # O(N) vector scaling (S^-1 g)
            ops.symv_lower(self.WR, n, self.gs, self.m)
        self.Rdiag.copy_(self.WR.diagonal()[:n])

    def _make_chol_R(self):
        """the happy path of ``_make_R`` + ``_chol_R(0.0)``: in the inverse form the steps between the SYRK and the
        factorisation -- shift the diagonal by s_A, save it, add the 1e-11, write f - m into the riding row -- are ONE launch
        (``gpmc_shift_diag_sub``: they were four elementwise launches, ~25 us of a 0.4 ms proposal at N = 455); the same
        additions in the same order, the same bits"""
        if self.form != 'inv':
            self._make_R()
            self._chol_R(0.0)
            return
        n, np_ = self.n, self.np_
        rows = self.WA[self.rows0:self.rows0 + np_]
        ops.syrk_tri(rows, self.WR, n, np_, 0.0, C2=self.R0)      # -P P^T (identity on the padding), kept twice
        ops.trmv_upper(rows, n, self.WA[self.g_row], self.g_in, -1.0, self.m)
        if self._sA != self.Sii:
            self.m.mul_(self._sA / self.Sii)
        ops.shift_diag_sub(self.WR, n, self.kpS[1:2], R_JITTER, self.Rdiag, self.f_in, self.m, self.WR[np_, :n])
        ops.potrf(self.WR, self.mpR, np_, self.infoR, self.scratchR, rhs_rows=1)

    def _chol_R(self, jitter):
        """C = chol(R + (1e-11 + jitter) I) in place, eta = C^-1 (f - m) riding"""
        n, np_ = self.n, self.np_
        self.WR.diagonal()[:n].add_(R_JITTER + jitter)
        torch.sub(self.f_in, self.m, out=self.WR[np_, :n])
        ops.potrf(self.WR, self.mpR, np_, self.infoR, self.scratchR, rhs_rows=1)                 # f - m

    def _results(self):
        """[G, logdet, quad, infoA], infoR and f' -> pinned host memory, written by a kernel (asynchronous; read after a
        stream synchronisation)"""
        ops.publish_host(self.out, self.infoR, self.prop_f, self.n, self.res_host)

    def _read(self):
        (self._last_stream or torch.cuda.current_stream()).synchronize()       # the one sync of the happy path
        self._last_stream = None
        r = self.res_host
        return float(r[0]), _lib.check_info(r[3]), _lib.check_info(r[4])

    def aux_var_model(self, f, hyp, g, need_eta, maxtries=5):
        """Factor everything for hyper-parameters ``hyp`` given latent f and auxiliary g (device vectors).

        Returns G = log N(g; 0, K + S); afterwards self.m, the lower triangle of self.WR (= chol_R_theta),
        self.eta and self.WA (L, alpha, and P = s L^-T or V^T) are valid.  A failed factorisation follows
        the jitchol schedule the reference pins (pyGPs 1.3.4): mean(diag) * 1e-9 added, x10 per retry, ``maxtries``
        retries, then np.linalg.LinAlgError propagates to the caller exactly as in the reference.  R has eigenvalues at
        rounding level, so that schedule is part of normal operation (about a third of the bench's points at N = 8192):
        the launch that forms R writes it twice (gpmc_syrk_tri2), and a failure restores R from the second copy for every retry.
        (``need_eta`` is kept for the signature: eta rides on the factorisation of R either way.)"""
        if f is not self.f_in:
            self.f_in.copy_(f)
        if g is not self.g_in:
            self.g_in.copy_(g)
        self._retried = False
        self._set_hyp(hyp)
        self._factor_A(0.0)
        self._make_chol_R()
        self._results()
        return self._finish(maxtries)

    def _finish(self, maxtries=5):
        """read the results of the happy-path sequence (eager or replayed) and run the jitter schedules if it failed"""
        G, infoA, infoR = self._read()
        if infoA != 0:
            self._dirty = True
            diag_mean = self._kp0_sf2 + self.Sii                 # diag(K + S) is constant and positive
            jitter = diag_mean * 1e-9
            for _ in range(maxtries):
                self._factor_A(jitter)
                self._make_chol_R()
                self._results()
                G, infoA, infoR = self._read()
                if infoA == 0:
                    break
                self._dirty = True
                jitter *= 10.
            else:
                raise np.linalg.LinAlgError('not positive definite, even with jitter.')
        if infoR != 0:
            rdiag_mean = float(self.Rdiag.mean())
            if not rdiag_mean > 0:
                raise np.linalg.LinAlgError('not pd: non-positive diagonal elements')
            jitter = rdiag_mean * 1e-9
            if self.form == 'inv':                               # the failed attempt overwrote R: R0 and Rdiag restore it
                self.R0.diagonal()[:self.n].copy_(self.Rdiag)
            else:
                self._make_R()
                if self.R0 is None:
                    self.R0 = torch.empty((self.np_, self.np_), dtype=self.dtype, device=self.WR.device)
                self.R0.copy_(self.WR[:self.np_])
            for t in range(maxtries):
                self.WR[:self.np_].copy_(self.R0)
                self._chol_R(jitter)
                if _lib.check_info(self.infoR.item()) == 0:
                    break
                jitter *= 10.
            else:
                raise np.linalg.LinAlgError('not positive definite, even with jitter.')
            self._retried = True
            self._jitR = jitter                                  # what the schedule added on top of the 1e-11
        return G

    # ------------------------------------------------------------------ one proposal of the sampler's inner loop
    def begin(self, f, g, ita):
        """fix the iteration's latent f, auxiliary g and whitened latent eta (sliceSample.py:107-108): the proposals of
        this iteration (sliceSample.py:131-163) only change the hyper-parameters"""
        if f is not self.f_in:
            self.f_in.copy_(f)
        if g is not self.g_in:
            self.g_in.copy_(g)
        self.ita_in.copy_(ita)

    def _sequence(self):
        """the happy path of one proposal, without a host value in any launch and without a sync: K' + S' with g and
        s I riding -> L', alpha', P' -> G'; R' = s I - P' P'^T, m'; chol(R' + 1e-11 I) with f - m' riding; f' = C' eta + m';
        results and f' to pinned host memory"""
        self._factor_A(0.0)
        self._make_chol_R()
        ops.trmv_lower(self.WR, self.n, self.ita_in, self.m, self.prop_f)
        self._results()

    def _record(self):
        """Record the proposal sequence once (``gpmc_seq_*``: stream capture around the same calls -- the library's
        launches and torch's elementwise kernels go to the capturing stream, the look-ahead's second stream joins the
        capture through its event edges) and replay it per proposal.  The sequence holds kernels and memsets only: a
        first version with copy nodes in it (torch ``copy_`` of contiguous vectors, D2H copies of the results) replayed
        out of order on HIP runtime 7.0 from the second replay on (round 3), hence ``gpmc_veccopy`` and
        ``gpmc_publish_host``.  The exec is retired through the library's graveyard (csrc/sched.hip)."""
        import ctypes
        L = _lib.lib()
        side = torch.cuda.Stream(device=self.X.device)
        side.wait_stream(torch.cuda.current_stream())
        h = ctypes.c_void_p()
        with torch.cuda.stream(side):
            _lib.check(L.gpmc_seq_begin(ctypes.byref(h), side.cuda_stream), 'gpmc_seq_begin')
            try:
                self._sequence()
            finally:
                rc = L.gpmc_seq_end(h)
            _lib.check(rc, 'gpmc_seq_end')
        torch.cuda.current_stream().wait_stream(side)
        self._seq, self._seq_stream = h, side

    def close(self):
        if getattr(self, '_seq', None) is not None:
            torch.cuda.synchronize()
            _lib.lib().gpmc_seq_retire(self._seq)
            self._seq = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def proposal(self, hyp, maxtries=5, record=None):
        """(G', f') for proposed hyper-parameters: the reference's loop body sliceSample.py:136-147 (K build, aux_var_model
        with the iteration's g, prop_f, propG).  ONE stream synchronisation per proposal either way; ``record`` (None =
        by size): the first proposal of a workspace launches eagerly, the second records the sequence, later ones
        replay it -- which pays from about 6000 padded columns on (N = 8192: 15.76 vs 15.80 ms) and costs below that: a
        graph launch reaches its first kernel 20-30 us later than the first of ~30 eager launches does, 6 % of a proposal
        at the reference's demo sizes (N = 455: 0.410 vs 0.386 ms, 1024: 0.711 vs 0.690, 2048: 1.354 vs 1.325).  If a
        factorisation fails the jitter schedules run eagerly, exactly as in ``aux_var_model``."""
        if record is None:
            record = self.np_ >= RECORD_MIN_NP
        if os.environ.get('GPMC_SDS_RECORD') == '0':     # development: eager launches
            record = False
        self._set_hyp(hyp)
        self._retried = False
        self._sA = self.Sii              # the recorded sequence is the jitter-free path (the eager one sets it itself)
        if self._seq is None and record and self._proposals >= 1 and self.form == 'inv':
            self._clean()                                    # (not inside the recording)
            torch.cuda.current_stream().synchronize()
            self._record()
        if self._seq is not None and record:
            self._clean()
            # replayed on the stream it was recorded on, behind whatever the current stream holds (the staged parameters,
            # begin()'s copies): launched into the legacy NULL stream -- torch's default -- every node of the exec
            # synchronises with it, and a proposal at N = 1024 took 1.8 instead of 0.8 ms
            side = self._seq_stream
            side.wait_stream(torch.cuda.current_stream())
            _lib.check(_lib.lib().gpmc_seq_launch(self._seq, side.cuda_stream), 'gpmc_seq_launch')
            self._last_stream = side
        else:
            self._sequence()
        self._proposals += 1
        # whatever follows a replay -- reading the results, the jitter schedules if a pivot failed -- stays on the replay's
        # stream (the eager retries of a replayed proposal took 12 ms on the default stream against 8 after eager launches)
        side = self._last_stream
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            G = self._finish(maxtries)
            if self._retried or self._sA != self.Sii:
                # a jitter retry re-factored R (or K + S) after the sequence had formed f': form it again from the final factors
                ops.trmv_lower(self.WR, self.n, self.ita_in, self.m, self.prop_f)
                fp = self.prop_f.cpu().numpy()
            else:
                fp = self.res_host[8:8 + self.n].numpy().astype(np.float64, copy=True)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return G, fp

    @property
    def eta(self):
        return self.WR[self.np_, :self.n]

    def propose_f(self, eta):
        """f' = C eta + m  (sliceSample.py:140)"""
        return ops.trmv_lower(self.WR, self.n, eta, self.m, self.prop_f)

    # ------------------------------------------------------------------ dense views (tests / drop-in protocol)
    def L_KS(self):
        return ops.unpack_lower(self.WA, self.np_, self.n)

    def chol_R(self):
        return ops.unpack_lower(self.WR, self.np_, self.n)


def jitter_schedule(attempt, diag, maxtries=5):
    """The failure handling of pyGPs 1.3.4 ``jitchol`` (the version the reference pins, requirements.txt:10; call
    sites sliceSample.py:196,205), shared by SDSWorkspace-free callers: ``attempt(jitter)`` factors the matrix with
    ``jitter`` added to its diagonal and returns LAPACK-style info.  attempt(0) first; on failure every diagonal
    element must be positive, then jitter = mean(diag) * 1e-9, x10 per retry, ``maxtries`` retries, then
    LinAlgError.  Returns the jitter that worked."""
    if attempt(0.0) == 0:
        return 0.0
    if bool((diag <= 0).any()):
        raise np.linalg.LinAlgError('not pd: non-positive diagonal elements')
    jitter = float(diag.mean()) * 1e-9
    for _ in range(maxtries):
        if attempt(jitter) == 0:
            return jitter
        jitter *= 10.
    raise np.linalg.LinAlgError('not positive definite, even with jitter.')


def aux_var_model(f, K, sn, g=None, kernel_ws=None):
    """Protocol-compatible ``aux_var_model(f, K, sn, g=None)`` -> (g, K+S, m_theta_g, chol_R_theta, L)
    for a dense K handed in by the caller (ndarray or device tensor); returns device tensors, or ndarrays after
    ``gpmc.set_return('numpy')``.  Slower than SDSWorkspace (K is packed instead of built in place) -- it exists so
    code written against the reference's function keeps working.  Same inverse form as SDSWorkspace, here with the
    general (vector) S of sliceSample.py:183-190: riding rows diag(S) come back as P = S L^-T, R = diag(S) - P P^T,
    m = g - P (L^-1 g).  Both factorisations go through ``jitter_schedule`` as the reference's go through jitchol
    (sliceSample.py:196,205).  As in the reference the K+S that is RETURNED is the unjittered one (:207); a jitter j
    that chol(K+S) needed acts as S' = S + j I in R and m = R S^-1 g (algebraically the reference's
    K - K (K+S+jI)^-1 K); the riding right-hand side is then D g with D = (S + jI) S^-1, which makes m exact for a
    non-constant diagonal S too (a scalar rescaling of m afterwards is only right for S = s I)."""
    Kd = ops.to_dev(K)
    n = Kd.shape[0]
    dev, dt = Kd.device, Kd.dtype
    f = ops.to_dev(f).reshape(-1)
    Kii = torch.diagonal(Kd)
    kinv = 1. / Kii
    Sii = torch.clamp(1. / (((sn ** 2) ** (-1) + kinv) - kinv), min=0.)
    if g is None:
        g = f + torch.sqrt(Sii) * ops.to_dev(np.random.standard_normal(n))    # legacy stream, see oracle.draw_g
    else:
        g = ops.to_dev(g).reshape(-1)
    np_ = ops.pad(n)
    KS = Kd + torch.diag(Sii)
    # workspace: [K + S ; g block ; diag(S) rows]  (layout of SDSWorkspace form 'inv')
    mpA = 2 * np_ + ops.PAD
    WA = torch.zeros((mpA, np_), dtype=dt, device=dev)
    rows = WA[np_ + ops.PAD:]
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = ops.potrf_scratch(np_, dev)

    def factor_A(jitter):
        # the riding row holds g' = D g, D = (S + jitter I) S^-1 (= g without jitter), so that g' - P (L^-1 g') below is
        # R S^-1 g for ANY diagonal S: R S^-1 g = (S' - S' A'^-1 S') S^-1 g = g' - S' A'^-1 g' with S' = S + jitter I
        ops.pack_lower(KS, WA, np_, jitter)
        WA[np_:].zero_()
        WA[np_, :n].copy_(g * ((Sii + jitter) / Sii) if jitter else g)
        torch.diagonal(rows)[:n].copy_(Sii + jitter)
        if np_ > n:
            torch.diagonal(rows)[n:].fill_(1.0)
        ops.potrf_tri(WA, mpA, np_, np_, info, scratch)
        return _lib.check_info(info.item())

    jA = jitter_schedule(factor_A, torch.diagonal(KS))
    L = ops.unpack_lower(WA, np_, n)
    SA = Sii + jA
    m = torch.empty(n, dtype=dt, device=dev)
    gp = g * (SA / Sii) if jA else g                         # D g, see factor_A
    ops.trmv_upper(rows, n, WA[np_], gp, -1.0, m)            # g' - P (L^-1 g') = R S^-1 g with the unjittered S (sliceSample.py:204)
    WR = torch.zeros((np_ + ops.PAD, np_), dtype=dt, device=dev)
    R0 = torch.empty((np_, np_), dtype=dt, device=dev)
    ops.syrk_tri(rows, R0, n, np_, 0.0)                      # -P P^T (identity on the padding)
    R0.diagonal()[:n].add_(SA)                               # R = diag(S') - P P^T

    def factor_R(jitter):
        WR[:np_].copy_(R0)
        WR[np_:].zero_()
        WR.diagonal()[:n].add_(R_JITTER + jitter)            # R + 1e-11 I (sliceSample.py:205) + the schedule's jitter
        ops.potrf(WR, np_ + ops.PAD, np_, info, scratch)
        return _lib.check_info(info.item())

    jitter_schedule(factor_R, R0.diagonal()[:n] + R_JITTER)
    C = ops.unpack_lower(WR, np_, n)
    return tuple(ops.ret(t) for t in (g, KS, m, C, L))
