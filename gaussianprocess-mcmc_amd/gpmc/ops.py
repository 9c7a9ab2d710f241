"""Typed wrappers over the C ABI: torch tensors in, raw device pointers + the current HIP stream out.

torch is the device container only (memory, streams); every arithmetic op below is a libgpmc kernel.
"""
import numpy as np
import torch

from . import _lib
from ._lib import KB_FULL, KB_LOWER, KB_PAD_IDENTITY, KERNEL_MATERN52_ARD, KERNEL_RBF, PAD, check, pad  # noqa: F401

KERNELS = {'rbf': KERNEL_RBF, 'matern52ard': KERNEL_MATERN52_ARD}


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.GpmcError('no HIP device visible: the gpmc product path is GPU-only (no CPU fallback)')


def device():
    require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


# ---- what the kcGP-protocol mirrors hand back: 'torch' (device tensors, default) or 'numpy' (fresh host ndarrays, so
# that NumPy code written against the reference's kcGP -- np.diagonal(K), np.linalg.solve(L, K), sliceSample.py:182-205 --
# runs on top unchanged; costs one D2H copy per returned matrix)
_RETURN = 'torch'


def set_return(mode):
    global _RETURN
    if mode not in ('torch', 'numpy'):
        raise ValueError("return mode must be 'torch' or 'numpy'")
    _RETURN = mode


def get_return():
    return _RETURN


def ret(t):
    """a result of a protocol mirror in the configured return form"""
    return t.cpu().numpy() if _RETURN == 'numpy' else t


def sfx(dtype):
    if dtype == torch.float64:
        return 'f64'
    if dtype == torch.float32:
        return 'f32'
    raise TypeError('gpmc supports float64 and float32, got %r' % (dtype,))


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def to_dev(a, dtype=torch.float64):
    """ndarray / list / tensor -> contiguous device tensor of dtype (no copy when already there)."""
    if isinstance(a, torch.Tensor):
        return a.to(device=device(), dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device())


def _fn(name, dtype):
    return getattr(_lib.lib(), 'gpmc_%s_%s' % (name, sfx(dtype)))


def kbuild(kernel, X, kparams_dev, out, n_rows, n_cols, flags, Z=None):
    """out[0:rows_out, 0:cols_out] <- k(X, Z) per include/gpmc.h (out is a 2-D row-major view)."""
    dt = X.dtype
    n, d = X.shape
    nz = 0 if Z is None else Z.shape[0]
    check(_fn('kbuild', dt)(KERNELS[kernel] if isinstance(kernel, str) else kernel,
                            X.data_ptr(), n, d, X.stride(0),
                            0 if Z is None else Z.data_ptr(), nz, 0 if Z is None else Z.stride(0),
                            kparams_dev.data_ptr(), out.data_ptr(), n_rows, n_cols, out.stride(0), flags,
                            stream_ptr()), 'gpmc_kbuild')
    return out


def potrf(W, mp, np_, info, scratch, rhs_rows=-1):
    """``rhs_rows``: rows in use of the riding block right below the matrix (-1: unknown), see gpmc_potrf_rhs"""
    check(_fn('potrf_rhs', W.dtype)(W.data_ptr(), mp, np_, W.stride(0), 0, rhs_rows, info.data_ptr(), scratch.data_ptr(),
                                    stream_ptr()), 'gpmc_potrf')


def potrf_tri(W, mp, np_, ntri, info, scratch, rhs_rows=-1):
    """gpmc_potrf_tri: the last ``ntri`` riding rows are upper-trapezoidal on entry (e.g. s*I -> s*L^-T)"""
    check(_fn('potrf_rhs', W.dtype)(W.data_ptr(), mp, np_, W.stride(0), ntri, rhs_rows, info.data_ptr(),
                                    scratch.data_ptr(), stream_ptr()), 'gpmc_potrf_tri')


def reset_tri(rows, np_, diag_dev):
    """rows <- upper-trapezoidal (diag_dev[0]) I: diagonal set, zero right of it, strict lower part untouched"""
    check(_fn('reset_tri', rows.dtype)(rows.data_ptr(), np_, rows.stride(0), diag_dev.data_ptr(), stream_ptr()),
          'gpmc_reset_tri')


def veccopy(dst, src, n):
    check(_fn('veccopy', dst.dtype)(dst.data_ptr(), src.data_ptr(), n, stream_ptr()), 'gpmc_veccopy')


def shift_diag_sub(R, n, shift_dev, jitter, diag_out, f, m, row_out):
    """diag(R) += shift_dev[0]; diag_out = diag(R); diag(R) += jitter; row_out = f - m -- one launch (SDS proposal)"""
    check(_fn('shift_diag_sub', R.dtype)(R.data_ptr(), R.stride(0), n, shift_dev.data_ptr(), float(jitter), diag_out.data_ptr(),
                                         f.data_ptr(), m.data_ptr(), row_out.data_ptr(), stream_ptr()), 'gpmc_shift_diag_sub')


def publish_host(out4, info2, vec, n, host_pinned):
    """out4, *info2 and vec[0:n] into pinned host memory (float64 tensor of >= 8 + n, device-mapped): no copy node"""
    check(_fn('publish_host', vec.dtype)(out4.data_ptr(), info2.data_ptr(), vec.data_ptr(), n, host_pinned.data_ptr(),
                                         stream_ptr()), 'gpmc_publish_host')


def logdet_quad(W, n, alpha, info, out):
    check(_fn('logdet_quad', W.dtype)(W.data_ptr(), n, W.stride(0), alpha.data_ptr(),
                                      0 if info is None else info.data_ptr(), out.data_ptr(), stream_ptr()),
          'gpmc_logdet_quad')


def pack_lower(A, W, np_, jitter=0.0):
    n = A.shape[0]
    check(_fn('pack_lower', A.dtype)(A.data_ptr(), n, A.stride(0), float(jitter), W.data_ptr(), np_, W.stride(0),
                                     stream_ptr()), 'gpmc_pack_lower')


def pack_rows(B, W, mp, np_, n):
    """B: (nrhs, n) rows, or None to zero the riding rows."""
    nrhs = 0 if B is None else B.shape[0]
    check(_fn('pack_rows', W.dtype)(0 if B is None else B.data_ptr(), nrhs, n, 0 if B is None else B.stride(0),
                                    W.data_ptr(), mp, np_, W.stride(0), stream_ptr()), 'gpmc_pack_rows')


def unpack_lower(W, np_, n, out=None):
    if out is None:
        out = torch.empty((n, n), dtype=W.dtype, device=W.device)
    check(_fn('unpack_lower', W.dtype)(W.data_ptr(), np_, W.stride(0), out.data_ptr(), n, out.stride(0),
                                       stream_ptr()), 'gpmc_unpack_lower')
    return out


def unpack_rows(W, np_, nrhs, n, out=None):
    if out is None:
        out = torch.empty((nrhs, n), dtype=W.dtype, device=W.device)
    check(_fn('unpack_rows', W.dtype)(W.data_ptr(), np_, W.stride(0), out.data_ptr(), nrhs, n, out.stride(0),
                                      stream_ptr()), 'gpmc_unpack_rows')
    return out


def syrk_tri(P, C, n, np_, diag_add=0.0, C2=None):
    """C = diag_add*I - P P^T for an upper-trapezoidal P (np_ x np_ rows of gpmc_potrf_tri); with ``C2`` the result is
    written to both matrices (same leading dimension)"""
    if C2 is not None:
        if C2.dtype != C.dtype or C2.stride(0) != C.stride(0) or C2.data_ptr() == C.data_ptr():
            raise ValueError('syrk_tri: C2 must be another matrix of the same dtype and leading dimension')
        check(_fn('syrk_tri2', C.dtype)(P.data_ptr(), P.stride(0), float(diag_add), C.data_ptr(), C2.data_ptr(), n, np_,
                                        C.stride(0), stream_ptr()), 'gpmc_syrk_tri2')
        return C
    check(_fn('syrk_tri', C.dtype)(P.data_ptr(), P.stride(0), float(diag_add), C.data_ptr(), n, np_, C.stride(0),
                                   stream_ptr()), 'gpmc_syrk_tri')
    return C


def syrk_diff(D, P, kp, C, n, np_, diag_add=0.0):
    check(_fn('syrk_diff', C.dtype)(D.data_ptr(), D.stride(0), P.data_ptr(), kp, P.stride(0), float(diag_add),
                                    C.data_ptr(), n, np_, C.stride(0), stream_ptr()), 'gpmc_syrk_diff')


def symv_lower(W, n, x, y):
    check(_fn('symv_lower', W.dtype)(W.data_ptr(), n, W.stride(0), x.data_ptr(), y.data_ptr(), stream_ptr()),
          'gpmc_symv_lower')
    return y


def trmv_lower(W, n, x, y0, y):
    check(_fn('trmv_lower', W.dtype)(W.data_ptr(), n, W.stride(0), x.data_ptr(),
                                     0 if y0 is None else y0.data_ptr(), y.data_ptr(), stream_ptr()),
          'gpmc_trmv_lower')
    return y


def trmv_upper(W, n, x, y0, alpha, y):
    """y = y0 + alpha * (U x), U = the upper-trapezoidal rows held in W (row i: columns >= i)"""
    check(_fn('trmv_upper', W.dtype)(W.data_ptr(), n, W.stride(0), x.data_ptr(),
                                     0 if y0 is None else y0.data_ptr(), float(alpha), y.data_ptr(), stream_ptr()),
          'gpmc_trmv_upper')
    return y


def trsm_rows(W, np_, R, nrows):
    check(_lib.lib().gpmc_trsm_rows_f64(W.data_ptr(), np_, W.stride(0), R.data_ptr(), nrows, R.stride(0), stream_ptr()),
          'gpmc_trsm_rows_f64')


def revt_lower(L, n, out):
    check(_lib.lib().gpmc_revt_lower_f64(L.data_ptr(), n, L.stride(0), out.data_ptr(), out.stride(0), stream_ptr()),
          'gpmc_revt_lower_f64')
    return out


def gemm_nt(A, B, C, M, N, K):
    """C[:M,:N] = A[:M,:K] @ B[:N,:K].T (padded sizes: M, N multiples of 64, K of 16)"""
    check(_lib.lib().gpmc_gemm_nt_f64(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0),
                                      M, N, K, stream_ptr()), 'gpmc_gemm_nt_f64')
    return C


def potrf_scratch(np_, dev):
    nbytes = _lib.lib().gpmc_potrf_scratch_bytes(np_)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)
