"""ctypes binding of libgpmc.so (the C ABI declared in include/gpmc.h).

The product path has no CPU fallback: if the library is missing or a call fails, this module raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GPMC_LIB') or os.path.join(_HERE, 'libgpmc.so')      # GPMC_LIB: another build of the same ABI (development A/B)
CSRC = os.path.join(os.path.dirname(_HERE), 'csrc')

KERNEL_RBF = 0
KERNEL_MATERN52_ARD = 1
KB_FULL = 0
KB_LOWER = 1
KB_PAD_IDENTITY = 2
PAD = 128

c_int, c_void_p, c_double, c_float, c_size_t, c_char_p = (ctypes.c_int, ctypes.c_void_p, ctypes.c_double,
                                                          ctypes.c_float, ctypes.c_size_t, ctypes.c_char_p)


class GpmcError(RuntimeError):
    pass


def build(force=False):
    """Compile libgpmc.so in-tree with hipcc for gfx950 (seconds; works without a GPU)."""
    if force:
        subprocess.check_call(['make', '-C', CSRC, 'clean'])
    subprocess.check_call(['make', '-C', CSRC, '-j4'])
    return LIB_PATH


ASAN_LIB_PATH = os.path.join(_HERE, 'libgpmc_asan.so')


def build_sanitized():
    """Host-side AddressSanitizer + UBSan build of the same sources (device code uninstrumented) for the no-GPU ABI
    tests: ``make SANITIZE=1`` -> libgpmc_asan.so.  Never loaded by the product path."""
    subprocess.check_call(['make', '-C', CSRC, 'SANITIZE=1', '-j4'])
    return ASAN_LIB_PATH


def _sig(lib):
    P, I = c_void_p, c_int
    kb = [I, P, I, I, I, P, I, I, P, P, I, I, I, I, P]
    table = {
        'gpmc_version': ([], I),
        'gpmc_last_error': ([], c_char_p),
        'gpmc_pad': ([I], I),
        'gpmc_potrf_scratch_bytes': ([I], c_size_t),
        'gpmc_kbuild_f64': (kb, I), 'gpmc_kbuild_f32': (kb, I),
        'gpmc_potrf_f64': ([P, I, I, I, P, P, P], I), 'gpmc_potrf_f32': ([P, I, I, I, P, P, P], I),
        'gpmc_logdet_quad_f64': ([P, I, I, P, P, P, P], I), 'gpmc_logdet_quad_f32': ([P, I, I, P, P, P, P], I),
        'gpmc_pack_lower_f64': ([P, I, I, c_double, P, I, I, P], I),
        'gpmc_pack_lower_f32': ([P, I, I, c_float, P, I, I, P], I),
        'gpmc_pack_rows_f64': ([P, I, I, I, P, I, I, I, P], I), 'gpmc_pack_rows_f32': ([P, I, I, I, P, I, I, I, P], I),
        'gpmc_unpack_lower_f64': ([P, I, I, P, I, I, P], I), 'gpmc_unpack_lower_f32': ([P, I, I, P, I, I, P], I),
        'gpmc_unpack_rows_f64': ([P, I, I, P, I, I, I, P], I), 'gpmc_unpack_rows_f32': ([P, I, I, P, I, I, I, P], I),
        'gpmc_potrf_rhs_f64': ([P, I, I, I, I, I, P, P, P], I), 'gpmc_potrf_rhs_f32': ([P, I, I, I, I, I, P, P, P], I),
        'gpmc_potrf_tri_f64': ([P, I, I, I, I, P, P, P], I),
        'gpmc_potrf_tri_f32': ([P, I, I, I, I, P, P, P], I),
        'gpmc_reset_tri_f64': ([P, I, I, P, P], I), 'gpmc_reset_tri_f32': ([P, I, I, P, P], I),
        'gpmc_veccopy_f64': ([P, P, I, P], I), 'gpmc_veccopy_f32': ([P, P, I, P], I),
        'gpmc_shift_diag_sub_f64': ([P, I, I, P, c_double, P, P, P, P, P], I),
        'gpmc_shift_diag_sub_f32': ([P, I, I, P, c_float, P, P, P, P, P], I),
        'gpmc_publish_host_f64': ([P, P, P, I, P, P], I), 'gpmc_publish_host_f32': ([P, P, P, I, P, P], I),
        'gpmc_seq_begin': ([ctypes.POINTER(c_void_p), P], I), 'gpmc_seq_end': ([P], I),
        'gpmc_seq_launch': ([P, P], I), 'gpmc_seq_retire': ([P], I),
        'gpmc_syrk_tri_f64': ([P, I, c_double, P, I, I, I, P], I),
        'gpmc_syrk_tri_f32': ([P, I, c_float, P, I, I, I, P], I),
        'gpmc_syrk_tri2_f64': ([P, I, c_double, P, P, I, I, I, P], I),
        'gpmc_syrk_tri2_f32': ([P, I, c_float, P, P, I, I, I, P], I),
        'gpmc_syrk_diff_f64': ([P, I, P, I, I, c_double, P, I, I, I, P], I),
        'gpmc_syrk_diff_f32': ([P, I, P, I, I, c_float, P, I, I, I, P], I),
        'gpmc_symv_lower_f64': ([P, I, I, P, P, P], I), 'gpmc_symv_lower_f32': ([P, I, I, P, P, P], I),
        'gpmc_trmv_lower_f64': ([P, I, I, P, P, P, P], I), 'gpmc_trmv_lower_f32': ([P, I, I, P, P, P, P], I),
        'gpmc_trmv_upper_f64': ([P, I, I, P, P, c_double, P, P], I),
        'gpmc_trmv_upper_f32': ([P, I, I, P, P, c_double, P, P], I),
        'gpmc_plan_workspace_bytes': ([I, I, I], c_size_t),
        'gpmc_plan_create': ([ctypes.POINTER(c_void_p), I, I, I, I, P, c_size_t, P], I),
        'gpmc_plan_set_data': ([P, P, I, P], I),
        'gpmc_plan_logml_async': ([P, ctypes.POINTER(c_double), I, I], I),
        'gpmc_plan_result': ([P, I, ctypes.POINTER(c_double)], I),
        'gpmc_plan_workspace': ([P, ctypes.POINTER(c_void_p), ctypes.POINTER(I), ctypes.POINTER(I),
                                 ctypes.POINTER(I)], I),
        'gpmc_plan_destroy': ([P], I),
        'gpmc_graveyard_stats': ([ctypes.POINTER(I), ctypes.POINTER(c_size_t)], I),
        'gpmc_plan_handover_in_launch': ([P], I),
        'gpmc_debug_panel_stamps': ([P, I, I, I, I, P, P, I, P, P], I),
        'gpmc_debug_tail_stamps': ([P], I),
        'gpmc_debug_graph_churn': ([P, I, I, I], I),
        'gpmc_debug_plan_record_info': ([P, ctypes.POINTER(c_double)], I),
        'gpmc_debug_plan_ops': ([P, ctypes.POINTER(I), ctypes.POINTER(I)], I),
        'gpmc_trsm_rows_f64': ([P, I, I, P, I, I, P], I),
        'gpmc_revt_lower_f64': ([P, I, I, P, I, P], I),
        'gpmc_gemm_nt_f64': ([P, I, P, I, P, I, I, I, I, P], I),
        'gpmc_profile_begin': ([], I),
        'gpmc_profile_read': ([ctypes.POINTER(c_double)], I),
        'gpmc_set_option': ([c_char_p, I], I),
        'gpmc_get_option': ([c_char_p], I),
    }
    for name, (argtypes, restype) in table.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = restype
    return sorted(table)


_lib = None
EXPORTS = None


def lib():
    """The loaded library; raises GpmcError when it has not been built."""
    global _lib, EXPORTS
    if _lib is None:
        # torch first: libgpmc.so must bind to the HIP runtime torch loads (its bundled one), not open a second copy of
        # the runtime beside it -- two runtimes in one process do not see each other's device
        # ("no ROCm-capable device is detected" from whichever came second)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise GpmcError('libgpmc.so not found at %s -- run `python -c "import __graft_entry__ as g; g.build()"` '
                            '(there is no CPU fallback)' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        EXPORTS = _sig(_lib)
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().gpmc_last_error()
        raise GpmcError('%s failed (rc=%d): %s' % (what or 'libgpmc call', rc, msg.decode() if msg else '?'))


TAIL_TIMEOUT = -1001     # *info when the tile-DAG kernel (csrc/tail.h) gave up waiting: an error, not "not positive definite"
TAIL_TIMEOUT_BASE = -1000000     # ... with the wait that ran out: -(1000000 + 10000 cause + tile column), csrc/tail.h
TAIL_WAIT_CAUSES = {1: "a hand-over flag of the trailing GEMM running beside the launch",
                    2: "an operand tile of an earlier column",
                    3: "the diagonal factor of a column"}


def decode_timeout(info):
    """(cause, tile column) of a tile-DAG timeout code, or None if ``info`` is not one"""
    info = int(info)
    if info == TAIL_TIMEOUT:
        return 0, -1
    if info <= TAIL_TIMEOUT_BASE:
        v = -info - 1000000
        return v // 10000, v % 10000
    return None


def check_info(info):
    """pass a factorisation's LAPACK-style info through; raise if it is a tile-DAG timeout code (the kernel's bounded
    spins ran out: the GPU was held by other work for about a second, or a producer kernel was never dispatched).  The
    message names the wait that expired."""
    info = int(info)
    t = decode_timeout(info)
    if t is not None:
        cause, col = t
        what = TAIL_WAIT_CAUSES.get(cause, 'another workgroup')
        raise GpmcError('tile-DAG factorisation kernel timed out waiting for %s%s (info=%d); if other processes share '
                        'this card call set_option("shared_card", 1); set_option("tail_m", 0) selects the launch-chain '
                        'schedule' % (what, (' (tile column %d)' % col) if col >= 0 else '', info))
    return info


def graveyard_stats():
    """(execs, nodes) parked instead of destroyed on HIP runtimes < 7.2 (include/gpmc.h, gpmc_graveyard_stats)"""
    e, n = c_int(), c_size_t()
    check(lib().gpmc_graveyard_stats(ctypes.byref(e), ctypes.byref(n)), 'gpmc_graveyard_stats')
    return int(e.value), int(n.value)


def set_option(name, value):
    check(lib().gpmc_set_option(name.encode(), int(value)), 'gpmc_set_option')


def get_option(name):
    return lib().gpmc_get_option(name.encode())


def profile_begin():
    check(lib().gpmc_profile_begin(), 'gpmc_profile_begin')


def profile_read():
    """((total_ms, launches, algorithmic_flops) of the trailing-update kernel, the same of the tile-DAG kernel) since
    profile_begin()"""
    out = (c_double * 6)()
    check(lib().gpmc_profile_read(out), 'gpmc_profile_read')
    return (float(out[0]), int(out[1]), float(out[2])), (float(out[3]), int(out[4]), float(out[5]))


def pad(n):
    return (int(n) + PAD - 1) // PAD * PAD
