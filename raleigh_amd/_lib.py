"""ctypes binding of librlhip.so (include/rlhip.h).

The product path has NO CPU fallback: if the shared library is missing or
cannot be loaded, or no MI355X is visible, every entry point raises.  Tests may
install a stand-in object with the same attributes through ``set_library``
(tests/fake_lib.py) to exercise the host logic on a CPU-only machine.
"""

import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_PKG, 'lib', 'librlhip.so')

RLH_S, RLH_D, RLH_C, RLH_Z = 0, 1, 2, 3

DTYPE_CODE = {np.float32: RLH_S, np.float64: RLH_D, np.complex64: RLH_C, np.complex128: RLH_Z}
DTYPE_SIZE = {np.float32: 4, np.float64: 8, np.complex64: 8, np.complex128: 16}

_i64 = ctypes.c_int64
_p = ctypes.c_void_p
_int = ctypes.c_int

# name -> argtypes, exactly the prototypes of include/rlhip.h
SIGNATURES = {
    'rlh_version': [],
    'rlh_last_error': [],
    'rlh_device_count': [ctypes.POINTER(_int)],
    'rlh_init': [_int],
    'rlh_finalize': [],
    'rlh_set_stream': [_p],
    'rlh_sync': [],
    'rlh_mem_info': [ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    'rlh_malloc': [ctypes.POINTER(_p), _i64],
    'rlh_free': [_p],
    'rlh_memset': [_p, _int, _i64],
    'rlh_h2d': [_p, _p, _i64],
    'rlh_d2h': [_p, _p, _i64],
    'rlh_d2d': [_p, _p, _i64],
    'rlh_fetch': [_p, _p, _i64],
    'rlh_copy2d': [_p, _i64, _p, _i64, _i64, _i64, _int],
    'rlh_gram': [_int, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _p],
    'rlh_gram_multi': [_int, _i64, _int, _p, _p, _p, _int, _p, _p, _p, _p, _p],
    'rlh_dots': [_int, _i64, _i64, _p, _i64, _p, _i64, _p, _p],
    'rlh_absmax': [_int, _i64, _i64, _p, _i64, ctypes.POINTER(ctypes.c_double)],
    'rlh_dots_transp': [_int, _i64, _i64, _p, _i64, _p, _i64, _p],
    'rlh_block_update': [_int, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _int],
    'rlh_block_update2': [_int, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _i64,
                          _p, _int],
    'rlh_block_update2x2': [_int, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _i64,
                            _i64, _p, _i64],
    'rlh_lincomb_cols': [_int, _i64, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i64],
    'rlh_axpy': [_int, _i64, _i64, _p, _p, _i64, _p, _i64],
    'rlh_axpy_cols': [_int, _i64, _i64, _p, _p, _i64, _p, _i64],
    'rlh_copy': [_int, _i64, _i64, _p, _i64, _p, _i64],
    'rlh_copy_cols': [_int, _i64, _i64, _p, _p, _i64, _p, _i64],
    'rlh_scale_cols': [_int, _i64, _i64, _p, _int, _p, _i64],
    'rlh_convert': [_int, _int, _i64, _i64, _p, _i64, _p, _i64],
    'rlh_fill_random': [_int, _i64, _i64, _p, _i64, ctypes.c_uint64, _i64, _i64],
    'rlh_conj': [_int, _i64, _i64, _p, _i64],
    'rlh_gather_rows': [_int, _i64, _p, _i64, _p, _i64, _p, _i64],
    'rlh_csr_create': [ctypes.POINTER(_p), _int, _i64, _i64, _p, _p, _p],
    'rlh_csr_create_upper': [ctypes.POINTER(_p), _int, _i64, _p, _p, _p],
    'rlh_csr_destroy': [_p],
    'rlh_csr_info': [_p, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                     ctypes.POINTER(_i64)],
    'rlh_csr_layout': [_p, ctypes.POINTER(_int), ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)],
    'rlh_csr_stacks': [_p, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
    'rlh_csr_bf16_ready': [_p, _i64, _i64, ctypes.POINTER(_int)],
    'rlh_spmm': [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _i64],
    'rlh_spmm_cheb': [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, ctypes.c_double, ctypes.c_double,
                      ctypes.c_double],
    'rlh_spmm_part': [_p, _int, _i64, _p, _i64, _i64, _p, _i64, _p, _i64],
    'rlh_spmm_cheb_part': [_p, _int, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, ctypes.c_double,
                           ctypes.c_double, ctypes.c_double],
    'rlh_bf16_pack': [_int, _i64, _i64, _p, _i64, ctypes.c_double, _p, _i64],
    'rlh_bf16_unpack': [_int, _i64, _i64, _p, _i64, _p, _i64],
    'rlh_spmm_cheb_bf16': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, ctypes.c_double, ctypes.c_double, ctypes.c_double],
    'rlh_spmm_cheb_bf16_part': [_p, _int, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, ctypes.c_double,
                                ctypes.c_double, ctypes.c_double],
    'rlh_gather_rows_bf16': [_i64, _p, _i64, _p, _i64, _p, _i64],
    'rlh_ilut_factor': [ctypes.POINTER(_p), _int, _i64, _p, _p, _p, ctypes.c_double, _i64],
    'rlh_factors_nnz': [_p, ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    'rlh_factors_get': [_p, _int, _p, _p, _p],
    'rlh_factors_destroy': [_p],
    'rlh_ldlt_factor': [ctypes.POINTER(_p), _int, _i64, _p, _p, _p, _p, ctypes.c_double, ctypes.c_double],
    'rlh_ldlt_info': [_p, _p],
    'rlh_ldlt_get': [_p, _p, _p, _p, _p, _p, _p, _p],
    'rlh_ldlt_get_transposed': [_p, _p, _p, _p],
    'rlh_ldlt_destroy': [_p],
    'rlh_bdiag_solve': [_int, _i64, _p, _p, _i64, _p, _i64],
    'rlh_shm_create': [ctypes.POINTER(_p), ctypes.c_char_p, _int, _int, _i64],
    'rlh_shm_unlink': [ctypes.c_char_p],
    'rlh_shm_allreduce': [_p, _int, _i64, _p],
    'rlh_shm_destroy': [_p],
    'rlh_sptrsv_create': [ctypes.POINTER(_p), _int, _i64, _p, _p, _p, _int, _int],
    'rlh_sptrsv_info': [_p, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    'rlh_sptrsv_solve_chain': [_int, _p, _p, _p, _i64, _p, _i64, _p, _i64],
    'rlh_sptrsv_destroy': [_p],
    'rlh_dense_apply': [_int, _i64, _i64, _p, _i64, _int, _int, _i64, _p, _i64, _p, _i64],
    'rlh_dense_apply_r1': [_int, _i64, _i64, _p, _i64, _int, _int, _i64, _p, _i64, _p, _i64, _p, _p],
    'rlh_timer_start': [],
    'rlh_timer_stop': [ctypes.POINTER(ctypes.c_float)],
}

_lib = None
_initialised_device = None


class RlhError(RuntimeError):
    """A librlhip call failed (the reference raises RuntimeError('cuda error %d'),
    raleigh/algebra/dense_cublas.py:779-781)."""


def _load():
    if not os.path.exists(LIBPATH):
        raise RlhError(
            'librlhip.so not found at %s: build it with `python -m raleigh_amd.build` '
            '(there is no CPU fallback for the hip backend)' % LIBPATH)
    try:
        dll = ctypes.CDLL(LIBPATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise RlhError('cannot load %s: %s' % (LIBPATH, e))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(dll, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == 'rlh_last_error' else _int
    return dll


def set_library(obj):
    """Installs a replacement for the shared library (test infrastructure only)."""
    global _lib, _initialised_device
    _lib = obj
    _initialised_device = None


def library():
    """The loaded shared library (no device initialisation)."""
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def check(rc):
    if rc != 0:
        msg = library().rlh_last_error()
        if isinstance(msg, bytes):
            msg = msg.decode('utf-8', 'replace')
        raise RlhError('rlhip error %d: %s' % (rc, msg))


def default_device():
    return int(os.environ.get('LOCAL_RANK', '0'))


def lib(device=None):
    """The library with the device context initialised (once per process)."""
    global _initialised_device
    L = library()
    if _initialised_device is None:
        dev = default_device() if device is None else int(device)
        check(L.rlh_init(dev))
        _initialised_device = dev
    elif device is not None and int(device) != _initialised_device:
        raise RlhError('process is bound to device %d, cannot switch to %d (one process per GPU)'
                       % (_initialised_device, int(device)))
    return L


def device():
    return _initialised_device


def synchronize():
    """Counterpart of cuda.synchronize() (raleigh/algebra/cuda_wrap.py)."""
    check(lib().rlh_sync())


def dtype_code(dt):
    dt = np.dtype(dt).type
    if dt not in DTYPE_CODE:
        raise ValueError('data type %s not supported' % repr(dt))
    return DTYPE_CODE[dt]


def host_ptr(a):
    """void* of a numpy array (borrowed for the duration of the call)."""
    return ctypes.c_void_p(a.ctypes.data)
