"""ctypes binding of libdspn_hip.so (the C ABI declared in include/*.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the
shared object is missing or a symbol cannot be resolved, importing an operator
raises immediately.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSPN_LIB: another build of the SAME library (A/B timing of two builds on one box); never a fallback
LIB_PATH = os.environ.get("DSPN_LIB") or os.path.join(_HERE, "libdspn_hip.so")

_c = ctypes
_f32p = _c.POINTER(_c.c_float)
_vp = _c.c_void_p

# name -> (restype, argtypes); mirrors include/dspn_multibox.h and include/dspn_nn.h
SIGNATURES = {
    "dspn_last_error": (_c.c_char_p, []),
    "dspn_abi_version": (_c.c_int, []),
    "dspn_profile_enable": (_c.c_int, [_c.c_int]),
    "dspn_conv_set_reserved_cus": (_c.c_int, [_c.c_int]),
    "dspn_conv_set_wide_tiles": (_c.c_int, [_c.c_int]),
    "dspn_conv_set_tile_spanning": (_c.c_int, [_c.c_int]),
    "dspn_affine_sampler_set_batched": (_c.c_int, [_c.c_int]),
    "dspn_profile_collect": (_c.c_int, [_c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_longlong)]),
    "dspn_multibox_prior_f32": (_c.c_int, [_f32p, _c.c_int, _f32p, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_float, _c.c_float, _c.c_float, _c.c_float,
                                           _c.c_int, _vp, _vp]),
    "dspn_multibox_target_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "dspn_multibox_target_f32": (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                            _c.c_int, _c.c_float, _c.c_float, _c.c_float,
                                            _c.c_float, _c.c_int, _f32p, _vp, _vp, _vp, _vp,
                                            _c.c_size_t, _vp]),
    "dspn_multibox_target_errors": (_c.c_int, [_vp, _c.c_int, _c.POINTER(_c.c_int), _vp]),
    "dspn_multibox_detection_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "dspn_multibox_detection_f32": (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int, _c.c_int,
                                               _c.c_float, _c.c_int, _f32p, _c.c_float, _c.c_int,
                                               _c.c_int, _vp, _vp, _c.c_size_t, _vp]),
}


class DspnError(RuntimeError):
    """Raised when a C-ABI call returns a non-zero status (the MXNetError analogue)."""


_lib = None


def register(signatures):
    """Used by sibling modules to add the signatures of further include/*.h files."""
    SIGNATURES.update(signatures)
    if _lib is not None:
        _bind(_lib, signatures)


def _bind(lib, signatures):
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild with "
                              f"`make -C dspnet_amd/csrc`") from e
        fn.restype = res
        fn.argtypes = args


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                f"Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                f"`make -C dspnet_amd/csrc`.")
        handle = ctypes.CDLL(LIB_PATH)
        _bind(handle, SIGNATURES)
        _lib = handle
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().dspn_last_error().decode("utf-8", "replace")
        raise DspnError(f"{what}: {msg} (status {status})" if what else f"{msg} (status {status})")


def floats(values):
    arr = (_c.c_float * len(values))(*[float(v) for v in values])
    return arr
