"""ctypes binding of the C-ABI in include/dil256.h.  Loads the in-tree libdil256.so.

There is NO CPU fallback: if the library is missing this raises, and every compute entry
point returns the hipError_t of the failed HIP call when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p
_sz = C.c_size_t

class GatherOp(C.Structure):
    """include/dil256.h dil_gather_op"""
    _fields_ = [("kind", C.c_int), ("rank", C.c_int), ("peer", C.c_int), ("offset", _sz), ("bytes", _sz)]


# name -> (argtypes); every function returns int unless listed in _RESTYPE
SIGNATURES = {
    "dil_init": [C.c_int],
    "dil_shutdown": [],
    "dil_device_count": [C.POINTER(C.c_int)],
    "dil_num_cus": [],
    "dil_set_option": [C.c_char_p, C.c_int],
    "dil_get_option": [C.c_char_p, C.POINTER(C.c_int)],
    "dil_error_string": [C.c_int],
    "dil_host_twiddle_tables": [_u32p, _u32p, _u32p],
    "dil_host_zetas": [_i32p],
    "dil_ntt_dev": [_vp, _sz, _vp],
    "dil_invntt_dev": [_vp, _sz, _vp],
    "dil_ntt_traffic_dev": [_vp, _sz, C.c_int, _vp],
    "dil_ntt_host": [_i32p, _sz],
    "dil_invntt_host": [_i32p, _sz],
    "dil_pointwise_dev": [_vp, _vp, _vp, _sz, _vp],
    "dil_pointwise_acc_dev": [_vp, _vp, _vp, _vp, _sz, _vp],
    "dil_poly_add_dev": [_vp, _vp, _vp, _sz, _vp],
    "dil_poly_sub_dev": [_vp, _vp, _vp, _sz, _vp],
    "dil_pointwise_host": [_i32p, _i32p, _i32p, _sz],
    "dil_polymul_dev": [_vp, _vp, _vp, _sz, _vp],
    "dil_polymul_host": [_i32p, _i32p, _i32p, _sz],
    "dil_bram_fwdntt_dev": [_vp, _sz, C.c_int, _vp],
    "dil_bram_invntt_dev": [_vp, _sz, C.c_int, _vp],
    "dil_bram_mul_dev": [_vp, _vp, _sz, C.c_int, _vp],
    "dil_bram_fwdntt_host": [_i32p, _sz, C.c_int],
    "dil_bram_invntt_host": [_i32p, _sz, C.c_int],
    "dil_bram_mul_host": [_i32p, _i32p, _sz, C.c_int],
    "dil_matvec_dev": [_vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_verify_core_dev": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_multi_info": [C.POINTER(C.c_int), C.c_char_p, _sz, C.POINTER(C.c_int)],
    "dil_multi_gather_plan": [_sz, _sz, C.c_int, C.c_int, C.c_int, _vp, _sz, C.POINTER(_sz)],
    "dil_verify_core_host": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int],
    "dil_sign_phase1_dev": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_sign_phase2_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_sign_phase2_early_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_clock_probe_dev": [_vp, C.c_uint, _vp],
    "dil_mailbox_stats": [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)],
    "dil_sign_phase2_skey_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int, _vp],
    "dil_launch_info": [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz), C.POINTER(_sz)],
    "dil_host_plan": [_sz, C.c_int, C.POINTER(C.c_int), C.POINTER(_sz)],
    "dil_sign_round_plan": [C.c_int, _sz, _sz, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(_sz)],
    "dil_shake256_dev": [_vp, _sz, _vp, _sz, _sz, _vp],
    "dil_expand_a_dev": [_vp, _vp, C.c_int, _sz, _vp],
    "dil_expand_mask_dev": [_vp, _vp, _vp, C.c_int, _sz, _vp],
    "dil_sample_in_ball_dev": [_vp, _vp, C.c_int, _sz, _vp],
    "dil_challenge_dev": [_vp, _vp, _vp, _vp, C.c_int, _sz, _vp],
    "dil_pack_w1_dev": [_vp, _vp, C.c_int, _sz, _vp],
    "dil_unpack_dev": [_vp, _vp, _sz, _sz, C.c_int, C.c_int, _sz, _vp],
    "dil_pack_dev": [_vp, _sz, _sz, _vp, C.c_int, C.c_int, _sz, _vp],
    "dil_hint_unpack_dev": [_vp, _vp, _vp, _sz, _sz, C.c_int, _sz, _vp],
    "dil_hint_pack_dev": [_vp, _sz, _sz, _vp, C.c_int, _sz, _vp],
    "dil_expand_s_dev": [_vp, _vp, _vp, _sz, C.c_int, _sz, _vp],
    "dil_keygen_dev": [_vp, _vp, _vp, C.c_int, _sz, _vp],
    "dil_pk_bytes": [C.c_int],
    "dil_sk_bytes": [C.c_int],
    "dil_sig_bytes": [C.c_int],
    "dil_verify_sig_dev": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_verify_sig_expanded_dev": [_vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_verify_sig_expanded2_dev": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_expand_t1_dev": [_vp, _vp, C.c_int, _sz, _vp],
    "dil_verify_wire_core_dev": [_vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_sign_dev": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int, _vp],
    "dil_mu_dev": [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _sz, _vp],
    "dil_sign_msg_dev": [_vp, _vp, _vp, _vp, _sz, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int, _vp],
    "dil_verify_msg_dev": [_vp, _vp, _vp, _vp, _sz, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_keygen_host": [_vp, _vp, _vp, C.c_int, _sz],
    "dil_sign_host": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int],
    "dil_verify_sig_host": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int],
    "dil_shard_range": [_sz, C.c_int, C.c_int, C.POINTER(_sz), C.POINTER(_sz)],
    "dil_ntt_multi_host": [_i32p, _sz, C.c_int, C.c_int],
    "dil_keygen_multi_host": [_vp, _vp, _vp, C.c_int, _sz, C.c_int],
    "dil_sign_multi_host": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int, C.c_int],
    "dil_verify_sig_multi_host": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int],
    "dil_multi_init": [C.c_int],
    "dil_multi_shutdown": [],
    "dil_multi_last_error": [],
    "dil_gather_slabs_multi_dev": [_vp, _sz, _sz, C.c_int, C.c_int],
    "dil_ntt_multi_dev": [_vp, _sz, C.c_int, C.c_int, C.c_int],
    "dil_sign_multi_dev": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int, C.c_int, C.c_int],
    "dil_verify_sig_multi_dev": [_vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, C.c_int, C.c_int],
    "dil_sign_phases_multi_dev": [_vp] * 11 + [C.c_int, _sz, C.c_int, C.c_int, C.c_int],
    "dil_verify_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_sign_attempt_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _sz, C.c_int, _vp],
    "dil_event_create": [C.POINTER(_vp)],
    "dil_event_destroy": [_vp],
    "dil_event_record": [_vp, _vp],
    "dil_event_elapsed_ms": [C.POINTER(C.c_float), _vp, _vp],
    "dil_stream_sync": [_vp],
}
_RESTYPE = {"dil_error_string": C.c_char_p, "dil_multi_last_error": C.c_char_p, "dil_host_twiddle_tables": None, "dil_host_zetas": None,
            "dil_shard_range": None, "dil_pk_bytes": C.c_size_t, "dil_sk_bytes": C.c_size_t, "dil_sig_bytes": C.c_size_t}

_lib = None


class DilError(RuntimeError):
    pass


def load(build_if_missing: bool = True) -> C.CDLL:
    """dlopen the in-tree library (building it first if hipcc is available and it is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("DIL_LIB_PATH", _build.LIB)      # tuning builds; default = the in-tree library
    if build_if_missing and path == _build.LIB:
        try:
            _build.build()
        except Exception:
            if not os.path.exists(path):
                raise
    if not os.path.exists(path):
        raise DilError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().dil_error_string(code)
        raise DilError(f"{what or 'libdil256'} failed: hipError {code} ({msg.decode() if msg else '?'})")
