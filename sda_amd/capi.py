"""ctypes loader for libsda_hip.so - every symbol include/sda_hip.h declares, with its signature.

Fails loudly: if the shared library is missing, ``load()`` raises (no fallback of any kind).

Two builds of the library exist (include/sda_hip_debug.h): the RELEASE library ``libsda_hip.so`` - what ``load()`` gives unless
told otherwise - and ``libsda_hip_test.so``, the same objects plus the test-only knob table and helpers (``HOOK_SIGNATURES``).
``use_test_hooks()`` makes the test library the active one for this process (parity tests of the non-default kernels, A/B
scripts); ``use_release()`` switches back.  ``load()`` returns a proxy that always forwards to the ACTIVE library, so a
reference taken before a switch stays valid."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDA_HIP_LIBRARY lets a developer A/B two builds of the same library; it is never a fallback
RELEASE_LIB_PATH = os.path.join(_HERE, "lib", "libsda_hip.so")
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libsda_hip_test.so")
LIB_PATH = os.environ.get("SDA_HIP_LIBRARY") or RELEASE_LIB_PATH

c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
c_sizep = C.POINTER(C.c_size_t)
c_i64pp = C.POINTER(c_i64p)
c_voidpp = C.POINTER(C.c_void_p)

# status codes (enum sda_status)
OK = 0
ERR_BATCH_INPUT_WRONG_LENGTH = -1
ERR_SHARING_FAILED = -2
ERR_INPUTS_MUST_HAVE_SAME_LENGTH = -3
ERR_NOT_ENOUGH_SHARES = -4
ERR_WRONG_DIMENSION = -5
ERR_MISMATCHING_DIMENSION = -6
ERR_ASSERTION = -7
ERR_INVALID_ARGUMENT = -8
ERR_UNSUPPORTED = -9
ERR_NO_DEVICE = -10
ERR_HIP = -11
ERR_ALLOC = -12
ERR_STATE = -13
ERR_ENTROPY = -14
ERR_COMM = -15
ERR_SODIUM_DECRYPTION = -16

SHARING_ADDITIVE, SHARING_PACKED_SHAMIR = 0, 1
MASKING_NONE, MASKING_FULL, MASKING_CHACHA = 0, 1, 2


class SharingScheme(C.Structure):
    _fields_ = [("kind", C.c_int32), ("share_count", C.c_uint64), ("modulus", C.c_int64),
                ("secret_count", C.c_uint64), ("privacy_threshold", C.c_uint64),
                ("omega_secrets", C.c_int64), ("omega_shares", C.c_int64)]


class JobLayout(C.Structure):
    _fields_ = [("payload_kind", C.c_uint32), ("rows", C.c_uint64), ("slot_bytes", C.c_uint64),
                ("lengths_offset", C.c_uint64), ("payload_offset", C.c_uint64), ("total_bytes", C.c_uint64)]


JOB_SEALED, JOB_VARINT, JOB_BASE64_TEXT = 0, 1, 2


class MaskingScheme(C.Structure):
    _fields_ = [("kind", C.c_int32), ("modulus", C.c_int64), ("dimension", C.c_uint64),
                ("seed_bitsize", C.c_uint64)]


_SS = C.POINTER(SharingScheme)
_MS = C.POINTER(MaskingScheme)
_H = C.c_void_p       # opaque handles
_HP = c_voidpp

# name -> (restype, argtypes); must list EVERY function of include/sda_hip.h (tests check this)
SIGNATURES = {
    "sda_scheme_input_size": (C.c_uint64, [_SS]),
    "sda_scheme_output_size": (C.c_uint64, [_SS]),
    "sda_scheme_privacy_threshold": (C.c_uint64, [_SS]),
    "sda_scheme_reconstruction_threshold": (C.c_uint64, [_SS]),
    "sda_masking_has_mask": (C.c_int, [_MS]),
    "sda_abi_version": (C.c_int, []),
    # include/sda_hip_debug.h - the two read-only queries both libraries export
    "sda_debug_hooks_compiled_in": (C.c_int, []),
    "sda_debug_last_kernel": (C.c_char_p, []),
    "sda_version": (C.c_char_p, []),
    "sda_build_id": (C.c_char_p, []),
    "sda_kernel_id": (C.c_char_p, []),
    "sda_share_generator_path_name": (C.c_char_p, [_H]),
    "sda_device_count": (C.c_int, []),
    "sda_set_device": (C.c_int, [C.c_int]),
    "sda_share_generator_set_value_mode": (C.c_int, [_H, C.c_int]),
    "sda_share_combiner_set_value_mode": (C.c_int, [_H, C.c_int]),
    "sda_secret_reconstructor_set_value_mode": (C.c_int, [_H, C.c_int]),
    "sda_secret_masker_set_value_mode": (C.c_int, [_H, C.c_int]),
    "sda_mask_combiner_set_value_mode": (C.c_int, [_H, C.c_int]),
    "sda_secret_unmasker_set_value_mode": (C.c_int, [_H, C.c_int]),
    "sda_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "sda_strerror": (C.c_char_p, [C.c_int]),
    "sda_last_error": (C.c_char_p, []),
    "sda_dev_malloc": (C.c_int, [c_voidpp, C.c_size_t]),
    "sda_dev_free": (C.c_int, [C.c_void_p]),
    "sda_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sda_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sda_dev_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "sda_dev_synchronize": (C.c_int, []),
    "sda_share_generator_new": (C.c_int, [_SS, _HP]),
    "sda_share_generator_free": (None, [_H]),
    "sda_share_generator_share_count": (C.c_uint64, [_H]),
    "sda_share_generator_batch_count": (C.c_uint64, [_H, C.c_size_t]),
    "sda_share_generator_rand_count": (C.c_uint64, [_H, C.c_size_t]),
    "sda_share_generator_set_drbg_key": (C.c_int, [_H, c_u8p]),
    "sda_share_generator_set_drbg_master_key": (C.c_int, [_H, c_u8p]),
    "sda_share_generator_set_drbg_rounds": (C.c_int, [_H, C.c_int]),
    "sda_drbg_draw_rule": (C.c_int, [C.c_int64]),
    "sda_share_generator_csprng_share_map": (C.c_int, [_H]),
    "sda_share_generator_set_csprng_share_map": (C.c_int, [_H, C.c_int]),
    "sda_share_generator_generate": (C.c_int, [_H, c_i64p, C.c_size_t, c_i64p, C.c_size_t, c_i64p, C.c_size_t]),
    "sda_share_generator_generate_batch_dev": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                                         C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p,
                                                         C.c_size_t, C.c_size_t, C.c_void_p]),
    "sda_share_generator_generate_combine_dev": (C.c_int, [_H, _H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64,
                                                           C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                           C.c_void_p]),
    "sda_share_combiner_new": (C.c_int, [_SS, _HP]),
    "sda_share_combiner_free": (None, [_H]),
    "sda_share_combiner_combine": (C.c_int, [_H, c_i64pp, c_sizep, C.c_size_t, c_i64p, C.c_size_t, c_sizep]),
    "sda_share_combiner_combine_dense": (C.c_int, [_H, c_i64p, C.c_size_t, C.c_size_t, C.c_size_t, c_i64p]),
    "sda_share_combiner_begin_dev": (C.c_int, [_H, C.c_size_t, C.c_size_t, C.c_void_p]),
    "sda_share_combiner_update_dev": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "sda_share_combiner_finish_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "sda_share_combiner_set_residency": (C.c_int, [_H, C.c_uint]),
    "sda_share_combiner_begin": (C.c_int, [_H, C.c_size_t]),
    "sda_share_combiner_update": (C.c_int, [_H, c_i64p, C.c_size_t, C.c_size_t]),
    "sda_share_combiner_finish": (C.c_int, [_H, c_i64p]),
    "sda_secret_reconstructor_new": (C.c_int, [_SS, C.c_size_t, _HP]),
    "sda_secret_reconstructor_free": (None, [_H]),
    "sda_secret_reconstructor_reconstruct": (C.c_int, [_H, c_sizep, c_i64pp, c_sizep, C.c_size_t, c_i64p,
                                                       C.c_size_t, c_sizep]),
    "sda_secret_reconstructor_reconstruct_dev": (C.c_int, [_H, c_sizep, C.c_size_t, C.c_void_p, C.c_size_t,
                                                           C.c_size_t, C.c_void_p, C.c_size_t, c_sizep, C.c_void_p]),
    "sda_secret_masker_new": (C.c_int, [_MS, _HP]),
    "sda_secret_masker_free": (None, [_H]),
    "sda_secret_masker_set_drbg_key": (C.c_int, [_H, c_u8p]),
    "sda_secret_masker_set_drbg_master_key": (C.c_int, [_H, c_u8p]),
    "sda_secret_masker_set_drbg_rounds": (C.c_int, [_H, C.c_int]),
    "sda_secret_masker_mask_len": (C.c_uint64, [_H, C.c_size_t]),
    "sda_secret_masker_mask": (C.c_int, [_H, c_i64p, C.c_size_t, c_i64p, C.c_size_t, c_i64p, C.c_size_t, c_sizep,
                                         c_i64p]),
    "sda_mask_combiner_new": (C.c_int, [_MS, _HP]),
    "sda_mask_combiner_free": (None, [_H]),
    "sda_mask_combiner_combine": (C.c_int, [_H, c_i64pp, c_sizep, C.c_size_t, c_i64p, C.c_size_t, c_sizep]),
    "sda_secret_unmasker_new": (C.c_int, [_MS, _HP]),
    "sda_secret_unmasker_free": (None, [_H]),
    "sda_secret_unmasker_unmask": (C.c_int, [_H, c_i64p, C.c_size_t, c_i64p, C.c_size_t, c_i64p]),
    "sda_positive": (C.c_int, [c_i64p, C.c_size_t, C.c_int64, c_i64p]),
    "sda_varint_codec_new": (C.c_int, [_HP]),
    "sda_varint_codec_free": (None, [_H]),
    "sda_varint_max_encoded_size": (C.c_size_t, [C.c_size_t]),
    "sda_varint_encode": (C.c_int, [_H, c_i64p, C.c_size_t, c_u8p, C.c_size_t, c_sizep]),
    "sda_varint_decode": (C.c_int, [_H, c_u8p, C.c_size_t, c_i64p, C.c_size_t, c_sizep]),
    "sda_varint_encode_dev": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "sda_varint_decode_dev": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_void_p]),
    "sda_share_combiner_update_varint_dev": (C.c_int, [_H, _H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                                       C.c_void_p]),
    "sda_share_combiner_update_varint": (C.c_int, [_H, _H, c_u8p, C.c_size_t]),
    "sda_secret_masker_mask_batch_dev": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64,
                                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sda_secret_unmasker_unmask_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sda_varint_slot_size": (C.c_size_t, [C.c_size_t]),
    "sda_varint_encode_rows_dev": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_void_p]),
    "sda_varint_decode_rows_dev": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_void_p]),
    "sda_share_combiner_update_varint_rows_dev": (C.c_int, [_H, _H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                            C.c_void_p, C.c_void_p]),
    "sda_job_slot_size": (C.c_size_t, [C.c_size_t]),
    "sda_job_container_size": (C.c_size_t, [C.c_size_t, C.c_size_t]),
    "sda_job_container_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_size_t, C.POINTER(JobLayout)]),
    "sda_job_container_set_row": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]),
    "sda_job_container_parse": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(JobLayout)]),
    "sda_job_container_get_row": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), c_sizep]),
    "sda_base64_encoded_size": (C.c_size_t, [C.c_size_t]),
    "sda_base64_decoded_max": (C.c_size_t, [C.c_size_t]),
    "sda_base64_decode_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sda_base64_encode_rows_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_void_p]),
    "sda_sealedbox_new": (C.c_int, [_HP]),
    "sda_sealedbox_free": (None, [_H]),
    "sda_sealedbox_open_rows_dev": (C.c_int, [_H, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sda_sealedbox_seal_rows_dev": (C.c_int, [_H, C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p, C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sda_sealedbox_public_key": (C.c_int, [_H, C.c_char_p, C.c_char_p]),
    "sda_sealedbox_seal": (C.c_int, [_H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "sda_sealedbox_open": (C.c_int, [_H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, c_sizep]),
    "sda_comm_unique_id": (C.c_int, [c_u8p]),
    "sda_comm_init": (C.c_int, [c_u8p, C.c_int, C.c_int, _HP]),
    "sda_comm_free": (None, [_H]),
    "sda_comm_rank": (C.c_int, [_H]),
    "sda_comm_world": (C.c_int, [_H]),
    "sda_comm_device": (C.c_int, [_H]),
    "sda_comm_rccl_version": (C.c_int, []),
    "sda_modular_allreduce_dev": (C.c_int, [_H, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sda_modsum_parts_dev": (C.c_int, [C.c_int64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                       C.c_void_p]),
    "sda_fill_synthetic_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64,
                                         C.c_int64, C.c_void_p]),
    "sda_event_create": (C.c_int, [c_voidpp]),
    "sda_event_destroy": (C.c_int, [C.c_void_p]),
    "sda_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sda_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
}

# include/sda_hip_debug.h - libsda_hip_test.so only (-DSDA_TEST_HOOKS): the knob table and the tests' helpers
HOOK_SIGNATURES = {
    "sda_debug_set_knob": (C.c_int, [C.c_char_p, C.c_long]),
    "sda_debug_reset_knobs": (None, []),
    "sda_debug_env_knobs_compiled_in": (C.c_int, []),
    "sda_debug_stream_create": (C.c_int, [c_voidpp]),
    "sda_debug_stream_destroy": (C.c_int, [C.c_void_p]),
    "sda_debug_stream_synchronize": (C.c_int, [C.c_void_p]),
    "sda_debug_mem_info": (C.c_int, [c_sizep, c_sizep]),
    "sda_debug_select_path": (C.c_int, [_SS, C.c_char_p, C.c_char_p, C.c_size_t]),
}

_loaded = {}          # path -> CDLL with signatures attached
_active_path = None   # None = LIB_PATH (the release library unless SDA_HIP_LIBRARY says otherwise)


def _load_path(path):
    lib = _loaded.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with __graft_entry__.build(); "
                          "there is no fallback implementation")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.sda_debug_hooks_compiled_in():
            for name, (res, args) in HOOK_SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
        _loaded[path] = lib
    return lib


class _ActiveLibrary:
    """forwards every attribute to the library that is active NOW (see use_test_hooks)"""

    def __getattr__(self, name):
        lib = _load_path(_active_path or LIB_PATH)
        if name in HOOK_SIGNATURES and not lib.sda_debug_hooks_compiled_in():
            raise AttributeError(f"{name} exists only in libsda_hip_test.so: call sda_amd.capi.use_test_hooks() first "
                                 f"(the release library {active_path()} has no test hooks)")
        return getattr(lib, name)


_proxy = _ActiveLibrary()


def load():
    """The active library (libsda_hip.so unless use_test_hooks() was called), signatures attached.  Raises OSError if it has
    not been built (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    _load_path(_active_path or LIB_PATH)
    return _proxy


def active_path() -> str:
    return _active_path or LIB_PATH


def has_test_hooks() -> bool:
    return bool(_load_path(active_path()).sda_debug_hooks_compiled_in())


def use_test_hooks():
    """make libsda_hip_test.so (same objects as the release library + the knob table) the active library of this process"""
    global _active_path
    if _active_path == TEST_LIB_PATH:
        return _proxy
    if LIB_PATH != RELEASE_LIB_PATH and _load_path(LIB_PATH).sda_debug_hooks_compiled_in():
        return _proxy                                   # an SDA_HIP_LIBRARY A/B build that carries the hooks itself
    _load_path(TEST_LIB_PATH)
    _active_path = TEST_LIB_PATH
    return _proxy


def hooks_library():
    """libsda_hip_test.so itself (signatures attached) WITHOUT making it the active library: for its stateless helpers - streams,
    memory figures, the selection table - beside handles that live in the release library (both sit on the one HIP runtime)"""
    return _load_path(TEST_LIB_PATH)


def use_release():
    """back to LIB_PATH; the knobs of the test library (if it was loaded) are reset first"""
    global _active_path
    if _active_path is not None:
        lib = _loaded.get(_active_path)
        if lib is not None and lib.sda_debug_hooks_compiled_in():
            lib.sda_debug_reset_knobs()
    _active_path = None
    return _proxy


class SdaError(RuntimeError):
    """A nonzero status from the C ABI.  ``code`` is the enum sda_status value."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


LAST_ERROR_SEEN = ""        # text of the last failure check() raised (sda_last_error() is per thread: a watchdog thread reads this)


def check(status: int) -> None:
    if status != OK:
        global LAST_ERROR_SEEN
        lib = load()
        msg = lib.sda_last_error().decode() or lib.sda_strerror(status).decode()
        LAST_ERROR_SEEN = msg
        raise SdaError(status, msg)
