"""ctypes loader for libsda_hip.so - every symbol include/sda_hip.h declares, with its signature.

Fails loudly: if the shared library is missing, ``load()`` raises (no fallback of any kind)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDA_HIP_LIBRARY lets a developer A/B two builds of the same library; it is never a fallback
LIB_PATH = os.environ.get("SDA_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libsda_hip.so")

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
    # include/sda_hip_debug.h - test / measurement only
    "sda_debug_set_knob": (C.c_int, [C.c_char_p, C.c_long]),
    "sda_debug_reset_knobs": (None, []),
    "sda_debug_env_knobs_compiled_in": (C.c_int, []),
    "sda_debug_last_kernel": (C.c_char_p, []),
    "sda_debug_stream_create": (C.c_int, [c_voidpp]),
    "sda_debug_stream_destroy": (C.c_int, [C.c_void_p]),
    "sda_debug_stream_synchronize": (C.c_int, [C.c_void_p]),
    "sda_debug_mem_info": (C.c_int, [c_sizep, c_sizep]),
    "sda_debug_select_path": (C.c_int, [_SS, C.c_char_p, C.c_char_p, C.c_size_t]),
    "sda_version": (C.c_char_p, []),
    "sda_build_id": (C.c_char_p, []),
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

_lib = None


def load():
    """Load libsda_hip.so and attach the signatures.  Raises OSError if it has not been built
    (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: build it with __graft_entry__.build(); "
                          "there is no fallback implementation")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class SdaError(RuntimeError):
    """A nonzero status from the C ABI.  ``code`` is the enum sda_status value."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


def check(status: int) -> None:
    if status != OK:
        lib = load()
        msg = lib.sda_last_error().decode() or lib.sda_strerror(status).decode()
        raise SdaError(status, msg)
