"""Host-side mirror of the reference's ``client::crypto`` sharing / masking interface, on top of
the C ABI (include/sda_hip.h).  Same names, argument meaning and error behaviour as the reference:

    CryptoModule.new_share_generator(scheme)         sharing/mod.rs:10-12,35-55
    ShareGenerator.generate(secrets)                 sharing/mod.rs:14-17
    CryptoModule.new_share_combiner(scheme)          sharing/mod.rs:19-21,57-73
    ShareCombiner.combine(shares)                    sharing/mod.rs:23-25
    CryptoModule.new_secret_reconstructor(scheme, d) sharing/mod.rs:27-29,75-96
    SecretReconstructor.reconstruct(indexed_shares)  sharing/mod.rs:31-33
    CryptoModule.new_secret_masker / new_mask_combiner / new_secret_unmasker   masking/mod.rs:9-31

``SdaClientResult`` errors surface as :class:`sda_amd.capi.SdaError` carrying the reference's message;
the masking traits' ``assert!`` panics surface as :class:`AssertionError`.  The only widening of the
interface is the optional ``rand`` argument (the reference draws from OsRng; see sda_hip.h).

Every method runs on the GPU through libsda_hip.so; nothing here computes on share data in Python.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import capi
from .capi import SdaError, check

Secret = Mask = MaskedSecret = Share = np.int64     # client/src/crypto/mod.rs:33-36


# ---- scheme enums (protocol/src/crypto.rs:43-155) ------------------------------------------------
@dataclass(frozen=True)
class Additive:
    share_count: int
    modulus: int

    def input_size(self): return 1                                   # crypto.rs:120-126
    def output_size(self): return self.share_count                   # crypto.rs:129-135
    def privacy_threshold_(self): return self.share_count - 1        # crypto.rs:138-144
    def reconstruction_threshold(self): return self.share_count      # crypto.rs:147-153

    def _c(self):
        return capi.SharingScheme(capi.SHARING_ADDITIVE, self.share_count, self.modulus, 0, 0, 0, 0)


@dataclass(frozen=True)
class PackedShamir:
    secret_count: int
    share_count: int
    privacy_threshold: int
    prime_modulus: int
    omega_secrets: int
    omega_shares: int

    def input_size(self): return self.secret_count
    def output_size(self): return self.share_count
    def privacy_threshold_(self): return self.privacy_threshold
    def reconstruction_threshold(self): return self.privacy_threshold + self.secret_count

    def _c(self):
        return capi.SharingScheme(capi.SHARING_PACKED_SHAMIR, self.share_count, self.prime_modulus,
                                  self.secret_count, self.privacy_threshold, self.omega_secrets,
                                  self.omega_shares)


LinearSecretSharingScheme = Union[Additive, PackedShamir]


@dataclass(frozen=True)
class NoMask:                       # LinearMaskingScheme::None
    def has_mask(self): return False
    def _c(self): return capi.MaskingScheme(capi.MASKING_NONE, 0, 0, 0)


@dataclass(frozen=True)
class Full:
    modulus: int
    def has_mask(self): return True
    def _c(self): return capi.MaskingScheme(capi.MASKING_FULL, self.modulus, 0, 0)


@dataclass(frozen=True)
class ChaCha:
    modulus: int
    dimension: int
    seed_bitsize: int
    def has_mask(self): return True
    def _c(self): return capi.MaskingScheme(capi.MASKING_CHACHA, self.modulus, self.dimension, self.seed_bitsize)


LinearMaskingScheme = Union[NoMask, Full, ChaCha]


# ---- helpers ------------------------------------------------------------------------------------------
def _vec(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64).reshape(-1)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(capi.c_i64p)


def _rows(vectors: Sequence) -> Tuple[List[np.ndarray], C.Array, C.Array]:
    if isinstance(vectors, np.ndarray) and vectors.ndim == 2 and vectors.dtype == np.int64 and vectors.shape[0] > 0 \
            and vectors.strides[1] == 8:
        # a matrix: row pointers by arithmetic instead of one ctypes object per row (thousands of seeds / vectors)
        n = vectors.shape[0]
        addr = (vectors.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(vectors.strides[0])).astype(np.uint64)
        lens_np = np.full(n, vectors.shape[1], dtype=np.uint64)
        ptrs = C.cast(addr.ctypes.data_as(C.POINTER(C.c_uint64)), C.POINTER(capi.c_i64p))
        lens = C.cast(lens_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t))
        rows = _MatrixRows(vectors, addr, lens_np)
        return rows, ptrs, lens
    arrs = [_vec(v) for v in vectors]
    ptrs = (capi.c_i64p * max(len(arrs), 1))(*[_ptr(a) for a in arrs])
    lens = (C.c_size_t * max(len(arrs), 1))(*[a.size for a in arrs])
    return arrs, ptrs, lens


class _MatrixRows(list):
    """keeps the matrix and the pointer/length arrays alive; len() = rows, [0].size = columns"""

    def __init__(self, matrix, addr, lens):
        super().__init__(matrix)
        self._keep = (matrix, addr, lens)


def _check_mask(status: int) -> None:
    """masking traits are infallible in the reference and panic on assert! - mirror as AssertionError"""
    if status == capi.ERR_ASSERTION:
        raise AssertionError(capi.load().sda_last_error().decode())
    check(status)


CANONICAL, RUST_SIGNED = 0, 1          # enum sda_value_mode


class _Handle:
    _free = None
    _value_mode = None                  # name of the C setter, for the six trait handles

    def __init__(self):
        self._h = C.c_void_p()
        self._lib = capi.load()

    def set_value_mode(self, mode) -> "_Handle":
        """CANONICAL (default): residues in [0, q).  RUST_SIGNED: the reference's own representatives, bit for bit
        (Rust's truncated `%`: values in (-q, q), history-dependent signs) - additive sharing, the combiner, the masks."""
        mode = {"canonical": CANONICAL, "rust_signed": RUST_SIGNED}.get(mode, mode)
        check(getattr(self._lib, self._value_mode)(self._h, int(mode)))
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            getattr(self._lib, self._free)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- sharing ----------------------------------------------------------------------------------------------
class ShareGenerator(_Handle):
    """sharing/mod.rs:14-17; impl batched.rs:18-53 over additive.rs / packed_shamir.rs."""
    _free = "sda_share_generator_free"
    _value_mode = "sda_share_generator_set_value_mode"

    def __init__(self, scheme: LinearSecretSharingScheme):
        super().__init__()
        self.scheme = scheme
        cs = scheme._c()
        check(self._lib.sda_share_generator_new(C.byref(cs), C.byref(self._h)))

    def set_drbg_key(self, key: bytes):
        """TEST / BENCH ONLY: deterministic mode (the caller's stream ids select the CSPRNG streams)"""
        if len(key) != 32:
            raise ValueError("the CSPRNG key is exactly 32 bytes")
        check(self._lib.sda_share_generator_set_drbg_key(self._h, (C.c_uint8 * 32)(*key)))

    def set_drbg_master_key(self, key: bytes):
        """TEST ONLY: fixed master key, per-call key derivation stays on"""
        if len(key) != 32:
            raise ValueError("the CSPRNG key is exactly 32 bytes")
        check(self._lib.sda_share_generator_set_drbg_master_key(self._h, (C.c_uint8 * 32)(*key)))

    def set_drbg_rounds(self, rounds: int):
        check(self._lib.sda_share_generator_set_drbg_rounds(self._h, rounds))

    SHARE_MAP_TSS_NODES, SHARE_MAP_SYSTEMATIC = 0, 1

    def csprng_share_map(self) -> int:
        """the parametrisation calls WITHOUT injected randomness use (include/sda_hip.h, "CSPRNG share map"): 1 = the t draws
        of a batch are its shares 0..t-1, 0 = tss's (draws are the values at omega_secrets^(k+1..k+t))"""
        return int(self._lib.sda_share_generator_csprng_share_map(self._h))

    def set_csprng_share_map(self, which: int):
        check(self._lib.sda_share_generator_set_csprng_share_map(self._h, which))

    def path_name(self) -> str:
        """the kernel family the library selected for this scheme ("l31", "fft+ngemm", ... - sda_amd/csrc/path_select.hpp)"""
        return self._lib.sda_share_generator_path_name(self._h).decode()

    def batch_count(self, length: int) -> int:
        return int(self._lib.sda_share_generator_batch_count(self._h, length))

    def rand_count(self, length: int) -> int:
        return int(self._lib.sda_share_generator_rand_count(self._h, length))

    def generate(self, secrets, rand=None) -> np.ndarray:
        """-> int64 array [share_count][batches], row j = the shares for clerk j (batched.rs:46-48)."""
        s = _vec(secrets)
        n = int(self._lib.sda_share_generator_share_count(self._h))
        B = self.batch_count(s.size)
        out = np.empty((n, B), dtype=np.int64)
        if rand is None:
            rp, rl = None, 0
        else:
            r = _vec(rand)
            rp, rl = _ptr(r), r.size
        check(self._lib.sda_share_generator_generate(self._h, _ptr(s), s.size, rp, rl, _ptr(out), out.size))
        return out

    def generate_batch_dev(self, d_secrets: int, participants: int, length: int, secrets_stride: int,
                           d_out: int, out_stride_participant: int, out_stride_clerk: int,
                           first_participant: int = 0, d_rand: int = 0, rand_stride: int = 0, stream: int = 0):
        check(self._lib.sda_share_generator_generate_batch_dev(
            self._h, d_secrets, participants, length, secrets_stride, d_rand or None, rand_stride,
            first_participant, d_out, out_stride_participant, out_stride_clerk, stream or None))


    def generate_combine_dev(self, combiner: "ShareCombiner", d_secrets: int, participants: int, length: int,
                             secrets_stride: int, d_out: int, out_stride_participant: int, out_stride_clerk: int,
                             d_prev: int = 0, prev_participants: int = 0, first_participant: int = 0, stream: int = 0):
        """software-pipelined step: generate this tile while the previous tile (d_prev) is summed into `combiner`"""
        check(self._lib.sda_share_generator_generate_combine_dev(
            self._h, combiner._h, d_secrets or None, participants, length, secrets_stride, first_participant,
            d_out or None, out_stride_participant, out_stride_clerk, d_prev or None, prev_participants, stream or None))


class ShareCombiner(_Handle):
    """sharing/mod.rs:23-25; impl combiner.rs:15-29."""
    _free = "sda_share_combiner_free"
    _value_mode = "sda_share_combiner_set_value_mode"

    def __init__(self, scheme: LinearSecretSharingScheme):
        super().__init__()
        self.scheme = scheme
        cs = scheme._c()
        check(self._lib.sda_share_combiner_new(C.byref(cs), C.byref(self._h)))

    def combine(self, shares: Sequence) -> np.ndarray:
        arrs, ptrs, lens = _rows(shares)
        cap = arrs[0].size if arrs else 0
        out = np.empty(max(cap, 1), dtype=np.int64)
        n_out = C.c_size_t()
        check(self._lib.sda_share_combiner_combine(self._h, ptrs, lens, len(arrs), _ptr(out), cap, C.byref(n_out)))
        return out[:n_out.value].copy()

    # streaming / accumulating form (fixes the FIXME at client/src/clerk.rs:71-72)
    def begin(self, dimension: int):
        check(self._lib.sda_share_combiner_begin(self._h, dimension))
        self._dimension = dimension

    def update(self, tile) -> None:
        t = np.ascontiguousarray(tile, dtype=np.int64)
        assert t.ndim == 2
        check(self._lib.sda_share_combiner_update(self._h, _ptr(t), t.shape[0], t.shape[1]))

    def finish(self, dimension: Optional[int] = None) -> np.ndarray:
        """the dimension is the one given to begin(); an argument is accepted only if it agrees"""
        have = getattr(self, "_dimension", None)
        if have is None:
            raise SdaError(capi.ERR_STATE, "finish before begin")
        if dimension is not None and dimension != have:
            raise ValueError(f"finish({dimension}) on a combiner begun with dimension {have}")
        out = np.empty(max(have, 1), dtype=np.int64)
        check(self._lib.sda_share_combiner_finish(self._h, _ptr(out)))
        return out[:have]

    def update_encoded(self, codec: "VarintCodec", raw: bytes) -> None:
        """one participant's wire-format share vector (the opened sealed-box payload, sodium.rs:83-89)"""
        b = np.frombuffer(raw, dtype=np.uint8)
        check(self._lib.sda_share_combiner_update_varint(self._h, codec._h, b.ctypes.data_as(capi.c_u8p), b.size))

    def update_encoded_dev(self, codec: "VarintCodec", d_bytes: int, n_bytes: int, d_row_offsets: int, rows: int,
                           d_status: int, stream: int = 0) -> None:
        check(self._lib.sda_share_combiner_update_varint_dev(self._h, codec._h, d_bytes, n_bytes, d_row_offsets or None,
                                                             rows, d_status, stream or None))

    def update_encoded_rows_dev(self, codec: "VarintCodec", d_bytes: int, slot_bytes: int, d_row_bytes: int, rows: int,
                                d_status: int, stream: int = 0) -> None:
        check(self._lib.sda_share_combiner_update_varint_rows_dev(self._h, codec._h, d_bytes, slot_bytes, d_row_bytes,
                                                                  rows, d_status, stream or None))

    def set_residency(self, max_workgroups_per_cu: int) -> None:
        check(self._lib.sda_share_combiner_set_residency(self._h, max_workgroups_per_cu))

    def begin_dev(self, jobs: int, dimension: int, stream: int = 0):
        check(self._lib.sda_share_combiner_begin_dev(self._h, jobs, dimension, stream or None))
        self._dimension = dimension if jobs == 1 else None

    def update_dev(self, d_shares: int, job_stride: int, n_rows: int, row_stride: int, stream: int = 0):
        check(self._lib.sda_share_combiner_update_dev(self._h, d_shares, job_stride, n_rows, row_stride, stream or None))

    def finish_dev(self, d_out: int, stream: int = 0):
        check(self._lib.sda_share_combiner_finish_dev(self._h, d_out, stream or None))


class SecretReconstructor(_Handle):
    """sharing/mod.rs:31-33; impl additive.rs:55-73 | batched.rs:68-97 + packed_shamir.rs:73-77."""
    _free = "sda_secret_reconstructor_free"
    _value_mode = "sda_secret_reconstructor_set_value_mode"

    def __init__(self, scheme: LinearSecretSharingScheme, dimension: int):
        super().__init__()
        self.scheme = scheme
        self.dimension = dimension
        cs = scheme._c()
        check(self._lib.sda_secret_reconstructor_new(C.byref(cs), dimension, C.byref(self._h)))

    def reconstruct(self, indexed_shares: Sequence[Tuple[int, Sequence]]) -> np.ndarray:
        idx = [int(i) for i, _ in indexed_shares]
        arrs, ptrs, lens = _rows([v for _, v in indexed_shares])
        cidx = (C.c_size_t * max(len(idx), 1))(*idx)
        cap = max(self.dimension, arrs[0].size if arrs else 0)
        out = np.empty(max(cap, 1), dtype=np.int64)
        n_out = C.c_size_t()
        check(self._lib.sda_secret_reconstructor_reconstruct(self._h, cidx, ptrs, lens, len(arrs), _ptr(out), cap,
                                                             C.byref(n_out)))
        return out[:n_out.value].copy()

    def reconstruct_dev(self, indices: Sequence[int], d_shares: int, row_len: int, row_stride: int, d_out: int,
                        out_cap: int, stream: int = 0) -> int:
        cidx = (C.c_size_t * max(len(indices), 1))(*[int(i) for i in indices])
        n_out = C.c_size_t()
        check(self._lib.sda_secret_reconstructor_reconstruct_dev(self._h, cidx, len(indices), d_shares, row_len,
                                                                 row_stride, d_out, out_cap, C.byref(n_out),
                                                                 stream or None))
        return n_out.value


# ---- masking -------------------------------------------------------------------------------------------------
class SecretMasker(_Handle):
    """masking/mod.rs:13-15; impl none.rs:13-19, full.rs:21-35, chacha.rs:24-54."""
    _free = "sda_secret_masker_free"
    _value_mode = "sda_secret_masker_set_value_mode"

    def __init__(self, scheme: LinearMaskingScheme):
        super().__init__()
        self.scheme = scheme
        cs = scheme._c()
        check(self._lib.sda_secret_masker_new(C.byref(cs), C.byref(self._h)))

    def set_drbg_key(self, key: bytes):
        """TEST / BENCH ONLY: deterministic mode"""
        if len(key) != 32:
            raise ValueError("the CSPRNG key is exactly 32 bytes")
        check(self._lib.sda_secret_masker_set_drbg_key(self._h, (C.c_uint8 * 32)(*key)))

    def set_drbg_master_key(self, key: bytes):
        if len(key) != 32:
            raise ValueError("the CSPRNG key is exactly 32 bytes")
        check(self._lib.sda_secret_masker_set_drbg_master_key(self._h, (C.c_uint8 * 32)(*key)))

    def set_drbg_rounds(self, rounds: int):
        check(self._lib.sda_secret_masker_set_drbg_rounds(self._h, rounds))

    def mask(self, secrets, rand=None) -> Tuple[np.ndarray, np.ndarray]:
        s = _vec(secrets)
        cap = int(self._lib.sda_secret_masker_mask_len(self._h, s.size))
        mask = np.empty(max(cap, 1), dtype=np.int64)
        masked = np.empty(max(s.size, 1), dtype=np.int64)
        n_mask = C.c_size_t()
        if rand is None:
            rp, rl = None, 0
        else:
            r = _vec(rand)
            rp, rl = _ptr(r), r.size
        _check_mask(self._lib.sda_secret_masker_mask(self._h, _ptr(s), s.size, rp, rl, _ptr(mask), cap,
                                                     C.byref(n_mask), _ptr(masked)))
        return mask[:n_mask.value].copy(), masked[:s.size].copy()

    def mask_batch_dev(self, d_secrets: int, participants: int, length: int, secrets_stride: int, d_masks: int,
                       mask_stride: int, d_masked: int, masked_stride: int, first_participant: int = 0,
                       stream: int = 0) -> None:
        """participate.rs:52-54 for a device-resident tile (Full / None schemes)"""
        check(self._lib.sda_secret_masker_mask_batch_dev(self._h, d_secrets, participants, length, secrets_stride,
                                                         first_participant, d_masks or None, mask_stride, d_masked,
                                                         masked_stride, stream or None))


class MaskCombiner(_Handle):
    """masking/mod.rs:21-23; impl none.rs:21-26, full.rs:37-52, chacha.rs:56-77."""
    _free = "sda_mask_combiner_free"
    _value_mode = "sda_mask_combiner_set_value_mode"

    def __init__(self, scheme: LinearMaskingScheme):
        super().__init__()
        self.scheme = scheme
        cs = scheme._c()
        check(self._lib.sda_mask_combiner_new(C.byref(cs), C.byref(self._h)))

    def combine(self, masks: Sequence) -> np.ndarray:
        arrs, ptrs, lens = _rows(masks)
        if isinstance(self.scheme, ChaCha):
            cap = self.scheme.dimension
        else:
            cap = arrs[0].size if arrs else 0
        out = np.empty(max(cap, 1), dtype=np.int64)
        n_out = C.c_size_t()
        _check_mask(self._lib.sda_mask_combiner_combine(self._h, ptrs, lens, len(arrs), _ptr(out), cap, C.byref(n_out)))
        return out[:n_out.value].copy()


class SecretUnmasker(_Handle):
    """masking/mod.rs:29-31; impl none.rs:28-33, full.rs:54-67, chacha.rs:79-93."""
    _free = "sda_secret_unmasker_free"
    _value_mode = "sda_secret_unmasker_set_value_mode"

    def __init__(self, scheme: LinearMaskingScheme):
        super().__init__()
        self.scheme = scheme
        cs = scheme._c()
        check(self._lib.sda_secret_unmasker_new(C.byref(cs), C.byref(self._h)))

    def unmask(self, values: Tuple[Sequence, Sequence]) -> np.ndarray:
        mask, masked = _vec(values[0]), _vec(values[1])
        out = np.empty(max(masked.size, 1), dtype=np.int64)
        _check_mask(self._lib.sda_secret_unmasker_unmask(self._h, _ptr(mask), mask.size, _ptr(masked), masked.size,
                                                         _ptr(out)))
        return out[:masked.size].copy()

    def unmask_dev(self, d_mask: int, d_masked: int, length: int, d_out: int, stream: int = 0) -> None:
        check(self._lib.sda_secret_unmasker_unmask_dev(self._h, d_mask or None, d_masked, length, d_out, stream or None))


# ---- factory (client/src/crypto/mod.rs:58-66) ---------------------------------------------------------------
class CryptoModule:
    """The reference's factory object; the keystore plays no part on this path."""

    def new_share_generator(self, scheme): return ShareGenerator(scheme)
    def new_share_combiner(self, scheme): return ShareCombiner(scheme)
    def new_secret_reconstructor(self, scheme, dimension): return SecretReconstructor(scheme, dimension)
    def new_secret_masker(self, scheme): return SecretMasker(scheme)
    def new_mask_combiner(self, scheme): return MaskCombiner(scheme)
    def new_secret_unmasker(self, scheme): return SecretUnmasker(scheme)


@dataclass
class RecipientOutput:
    """client/src/receive.rs:7-21."""
    modulus: int
    values: np.ndarray

    def positive(self) -> "RecipientOutput":
        v = _vec(self.values)
        out = np.empty(max(v.size, 1), dtype=np.int64)
        check(capi.load().sda_positive(_ptr(v), v.size, self.modulus, _ptr(out)))
        return RecipientOutput(self.modulus, out[:v.size].copy())


@dataclass
class Aggregation:
    """The compute-relevant fields of protocol/src/resources.rs:44-67."""
    vector_dimension: int
    modulus: int
    masking_scheme: LinearMaskingScheme
    committee_sharing_scheme: LinearSecretSharingScheme


def full_aggregation(aggregation: Aggregation, inputs: Sequence[Sequence[int]], mask_rand=None, share_rand=None,
                     clerk_subset: Optional[Sequence[int]] = None, value_mode=CANONICAL) -> dict:
    """The three callers' data flow - participate.rs:52-76, the snapshot transposition
    (server/src/stores.rs:86-101), clerk.rs:85-86, receive.rs:101-156 - with the HIP core in place of
    the reference's crypto module.  Returns every intermediate.  value_mode=RUST_SIGNED asks every handle for the
    reference's own signed representatives (packed Shamir's generator / reconstructor stay canonical: tss's values)."""
    crypto = CryptoModule()
    a = aggregation
    n = a.committee_sharing_scheme.output_size()
    masks, maskeds, shares = [], [], []
    signed_sharing = value_mode == RUST_SIGNED and isinstance(a.committee_sharing_scheme, Additive)
    masker = crypto.new_secret_masker(a.masking_scheme).set_value_mode(value_mode)
    generator = crypto.new_share_generator(a.committee_sharing_scheme)
    if signed_sharing:
        generator.set_value_mode(RUST_SIGNED)
    for p, secrets in enumerate(inputs):
        if len(secrets) != a.vector_dimension:
            raise ValueError("The input length does not match the aggregation.")      # participate.rs:44-46
        m, ms = masker.mask(secrets, None if mask_rand is None else mask_rand[p])      # participate.rs:53-54
        masks.append(m); maskeds.append(ms)
        shares.append(generator.generate(ms, None if share_rand is None else share_rand[p]))   # :75-76
    combiner = crypto.new_share_combiner(a.committee_sharing_scheme).set_value_mode(value_mode)
    clerk_sums = [combiner.combine([shares[p][c] for p in range(len(inputs))]) for c in range(n)]   # clerk.rs:85-86
    mask = (crypto.new_mask_combiner(a.masking_scheme).set_value_mode(value_mode).combine(masks)
            if a.masking_scheme.has_mask() else np.empty(0, dtype=np.int64))             # receive.rs:102-118
    subset = list(range(n)) if clerk_subset is None else list(clerk_subset)
    rec = crypto.new_secret_reconstructor(a.committee_sharing_scheme, a.vector_dimension)
    if signed_sharing:
        rec.set_value_mode(RUST_SIGNED)
    masked_output = rec.reconstruct([(c, clerk_sums[c]) for c in subset])                # receive.rs:140-144
    output = crypto.new_secret_unmasker(a.masking_scheme).set_value_mode(value_mode).unmask((mask, masked_output))  # receive.rs:149-152
    return {"masks": masks, "masked": maskeds, "shares": shares, "clerk_sums": clerk_sums,
            "combined_mask": mask, "masked_output": masked_output, "output": output,
            "positive": RecipientOutput(a.modulus, output).positive().values}


# ---- share-vector wire codec -------------------------------------------------------------
class VarintCodec(_Handle):
    """zig-zag LEB128 codec of share vectors: what `ShareEncryptor::encrypt` does before sealing
    (encryption/sodium.rs:36-41) and `ShareDecryptor::decrypt` after opening (:83-89)."""
    _free = "sda_varint_codec_free"

    def __init__(self):
        super().__init__()
        check(self._lib.sda_varint_codec_new(C.byref(self._h)))

    def encode(self, shares) -> bytes:
        v = _vec(shares)
        out = np.empty(max(v.size * 10, 1), dtype=np.uint8)
        n = C.c_size_t()
        check(self._lib.sda_varint_encode(self._h, _ptr(v), v.size, out.ctypes.data_as(capi.c_u8p), v.size * 10, C.byref(n)))
        return out[:n.value].tobytes()

    def decode(self, raw: bytes) -> np.ndarray:
        b = np.frombuffer(raw, dtype=np.uint8)
        out = np.empty(max(b.size, 1), dtype=np.int64)
        n = C.c_size_t()
        check(self._lib.sda_varint_decode(self._h, b.ctypes.data_as(capi.c_u8p), b.size, _ptr(out), b.size, C.byref(n)))
        return out[:n.value].copy()

    def encode_dev(self, d_values: int, rows: int, length: int, row_stride: int, d_out: int, out_cap: int,
                   d_row_offsets: int = 0, stream: int = 0) -> int:
        total = C.c_uint64()
        check(self._lib.sda_varint_encode_dev(self._h, d_values, rows, length, row_stride, d_out, out_cap,
                                              d_row_offsets or None, C.byref(total), stream or None))
        return total.value

    def slot_size(self, length: int) -> int:
        return self._lib.sda_varint_slot_size(length)

    def encode_rows_dev(self, d_values: int, rows: int, length: int, row_stride: int, d_out: int, slot_bytes: int,
                        d_row_bytes: int, stream: int = 0) -> None:
        """single pass: row r -> d_out + r*slot_bytes, its length -> d_row_bytes[r]"""
        check(self._lib.sda_varint_encode_rows_dev(self._h, d_values, rows, length, row_stride, d_out, slot_bytes,
                                                   d_row_bytes, stream or None))

    def decode_rows_dev(self, d_bytes: int, slot_bytes: int, d_row_bytes: int, rows: int, length: int, d_values: int,
                        row_stride: int, d_status: int, stream: int = 0) -> None:
        check(self._lib.sda_varint_decode_rows_dev(self._h, d_bytes, slot_bytes, d_row_bytes, rows, length, d_values,
                                                   row_stride, d_status, stream or None))

    def decode_dev(self, d_bytes: int, n_bytes: int, d_row_offsets: int, rows: int, length: int, d_values: int,
                   row_stride: int, d_status: int, stream: int = 0) -> None:
        check(self._lib.sda_varint_decode_dev(self._h, d_bytes, n_bytes, d_row_offsets or None, rows, length, d_values,
                                              row_stride, d_status, stream or None))


# ---- clerking-job container + base64 `Binary` payloads -------------------------------------
class JobContainer:
    """SDAJOBv1: a ClerkingJob's `encryptions: Vec<Encryption>` (protocol/src/resources.rs:128-139) as one binary blob
    with fixed slots, laid out so the device calls consume it in place (include/sda_hip.h).  Host-side byte layout."""

    def __init__(self, blob: bytearray, layout: "capi.JobLayout"):
        self.blob, self.layout = blob, layout

    @classmethod
    def build(cls, kind: int, payloads: Sequence[bytes], slot_bytes: Optional[int] = None) -> "JobContainer":
        lib = capi.load()
        rows = len(payloads)
        slot = lib.sda_job_slot_size(max((len(p) for p in payloads), default=0)) if slot_bytes is None else slot_bytes
        size = lib.sda_job_container_size(rows, slot)
        if rows and size == 0:
            raise ValueError("slot_bytes must be a multiple of 16")
        blob = bytearray(max(size, 64))
        buf = (C.c_uint8 * len(blob)).from_buffer(blob)
        lay = capi.JobLayout()
        check(lib.sda_job_container_init(buf, len(blob), kind, rows, slot, C.byref(lay)))
        for r, p in enumerate(payloads):
            check(lib.sda_job_container_set_row(buf, len(blob), r, bytes(p), len(p)))
        del buf
        return cls(blob, lay)

    @classmethod
    def parse(cls, blob: bytes) -> "JobContainer":
        b = bytearray(blob)
        lay = capi.JobLayout()
        check(capi.load().sda_job_container_parse((C.c_uint8 * len(b)).from_buffer(b), len(b), C.byref(lay)))
        return cls(b, lay)

    def rows(self) -> List[bytes]:
        L = self.layout
        lens = np.frombuffer(self.blob, dtype="<u8", count=L.rows, offset=L.lengths_offset)
        return [bytes(self.blob[L.payload_offset + r * L.slot_bytes:L.payload_offset + r * L.slot_bytes + int(lens[r])])
                for r in range(L.rows)]

    def __bytes__(self):
        return bytes(self.blob)


def base64_decode_rows_dev(d_text: int, text_slot: int, d_text_bytes: int, rows: int, max_chars: int, d_out: int,
                           out_slot: int, d_out_bytes: int, d_status: int, d_row_status: int = 0, d_text_offsets: int = 0,
                           stream: int = 0) -> None:
    """Binary::from_base64 (helpers.rs:182-184) for `rows` payloads resident in HBM"""
    check(capi.load().sda_base64_decode_rows_dev(d_text, d_text_offsets or None, text_slot, d_text_bytes, rows, max_chars,
                                                 d_out, out_slot, d_out_bytes, d_status, d_row_status or None,
                                                 stream or None))


def base64_encode_rows_dev(d_in: int, in_slot: int, d_in_bytes: int, rows: int, max_bytes: int, d_text: int,
                           text_slot: int, d_text_bytes: int, stream: int = 0) -> None:
    """Binary::to_base64 (helpers.rs:178-180) for `rows` payloads resident in HBM"""
    check(capi.load().sda_base64_encode_rows_dev(d_in, in_slot, d_in_bytes, rows, max_bytes, d_text, text_slot,
                                                 d_text_bytes, stream or None))


# ---- sealed boxes: encryption/sodium.rs:33-46 (encrypt), :72-92 (decrypt) ----------------------
class SealedBox(_Handle):
    """libsodium crypto_box_seal / crypto_box_seal_open on the GPU, one payload (host forms) or a whole job (rows)."""
    _free = "sda_sealedbox_free"
    SEALBYTES = 48

    def __init__(self):
        super().__init__()
        check(self._lib.sda_sealedbox_new(C.byref(self._h)))

    def public_key(self, sk: bytes) -> bytes:
        """X25519(sk, 9): the public half of a key pair (crypto_scalarmult_base) - for tests and tools"""
        assert len(sk) == 32
        out = C.create_string_buffer(32)
        check(self._lib.sda_sealedbox_public_key(self._h, sk, out))
        return out.raw

    def seal(self, message: bytes, pk: bytes, esk: Optional[bytes] = None) -> bytes:
        """sealedbox::seal (sodium.rs:43); esk injects the ephemeral secret key (tests only)"""
        assert len(pk) == 32 and (esk is None or len(esk) == 32)
        out = np.empty(len(message) + 48, dtype=np.uint8)
        check(self._lib.sda_sealedbox_seal(self._h, pk, esk, bytes(message), len(message), out.ctypes.data_as(C.c_void_p), out.size))
        return out.tobytes()

    def open(self, box: bytes, pk: bytes, sk: bytes) -> bytes:
        """sealedbox::open (sodium.rs:78); raises SdaError("Sodium decryption failure") like the reference's Err (:80)"""
        assert len(pk) == 32 and len(sk) == 32
        out = np.empty(max(len(box), 1), dtype=np.uint8)
        n = C.c_size_t()
        check(self._lib.sda_sealedbox_open(self._h, pk, sk, bytes(box), len(box), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out[:n.value].tobytes()

    def open_rows_dev(self, pk: bytes, sk: bytes, d_boxes: int, slot_bytes: int, d_row_bytes: int, rows: int, max_box_bytes: int,
                      d_out: int, out_slot: int, d_out_bytes: int, d_status: int, d_ok: int = 0, stream: int = 0) -> None:
        check(self._lib.sda_sealedbox_open_rows_dev(self._h, pk, sk, d_boxes, slot_bytes, d_row_bytes, rows, max_box_bytes, d_out,
                                                    out_slot, d_out_bytes, d_ok or None, d_status, stream or None))

    def seal_rows_dev(self, pks: Sequence[bytes], rows_per_key: int, d_msgs: int, msg_slot: int, d_msg_bytes: int, rows: int,
                      max_msg_bytes: int, d_boxes: int, slot_bytes: int, d_row_bytes: int, esk: Optional[bytes] = None,
                      stream: int = 0) -> None:
        allpk = b"".join(pks)
        assert len(allpk) == 32 * len(pks) and (esk is None or len(esk) == 32 * rows)
        check(self._lib.sda_sealedbox_seal_rows_dev(self._h, allpk, len(pks), rows_per_key, esk, d_msgs, msg_slot, d_msg_bytes, rows,
                                                    max_msg_bytes, d_boxes, slot_bytes, d_row_bytes, stream or None))


class ShareEncryptor:
    """encryption/sodium.rs:33-46: zig-zag varint encode every share, then seal the bytes for the clerk's key."""

    def __init__(self, pk: bytes):
        self.pk, self._box, self._codec = pk, SealedBox(), VarintCodec()

    def encrypt(self, shares, esk: Optional[bytes] = None) -> bytes:
        return self._box.seal(self._codec.encode(shares), self.pk, esk)


class ShareDecryptor:
    """encryption/sodium.rs:72-92: open the sealed box, decode varints until the reader is empty."""

    def __init__(self, pk: bytes, sk: bytes):
        self.pk, self.sk, self._box, self._codec = pk, sk, SealedBox(), VarintCodec()

    def decrypt(self, encryption: bytes) -> np.ndarray:
        return self._codec.decode(self._box.open(encryption, self.pk, self.sk))
