"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/sda_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    """every function the two headers under include/ declare: the boundary (sda_hip.h) and the test-only knobs (sda_hip_debug.h)"""
    out = set()
    for name in ("sda_hip.h", "sda_hip_debug.h"):
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        out |= set(re.findall(r"\b(sda_[a-z0-9_]+)\s*\(", hdr))
    return out


def test_library_exports_every_declared_symbol(built):
    from sda_amd import capi
    lib = capi.load()
    declared = _declared()
    assert len(declared) >= 50
    everything = set(capi.SIGNATURES) | set(capi.HOOK_SIGNATURES)
    assert declared == everything, (declared ^ everything)
    # the release library exports the boundary (+ the two read-only queries), the test library the hooks as well
    for path, want in ((capi.RELEASE_LIB_PATH, set(capi.SIGNATURES)), (capi.TEST_LIB_PATH, everything)):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        exported = set(re.findall(r" T (sda_[a-z0-9_]+)", out))
        assert want <= exported, (path, want - exported)
    assert lib.sda_abi_version() == 6 and b"gfx950" in lib.sda_version()


def test_release_library_reads_no_environment_variable(built):
    """kernel selection of a crypto library is not steered by the process environment: the shipped build has no getenv
    import at all (the A/B variant, -DSDA_AB_KNOBS, is built by tools/build_ab_variant.sh only), knobs move only through
    the test-only sda_debug_set_knob OF THE TEST LIBRARY, unknown names are refused, and the Python package and the library
    agree on the version."""
    import sda_amd
    from sda_amd import capi
    lib = capi.load()
    for path in (capi.RELEASE_LIB_PATH, capi.TEST_LIB_PATH):
        und = subprocess.check_output(["nm", "-D", "--undefined-only", path], text=True)
        assert not re.search(r"\bU (secure_)?getenv\b", und), path + " imports getenv"
    assert sda_amd.__version__.encode() in lib.sda_version()
    hooks = capi.use_test_hooks()
    try:
        assert hooks.sda_debug_env_knobs_compiled_in() == 0
        assert hooks.sda_debug_set_knob(b"SDA_FORCE_MFMA", 1) == capi.OK
        hooks.sda_debug_reset_knobs()
        assert hooks.sda_debug_set_knob(b"SDA_NO_SUCH_KNOB", 1) == capi.ERR_INVALID_ARGUMENT
        assert sda_amd.__version__.encode() in hooks.sda_version() and b"+test-hooks" in hooks.sda_version()
    finally:
        capi.use_release()
    build_py = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "SDA_AB_KNOBS" not in build_py.split("def smoke")[0].replace("# SDA_AB_KNOBS", "")
    # nothing outside the C ABI reads the environment to steer the product either (bench.py and tools/ are measurement tools)
    for f in ("distributed.py", "crypto.py", "capi.py", "device.py"):
        text = open(os.path.join(ROOT, "sda_amd", f)).read()
        names = set(re.findall(r"environ(?:\.get)?\(?\[?[\"']([A-Z_]+)", text))
        assert names <= {"SDA_HIP_LIBRARY"}, (f, names)


def test_release_library_has_no_test_hooks(built):
    """`nm -D libsda_hip.so | grep sda_debug_set_knob` is empty.  The release library exports none of the
    entry points of include/sda_hip_debug.h that change or create anything (the knob table, the stream helpers, the selection
    table); only the two read-only queries.  libsda_hip_test.so - same objects, sda_capi.cpp rebuilt with -DSDA_TEST_HOOKS -
    exports all of them; both carry the build id of the tree; the loader refuses a hook on the release library by name."""
    import ctypes
    from sda_amd import capi
    import __graft_entry__ as g

    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return set(re.findall(r" T (sda_[a-z0-9_]+)", out))
    rel, tst = exported(capi.RELEASE_LIB_PATH), exported(capi.TEST_LIB_PATH)
    assert {n for n in rel if n.startswith("sda_debug_")} == {"sda_debug_last_kernel", "sda_debug_hooks_compiled_in"}
    assert set(capi.HOOK_SIGNATURES) <= tst and not (set(capi.HOOK_SIGNATURES) & rel)
    assert tst - rel == set(capi.HOOK_SIGNATURES), "the two libraries differ in the hooks only"
    for path, want in ((capi.RELEASE_LIB_PATH, 0), (capi.TEST_LIB_PATH, 1)):
        so = ctypes.CDLL(path)
        so.sda_build_id.restype = ctypes.c_char_p
        so.sda_kernel_id.restype = ctypes.c_char_p
        assert so.sda_debug_hooks_compiled_in() == want and so.sda_build_id().decode() == g.source_digest()
        assert so.sda_kernel_id().decode() == g.kernel_digest() != g.source_digest()          # the device code's own digest
    assert capi.active_path() == capi.RELEASE_LIB_PATH and not capi.has_test_hooks()
    with pytest.raises(AttributeError, match="libsda_hip_test.so"):
        capi.load().sda_debug_set_knob
    # every function declared in the debug header is one of the two kinds
    hdr = open(os.path.join(ROOT, "include", "sda_hip_debug.h")).read()
    declared = set(re.findall(r"\b(sda_debug_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.HOOK_SIGNATURES) | {"sda_debug_last_kernel", "sda_debug_hooks_compiled_in"}, declared


def test_host_side_under_address_and_undefined_behaviour_sanitizers(built):
    """make -C tests/cpp check-sanitize: the four host translation units of the library rebuilt with
    -fsanitize=address,undefined, linked with the in-tree kernel objects, and driven through every host-only entry point
    (scheme sizes, constructors on hostile descriptors, positive(), the SDAJOBv1 parser on 20,000 mutated blobs and every
    truncation, NULL / short-buffer misuse).  Round 4's first run found a signed overflow in sda_positive."""
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "check-sanitize"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "host side clean under ASan + UBSan" in out.stdout


def test_library_is_gfx950_only_and_links_no_oracle(built):
    from sda_amd import capi
    out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-d", capi.LIB_PATH], text=True)
    assert "libamdhip64" in out and "oracle" not in out
    raw = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in raw and b"gfx942" not in raw and b"sm_" not in raw


def test_product_does_not_import_oracle():
    """The product path must not route through the oracle or any CPU fallback: nothing under sda_amd/
    (Python, C++ or HIP) may mention oracle/, and the loader has no alternative implementation."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sda_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read().lower()
                assert "oracle" not in text, (f, "mentions the oracle")
    assert "oracle" not in open(os.path.join(ROOT, "include", "sda_hip.h")).read().lower()


def test_scheme_derived_sizes(built):
    """protocol/src/crypto.rs:120-153."""
    from sda_amd import capi
    lib = capi.load()
    add = capi.SharingScheme(capi.SHARING_ADDITIVE, 3, 433, 0, 0, 0, 0)
    pss = capi.SharingScheme(capi.SHARING_PACKED_SHAMIR, 8, 433, 3, 4, 354, 150)
    assert (lib.sda_scheme_input_size(C.byref(add)), lib.sda_scheme_output_size(C.byref(add)),
            lib.sda_scheme_privacy_threshold(C.byref(add)), lib.sda_scheme_reconstruction_threshold(C.byref(add))) == (1, 3, 2, 3)
    assert (lib.sda_scheme_input_size(C.byref(pss)), lib.sda_scheme_output_size(C.byref(pss)),
            lib.sda_scheme_privacy_threshold(C.byref(pss)), lib.sda_scheme_reconstruction_threshold(C.byref(pss))) == (3, 8, 4, 7)
    for kind, want in ((capi.MASKING_NONE, 0), (capi.MASKING_FULL, 1), (capi.MASKING_CHACHA, 1)):
        ms = capi.MaskingScheme(kind, 433, 4, 128)
        assert lib.sda_masking_has_mask(C.byref(ms)) == want


def test_positive_is_receive_rs(built):
    from sda_amd import crypto
    out = crypto.RecipientOutput(433, np.array([-1, 0, 432, -432], dtype=np.int64)).positive()
    assert out.values.tolist() == [432, 0, 432, 1]                           # receive.rs:15


def test_error_strings_mirror_the_reference(built):
    from sda_amd import capi
    lib = capi.load()
    want = {capi.ERR_BATCH_INPUT_WRONG_LENGTH: "Batch input wrong length",                      # additive.rs:33
            capi.ERR_SHARING_FAILED: "Sharing failed for packed secret sharing scheme",        # packed_shamir.rs:41
            capi.ERR_INPUTS_MUST_HAVE_SAME_LENGTH: "Inputs must have same length",             # packed_shamir.rs:74
            capi.ERR_NOT_ENOUGH_SHARES: "Not enough shares to reconstruct",                    # packed_shamir.rs:75
            capi.ERR_WRONG_DIMENSION: "Wrong dimension",                                       # combiner.rs:21
            capi.ERR_MISMATCHING_DIMENSION: "Mismatching dimension"}                           # additive.rs:64
    for code, msg in want.items():
        assert lib.sda_strerror(code).decode() == msg


def test_fails_loudly_without_a_gpu(built):
    """No silent CPU path: on a box without a GPU every handle constructor reports NO_DEVICE."""
    from sda_amd import capi, crypto
    lib = capi.load()
    if lib.sda_device_count() > 0:
        pytest.skip("a GPU is visible here")
    for make in (lambda: crypto.ShareGenerator(crypto.Additive(3, 433)),
                 lambda: crypto.ShareCombiner(crypto.Additive(3, 433)),
                 lambda: crypto.SecretReconstructor(crypto.Additive(3, 433), 4),
                 lambda: crypto.SecretMasker(crypto.Full(433)),
                 lambda: crypto.MaskCombiner(crypto.ChaCha(433, 4, 128)),
                 lambda: crypto.SecretUnmasker(crypto.Full(433))):
        with pytest.raises(capi.SdaError) as e:
            make()
        assert e.value.code == capi.ERR_NO_DEVICE and "no CPU fallback" in e.value.message
    p = C.c_void_p()
    assert lib.sda_dev_malloc(C.byref(p), 64) == capi.ERR_NO_DEVICE
    # parameter validation happens before the device is touched
    with pytest.raises(capi.SdaError) as e:
        crypto.ShareGenerator(crypto.Additive(3, 1))
    assert e.value.code == capi.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.SdaError) as e:
        crypto.ShareGenerator(crypto.PackedShamir(3, 8, 4, 435, 354, 150))
    assert e.value.code == capi.ERR_INVALID_ARGUMENT and "prime" in e.value.message


def test_missing_library_raises(built, monkeypatch):
    from sda_amd import capi
    monkeypatch.setattr(capi, "_loaded", {})
    monkeypatch.setattr(capi, "LIB_PATH", os.path.join(ROOT, "sda_amd", "lib", "nope.so"))
    with pytest.raises(OSError, match="no fallback"):
        capi.load()
    monkeypatch.setattr(capi, "TEST_LIB_PATH", os.path.join(ROOT, "sda_amd", "lib", "nope_test.so"))
    with pytest.raises(OSError, match="no fallback"):
        capi.use_test_hooks()


def test_abi_misuse_returns_codes_not_crashes(built):
    """NULL handles / NULL out-pointers are answered with SDA_ERR_INVALID_ARGUMENT (nothing throws or aborts
    across the ABI; argument checks come before any device access, so this runs without a GPU)."""
    from sda_amd import capi
    lib = capi.load()
    n = C.c_size_t()
    buf = (C.c_int64 * 4)()
    u8 = (C.c_uint8 * 16)()
    bad = capi.ERR_INVALID_ARGUMENT
    assert lib.sda_share_generator_new(None, None) == bad
    assert lib.sda_share_generator_generate(None, buf, 4, None, 0, buf, 4) == bad
    assert lib.sda_share_generator_generate_batch_dev(None, None, 1, 1, 1, None, 0, 0, None, 1, 1, None) == bad
    assert lib.sda_share_generator_set_drbg_key(None, None) == bad
    assert lib.sda_share_combiner_new(None, None) == bad
    assert lib.sda_share_combiner_combine(None, None, None, 0, buf, 4, C.byref(n)) == bad
    assert lib.sda_share_combiner_begin(None, 4) == bad and lib.sda_share_combiner_update(None, buf, 1, 4) == bad
    assert lib.sda_share_combiner_finish(None, buf) == bad and lib.sda_share_combiner_set_residency(None, 2) == bad
    assert lib.sda_share_combiner_update_varint(None, None, u8, 1) == bad
    assert lib.sda_secret_reconstructor_new(None, 4, None) == bad
    assert lib.sda_secret_reconstructor_reconstruct(None, None, None, None, 0, buf, 4, C.byref(n)) == bad
    assert lib.sda_secret_masker_new(None, None) == bad
    assert lib.sda_secret_masker_mask(None, buf, 4, None, 0, buf, 4, C.byref(n), buf) == bad
    assert lib.sda_mask_combiner_combine(None, None, None, 0, buf, 4, C.byref(n)) == bad
    assert lib.sda_secret_unmasker_unmask(None, buf, 4, buf, 4, buf) == bad
    assert lib.sda_varint_codec_new(None) == bad
    assert lib.sda_varint_encode(None, buf, 4, u8, 16, C.byref(n)) == bad
    assert lib.sda_varint_decode(None, u8, 4, buf, 4, C.byref(n)) == bad
    assert lib.sda_positive(None, 4, 433, None) == bad and lib.sda_positive(None, 0, 433, None) == capi.OK
    assert lib.sda_event_create(None) == bad and lib.sda_dev_malloc(None, 8) == bad
    # the entry points added for the pipelined / wire-format / device-resident forms
    assert lib.sda_share_generator_generate_combine_dev(None, None, None, 0, 0, 0, 0, None, 0, 0, None, 0, None) == bad
    assert lib.sda_varint_encode_dev(None, None, 1, 1, 1, None, 0, None, None, None) == bad
    assert lib.sda_varint_decode_dev(None, None, 0, None, 1, 1, None, 1, None, None) == bad
    assert lib.sda_varint_encode_rows_dev(None, None, 1, 1, 1, None, 16, None, None) == bad
    assert lib.sda_varint_decode_rows_dev(None, None, 16, None, 1, 1, None, 1, None, None) == bad
    assert lib.sda_share_combiner_update_varint_dev(None, None, None, 0, None, 1, None, None) == bad
    assert lib.sda_share_combiner_update_varint_rows_dev(None, None, None, 16, None, 1, None, None) == bad
    assert lib.sda_secret_masker_mask_batch_dev(None, None, 1, 1, 1, 0, None, 1, None, 1, None) == bad
    assert lib.sda_secret_unmasker_unmask_dev(None, None, None, 1, None, None) == bad
    # round 3: value mode setters, sealed-box key helper, communicator queries, device identity
    for f in ("sda_share_generator_set_value_mode", "sda_share_combiner_set_value_mode", "sda_secret_reconstructor_set_value_mode",
              "sda_secret_masker_set_value_mode", "sda_mask_combiner_set_value_mode", "sda_secret_unmasker_set_value_mode"):
        assert getattr(lib, f)(None, 1) == bad
    assert lib.sda_sealedbox_public_key(None, None, None) == bad
    assert lib.sda_comm_device(None) == -1 and lib.sda_comm_world(None) == 0 and lib.sda_comm_rank(None) == -1
    assert lib.sda_device_pci_bus_id(0, None, 0) == bad
    assert lib.sda_varint_slot_size(3) == 32 and lib.sda_varint_slot_size(0) == 0 and lib.sda_varint_max_encoded_size(3) == 30
    assert lib.sda_last_error() != b""
    # free functions accept NULL
    for f in ("sda_share_generator_free", "sda_share_combiner_free", "sda_secret_reconstructor_free",
              "sda_secret_masker_free", "sda_mask_combiner_free", "sda_secret_unmasker_free", "sda_varint_codec_free"):
        getattr(lib, f)(None)
    # unknown scheme kinds and out-of-range parameters are refused before any device work
    s = capi.SharingScheme(7, 3, 433, 0, 0, 0, 0)
    h = C.c_void_p()
    assert lib.sda_share_generator_new(C.byref(s), C.byref(h)) == bad and not h.value
    m = capi.MaskingScheme(9, 433, 4, 128)
    assert lib.sda_secret_masker_new(C.byref(m), C.byref(h)) == bad and not h.value
    s = capi.SharingScheme(capi.SHARING_ADDITIVE, 3, 1 << 62, 0, 0, 0, 0)
    assert lib.sda_share_combiner_new(C.byref(s), C.byref(h)) == capi.ERR_UNSUPPORTED


def test_header_is_plain_c(built):
    """the boundary is a C ABI: include/sda_hip.h must compile as C99 (what bindgen / cgo / a C caller reads) and as
    C++11, warning-free under -pedantic."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "sda_hip.h")
    for cmd in (["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", hdr],
                ["g++", "-x", "c++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


# ---- the Rust shim of INTEGRATION.md is the deliverable a maintainer pastes: keep it equal to the header --------------------
_C2RUST = {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "int64_t": "i64", "uint32_t": "u32",
           "int32_t": "i32", "uint8_t": "u8", "char": "c_char", "void": "c_void", "float": "f32"}


def _c_type_to_rust(ctype: str) -> str:
    """'const int64_t* const*' -> '*const *const i64' (what a Rust extern "C" declaration must say for that C parameter)"""
    import re
    t = ctype.strip()
    arr = re.search(r"\[[^\]]*\]\s*$", t)                       # `uint8_t id[128]` decays to a pointer
    if arr:
        t = t[:arr.start()].strip() + "*"
    toks = re.findall(r"\*|const|[A-Za-z_][A-Za-z_0-9]*", t)
    base, stars, const_base = None, [], False
    pending_const = False
    for tok in toks:
        if tok == "const":
            if base is None:
                const_base = True
            else:
                pending_const = True                              # qualifies the pointer to its left
        elif tok == "*":
            stars.append(False)
            pending_const = False
        elif tok == "struct" or tok == "unsigned" and base is not None:
            continue
        elif base is None:
            base = tok
        # a trailing identifier is the parameter name: ignored
        if tok == "const" and stars:
            stars[-1] = True
    name = base[:-2] if base.endswith("_t") and base.startswith("sda_") and base not in ("sda_sharing_scheme_t", "sda_masking_scheme_t", "sda_job_layout_t") else base
    rust = _C2RUST.get(name, name)
    # innermost pointee constness is the base's; each further level takes the const that FOLLOWS the previous star
    quals = [const_base] + stars[:-1] if stars else []
    for q in quals:
        rust = ("*const " if q else "*mut ") + rust
    return rust


def _split_args(arglist: str):
    args, depth, cur = [], 0, ""
    for ch in arglist:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur)
    return [a.strip() for a in args]


def _header_functions():
    import re
    text = open(os.path.join(ROOT, "include", "sda_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)
    funcs = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9\s\*]*?)\b(sda_[a-z0-9_]+)\s*\(([^;{}]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        params = [] if args in ("", "void") else [_c_type_to_rust(a) for a in _split_args(args)]
        funcs[name] = (None if ret == "void" else _c_type_to_rust(ret), params)
    return funcs


def _rust_shim():
    import re
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```rust\n(.*?)```", md, flags=re.S)
    fns, structs = {}, {}
    for b in blocks:
        b = re.sub(r"//[^\n]*", "", b)
        b = re.sub(r"/\*.*?\*/", "", b, flags=re.S)
        for ext in re.findall(r'extern "C" \{(.*?)\n\}', b, flags=re.S):
            for m in re.finditer(r"pub fn (sda_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
                params = [re.sub(r"\s+", " ", a.split(":", 1)[1]).strip() for a in _split_args(m.group(2)) if ":" in a]
                fns[m.group(1)] = (m.group(3).strip() if m.group(3) else None, params)
        for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\}", b, flags=re.S):
            structs[m.group(1)] = [(f.split(":")[0].replace("pub", "").strip(), f.split(":")[1].strip())
                                   for f in _split_args(m.group(2)) if ":" in f]
    return fns, structs


def test_c_type_translation_examples():
    assert _c_type_to_rust("const int64_t* secrets") == "*const i64"
    assert _c_type_to_rust("const int64_t* const* rows") == "*const *const i64"
    assert _c_type_to_rust("sda_share_generator_t** out") == "*mut *mut sda_share_generator"
    assert _c_type_to_rust("const uint8_t id[SDA_COMM_ID_BYTES]") == "*const u8"
    assert _c_type_to_rust("uint8_t id[128]") == "*mut u8"
    assert _c_type_to_rust("const sda_sharing_scheme_t* s") == "*const sda_sharing_scheme_t"
    assert _c_type_to_rust("const uint8_t** payload") == "*mut *const u8"
    assert _c_type_to_rust("void* stream") == "*mut c_void" and _c_type_to_rust("size_t") == "usize"


def test_integration_md_rust_shim_matches_the_header(built, tmp_path):
    """Every `pub fn` of INTEGRATION.md's extern "C" blocks (the binding for the traits of sharing/mod.rs:10-33 and
    masking/mod.rs:9-31, plus the codec, sealed-box, job and communicator calls) exists in include/sda_hip.h with the same
    arity, the same integer widths and the same pointer constness; the #[repr(C)] structs have the C structs' size and
    field offsets (checked with a compiled C program)."""
    import subprocess
    header = _header_functions()
    fns, structs = _rust_shim()
    assert len(fns) >= 45 and {"sda_share_generator_generate", "sda_share_combiner_combine", "sda_secret_reconstructor_reconstruct",
                               "sda_secret_masker_mask", "sda_mask_combiner_combine", "sda_secret_unmasker_unmask"} <= set(fns)
    for name, (ret, params) in sorted(fns.items()):
        assert name in header, f"{name} is declared in INTEGRATION.md but not in include/sda_hip.h"
        cret, cparams = header[name]
        assert ret == cret, f"{name}: return type {ret} vs C {cret}"
        assert len(params) == len(cparams), f"{name}: {len(params)} parameters in the Rust shim, {len(cparams)} in the header"
        for i, (r, c) in enumerate(zip(params, cparams)):
            assert r == c, f"{name}, parameter {i}: Rust `{r}` vs header `{c}`"
    # struct layouts: what #[repr(C)] gives for the Rust field lists == what the C compiler gives for the header's structs
    size = {"i32": 4, "u32": 4, "i64": 8, "u64": 8, "usize": 8, "u8": 1}
    assert {"sda_sharing_scheme_t", "sda_masking_scheme_t", "sda_job_layout_t"} <= set(structs)
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "sda_hip.h"', "int main(void) {"]
    for sname, fields in structs.items():
        prog.append(f'printf("{sname} size %zu\\n", sizeof({sname}));')
        for f, _ in fields:
            prog.append(f'printf("{sname} {f} %zu\\n", offsetof({sname}, {f}));')
    prog.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for sname, fields in structs.items():
        off, align = 0, 1
        for f, t in fields:
            a = size[t]
            off = (off + a - 1) // a * a
            assert got[(sname, f)] == off, f"{sname}.{f}: repr(C) offset {off}, C offset {got[(sname, f)]}"
            off += size[t]
            align = max(align, a)
        assert got[(sname, "size")] == (off + align - 1) // align * align, sname


def test_drbg_draw_rule_query(built):
    """ABI 6: which draw rule of sda-drbg-v1 serves a modulus - the paired rule (one candidate word per two draws) up to 0x7F7F7F,
    one word per draw above; out-of-range moduli come back as status codes.  No device needed."""
    from sda_amd import capi
    lib = capi.load()
    assert lib.sda_drbg_draw_rule(433) == 2 and lib.sda_drbg_draw_rule(746497) == 2 and lib.sda_drbg_draw_rule(0x7F7F7F) == 2
    assert lib.sda_drbg_draw_rule(0x7F7F7F + 1) == 1 and lib.sda_drbg_draw_rule(2147482801) == 1
    assert lib.sda_drbg_draw_rule(4611686006577364993) == 1
    assert lib.sda_drbg_draw_rule(1) == capi.ERR_INVALID_ARGUMENT and lib.sda_drbg_draw_rule(-5) == capi.ERR_INVALID_ARGUMENT
    assert lib.sda_drbg_draw_rule(1 << 62) == capi.ERR_UNSUPPORTED
    assert lib.sda_comm_rccl_version() == 0                      # nothing has bound RCCL in this process


def test_loader_switches_between_the_two_libraries(built):
    """sda_amd.capi: load() is a proxy for the ACTIVE library - the release one unless use_test_hooks() was called; a reference
    taken before a switch follows it; use_release() resets the knobs it leaves behind."""
    from sda_amd import capi
    lib = capi.load()
    assert capi.active_path() == capi.RELEASE_LIB_PATH and lib.sda_debug_hooks_compiled_in() == 0
    assert b"+test-hooks" not in lib.sda_version()
    capi.use_test_hooks()
    try:
        assert capi.active_path() == capi.TEST_LIB_PATH and capi.has_test_hooks() and lib.sda_debug_hooks_compiled_in() == 1
        assert b"+test-hooks" in lib.sda_version()
        assert lib.sda_debug_set_knob(b"SDA_FORCE_GENERIC", 1) == capi.OK
        assert capi.hooks_library().sda_debug_hooks_compiled_in() == 1
    finally:
        capi.use_release()
    assert capi.active_path() == capi.RELEASE_LIB_PATH and lib.sda_debug_hooks_compiled_in() == 0
    # the knob did not survive: the selection table of the test library (stateless call) still says what the defaults say
    import ctypes as C
    s = capi.SharingScheme(capi.SHARING_PACKED_SHAMIR, 8, 4611686006577364993, 3, 1, 631229665360524489, 3451275676410824977)
    buf = C.create_string_buffer(256)
    assert capi.hooks_library().sda_debug_select_path(C.byref(s), None, buf, 256) == capi.OK and b"wide=l31" in buf.value
