"""ctypes binding of libfmx_gfx950.so (C-ABI declared in include/fmx.h).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) so that it travels with the
repo snapshot.  There is deliberately NO fallback: if the shared object is missing or a symbol is
absent, importing `lib()` raises -- a Forge install that selected this backend must fail loudly rather
than silently run PyTorch eager ops.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))


def knob(name, default=None):
    """Development A/B knob: the value of environment variable `name` ONLY in a process that also carries FMX_ALLOW_KNOBS=1 (the same rule as
    fmx_knob() inside the library) -- a stray FMX_* variable must not change what a production process runs.  Knobs that took effect are recorded
    in ACTIVE_KNOBS (bench.py prints them, together with the library's own list, into its line)."""
    v = os.environ.get(name)
    if v is None:
        return default
    if os.environ.get("FMX_ALLOW_KNOBS") != "1":
        IGNORED_KNOBS[name] = v
        if name == "FMX_LIB":   # a tool that meant to measure another binary must not silently measure the production one
            import warnings
            warnings.warn(f"FMX_LIB={v} is IGNORED (set FMX_ALLOW_KNOBS=1 to load it); using the in-tree libfmx_gfx950.so")
        return default
    ACTIVE_KNOBS[name] = v
    return v


ACTIVE_KNOBS, IGNORED_KNOBS = {}, {}
# FMX_LIB: developer hook for tools/ (timing builds of the same library, tools/build_patched*.sh); a knob like the others
LIB_PATH = knob("FMX_LIB") or os.path.join(_HERE, "libfmx_gfx950.so")
CSRC = os.path.join(_HERE, "csrc")

_lock = threading.Lock()
_lib = None


class FmxError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("c0", C.c_int32), ("c1", C.c_int32),
        ("a0_stride", C.c_int32), ("a1_stride", C.c_int32),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("oh", C.c_int32), ("ow", C.c_int32),
        ("kh", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("up_h", C.c_int32), ("up_w", C.c_int32),
        ("wgt", C.c_void_p), ("ldw", C.c_int32), ("nout", C.c_int32),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("ld_rowvec", C.c_int32),
        ("residual", C.c_void_p), ("ld_res", C.c_int32),
        ("alpha", C.c_float), ("act", C.c_int32),
        ("out", C.c_void_p), ("ld_out", C.c_int32), ("out_f32", C.c_int32),
        ("zero_page", C.c_void_p), ("gate", C.c_void_p), ("ld_gate", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("ln_partial", C.c_void_p), ("ln_parts", C.c_int32), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
        ("ln_col_ab", C.c_void_p), ("ln_row_cb", C.c_void_p), ("ln_ab_out", C.c_void_p),
        ("xa_k", C.c_void_p), ("xa_vt", C.c_void_p), ("xa_k_rs", C.c_int32), ("xa_k_bs", C.c_int32), ("xa_vt_ds", C.c_int32), ("xa_vt_bs", C.c_int32),
        ("xa_nk", C.c_int32), ("xa_rows", C.c_int32), ("xa_scale", C.c_float), ("xa_k_bytes", C.c_int64), ("xa_vt_bytes", C.c_int64),
    ]


class ConvGnArgs(C.Structure):
    """fmx_conv_gn_args (include/fmx.h): GroupNorm + SiLU + 3x3 convolution in one kernel"""
    _fields_ = [
        ("x", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32),
        ("x_partial", C.c_void_p), ("x_nchunks", C.c_int32), ("groups", C.c_int32), ("eps", C.c_float),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("scale_shift", C.c_void_p),
        ("wgt", C.c_void_p), ("cout", C.c_int32), ("bias", C.c_void_p), ("residual", C.c_void_p), ("ld_res", C.c_int64),
        ("out", C.c_void_p), ("ld_out", C.c_int64), ("stats", C.c_void_p), ("stats_cap", C.c_int32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("o", C.c_void_p),
        ("q_bs", C.c_int64), ("q_rs", C.c_int64), ("k_bs", C.c_int64), ("k_rs", C.c_int64),
        ("vt_bs", C.c_int64), ("vt_hs", C.c_int64), ("vt_ds", C.c_int64), ("o_bs", C.c_int64), ("o_rs", C.c_int64),
        ("batch", C.c_int32), ("heads", C.c_int32), ("nq", C.c_int32), ("nk", C.c_int32), ("nk_pad", C.c_int32),
        ("dpad", C.c_int32), ("scale", C.c_float), ("causal", C.c_int32), ("zero_page", C.c_void_p),
        ("mask", C.c_void_p), ("mask_bs", C.c_int64), ("mask_hs", C.c_int64), ("mask_qs", C.c_int64),
    ]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes; every function returns int (0 = ok).  Must list every symbol of include/fmx.h
# (tests/test_capi_symbols.py parses the header and checks both directions).
SIGNATURES = {
    "fmx_abi_version": [],
    "fmx_device_info": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int],
    "fmx_active_knobs": [C.c_char_p, C.c_int, C.c_int],
    "fmx_gemm_conv_f16": [C.POINTER(GemmArgs), _vp],
    "fmx_gemm_linear_rowstats_f16": [C.POINTER(GemmArgs), _vp, _i32, C.POINTER(C.c_int32), _vp],
    "fmx_layernorm_rowstats_finalize": [_vp, _i32, _i64, _i32, _f32, _vp, _vp],
    "fmx_geglu_interleave_rows": [_vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "fmx_attention_f16": [C.POINTER(AttnArgs), _vp],
    "fmx_softmax_rows_f16": [_vp, _i64, _i32, _i64, _vp],
    "fmx_attention_single_head512_f16": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp],
    "fmx_groupnorm_stats_f16": [_vp, _i32, _i64, _i32, _i32, _vp, _i32, _vp],
    "fmx_groupnorm_apply_f16": [_vp, _vp, _i32, _i32, _i64, _i64, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _f32, _vp, _vp, _i32, _vp, _vp, _vp],
    "fmx_gemm_conv_stats_f16": [C.POINTER(GemmArgs), _vp, _i32, _i32, C.POINTER(C.c_int32), _vp],
    "fmx_layernorm_f16": [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "fmx_layernorm_padded_f16": [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _i64, _i64, _vp],
    "fmx_rmsnorm_f16": [_vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "fmx_layernorm_mod_f16": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i32, _f32, _vp],
    "fmx_flux_qk_norm_rope_f16": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "fmx_timestep_embedding": [_vp, _vp, _i32, _i32, _f32, _vp],
    "fmx_silu_f16": [_vp, _vp, _i64, _vp],
    "fmx_act_f16": [_vp, _vp, _i64, _i32, _vp],
    "fmx_embed_tokens": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "fmx_add_control_nchw": [_vp, _vp, _i32, _i32, _i64, _vp],
    "fmx_cast_f32_to_f16": [_vp, _vp, _i64, _vp],
    "fmx_strided_copy4": [_vp, _i32, C.POINTER(C.c_int64), _vp, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _vp],
    "fmx_unet_pack_input": [_vp, _vp, _f32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "fmx_cfg_combine": [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _f32, _vp],
    "fmx_sampler_euler_step": [_vp, _vp, _f32, _f32, _vp, _f32, _vp, _i64, _vp],
    "fmx_sampler_lincomb": [_vp, _vp, _i32, _vp, _i64, _vp],
    "fmx_sampler_error_norm": [_vp, _vp, _vp, _f32, _f32, _vp, _vp, _i64, _vp],
    "fmx_resize_separable_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "fmx_avgpool2x2_nhwc_f16": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "fmx_add_scaled_f16": [_vp, _vp, _i32, _f32, _i64, _vp],
    "fmx_sampler_lincomb3": [_vp, _vp, _vp, _f32, _f32, _f32, _vp, _i64, _vp],
    "fmx_scale_f32": [_vp, _f32, _vp, _i64, _vp],
    "fmx_vae_pack_latent": [_vp, _f32, _f32, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "fmx_im2col3x3_smallc": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "fmx_vae_unpack_image": [_vp, _i32, _i64, _i32, _vp, _vp],
    "fmx_conv3x3_narrow_f16": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp],
    "fmx_conv3x3_narrow_gn_silu_f16": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp],
    "fmx_conv3x3_up2x_f16": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, C.POINTER(C.c_int32), _vp, _vp],
    "fmx_conv3x3_gn_silu_f16": [C.POINTER(ConvGnArgs), C.POINTER(C.c_int32), _vp],
    "fmx_blend_masked": [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "fmx_count_nonfinite_f16": [_vp, _i64, _vp, _vp],
    "fmx_vae_sample_posterior": [_vp, _i32, _vp, _i32, _i32, _i64, _f32, _f32, _vp, _vp],
    "fmx_philox_randn": [C.c_uint64, C.c_uint32, _vp, _vp, _i64, _vp],
    "fmx_graph_begin": [_vp],
    "fmx_graph_end": [_vp, C.POINTER(C.c_void_p)],
    "fmx_graph_launch": [_vp, _vp],
    "fmx_graph_destroy": [_vp],
    "fmx_event_create": [C.POINTER(C.c_void_p)],
    "fmx_event_record": [_vp, _vp],
    "fmx_event_elapsed_ms": [_vp, _vp, C.POINTER(C.c_float)],
    "fmx_event_destroy": [_vp],
}


# bfloat16 build of the Flux path's kernels: same signatures as the _f16 entries (include/fmx.h, last section)
for _n in ("fmx_gemm_conv", "fmx_attention", "fmx_softmax_rows", "fmx_layernorm", "fmx_layernorm_padded", "fmx_rmsnorm",
           "fmx_layernorm_mod", "fmx_flux_qk_norm_rope", "fmx_silu"):
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n + "_f16"]
SIGNATURES["fmx_timestep_embedding_bf16"] = SIGNATURES["fmx_timestep_embedding"]
# bfloat16 build of the VAE (ABI 6)
for _n in ("fmx_gemm_conv_stats", "fmx_groupnorm_stats", "fmx_groupnorm_apply", "fmx_attention_single_head512", "fmx_conv3x3_narrow", "fmx_conv3x3_gn_silu", "fmx_conv3x3_up2x", "fmx_conv3x3_narrow_gn_silu"):
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n + "_f16"]
for _n in ("fmx_vae_pack_latent", "fmx_vae_unpack_image", "fmx_vae_sample_posterior"):
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n]


def source_tree_hash():
    """hash of the kernel sources in the tree (csrc/src_hash.py: what the Makefile bakes into fmx_build_info of the binary it builds)"""
    res = subprocess.run(["python3", os.path.join(CSRC, "src_hash.py")], capture_output=True, text=True, cwd=CSRC)
    if res.returncode != 0:
        raise FmxError("csrc/src_hash.py failed: " + res.stderr[-1000:])
    return res.stdout.strip()


def binary_matches_sources(path=None):
    """True when the shared object carries the hash of the sources in the tree (read from the file: nothing is loaded)"""
    path = path or os.path.join(_HERE, "libfmx_gfx950.so")
    if not os.path.exists(path):
        return False
    with open(path, "rb") as f:
        return ("src=" + source_tree_hash()).encode() in f.read()


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into libfmx_gfx950.so.  make is incremental (a prebuilt binary that travelled with the tree is reused), but the
    result is only accepted if the hash baked into it equals the hash of the sources in the tree: file times do not survive every way a tree is
    copied, so on a mismatch the objects are thrown away and everything is compiled again.  -> path of the library;
    LAST_BUILD says what happened ("reused" / "incremental" / "full rebuild after hash mismatch" / "forced")."""
    global LAST_BUILD
    lib_path = os.path.join(_HERE, "libfmx_gfx950.so")
    existed = os.path.exists(lib_path)
    mtime = os.path.getmtime(lib_path) if existed else None

    def make(*extra):
        res = subprocess.run(["make", "-C", CSRC, "-j8", *extra], capture_output=True, text=True)
        if verbose:
            print(res.stdout[-2000:])
        if res.returncode != 0:
            raise FmxError("building libfmx_gfx950.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])

    if force:
        make("clean")
    make()
    LAST_BUILD = "forced" if force else ("reused" if existed and os.path.getmtime(lib_path) == mtime else "incremental")
    if not binary_matches_sources(lib_path):
        make("clean")
        make()
        LAST_BUILD = "full rebuild after hash mismatch"
        if not binary_matches_sources(lib_path):
            raise FmxError("libfmx_gfx950.so does not carry the hash of the sources it was just built from")
    if not os.path.exists(lib_path):
        raise FmxError("build finished but %s is missing" % lib_path)
    return lib_path


LAST_BUILD = None


def lib():
    """Load the shared library (once) and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise FmxError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback by design.")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise FmxError(f"symbol {name} missing from {LIB_PATH}") from e
            fn.argtypes = argtypes
            fn.restype = C.c_int
        handle.fmx_last_error.argtypes = []
        handle.fmx_last_error.restype = C.c_char_p
        try:
            handle.fmx_build_info.argtypes = []
            handle.fmx_build_info.restype = C.c_char_p
        except AttributeError as e:
            raise FmxError(f"symbol fmx_build_info missing from {LIB_PATH}") from e
        if handle.fmx_abi_version() != 11:
            raise FmxError("libfmx ABI version mismatch")
        _lib = handle
    return _lib


def build_info():
    """{'src': hash of the kernel sources the LOADED binary was built from, 'abi': ..., 'arch': ...} (fmx_build_info)."""
    return dict(kv.split("=", 1) for kv in lib().fmx_build_info().decode().split())


def active_knobs(ignored=False):
    """Development knobs that took effect in this process (library side + Python side), or -- ignored=True -- that were set without FMX_ALLOW_KNOBS=1."""
    buf = C.create_string_buffer(2048)
    check(lib().fmx_active_knobs(buf, len(buf), 1 if ignored else 0), "fmx_active_knobs")
    out = dict(kv.split("=", 1) for kv in buf.value.decode().split(",") if "=" in kv)
    out.update(IGNORED_KNOBS if ignored else ACTIVE_KNOBS)
    return out


def check(code, what=""):
    if code != 0:
        msg = lib().fmx_last_error()
        raise FmxError(f"{what} failed with code {code}: {msg.decode() if msg else ''}")
