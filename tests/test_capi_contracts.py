"""Argument contracts of the C-ABI (include/fmx.h): every entry point validates on the HOST before it launches anything, returns
FMX_E_BADARG (10001) and leaves a message in fmx_last_error().  No GPU is needed to check that -- nothing here gets as far as a launch."""
import ctypes as C

import pytest

import forge_amd  # noqa: F401
from forge_amd import _lib
from forge_amd._lib import AttnArgs, GemmArgs

BADARG, UNSUPPORTED = 10001, 10002
# entry points whose first act is a HIP runtime call (device query, graph / event objects): not exercised without a device
RUNTIME = {"fmx_abi_version", "fmx_device_info", "fmx_graph_begin", "fmx_graph_end", "fmx_graph_launch", "fmx_graph_destroy", "fmx_event_create",
           "fmx_event_record", "fmx_event_elapsed_ms", "fmx_event_destroy"}
FAKE = 0x7F0000001000  # a 16-byte aligned non-null "device pointer": validation never dereferences it


@pytest.fixture(scope="module")
def lib():
    try:
        return _lib.lib()
    except _lib.FmxError as e:
        pytest.skip(f"libfmx not built: {e}")


def _zero_args(argtypes):
    out = []
    for t in argtypes:
        if t in (C.c_void_p, C.c_char_p) or (hasattr(t, "_type_") and not isinstance(t._type_, str)):
            out.append(None)
        elif t in (C.c_float, C.c_double):
            out.append(0.0)
        else:
            out.append(0)
    return out


def test_every_entry_point_rejects_null_pointers_and_zero_sizes(lib):
    checked = 0
    for name, argtypes in _lib.SIGNATURES.items():
        if name in RUNTIME:
            continue
        rc = getattr(lib, name)(*_zero_args(argtypes))
        assert rc == BADARG, (name, rc)
        assert lib.fmx_last_error().decode().strip(), name
        checked += 1
    assert checked >= 40


def _gemm(**kw):
    a = GemmArgs()
    a.a0 = a.wgt = a.out = a.zero_page = FAKE
    a.c0, a.c1, a.n, a.h, a.w, a.oh, a.ow, a.kh, a.stride, a.pad, a.nout = 64, 0, 1, 1, 128, 1, 128, 1, 1, 0, 64
    a.alpha = 1.0
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("suffix", ["_f16", "_bf16"])
@pytest.mark.parametrize("kw,msg", [
    (dict(c0=48), "multiples of 64"),                       # K granule of every GEMM kernel
    (dict(c1=64), "a1 missing"),                            # second source announced but not given
    (dict(kh=5), "kh must be 1 or 3"),
    (dict(act=1, nout=48), "bad act/nout"),                 # GEGLU interleave granule
    (dict(a0=FAKE + 8), "16-byte aligned"),
    (dict(a0_stride=68), "multiples of 8"),
    (dict(nout=0), "bad dims"),
])
def test_gemm_contract_violations(lib, suffix, kw, msg):
    rc = getattr(lib, "fmx_gemm_conv" + suffix)(C.byref(_gemm(**kw)), None)
    assert rc == BADARG and msg in lib.fmx_last_error().decode()


def test_other_contract_violations(lib):
    err = lambda: lib.fmx_last_error().decode()  # noqa: E731
    p = C.c_void_p(FAKE)
    assert lib.fmx_layernorm_f16(p, p, p, p, 4, 4104, 1e-5, None) == BADARG                      # c > 4096
    assert lib.fmx_layernorm_f16(p, p, p, p, 4, 100, 1e-5, None) == BADARG                       # c % 8
    assert lib.fmx_layernorm_padded_f16(p, p, p, p, 10, 64, 1e-5, 4, 8, None) == BADARG and "row geometry" in err()   # rows % rows_per_image
    assert lib.fmx_layernorm_padded_f16(p, p, p, p, 8, 64, 1e-5, 4, 2, None) == BADARG                                # out stride < rows per image
    assert lib.fmx_flux_qk_norm_rope_f16(p, 384, p, p, p, p, p, p, 1, 8, 1, 64, 0, 64, 1e-6, None) == BADARG and "128" in err()
    assert lib.fmx_flux_qk_norm_rope_bf16(p, 384, p, p, p, p, p, p, 1, 80, 1, 128, 0, 64, 1e-6, None) == BADARG       # l_pad < row_off + tokens
    assert lib.fmx_sampler_lincomb(p, p, 9, p, 16, None) == BADARG and "1..8" in err()
    assert lib.fmx_avgpool2x2_nhwc_f16(p, p, 1, 5, 4, 64, None) == BADARG and "even" in err()
    assert lib.fmx_act_f16(p, p, 16, 7, None) == BADARG                                          # unknown activation kind
    # the direct narrow-output 3x3 convolution (ABI 8): 1..4 output channels, 32 / 64 / 128 input channels, one launch's 32-bit offset range
    assert lib.fmx_conv3x3_narrow_f16(p, 1, 8, 8, 128, p, None, 5, p, 8, None) == BADARG and "1..4 output" in err()
    assert lib.fmx_conv3x3_narrow_f16(p, 1, 8, 8, 80, p, None, 3, p, 4, None) == BADARG and "multiple of 32" in err()
    assert lib.fmx_conv3x3_narrow_bf16(p, 1, 8, 8, 128, p, None, 3, p, 2, None) == BADARG                              # ld_out < nout
    assert lib.fmx_conv3x3_narrow_f16(p, 16, 1024, 1024, 128, p, None, 3, p, 4, None) == BADARG and "split the batch" in err()
    assert lib.fmx_conv3x3_narrow_f16(None, 1, 8, 8, 128, p, None, 3, p, 4, None) == BADARG
    # GroupNorm + SiLU + 3x3 convolution in one kernel (ABI 11): 128 output channels, input channels in 64-channel chunks, a statistics buffer with one
    # record per 8 x 32 tile
    def cg(**kw):
        a = _lib.ConvGnArgs()
        a.x = a.x_partial = a.gamma = a.beta = a.scale_shift = a.wgt = a.out = FAKE
        a.n, a.h, a.w, a.cin, a.x_nchunks, a.groups, a.eps, a.cout, a.ld_out = 1, 64, 64, 128, 4, 32, 1e-6, 128, 128
        for k, v in kw.items():
            setattr(a, k, v)
        return C.byref(a)
    for sfx in ("_f16", "_bf16"):
        fn = getattr(lib, "fmx_conv3x3_gn_silu" + sfx)
        assert fn(cg(cout=256), None, None) == BADARG and "128 output channels" in err()
        assert fn(cg(cin=96), None, None) == BADARG and "multiple of 64" in err()
        assert fn(cg(x_partial=None), None, None) == BADARG and "null pointer" in err()
        assert fn(cg(groups=24), None, None) == BADARG and "groups" in err()
        assert fn(cg(ld_out=64), None, None) == BADARG and "leading dimensions" in err()
        assert fn(cg(out=FAKE + 2), None, None) == BADARG and "alignment" in err()
        assert fn(cg(stats=FAKE, stats_cap=8), None, None) == BADARG and "16 tiles" in err()       # 64 x 64 pixels = 8 x 2 tiles
    # ... and with GroupNorm + SiLU in its staging (ABI 11)
    f0 = C.c_void_p(FAKE)
    assert lib.fmx_conv3x3_narrow_gn_silu_f16(p, 1, 8, 8, 128, f0, 1, 32, 1e-6, p, p, f0, p, None, 5, p, 8, None) == BADARG and "1..4 output" in err()
    assert lib.fmx_conv3x3_narrow_gn_silu_bf16(p, 1, 8, 8, 96, f0, 1, 64, 1e-6, p, p, f0, p, None, 3, p, 4, None) == BADARG and "group count" in err()
    assert lib.fmx_conv3x3_narrow_gn_silu_f16(p, 1, 8, 8, 128, None, 1, 32, 1e-6, p, p, f0, p, None, 3, p, 4, None) == BADARG
    # the Upsample convolution as four phase convolutions (ABI 11): channel granules, an input width in 32s, whole statistics chunks
    f = C.c_void_p(FAKE)
    for sfx in ("_f16", "_bf16"):
        fn = getattr(lib, "fmx_conv3x3_up2x" + sfx)
        assert fn(p, 1, 16, 32, 96, p, None, 64, p, None, 0, None, p, None) == BADARG and "multiple of 64" in err()
        assert fn(p, 1, 16, 40, 64, p, None, 64, p, None, 0, None, p, None) == BADARG and "multiple of 32" in err()
        assert fn(p, 1, 5, 32, 64, p, None, 64, p, f, 64, None, p, None) == BADARG and "multiple of 256" in err()
        assert fn(p, 1, 16, 32, 64, p, None, 64, p, f, 4, None, p, None) == BADARG and "four phases write 8" in err()
        assert fn(p, 16, 1024, 1024, 128, p, None, 64, p, None, 0, None, p, None) == BADARG and "split the batch" in err()
        assert fn(p, 1, 16, 32, 64, p, None, 64, p, None, 0, None, None, None) == BADARG
    # GroupNorm: a second source needs its own statistics; channel / group / stride geometry
    f = C.c_void_p(FAKE)
    assert lib.fmx_groupnorm_apply_f16(p, p, 64, 64, 64, 64, 1, 16, f, 1, None, 0, 32, 1e-5, p, p, 0, f, p, None) == BADARG and "second source" in err()
    assert lib.fmx_groupnorm_apply_f16(p, None, 72, 0, 72, 0, 1, 16, f, 1, None, 0, 32, 1e-5, p, p, 0, f, p, None) == BADARG and "bad channels" in err()
    assert lib.fmx_groupnorm_apply_f16(p, None, 64, 0, 60, 0, 1, 16, f, 1, None, 0, 32, 1e-5, p, p, 0, f, p, None) == BADARG and "geometry" in err()
    assert lib.fmx_groupnorm_stats_f16(p, 64, 64, 1, 16, f, 2000, None) == BADARG and "nchunks" in err()
    # statistics out of the GEMM: dense fp16 [M][nout] output without activation, sane chunk counts
    nch = C.c_int32(0)
    assert lib.fmx_gemm_conv_stats_f16(C.byref(_gemm(ld_out=128)), f, 4, 2, C.byref(nch), None) == BADARG and "statistics" in err()
    assert lib.fmx_gemm_conv_stats_f16(C.byref(_gemm(ld_out=64, act=2)), f, 4, 2, C.byref(nch), None) == BADARG and "statistics" in err()
    assert lib.fmx_gemm_conv_stats_f16(C.byref(_gemm(ld_out=64)), f, 1, 2, C.byref(nch), None) == BADARG and "chunk counts" in err()
    assert lib.fmx_gemm_conv_stats_f16(C.byref(_gemm(ld_out=64)), None, 4, 2, C.byref(nch), None) == BADARG
    assert lib.fmx_gemm_conv_f16(C.byref(_gemm(ld_out=64, out_f32=-4)), None) == BADARG and "no longer part" in err()   # the retired ping-pong tile id
    # LayerNorm folding: a GEMM is the producer OR the consumer; the consumer is a plain linear with column sums and 1..8 parts (one per 160 output columns of the producer)
    parts = C.c_int32(0)
    assert lib.fmx_gemm_linear_rowstats_f16(C.byref(_gemm(ld_out=64)), None, 8, C.byref(parts), None) == BADARG
    assert lib.fmx_gemm_linear_rowstats_f16(C.byref(_gemm(ld_out=64)), f, 1, C.byref(parts), None) == BADARG
    assert lib.fmx_gemm_linear_rowstats_f16(C.byref(_gemm(ld_out=64, ln_partial=FAKE, ln_parts=2, ln_colsum=FAKE)), f, 8, C.byref(parts), None) == BADARG \
        and "producer or the consumer" in err()
    for bad in (dict(ln_parts=0), dict(ln_parts=10), dict(ln_colsum=0), dict(residual=FAKE, ld_res=64), dict(kh=3, pad=1), dict(act=2)):
        kw = dict(ld_out=64, ln_partial=FAKE, ln_parts=2, ln_colsum=FAKE, ln_eps=1e-5)
        kw.update(bad)
        assert lib.fmx_gemm_conv_f16(C.byref(_gemm(**kw)), None) == BADARG and "LayerNorm-folded" in err(), bad
    # the operand-swapped fold (ABI 7): no bias / residual, both operand arrays, M a multiple of 320, not together with the ordinary fold
    for bad in (dict(ln_row_cb=0), dict(bias=FAKE), dict(residual=FAKE, ld_res=64), dict(ln_partial=FAKE, ln_parts=2, ln_colsum=FAKE), dict(act=1)):
        kw = dict(ld_out=64, ln_col_ab=FAKE, ln_row_cb=FAKE)
        kw.update(bad)
        assert lib.fmx_gemm_conv_f16(C.byref(_gemm(**kw)), None) == BADARG, bad
    assert lib.fmx_layernorm_rowstats_finalize(None, 8, 64, 1280, 1e-5, f, None) == BADARG
    assert lib.fmx_layernorm_rowstats_finalize(f, 0, 64, 1280, 1e-5, f, None) == BADARG
    a = AttnArgs()
    a.q = a.k = a.vt = a.o = a.zero_page = FAKE
    a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.dpad, a.scale = 1, 1, 64, 64, 64, 200, 0.1
    assert lib.fmx_attention_f16(C.byref(a), None) == UNSUPPORTED                                # head dim the kernels are not built for
    a.dpad, a.nk_pad = 64, 60
    assert lib.fmx_attention_f16(C.byref(a), None) == BADARG                                     # key padding granule


def test_host_tensors_and_mixed_element_types_are_rejected_not_converted():
    """hipops never falls back to a torch implementation: host tensors, fp32 activations or a mix of fp16 and bf16 operands raise."""
    import torch
    from forge_amd import hipops as ops
    x, w = torch.zeros(64, 64, dtype=torch.float16), torch.zeros(64, 64, dtype=torch.float16)
    with pytest.raises(TypeError):
        ops.linear(x, w)                                   # host memory
    with pytest.raises(TypeError):
        ops.silu(torch.zeros(8))                           # fp32 on the host
    with pytest.raises(TypeError):
        ops.layernorm_mod(x, w.bfloat16(), w, 64)
