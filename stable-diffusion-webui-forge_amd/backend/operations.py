"""Op-level drop-in: `ForgeOperations` / `using_forge_operations` with the reference's shape (backend/operations.py:125-330, :442-467) --
`torch.nn.Module`s that hold their parameters under the usual names in the LDM / torch layout (so checkpoints load unchanged) and run
their forward on the MI355X kernels through the C-ABI:

    Linear     F.linear      (operations.py:153,156)   -> fmx_gemm_conv_f16 (MFMA GEMM, bias in the epilogue)
    Conv2d     F.conv2d      (operations.py:173,176)   -> fmx_gemm_conv_f16 (implicit GEMM: 1x1 / 3x3, stride 1 / 2, zero padding)
    GroupNorm  F.group_norm  (operations.py:308)       -> fmx_groupnorm_stats_f16 + fmx_groupnorm_apply_f16
    LayerNorm  F.layer_norm  (operations.py:327)       -> fmx_layernorm_f16

A model built under `using_forge_operations()` therefore runs these four op types natively with no other change.  This is the boundary for
code that must keep the reference's nn.Module graph (a third-party network, a ControlNet variant the native executors do not know); the native
UNet / VAE / Flux executors (backend/nn/*.py) do NOT go through it -- they keep activations in fp16 NHWC between kernels and fuse what this
level cannot see (norm + activation + concat, residual adds, GEGLU, upsample-on-load).  Each forward here pays a layout round trip
(NCHW any-float <-> NHWC fp16, one strided-copy kernel each way).

No fallback: tensors on the host, dilation / groups / kernel sizes the implicit GEMM is not built for raise; nothing routes to
torch.nn.functional.  Weights are converted to the kernel layout on first use and again whenever the parameter is replaced or modified
in place (LoRA merges patch `weight` after load).  Conv1d / Conv3d / ConvTranspose* / Embedding are left as torch's own classes: they do
not occur on the hot path named by SURVEY section 8.
"""
import contextlib

import torch

from .. import hipops as ops

current_device = None
current_dtype = None


def _need_cuda(x):
    if not x.is_cuda:
        raise TypeError("forge_amd.backend.operations runs on the MI355X only: got a tensor on %s (there is no CPU path)" % x.device)


def _pad64(c):
    return -(-c // 64) * 64


class _Cached:
    """fp16 kernel-layout copy of a parameter, rebuilt when the parameter object, its storage or its version counter changes."""

    def __init__(self):
        self.key, self.value = None, None

    def get(self, p, build):
        try:
            ver = p._version
        except RuntimeError:
            ver = -1
        key = (id(p), p.data_ptr(), ver, p.device)
        if key != self.key:
            self.key, self.value = key, build(p.detach())
        return self.value


def _f16_vector(t):
    out = torch.empty(t.numel(), dtype=torch.float16, device=t.device)
    ops.strided_copy4(t, out, (1, 1, 1, t.numel()), (0, 0, 0, t.stride(-1) if t.dim() else 1), (0, 0, 0, 1))
    return out


class ForgeOperations:
    class Linear(torch.nn.Module):
        def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
            super().__init__()
            self.in_features, self.out_features = in_features, out_features
            kw = dict(device=device if device is not None else current_device, dtype=dtype if dtype is not None else current_dtype)
            self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, **kw), requires_grad=False)
            self.bias = torch.nn.Parameter(torch.empty(out_features, **kw), requires_grad=False) if bias else None
            self._w, self._b = _Cached(), _Cached()

        def _weight_f16(self, w):
            kp = _pad64(self.in_features)
            out = torch.zeros(self.out_features, kp, dtype=torch.float16, device=w.device)
            ops.strided_copy4(w, out, (1, 1, self.out_features, self.in_features), (0, 0, w.stride(0), w.stride(1)), (0, 0, kp, 1))
            return out

        def forward(self, x):
            _need_cuda(x)
            k, n = self.in_features, self.out_features
            lead = x.shape[:-1]
            x2 = x.reshape(-1, k)
            m, kp = x2.shape[0], _pad64(k)
            xh = torch.zeros(m, kp, dtype=torch.float16, device=x.device) if kp != k else torch.empty(m, kp, dtype=torch.float16, device=x.device)
            ops.strided_copy4(x2, xh, (1, 1, m, k), (0, 0, x2.stride(0), x2.stride(1)), (0, 0, kp, 1))
            w = self._w.get(self.weight, self._weight_f16)
            b = self._b.get(self.bias, _f16_vector) if self.bias is not None else None
            y = ops.conv_gemm(xh, w, n, bias=b)
            out = torch.empty(m, n, dtype=x.dtype, device=x.device)
            ops.strided_copy4(y, out, (1, 1, m, n), (0, 0, y.stride(0), 1), (0, 0, n, 1))
            return out.reshape(*lead, n)

    class Conv2d(torch.nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, padding_mode="zeros",
                     device=None, dtype=None):
            super().__init__()
            pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
            self.in_channels, self.out_channels = in_channels, out_channels
            self.kernel_size, self.stride, self.padding, self.dilation, self.groups = pair(kernel_size), pair(stride), pair(padding), pair(dilation), groups
            kh, kw_ = self.kernel_size
            if kh != kw_ or kh not in (1, 3) or self.stride[0] != self.stride[1] or self.stride[0] not in (1, 2) or self.padding[0] != self.padding[1] \
                    or self.dilation != (1, 1) or groups != 1 or padding_mode != "zeros":
                raise NotImplementedError("the MI355X implicit-GEMM convolution covers square 1x1 / 3x3 kernels, stride 1 / 2, symmetric zero padding, "
                                          f"no dilation, no groups; got kernel {self.kernel_size} stride {self.stride} padding {self.padding} "
                                          f"dilation {self.dilation} groups {groups} ({padding_mode})")
            kw = dict(device=device if device is not None else current_device, dtype=dtype if dtype is not None else current_dtype)
            self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels, kh, kh, **kw), requires_grad=False)
            self.bias = torch.nn.Parameter(torch.empty(out_channels, **kw), requires_grad=False) if bias else None
            self._w, self._b = _Cached(), _Cached()

        def _weight_f16(self, w):
            co, ci, kh, _ = w.shape
            cp = _pad64(ci)
            out = torch.zeros(co, kh * kh * cp, dtype=torch.float16, device=w.device)      # [Cout][ky][kx][Cin padded]: the GEMM's K order
            ops.strided_copy4(w, out, (co, kh * kh, ci, 1), (w.stride(0), w.stride(3), w.stride(1), 0), (kh * kh * cp, cp, 1, 0))
            return out

        def forward(self, x):
            _need_cuda(x)
            b, ci, hh, ww = x.shape
            assert ci == self.in_channels
            cp = _pad64(ci)
            xh = torch.zeros(b, hh, ww, cp, dtype=torch.float16, device=x.device) if cp != ci else torch.empty(b, hh, ww, cp, dtype=torch.float16, device=x.device)
            ops.strided_copy4(x, xh, (b, hh, ww, ci), (x.stride(0), x.stride(2), x.stride(3), x.stride(1)), xh.stride())
            w = self._w.get(self.weight, self._weight_f16)
            bb = self._b.get(self.bias, _f16_vector) if self.bias is not None else None
            kh, st, pad = self.kernel_size[0], self.stride[0], self.padding[0]
            oh, ow = (hh + 2 * pad - kh) // st + 1, (ww + 2 * pad - kh) // st + 1
            y = ops.conv_gemm(xh, w, self.out_channels, kh=kh, stride=st, pad=pad, bias=bb)        # [B*oh*ow, Cout]
            out = torch.empty(b, self.out_channels, oh, ow, dtype=x.dtype, device=x.device)
            y4 = y.view(b, oh, ow, self.out_channels)
            ops.strided_copy4(y4, out, (b, oh, ow, self.out_channels), y4.stride(), (out.stride(0), out.stride(2), out.stride(3), out.stride(1)))
            return out

    class GroupNorm(torch.nn.Module):
        def __init__(self, num_groups, num_channels, eps=1e-5, affine=True, device=None, dtype=None):
            super().__init__()
            if num_channels % 8 or num_channels % num_groups:
                raise NotImplementedError(f"GroupNorm on the MI355X kernels needs channels % 8 == 0 and % groups == 0, got {num_channels} / {num_groups}")
            self.num_groups, self.num_channels, self.eps, self.affine = num_groups, num_channels, eps, affine
            kw = dict(device=device if device is not None else current_device, dtype=dtype if dtype is not None else current_dtype)
            self.weight = torch.nn.Parameter(torch.ones(num_channels, **kw), requires_grad=False) if affine else None
            self.bias = torch.nn.Parameter(torch.zeros(num_channels, **kw), requires_grad=False) if affine else None
            self._w, self._b = _Cached(), _Cached()

        def forward(self, x):
            _need_cuda(x)
            shape = x.shape
            b, c = shape[0], shape[1]
            x4 = x.reshape(b, c, -1, 1)                                              # [B, C, L, 1]: any number of trailing spatial dims
            n = x4.shape[2]
            xh = torch.empty(b, n, 1, c, dtype=torch.float16, device=x.device)
            ops.strided_copy4(x4, xh, (b, n, 1, c), (x4.stride(0), x4.stride(2), 0, x4.stride(1)), xh.stride())
            if self.affine:
                g, bt = self._w.get(self.weight, _f16_vector), self._b.get(self.bias, _f16_vector)
            else:
                g, bt = torch.ones(c, dtype=torch.float16, device=x.device), torch.zeros(c, dtype=torch.float16, device=x.device)
            y = ops.groupnorm(xh, g, bt, self.eps, groups=self.num_groups)
            out = torch.empty(shape, dtype=x.dtype, device=x.device)
            o4 = out.reshape(b, c, n, 1)
            ops.strided_copy4(y, o4, (b, n, 1, c), y.stride(), (o4.stride(0), o4.stride(2), 0, o4.stride(1)))
            return out

    class LayerNorm(torch.nn.Module):
        def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, bias=True, device=None, dtype=None):
            super().__init__()
            ns = (normalized_shape,) if isinstance(normalized_shape, int) else tuple(normalized_shape)
            if len(ns) != 1 or ns[0] % 8 or ns[0] > 4096:
                raise NotImplementedError(f"LayerNorm on the MI355X kernel normalises the last dimension, width % 8 == 0 and <= 4096; got {ns}")
            self.normalized_shape, self.eps, self.elementwise_affine = ns, eps, elementwise_affine
            kw = dict(device=device if device is not None else current_device, dtype=dtype if dtype is not None else current_dtype)
            self.weight = torch.nn.Parameter(torch.ones(ns, **kw), requires_grad=False) if elementwise_affine else None
            self.bias = torch.nn.Parameter(torch.zeros(ns, **kw), requires_grad=False) if elementwise_affine and bias else None
            self._w, self._b = _Cached(), _Cached()

        def forward(self, x):
            _need_cuda(x)
            c = self.normalized_shape[0]
            x2 = x.reshape(-1, c)
            m = x2.shape[0]
            xh = torch.empty(m, c, dtype=torch.float16, device=x.device)
            ops.strided_copy4(x2, xh, (1, 1, m, c), (0, 0, x2.stride(0), x2.stride(1)), (0, 0, c, 1))
            g = self._w.get(self.weight, _f16_vector) if self.weight is not None else torch.ones(c, dtype=torch.float16, device=x.device)
            bt = self._b.get(self.bias, _f16_vector) if self.bias is not None else torch.zeros(c, dtype=torch.float16, device=x.device)
            y = ops.layernorm(xh, g, bt, self.eps)
            out = torch.empty(m, c, dtype=x.dtype, device=x.device)
            ops.strided_copy4(y, out, (1, 1, m, c), (0, 0, c, 1), (0, 0, c, 1))
            return out.reshape(x.shape)


NATIVE_OPS = ("Linear", "Conv2d", "GroupNorm", "LayerNorm")


@contextlib.contextmanager
def using_forge_operations(operations=None, device=None, dtype=None, manual_cast_enabled=False, bnb_dtype=None):
    """Reference: backend/operations.py:442-467 -- while active, `torch.nn.{Linear, Conv2d, GroupNorm, LayerNorm}` construct the classes above, so a
    network definition written against torch.nn is instantiated on the native kernels.  `bnb_dtype` (GGUF / NF4 quantised weights) is outside
    the hot path and rejected."""
    global current_device, current_dtype
    if bnb_dtype is not None:
        raise NotImplementedError("quantised (bnb / GGUF) operations are not part of the MI355X hot path")
    operations = ForgeOperations if operations is None else operations
    saved = (current_device, current_dtype)
    current_device, current_dtype = device, dtype
    backups = {n: getattr(torch.nn, n) for n in NATIVE_OPS}
    try:
        for n in NATIVE_OPS:
            setattr(torch.nn, n, getattr(operations, n))
        yield
    finally:
        for n in NATIVE_OPS:
            setattr(torch.nn, n, backups[n])
        current_device, current_dtype = saved
