"""MI355X-native T2I-Adapter -- drop-in for `Adapter` of backend/nn/cnets/t2i_adapter.py:103-164 (same checkpoint keys: conv_in, body.N.{in_conv,
block1, block2, skep, down_opt.op}).  Pixel-unshuffle (x8, x16 for SDXL) -> conv_in -> per stage `nums_rb` ResnetBlocks (conv, ReLU, conv,
+ skip; the first block of a stage downsamples by a stride-2 conv or a 2x2 average pool) -> one feature map per stage, placed in the list
positions that line them up with the UNet's input blocks (:140-160).  All convolutions are the implicit-GEMM kernel on fp16 NHWC; the
features are handed out as NCHW views of channels-last buffers, like the native ControlNet's residuals.

The adapter runs ONCE per hint image (the patcher-level T2IAdapter caches `control_input`, patcher/controlnet.py:518-535), so this is
job set-up, not per-step work.  `Adapter_light` (:226-293, the colour adapter's network: per stage avg-pool -> 1x1 in_conv to a quarter of the
width -> `nums_rb` x (3x3, ReLU, 3x3, + x) -> 1x1 out_conv) is built below on the same kernels; its quarter widths (80, 160, 320) are zero-padded
to the GEMM's 64-channel granule.  `StyleAdapter` (:166-224) is a CLIP-vision transformer head, a different input path, and is not built."""
import torch

from .... import hipops as ops
from ..unet import _conv_w


def adapter_param_shapes(channels=(320, 640, 1280, 1280), nums_rb=2, cin=192, ksize=1, sk=True, use_conv=False, xl=False):
    """LDM / TencentARC key names and shapes of `Adapter` (what load_t2i_adapter feeds it, patcher/controlnet.py:564-575)."""
    down_at, nodown_at = ((2,), (1,)) if xl else ((3, 2, 1), ())
    s = {"conv_in.weight": (channels[0], cin, 3, 3), "conv_in.bias": (channels[0],)}
    for i in range(len(channels)):
        for j in range(nums_rb):
            k = f"body.{i * nums_rb + j}"
            first = j == 0 and (i in down_at or i in nodown_at)
            in_c, out_c = (channels[i - 1], channels[i]) if first else (channels[i], channels[i])
            if in_c != out_c or not sk:
                s[k + ".in_conv.weight"], s[k + ".in_conv.bias"] = (out_c, in_c, ksize, ksize), (out_c,)
            s[k + ".block1.weight"], s[k + ".block1.bias"] = (out_c, out_c, 3, 3), (out_c,)
            s[k + ".block2.weight"], s[k + ".block2.bias"] = (out_c, out_c, ksize, ksize), (out_c,)
            if not sk:
                s[k + ".skep.weight"], s[k + ".skep.bias"] = (out_c, in_c, ksize, ksize), (out_c,)
            if j == 0 and i in down_at and use_conv:
                s[k + ".down_opt.op.weight"], s[k + ".down_opt.op.bias"] = (in_c, in_c, 3, 3), (in_c,)
    return s


class Adapter:
    def __init__(self, state_dict, channels=(320, 640, 1280, 1280), nums_rb=2, cin=192, ksize=1, sk=True, use_conv=False, xl=False, device="cuda"):
        self.device = torch.device(device)
        self.channels, self.nums_rb, self.xl, self.ksize, self.use_conv = list(channels), nums_rb, xl, ksize, use_conv
        self.unshuffle_amount = 16 if xl else 8
        self.input_channels = cin // (self.unshuffle_amount ** 2)
        self.down_at = (2,) if xl else (3, 2, 1)
        if cin % 64 or any(c % 64 for c in channels):
            raise NotImplementedError("T2I-Adapter channel counts must be multiples of 64 (GEMM channel granule)")
        T = lambda k: state_dict[k].to(device=self.device, dtype=torch.float16).contiguous()
        conv = lambda k: (_conv_w(state_dict[k + ".weight"].to(self.device, torch.float16)), T(k + ".bias")) if k + ".weight" in state_dict else None
        self.w = {"conv_in": conv("conv_in")}
        for n in range(len(channels) * nums_rb):
            for part in ("in_conv", "block1", "block2", "skep", "down_opt.op"):
                self.w[f"body.{n}.{part}"] = conv(f"body.{n}.{part}")

    def _conv(self, name, x, k, stride=1, residual=None):
        wt, b = self.w[name]
        bb, h, w, _ = x.shape
        oh, ow = ((h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1) if k == 3 else (h, w)
        out = ops.conv_gemm(x, wt, wt.shape[0], kh=k, stride=stride, pad=1 if k == 3 else 0, bias=b, residual=residual)
        return out.view(bb, oh, ow, wt.shape[0])

    def _block(self, n, x, down):
        p = f"body.{n}"
        if down:
            x = self._conv(p + ".down_opt.op", x, 3, stride=2) if self.use_conv else ops.avgpool2x2(x)
        if self.w[p + ".in_conv"] is not None:
            x = self._conv(p + ".in_conv", x, self.ksize)
        h = ops.act(self._conv(p + ".block1", x, 3), ops.ACT_RELU)
        skip = self._conv(p + ".skep", x, self.ksize) if self.w[p + ".skep"] is not None else x
        return self._conv(p + ".block2", h, self.ksize, residual=skip.reshape(-1, skip.shape[-1]))

    @torch.inference_mode()
    def forward(self, x):
        """x [B, input_channels, H, W] -> the reference's feature list (t2i_adapter.py:133-160): None placeholders + one NCHW fp16 feature map
        per stage (channels-last memory)."""
        r = self.unshuffle_amount
        xs = torch.nn.functional.pixel_unshuffle(x.to(device=self.device, dtype=torch.float32), r)   # layout prep of the input image
        h = self._conv("conv_in", xs.permute(0, 2, 3, 1).contiguous().half(), 3)
        features = []
        for i in range(len(self.channels)):
            for j in range(self.nums_rb):
                h = self._block(i * self.nums_rb + j, h, down=(j == 0 and i in self.down_at))
            if self.xl:
                features.append(None)
                if i == 0:
                    features += [None, None]
                if i == 2:
                    features.append(None)
            else:
                features += [None, None]
            features.append(h.permute(0, 3, 1, 2))
        return features

    __call__ = forward



def adapter_light_param_shapes(channels=(320, 640, 1280, 1280), nums_rb=4, cin=64):
    """Key names and shapes of `Adapter_light` (t2i_adapter.py:241-281)."""
    s = {}
    for i, c in enumerate(channels):
        in_c, inter = (cin if i == 0 else channels[i - 1]), c // 4
        s[f"body.{i}.in_conv.weight"], s[f"body.{i}.in_conv.bias"] = (inter, in_c, 1, 1), (inter,)
        for j in range(nums_rb):
            for blk in ("block1", "block2"):
                s[f"body.{i}.body.{j}.{blk}.weight"], s[f"body.{i}.body.{j}.{blk}.bias"] = (inter, inter, 3, 3), (inter,)
        s[f"body.{i}.out_conv.weight"], s[f"body.{i}.out_conv.bias"] = (c, inter, 1, 1), (c,)
    return s


class Adapter_light:
    def __init__(self, state_dict, channels=(320, 640, 1280, 1280), nums_rb=4, cin=64, device="cuda"):
        self.device = torch.device(device)
        self.channels, self.nums_rb, self.xl = list(channels), nums_rb, False
        self.unshuffle_amount = 8
        self.input_channels = cin // 64
        if cin % 64 or any(c % 64 for c in channels):
            raise NotImplementedError("T2I-Adapter channel counts must be multiples of 64 (GEMM channel granule)")
        pad64 = lambda c: -(-c // 64) * 64

        def conv(key, pad_out, pad_in):
            """weights zero-padded on the quarter-width side(s): padded channels stay exactly 0 through bias 0, ReLU and the residual add"""
            wt, b = state_dict[key + ".weight"].to(self.device, torch.float16), state_dict[key + ".bias"].to(self.device, torch.float16)
            co, ci = (pad64(wt.shape[0]) if pad_out else wt.shape[0]), (pad64(wt.shape[1]) if pad_in else wt.shape[1])
            full = wt.new_zeros(co, ci, wt.shape[2], wt.shape[3])
            full[:wt.shape[0], :wt.shape[1]] = wt
            bias = b.new_zeros(co)
            bias[:b.shape[0]] = b
            return _conv_w(full), bias.contiguous()
        self.w = {}
        for i in range(len(channels)):
            self.w[f"body.{i}.in_conv"] = conv(f"body.{i}.in_conv", True, False)
            for j in range(nums_rb):
                for blk in ("block1", "block2"):
                    self.w[f"body.{i}.body.{j}.{blk}"] = conv(f"body.{i}.body.{j}.{blk}", True, True)
            self.w[f"body.{i}.out_conv"] = conv(f"body.{i}.out_conv", False, True)

    def _conv(self, name, x, k, residual=None):
        wt, b = self.w[name]
        bb, h, w, _ = x.shape
        return ops.conv_gemm(x, wt, wt.shape[0], kh=k, stride=1, pad=1 if k == 3 else 0, bias=b, residual=residual).view(bb, h, w, wt.shape[0])

    @torch.inference_mode()
    def forward(self, x):
        """x [B, input_channels, H, W] -> [None, None, f0, None, None, f1, ...] (t2i_adapter.py:282-293), NCHW fp16 views of channels-last buffers."""
        xs = torch.nn.functional.pixel_unshuffle(x.to(device=self.device, dtype=torch.float32), self.unshuffle_amount)
        h = xs.permute(0, 2, 3, 1).contiguous().half()
        features = []
        for i in range(len(self.channels)):
            if i > 0:
                h = ops.avgpool2x2(h)
            h = self._conv(f"body.{i}.in_conv", h, 1)
            for j in range(self.nums_rb):
                t = ops.act(self._conv(f"body.{i}.body.{j}.block1", h, 3), ops.ACT_RELU)
                h = self._conv(f"body.{i}.body.{j}.block2", t, 3, residual=h.reshape(-1, h.shape[-1]))
            h = self._conv(f"body.{i}.out_conv", h, 1)
            features += [None, None, h.permute(0, 3, 1, 2)]
        return features

    __call__ = forward
