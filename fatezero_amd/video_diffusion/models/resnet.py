"""Pseudo-3D convolution and ResNet blocks (reference: video_diffusion/models/resnet.py).

Token-major fp16 engine: x is [N = B*F, H*W, C].  GroupNorm(+SiLU) is the HIP kernel (statistics span all F
frames of a batch element, exactly like torch.nn.GroupNorm on the reference's 5-D [b,c,f,h,w] input,
resnet.py:338,369); every convolution and projection is the MFMA implicit-GEMM kernel of csrc/igemm.hip (no layout
copies, no library GEMM / convolution on the path).
"""
import copy
import os

import torch
from torch import nn

from ... import dist as D
from ... import kernels as K
from .lora import LoRALinearLayer, _Conv1dParams, pack_temporal_weight, temporal_conv_tokens


# GroupNorm statistics out of the producing launch's epilogue (fz_temporal_conv3_gn / fz_gemm_gn) where that launch runs on a 320-wide tile
# without split-K -- the 64^2 level: the consuming GroupNorm then runs finalize + normalise only (no statistics pass over the tensor).
# The env switch is for same-box A/B runs of bench.py.
GN_FROM_EPILOGUE = os.environ.get("FZ_NO_GN_EPILOGUE") is None
# nearest-2x + 3x3 convolution as four 2x2 convolutions of the input on summed weights (fz_conv3x3_up2: 4 / 9 of the multiply-adds) where the
# kernel carries the shape; FZ_NO_CONV_UP2=1: the nine-tap form through fz_conv3x3(upsample=1) everywhere
CONV_UP2 = os.environ.get("FZ_NO_CONV_UP2") is None


class Tokens:
    """A token-major activation: data [N, H*W, C] fp16 plus its frame geometry.  `gn` (optional): (partial, groups) -- the Welford partials
    of this tensor's GroupNorm statistics, written by the epilogue of the launch that produced it (fz_temporal_conv3_gn / fz_gemm_gn): the
    GroupNorm that consumes the tensor then skips its statistics pass.  Never inherited by `like()`: it describes exactly this data."""
    __slots__ = ("data", "b", "f", "h", "w", "gn")

    def __init__(self, data, b, f, h, w, gn=None):
        self.data, self.b, self.f, self.h, self.w, self.gn = data, b, f, h, w, gn

    @property
    def c(self):
        return self.data.shape[-1]

    def like(self, data, h=None, w=None):
        return Tokens(data, self.b, self.f, self.h if h is None else h, self.w if w is None else w)

    @staticmethod
    def cat(a, b):
        """torch.cat([a, b], channel) as a LAZY pair (unet_3d_blocks.py:384-395: the skip connections of the up blocks): the two
        consumers of the concatenation -- the resnet's first GroupNorm and its 1x1 shortcut convolution -- read the two tensors in
        place (fz_groupnorm_cat; two accumulating GEMMs over the two halves of K); anything else that touches `.data` gets the
        materialised concatenation."""
        return CatTokens(a, b)

    @staticmethod
    def from_bcfhw(x):
        b, c, f, h, w = x.shape
        return Tokens(x.permute(0, 2, 3, 4, 1).reshape(b * f, h * w, c).contiguous(), b, f, h, w)

    def to_bcfhw(self):
        return self.data.view(self.b, self.f, self.h, self.w, self.c).permute(0, 4, 1, 2, 3)


class CatTokens(Tokens):
    __slots__ = ("parts", "_cat")

    def __init__(self, a: Tokens, b: Tokens):
        self.parts, self._cat = (a.data, b.data), None
        self.b, self.f, self.h, self.w, self.gn = a.b, a.f, a.h, a.w, None

    @property
    def data(self):
        if self._cat is None:
            self._cat = torch.cat(self.parts, dim=-1)  # channel concat == last-dim concat in token-major
        return self._cat

    @property
    def c(self):
        return self.parts[0].shape[-1] + self.parts[1].shape[-1]


class PseudoConv3d(nn.Module):
    """resnet.py:12-80. Parameters: weight/bias as nn.Conv2d, `conv_temporal` as in the reference."""

    def __init__(self, in_channels, out_channels, kernel_size, temporal_kernel_size=None, model_config: dict = {},
                 temporal_downsample=False, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride = stride
        self.padding = padding[0] if isinstance(padding, (tuple, list)) else padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if temporal_kernel_size is None:
            temporal_kernel_size = kernel_size
        assert not temporal_downsample, "temporal_downsample is not used by any shipped config"
        if kernel_size > 1:
            if "lora" in model_config:
                self.conv_temporal = LoRALinearLayer(out_channels, out_channels, rank=model_config["lora"])
            else:  # plain temporal Conv1d initialised to identity (resnet.py:42-55)
                self.conv_temporal = _Conv1dParams(out_channels, out_channels, temporal_kernel_size, bias=True)
                nn.init.dirac_(self.conv_temporal.weight.data)
        else:
            self.conv_temporal = None
        self._packed = None
        self._packed_up = None

    def _pack(self, dtype, device):
        if self._packed is None or self._packed[0].dtype != dtype or self._packed[0].device != device:
            if dtype != torch.float16:
                raise RuntimeError("the MI355X engine computes in fp16 storage / fp32 accumulation: call unet.half()")
            w = self.weight.detach().to(device=device, dtype=dtype)
            if self.kernel_size == 1:
                w = w.reshape(self.out_channels, self.in_channels).contiguous()
            else:
                assert self.kernel_size == 3 and self.padding == 1, "the pseudo-3D UNet only uses 3x3 pad-1 and 1x1 convolutions"
                w = K.pack_conv3x3_weight(w)  # [Cout][tap][Cin]: K contiguous per output channel (csrc/igemm.hip)
            bias = self.bias.detach().to(device=device, dtype=dtype)
            wtt = btt = None
            if self.conv_temporal is not None and not isinstance(self.conv_temporal, LoRALinearLayer):
                # plain nn.Conv1d over the frames (resnet.py:42-55), initialised to the identity: the dirac weight with a zero
                # bias is an exact no-op and is skipped (wtt None); anything else runs on fz_temporal_conv3, the bias as a row
                # added per batch element (two rows: the CFG batch)
                tw, tb = self.conv_temporal.weight.detach(), self.conv_temporal.bias.detach()
                ident = torch.zeros_like(tw)
                nn.init.dirac_(ident)
                if not (bool((tw == ident).all()) and bool((tb == 0).all())):
                    wtt = pack_temporal_weight(tw, dtype, device)
                    btt = tb.to(device=device, dtype=dtype)
            self._packed = (w, bias, wtt, btt)
        return self._packed

    def forward_tokens(self, x: Tokens, residual=None, temb=None, upsample=False, gn_groups: int = 0) -> Tokens:
        """conv (+ temporal conv) (+ temb[b] per batch element) (+ residual). temb: [B, Cout] view, residual: [N, T, Cout].
        gn_groups > 0: the caller will GroupNorm the result with that many groups -- where the layer's LAST launch can emit the statistics
        from its epilogue (the LoRA up convolution on a 320-wide tile) the result carries them (`Tokens.gn`).
        Every spatial convolution of the UNet -- all pyramid levels, conv_in, conv_out, the 1x1 shortcuts -- runs through
        the hand-written implicit-GEMM kernel (fz_conv3x3 / fz_gemm); the elementwise tail (time embedding, residual)
        rides in the epilogue of the LAST linear op of the layer (the temporal conv when it is active)."""
        probe = x.parts[0] if isinstance(x, CatTokens) and x._cat is None else x.data  # (do not materialise a lazy concatenation)
        w, bias, wtt, btt = self._pack(probe.dtype, probe.device)
        n, hw = probe.shape[0], probe.shape[1]
        lora = self.conv_temporal if isinstance(self.conv_temporal, LoRALinearLayer) else None
        plain_t = self.conv_temporal is not None and lora is None and wtt is not None
        temporal_active = plain_t or (lora is not None and not lora.is_noop(probe.dtype, probe.device))
        fuse_tail = not temporal_active  # the elementwise tail commutes with nothing but the last linear op
        if self.kernel_size == 1 and isinstance(x, CatTokens) and x._cat is None and residual is None and temb is None \
                and x.parts[0].shape[-1] % 8 == 0:
            # 1x1 convolution of a lazy channel concatenation: W = [W1 | W2] along K, y = x1 W1^T + bias, then += x2 W2^T
            # (the second GEMM takes the first one's output as its residual)
            c1 = x.parts[0].shape[-1]
            y = K.gemm(x.parts[1], w[:, c1:], None, res=K.gemm(x.parts[0], w[:, :c1], bias))
            oh, ow = x.h, x.w
            fused = fuse_tail
        elif self.kernel_size == 1:
            y = K.gemm(x.data, w, bias, res=residual if (fuse_tail and temb is None) else None)
            oh, ow = x.h, x.w
            fused = fuse_tail and temb is None
        else:
            xin = x.data if x.data.is_contiguous() else x.data.contiguous()
            up2 = upsample and CONV_UP2 and self.stride == 1 and self.in_channels % 64 == 0 and self.out_channels % 160 == 0
            if up2 and (self._packed_up is None or self._packed_up[0] is not w):
                # once per packed weight, in the FIRST forward that reaches the upsampler whether or not this launch takes the form (the 8-frame
                # launch of the 8 x 8 level does not, the 16-frame one does): a weight pack must exist before any forward is recorded as an issue
                # plan (issue.py: a block the plans' pool hands out later can alias what an older plan's replay overwrites)
                self._packed_up = (w, K.pack_conv3x3_up2_weight(w))
            if up2 and K.conv3x3_up2_preferred(n, x.h, x.w, self.in_channels, self.out_channels):
                y, (oh, ow) = K.conv3x3_up2(xin, self._packed_up[1], bias, hw=(x.h, x.w))
                fused = fuse_tail and temb is None and residual is None
            else:
                y, (oh, ow) = K.conv3x3(xin, w, bias, hw=(x.h, x.w), stride=self.stride, upsample=upsample,
                                        temb=temb if fuse_tail else None, frames_per_batch=x.f,
                                        res=residual if fuse_tail else None)
                fused = fuse_tail
        gn = None
        if lora is not None and temporal_active:
            y4 = lora.forward_tokens(y.view(x.b, x.f, oh * ow, self.out_channels), temb=temb, residual=residual, gn_groups=gn_groups)
            if isinstance(y4, tuple):
                y4, part = y4
                gn = None if part is None else (part, gn_groups)
            y = y4.reshape(n, oh * ow, self.out_channels)
            fused = True
        elif plain_t:
            rows = btt[None, :].expand(x.b, -1) if temb is None else temb + btt
            shard = D.active_shard()
            y4 = y.view(x.b, x.f, oh * ow, self.out_channels)
            if shard is not None:  # frames split over ranks: one-frame halo from both neighbours, halo outputs dropped
                y4 = temporal_conv_tokens(shard.with_halo(y4, 1, 1, zero_outside=True, tag="temporal_conv"), wtt, rows_add=rows.contiguous())[:, 1:-1]
                y = y4.reshape(n, oh * ow, self.out_channels)
                if residual is not None:
                    y = y + residual
            else:
                y = temporal_conv_tokens(y4, wtt, rows_add=rows.contiguous(), residual=residual).reshape(n, oh * ow, self.out_channels)
            fused = True
        if not fused:
            if temb is not None:
                y = (y.view(x.b, x.f * oh * ow, self.out_channels) + temb[:, None, :]).view(n, oh * ow, self.out_channels)
            if residual is not None:
                y = y + residual
        out = x.like(y, oh, ow)
        out.gn = gn
        return out


class _NormParams(nn.Module):
    """Holds GroupNorm / LayerNorm affine parameters under the reference's names."""

    def __init__(self, channels, groups=None, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.num_groups, self.eps = groups, eps
        self._packed = None

    def packed(self, device):
        if self._packed is None or self._packed[0].device != device:
            self._packed = (self.weight.detach().to(device=device, dtype=torch.float16).contiguous(),
                            self.bias.detach().to(device=device, dtype=torch.float16).contiguous())
        return self._packed


def group_norm_tokens(norm: _NormParams, x: Tokens, *, span_frames: bool, silu: bool) -> Tokens:
    lazy = isinstance(x, CatTokens) and x._cat is None
    g, b = norm.packed((x.parts[0] if lazy else x.data).device)
    shard = D.active_shard()
    if shard is not None and span_frames:
        # statistics span ALL frames of the clip, this rank holds x.f of them: exchange the Welford partials (a few KB)
        # and merge them in the same fixed order on every rank
        n, _, _ = x.data.shape
        part = K.groupnorm_stats(x.data, groups=norm.num_groups)
        allp = shard.all_gather_frames(part.view(n // x.f, x.f, *part.shape[1:]), tag="groupnorm").contiguous()
        y = K.groupnorm_apply(x.data, g, b, allp, span=x.f, groups=norm.num_groups, eps=norm.eps, silu=silu)
        return x.like(y)
    if not lazy and x.gn is not None and x.gn[1] == norm.num_groups and x.gn[0].shape[0] == x.data.shape[0]:
        # the producer's epilogue already wrote the statistics partials: merge + normalise only
        y = K.groupnorm_from_partial(x.data, g, b, x.gn[0], span=(x.f if span_frames else 1), groups=norm.num_groups, eps=norm.eps, silu=silu)
        return Tokens(y, x.b, x.f, x.h, x.w)
    if isinstance(x, CatTokens) and x._cat is None and x.parts[0].shape[-1] % 8 == 0 and x.parts[1].shape[-1] % 8 == 0:
        y = K.groupnorm_cat(x.parts[0], x.parts[1], g, b, span=(x.f if span_frames else 1), groups=norm.num_groups, eps=norm.eps,
                            silu=silu)
        return Tokens(y, x.b, x.f, x.h, x.w)
    y = K.groupnorm(x.data, g, b, span=(x.f if span_frames else 1), groups=norm.num_groups, eps=norm.eps, silu=silu)
    return Tokens(y, x.b, x.f, x.h, x.w)


def upsample_nearest2x(x: Tokens) -> Tokens:
    n, hw, c = x.data.shape
    y = x.data.view(n, x.h, 1, x.w, 1, c).expand(n, x.h, 2, x.w, 2, c).reshape(n, 4 * hw, c)
    return x.like(y, 2 * x.h, 2 * x.w)


class UpsamplePseudo3D(nn.Module):
    """resnet.py:83-175 (nearest 2x per frame, then conv3x3 + temporal)."""

    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv",
                 model_config: dict = {}, **kwargs):
        super().__init__()
        assert use_conv and not use_conv_transpose
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = PseudoConv3d(self.channels, self.out_channels, 3, padding=1, model_config=model_config)

    def forward_tokens(self, x: Tokens) -> Tokens:
        return self.conv.forward_tokens(x, upsample=True)


class DownsamplePseudo3D(nn.Module):
    """resnet.py:178-236 (stride-2 conv3x3 + temporal)."""

    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, model_config: dict = {}, name="conv"):
        super().__init__()
        assert use_conv and padding == 1
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = PseudoConv3d(self.channels, self.out_channels, 3, stride=2, padding=padding, model_config=model_config)

    def forward_tokens(self, x: Tokens) -> Tokens:
        return self.conv.forward_tokens(x)


class _LinearParams(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self._packed = None

    def packed(self, dtype, device):
        if self._packed is None or self._packed[0].dtype != dtype or self._packed[0].device != device:
            self._packed = (self.weight.detach().to(device=device, dtype=dtype).contiguous(),
                            None if self.bias is None else self.bias.detach().to(device=device, dtype=dtype))
        return self._packed

    def apply(self, x, res=None, want_stats=None):
        """x @ W^T + b (+ res) through fz_gemm (csrc/igemm.hip).  want_stats (True / False; None = plain call returning y): returns (y, stats) with the per-row block sums of
        y for a LayerNorm fused into the NEXT Linear (fz_gemm_ln), or (y, None) where that form does not apply."""
        w, b = self.packed(x.dtype, x.device)
        if want_stats is None:
            return K.gemm(x, w, b, res=res)
        if not want_stats or w.shape[0] % 64 or x.dtype != torch.float16:
            return K.gemm(x, w, b, res=res), None
        return K.gemm(x, w, b, res=res, want_stats=True)


class ResnetBlockPseudo3D(nn.Module):
    """resnet.py:239-394 with time_embedding_norm='default', non_linearity swish, output_scale_factor 1."""

    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6,
                 output_scale_factor=1.0, model_config: dict = {}, **unused):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = _NormParams(in_channels, groups, eps)
        self.conv1 = PseudoConv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, model_config=model_config)
        self.time_emb_proj = _LinearParams(temb_channels, out_channels)
        self.norm2 = _NormParams(out_channels, groups, eps)
        self.conv2 = PseudoConv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, model_config=model_config)
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = PseudoConv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0,
                                              model_config=model_config)

    def forward_tokens(self, x: Tokens, temb_act, temb_proj=None) -> Tokens:
        """temb_act = silu(temb) [B, temb_channels] fp16; temb_proj (optional) = this block's time_emb_proj output
        [B, Cout] precomputed by the UNet in one batched GEMM."""
        h = group_norm_tokens(self.norm1, x, span_frames=True, silu=True)
        if temb_proj is None:
            temb_proj = getattr(self, "_temb_cached", None)  # set by the UNet: all blocks' projections in one GEMM
            self._temb_cached = None
        t = temb_proj if temb_proj is not None else self.time_emb_proj.apply(temb_act)  # [B, Cout]
        gg = self.norm2.num_groups if GN_FROM_EPILOGUE else 0
        h = self.conv1.forward_tokens(h, temb=t, gn_groups=gg)
        h = group_norm_tokens(self.norm2, h, span_frames=True, silu=True)
        skip = x if self.conv_shortcut is None else self.conv_shortcut.forward_tokens(x)
        out = self.conv2.forward_tokens(h, residual=skip.data, gn_groups=gg)  # (its consumer: the next GroupNorm, same group count)
        if self.output_scale_factor != 1.0:
            out = out.like(out.data / self.output_scale_factor)
        return out
