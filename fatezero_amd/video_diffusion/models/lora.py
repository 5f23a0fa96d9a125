"""Temporal LoRA convolution (reference: video_diffusion/models/lora.py:22-54, LoRALinearLayer).

x[(b h w), c, f] -> up(down(x)) + x with two bias-free Conv1d(k=3, pad=1) over the frame axis.  On token-major
activations [B, F, HW, C] a k=3 temporal conv is an implicit GEMM whose three taps read shifted frames (zero padding
at the clip ends = lanes pointing at a page of zeros): fz_temporal_conv3 -- no '(b h w) c f' rearrange is materialised
and no library GEMM is involved.  Parameter names/shapes are the reference's (`down.weight [r, C, 3]`, `up.weight [C, r, 3]`) so checkpoints load unchanged.
"""
import os

import torch
from torch import nn

from ... import dist as D
from ... import kernels as K


class _Conv1dParams(nn.Module):
    def __init__(self, cin, cout, k, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None


LORA_PAIR_FUSION = os.environ.get("FZ_NO_LORA_PAIR") is None   # A/B switches (scripts/r04v.sh)
LORA_PAIR_GN = os.environ.get("FZ_NO_LORA_PAIR_GN") is None
LORA_PAIR_ALWAYS = os.environ.get("FZ_LORA_PAIR_ALWAYS") is not None  # tests: every shape the launch carries, preferred or not


def temporal_conv_native_ok(cin, cout):
    """Shapes fz_temporal_conv3 carries: the MFMA implicit GEMM (cin % 8 == 0, cout >= 8) or the direct kernel (both <= 8)."""
    return (cin % 8 == 0 and cout >= 8) or (cin <= 8 and cout <= 8)


def pack_temporal_weight(w, dtype, device):
    """nn.Conv1d weight [Cout, Cin, 3] -> [Cout][3][Cin] (K = (tap, Cin) contiguous per output channel, csrc/igemm.hip)."""
    return w.detach().permute(0, 2, 1).to(device=device, dtype=dtype).contiguous()


def temporal_conv_tokens(x4, w_native, *, rows_add=None, residual=None, residual2=None, gn_groups=0):
    """x4: [B, F, T, Cin] token-major; w_native: [Cout][3][Cin]; zero padded 'same' k=3 convolution over F on the native kernel
    (fz_temporal_conv3).  rows_add: [B, Cout] added per batch element (time embedding and / or the Conv1d bias); residuals
    [B*F, T, Cout].  Returns [B, F, T, Cout]."""
    b, f, t, cin = x4.shape
    cout = w_native.shape[0]
    if not temporal_conv_native_ok(cin, cout):
        raise NotImplementedError(f"temporal convolution {cin} -> {cout}: fz_temporal_conv3 needs cin % 8 == 0 (or <= 8 channels)")
    x3 = x4.reshape(b * f, t, cin)
    if not x3.is_contiguous():
        x3 = x3.contiguous()
    if gn_groups > 0:  # (y, partial or None): the GroupNorm statistics of y out of the launch's epilogue (fz_temporal_conv3_gn)
        y, part = K.temporal_conv3(x3, w_native, clip_len=f, res=residual, res2=residual2, temb=rows_add, gn_groups=gn_groups)
        return y.view(b, f, t, cout), part
    y = K.temporal_conv3(x3, w_native, clip_len=f, res=residual, res2=residual2, temb=rows_add)
    return y.view(b, f, t, cout)


class LoRALinearLayer(nn.Module):
    def __init__(self, in_features, out_features, rank=4, stride=1):
        super().__init__()
        if rank > min(in_features, out_features):
            rank = min(in_features, out_features) // 2  # lora.py:26-28
        assert stride == 1, "temporal_downsample is not used by any shipped config (SURVEY §8a-12)"
        self.down = _Conv1dParams(in_features, rank, 3)
        self.up = _Conv1dParams(rank, out_features, 3)
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)
        self._packed = None

    def _pack(self, dtype, device):
        if self._packed is None or self._packed[0].dtype != dtype or self._packed[0].device != device:
            if dtype != torch.float16:
                raise RuntimeError("the MI355X engine computes in fp16 storage / fp32 accumulation: call unet.half()")
            is_noop = bool((self.up.weight == 0).all())  # un-tuned SD: up == 0 -> exact identity
            # [Cout][3][Cin] packing of fz_temporal_conv3: the MFMA implicit GEMM for the rank-160 pairs, its direct kernel for
            # conv_out's rank-2 pair (4 -> 2 -> 4 channels)
            self._packed = (pack_temporal_weight(self.down.weight, dtype, device),
                            pack_temporal_weight(self.up.weight, dtype, device), is_noop)
        return self._packed

    def is_noop(self, dtype, device):
        return self._pack(dtype, device)[2]

    def forward_tokens(self, x4, temb=None, residual=None, gn_groups=0):
        """x4: [B, F, T, C] -> up(down(x)) + x (+ temb[b] broadcast) (+ residual), same shape.  gn_groups > 0 (single-GPU path): returns
        (y, partial or None) -- the GroupNorm statistics partials of y out of the up convolution's epilogue."""
        wdn, wun, is_noop = self._pack(x4.dtype, x4.device)
        b, f, t, c = x4.shape
        if is_noop:
            y = x4
            if temb is not None:
                y = y + temb[:, None, None, :]
            if residual is not None:
                y = y + residual.view(b, f, t, c)
            return y
        shard = D.active_shard()
        if shard is not None:
            # the clip's frames are split over ranks: ONE exchange -- a two-frame halo of x from both neighbours (zeros beyond the clip
            # ends, the conv's own padding).  down() over the extended clip yields down(x) of the own frames AND of the one-frame halo
            # the up convolution needs (recomputed from the neighbours' x instead of exchanged a second time: the same fp16 arithmetic
            # the owner runs); halo frames that lie OUTSIDE the clip are the up convolution's zero padding, not down(zeros | x)
            x_ext = shard.with_halo(x4, 2, 2, zero_outside=True, tag="temporal_conv")
            d_ext = self._conv(x_ext, wdn)[:, 1:-1]
            if shard.f0 == 0 or shard.f1 == shard.clip_len:
                d_ext = d_ext.clone()
                if shard.f0 == 0:
                    d_ext[:, 0] = 0
                if shard.f1 == shard.clip_len:
                    d_ext[:, -1] = 0
            y = self._conv(d_ext.contiguous(), wun)[:, 1:-1] + x4
            if temb is not None:
                y = y + temb[:, None, None, :]
            if residual is not None:
                y = y + residual.view(b, f, t, c)
            return y
        if LORA_PAIR_FUSION and (K.lora_pair_preferred(b * f, t, c, wdn.shape[0], f)
                                 or (LORA_PAIR_ALWAYS and K.lora_pair_ok(b * f, t, c, wdn.shape[0], f))):
            # both convolutions in one launch (fz_lora_pair) where that is the faster form -- the 64^2 level --: the rank-160
            # intermediate never leaves the workgroup's LDS
            x3 = x4.reshape(b * f, t, c)
            if not x3.is_contiguous():
                x3 = x3.contiguous()
            if gn_groups > 0 and LORA_PAIR_GN:   # + the GroupNorm partials of y out of the same launch (fz_lora_pair_gn)
                y, part = K.lora_pair(x3, wdn, wun, clip_len=f, res2=residual, temb=temb, gn_groups=gn_groups)
                return y.view(b, f, t, c), part
            y = K.lora_pair(x3, wdn, wun, clip_len=f, res2=residual, temb=temb).view(b, f, t, c)
            return (y, None) if gn_groups > 0 else y
        d = temporal_conv_tokens(x4, wdn)
        return temporal_conv_tokens(d, wun, rows_add=temb, residual=x4.reshape(b * f, t, c), residual2=residual, gn_groups=gn_groups)

    @staticmethod
    def _conv(x_ext, w_native):
        """k=3 temporal conv of a haloed clip [B, F+2, T, C] (the extended clip is its own zero-padded sequence; the halo
        frames' outputs are dropped by the caller)."""
        return temporal_conv_tokens(x_ext, w_native)
