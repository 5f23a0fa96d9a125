"""Temporal LoRA convolution (reference: video_diffusion/models/lora.py:22-54, LoRALinearLayer).

x[(b h w), c, f] -> up(down(x)) + x with two bias-free Conv1d(k=3, pad=1) over the frame axis.  On token-major
activations [B, F, HW, C] a k=3 temporal conv is three GEMMs over shifted frame views (zero padding at the clip
ends), accumulated in place -- no '(b h w) c f' rearrange is materialised.  Parameter names/shapes are the
reference's (`down.weight [r, C, 3]`, `up.weight [C, r, 3]`) so checkpoints load unchanged.
"""
import torch
from torch import nn

from ... import dist as D
from ... import kernels as K


class _Conv1dParams(nn.Module):
    def __init__(self, cin, cout, k, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None


def temporal_conv_tokens(x4, w_taps, bias=None, residual=None):
    """x4: [B, F, T, Cin]; w_taps: [k, Cin, Cout] (tap-major, transposed for GEMM); zero padded 'same' conv over F.
    Returns residual + conv(x) (+bias)."""
    b, f, t, cin = x4.shape
    k = w_taps.shape[0]
    half = k // 2
    cout = w_taps.shape[2]
    x2 = x4.reshape(b, f * t, cin)
    if residual is not None:
        y = torch.baddbmm(residual.reshape(b, f * t, cout), x2, w_taps[half].expand(b, cin, cout))
    else:
        y = torch.matmul(x2, w_taps[half])
    if bias is not None:
        y = y + bias
    y4 = y.view(b, f, t, cout)
    for tap in range(k):
        sh = tap - half  # output frame i takes input frame i + sh
        if sh == 0 or abs(sh) >= f:
            continue
        if sh < 0:
            src, dst = x4[:, : f + sh], y4[:, -sh:]
        else:
            src, dst = x4[:, sh:], y4[:, : f - sh]
        n = src.shape[1] * t
        dst2 = dst.reshape(b, n, cout)  # a view: frames are contiguous blocks
        dst2.baddbmm_(src.reshape(b, n, cin), w_taps[tap].expand(b, cin, cout))
    return y4


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
            dw, uw = self.down.weight.detach(), self.up.weight.detach()
            rank, cin = dw.shape[0], dw.shape[1]
            cout = uw.shape[0]
            is_noop = bool((self.up.weight == 0).all())  # un-tuned SD: up == 0 -> exact identity
            native = (rank % 8 == 0 and cin % 8 == 0 and dtype == torch.float16)
            # [Cout][3][Cin] packing of the implicit-GEMM kernel (fz_temporal_conv3) ...
            wdn = dw.permute(0, 2, 1).to(device=device, dtype=dtype).contiguous() if native else None
            wun = uw.permute(0, 2, 1).to(device=device, dtype=dtype).contiguous() if native else None
            # ... and tap-major [3, Cin, Cout] for conv_out's rank-2 LoRA (4 -> 2 -> 4 channels: below the 16-byte chunk of the
            # MFMA kernel; three [rows, 4] x [4, 2] products, only alive with a tuned checkpoint)
            wd = dw.to(device=device, dtype=dtype).permute(2, 1, 0).contiguous()
            wu = uw.to(device=device, dtype=dtype).permute(2, 1, 0).contiguous()
            self._packed = (wd, wu, is_noop, native, wdn, wun)
        return self._packed

    def is_noop(self, dtype, device):
        return self._pack(dtype, device)[2]

    def forward_tokens(self, x4, temb=None, residual=None):
        """x4: [B, F, T, C] -> up(down(x)) + x (+ temb[b] broadcast) (+ residual), same shape."""
        wd, wu, is_noop, native, wdn, wun = self._pack(x4.dtype, x4.device)
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
            # the clip's frames are split over ranks: one-frame halos from the neighbours (zeros beyond the clip ends, the
            # conv's own padding), first of x, then of down(x); the halo frames' outputs are dropped
            x_ext = shard.with_halo(x4, 1, 1, zero_outside=True)
            d = self._conv(x_ext, wd, wdn, native)[:, 1:-1].contiguous()
            d_ext = shard.with_halo(d, 1, 1, zero_outside=True)
            y = self._conv(d_ext, wu, wun, native)[:, 1:-1] + x4
            if temb is not None:
                y = y + temb[:, None, None, :]
            if residual is not None:
                y = y + residual.view(b, f, t, c)
            return y
        if native:
            x3 = x4.reshape(b * f, t, c)
            d = K.temporal_conv3(x3, wdn, clip_len=f)
            y = K.temporal_conv3(d, wun, clip_len=f, res=x3, res2=residual, temb=temb)
            return y.view(b, f, t, c)
        d = temporal_conv_tokens(x4, wd)
        y = temporal_conv_tokens(d, wu, residual=x4)
        if temb is not None:
            y = y + temb[:, None, None, :]
        if residual is not None:
            y = y + residual.view(b, f, t, c)
        return y

    @staticmethod
    def _conv(x_ext, w_taps, w_native, native):
        """k=3 temporal conv of a haloed clip [B, F+2, T, C] (the extended clip is its own zero-padded sequence; the halo
        frames' outputs are dropped by the caller)."""
        if not native:
            return temporal_conv_tokens(x_ext, w_taps)
        b, fe, t, c = x_ext.shape
        y = K.temporal_conv3(x_ext.reshape(b * fe, t, c).contiguous(), w_native, clip_len=fe)
        return y.view(b, fe, t, w_native.shape[0])
