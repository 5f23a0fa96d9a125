"""AutoencoderKL of Stable Diffusion 1.x on the MI355X engine.

The reference takes it from diffusers ([3P] `diffusers==0.11.1`, `models/vae.py` + `models/unet_2d_blocks.py` +
`models/resnet.py` + `models/attention.py:AttentionBlock`) and touches it in exactly two places:
`vae.encode(images).latent_dist.sample() * 0.18215` (video_diffusion/pipelines/p2p_ddim_spatial_temporal.py:88-96) and
`vae.decode(latents / 0.18215).sample` in chunks of 16 frames (video_diffusion/pipelines/stable_diffusion.py:297-319);
`test_fatezero.py:96-99` loads it with `AutoencoderKL.from_pretrained(path, subfolder="vae")`.  This module keeps that
surface -- class name, `from_pretrained`, `config`, `encode(...).latent_dist`, `decode(...).sample`, the state-dict key
names of the diffusers checkpoint -- and runs the network token-major in fp16 on the hand-written kernels: every 3x3 /
1x1 convolution and every Linear is the MFMA implicit-GEMM kernel (csrc/igemm.hip), GroupNorm(+SiLU) is csrc/norms.hip,
the single-head 512-wide mid-block attention is two fz_gemm launches per frame around fz_softmax_rows.

Layout plumbing that stays in PyTorch (copies, no arithmetic): NCHW <-> token-major at the two ends, zero-padding of the
3- / 4-channel inputs to the 8-channel granule of the kernels, the one-pixel shift that turns the kernels' symmetric
stride-2 convolution into diffusers' bottom/right-padded Downsample2D.
"""
import json
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn

from ... import kernels as K
from .resnet import Tokens, _LinearParams, _NormParams, group_norm_tokens


def _pad_channels(x: torch.Tensor, to: int) -> torch.Tensor:
    c = x.shape[-1]
    return x if c == to else torch.nn.functional.pad(x, (0, to - c))


class _Conv2d(nn.Module):
    """nn.Conv2d parameters ([Cout, Cin, k, k] + bias) applied to token-major fp16 data through fz_conv3x3 / fz_gemm."""

    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = nn.Parameter(torch.zeros(cout))
        self._packed = None

    def _pack(self, device):
        if self._packed is None or self._packed[0].device != device:
            w = self.weight.detach().to(device=device, dtype=torch.float16)
            cin8 = (self.cin + 7) // 8 * 8  # the kernels address K in 16-byte chunks: pad Cin with zero weights
            if cin8 != self.cin:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin8 - self.cin))
            w = w.reshape(self.cout, cin8).contiguous() if self.k == 1 else K.pack_conv3x3_weight(w)
            self._packed = (w, self.bias.detach().to(device=device, dtype=torch.float16), cin8)
        return self._packed

    def forward_tokens(self, x: Tokens, residual=None, upsample=False) -> Tokens:
        w, b, cin8 = self._pack(x.data.device)
        xin = _pad_channels(x.data, cin8).contiguous()
        if self.k == 1:
            return x.like(K.gemm(xin, w, b, res=residual))
        if self.stride == 1:
            y, (oh, ow) = K.conv3x3(xin, w, b, hw=(x.h, x.w), upsample=upsample, res=residual)
            return x.like(y, oh, ow)
        # diffusers Downsample2D(padding=0): pad (0,1,0,1), then a stride-2 conv WITHOUT padding, i.e. output (oy, ox) reads
        # input rows 2oy .. 2oy+2.  The kernel's stride-2 conv reads 2oy-1 .. 2oy+1 (symmetric padding of the UNet): run it on
        # the image shifted by one pixel (one zero row / column in front) and drop output row / column 0.
        n, _, c = xin.shape
        img = torch.nn.functional.pad(xin.view(n, x.h, x.w, c), (0, 0, 1, 0, 1, 0))
        y, (oh, ow) = K.conv3x3(img.reshape(n, (x.h + 1) * (x.w + 1), c), w, b, hw=(x.h + 1, x.w + 1), stride=2)
        y = y.view(n, oh, ow, self.cout)[:, 1:, 1:, :].reshape(n, (oh - 1) * (ow - 1), self.cout).contiguous()
        return x.like(y, oh - 1, ow - 1)


class ResnetBlock2D(nn.Module):
    """[3P] diffusers ResnetBlock2D without time embedding: GN -> SiLU -> conv -> GN -> SiLU -> conv (+ 1x1 shortcut)."""

    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = _NormParams(cin, groups, eps)
        self.conv1 = _Conv2d(cin, cout, 3)
        self.norm2 = _NormParams(cout, groups, eps)
        self.conv2 = _Conv2d(cout, cout, 3)
        self.conv_shortcut = _Conv2d(cin, cout, 1) if cin != cout else None

    def forward_tokens(self, x: Tokens) -> Tokens:
        h = group_norm_tokens(self.norm1, x, span_frames=False, silu=True)
        h = self.conv1.forward_tokens(h)
        h = group_norm_tokens(self.norm2, h, span_frames=False, silu=True)
        skip = x if self.conv_shortcut is None else self.conv_shortcut.forward_tokens(x)
        return self.conv2.forward_tokens(h, residual=skip.data)  # residual add in the conv epilogue


class AttentionBlock(nn.Module):
    """[3P] diffusers 0.11.1 AttentionBlock (one head of `channels`): x + proj(softmax(q k^T / sqrt(c)) v), q/k/v = Linear(GN(x))."""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.channels = channels
        self.group_norm = _NormParams(channels, groups, eps)
        self.query = _LinearParams(channels, channels)
        self.key = _LinearParams(channels, channels)
        self.value = _LinearParams(channels, channels)
        self.proj_attn = _LinearParams(channels, channels)
        self._qk = None

    def forward_tokens(self, x: Tokens) -> Tokens:
        n, l, c = x.data.shape
        h = group_norm_tokens(self.group_norm, x, span_frames=False, silu=False).data
        dev = h.device
        if self._qk is None or self._qk[0].device != dev:
            # diffusers scales q AND k by c^-1/4 (scores then need no factor and stay in fp16 range): fold it into the weights
            s = float(c) ** -0.25
            wq, bq = self.query.weight.detach().float() * s, self.query.bias.detach().float() * s
            wk, bk = self.key.weight.detach().float() * s, self.key.bias.detach().float() * s
            self._qk = (torch.cat([wq, wk], 0).to(device=dev, dtype=torch.float16).contiguous(),
                        torch.cat([bq, bk], 0).to(device=dev, dtype=torch.float16).contiguous())
        qk = K.gemm(h, self._qk[0], self._qk[1])                      # [n, l, 2c]
        lp = (l + 7) // 8 * 8
        wv, bv = self.value.packed(torch.float16, dev)
        vt = K.gemm_vt(h, wv, lp)                                     # V^T [n, c, lp] straight out of the GEMM (bias added below)
        out = torch.empty(n, l, c, dtype=torch.float16, device=dev)
        # three launches for a chunk of frames (batched q k^T, row softmax, batched P V) instead of three per frame; the [l, l]
        # score matrix of a 512^2 frame is 32 MB in fp16: chunks of at most 8 frames (2^27 score elements) keep scores +
        # probabilities at 0.5 GB (+ the -inf padded copy when l is not a multiple of 8)
        step = max(1, min(n, (1 << 27) // max(1, l * lp)))
        for i0 in range(0, n, step):
            i1 = min(n, i0 + step)
            s = K.gemm_batched(qk[i0:i1, :, :c], qk[i0:i1, :, c:])     # scores [f, l, l] = q k^T
            if lp != l:
                s = torch.nn.functional.pad(s, (0, lp - l), value=float("-inf"))
            p = K.softmax_rows(s.view(-1, lp)).view(i1 - i0, l, lp)
            K.gemm_batched(p, vt[i0:i1], out=out[i0:i1])                # P V
        # softmax rows sum to one, so the value bias passes through the attention unchanged: add it here
        out = out + bv
        return x.like(self.proj_attn.apply(out, res=x.data))

    def load_state_dict(self, *a, **k):
        self._qk = None
        return super().load_state_dict(*a, **k)


class _Downsampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv2d(c, c, 3, stride=2)

    def forward_tokens(self, x):
        return self.conv.forward_tokens(x)


class _Upsampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv2d(c, c, 3)

    def forward_tokens(self, x):
        return self.conv.forward_tokens(x, upsample=True)  # nearest 2x folded into the conv's addressing


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Downsampler(cout)]) if add_downsample else None

    def forward_tokens(self, x):
        for r in self.resnets:
            x = r.forward_tokens(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].forward_tokens(x)
        return x


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([_Upsampler(cout)]) if add_upsample else None

    def forward_tokens(self, x):
        for r in self.resnets:
            x = r.forward_tokens(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].forward_tokens(x)
        return x


class _MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([AttentionBlock(c, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])

    def forward_tokens(self, x):
        x = self.resnets[0].forward_tokens(x)
        x = self.attentions[0].forward_tokens(x)
        return self.resnets[1].forward_tokens(x)


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        self.conv_in = _Conv2d(in_channels, block_out_channels[0], 3)
        blocks, c = [], block_out_channels[0]
        for i, co in enumerate(block_out_channels):
            blocks.append(_DownBlock(c, co, layers_per_block, groups, add_downsample=i != len(block_out_channels) - 1))
            c = co
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _MidBlock(c, groups)
        self.conv_norm_out = _NormParams(c, groups, 1e-6)
        self.conv_out = _Conv2d(c, 2 * out_channels, 3)

    def forward_tokens(self, x):
        x = self.conv_in.forward_tokens(x)
        for b in self.down_blocks:
            x = b.forward_tokens(x)
        x = self.mid_block.forward_tokens(x)
        x = group_norm_tokens(self.conv_norm_out, x, span_frames=False, silu=True)
        return self.conv_out.forward_tokens(x)


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = _Conv2d(in_channels, rev[0], 3)
        self.mid_block = _MidBlock(rev[0], groups)
        blocks, c = [], rev[0]
        for i, co in enumerate(rev):
            blocks.append(_UpBlock(c, co, layers_per_block + 1, groups, add_upsample=i != len(rev) - 1))
            c = co
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = _NormParams(c, groups, 1e-6)
        self.conv_out = _Conv2d(c, out_channels, 3)

    def forward_tokens(self, x):
        x = self.conv_in.forward_tokens(x)
        x = self.mid_block.forward_tokens(x)
        for b in self.up_blocks:
            x = b.forward_tokens(x)
        x = group_norm_tokens(self.conv_norm_out, x, span_frames=False, silu=True)
        return self.conv_out.forward_tokens(x)


class DiagonalGaussianDistribution:
    """[3P] diffusers DiagonalGaussianDistribution: parameters [N, 2 C, h, w] = (mean, logvar clamped to [-30, 20])."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        # the reference passes no generator (p2p_ddim_spatial_temporal.py:94): the global RNG of the parameters' device
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class AutoencoderKL(nn.Module):
    """Drop-in for `diffusers.AutoencoderKL` as the reference uses it (see the module docstring)."""

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, **unused):
        super().__init__()
        if act_fn not in ("silu", "swish"):
            raise NotImplementedError(f"act_fn={act_fn}: the SD-1.x VAE uses silu")
        if any(t != "DownEncoderBlock2D" for t in down_block_types) or any(t != "UpDecoderBlock2D" for t in up_block_types):
            raise NotImplementedError("only DownEncoderBlock2D / UpDecoderBlock2D (the SD-1.x VAE)")
        block_out_channels = tuple(block_out_channels)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, down_block_types=tuple(down_block_types),
                                      up_block_types=tuple(up_block_types), block_out_channels=block_out_channels,
                                      layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
                                      norm_num_groups=norm_num_groups, sample_size=sample_size)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = _Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = _Conv2d(latent_channels, latent_channels, 1)
        self.use_slicing = False

    # -- diffusers surface ---------------------------------------------------------------------------------
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    @classmethod
    def from_config(cls, config: dict):
        return cls(**{k: v for k, v in config.items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **unused):
        """`<path>[/<subfolder>]/config.json` + `diffusion_pytorch_model.{safetensors,bin}` (the diffusers layout)."""
        root = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(root, "config.json")) as f:
            model = cls.from_config(json.load(f))
        st = os.path.join(root, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(root, "diffusion_pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model.eval()

    _ATTN_KEY_MAP = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}

    def load_state_dict(self, state_dict, strict=True):
        # VAE folders saved by diffusers >= 0.15 name the mid-block attention to_q / to_k / to_v / to_out.0 (and older conversion
        # scripts stored its projections as 1x1 convolutions [C, C, 1, 1]); diffusers remaps both on load, so do we
        remapped = {}
        for k, v in state_dict.items():
            if ".attentions." in k:
                for new_name, old_name in self._ATTN_KEY_MAP.items():
                    if f".{new_name}." in k:
                        k = k.replace(f".{new_name}.", f".{old_name}.")
                        break
                if k.endswith(".weight") and v.dim() == 4 and v.shape[2:] == (1, 1):
                    v = v[:, :, 0, 0]
            remapped[k] = v
        state_dict = remapped
        for m in self.modules():
            if isinstance(m, (_Conv2d, _LinearParams, _NormParams)):
                m._packed = None
            if isinstance(m, AttentionBlock):
                m._qk = None
        return super().load_state_dict(state_dict, strict=strict)

    # -- the two calls of the reference --------------------------------------------------------------------
    @staticmethod
    def _tokens(x: torch.Tensor) -> Tokens:
        n, c, h, w = x.shape
        return Tokens(x.permute(0, 2, 3, 1).reshape(n, h * w, c).to(torch.float16).contiguous(), n, 1, h, w)

    @staticmethod
    def _image(t: Tokens, dtype) -> torch.Tensor:
        n, _, c = t.data.shape
        return t.data.view(n, t.h, t.w, c).permute(0, 3, 1, 2).to(dtype)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x: [N, 3, H, W] in [-1, 1] -> posterior over [N, latent_channels, H/8, W/8]."""
        chunks = x.split(1) if (self.use_slicing and x.shape[0] > 1) else [x]
        moments = []
        for c in chunks:
            h = self.encoder.forward_tokens(self._tokens(c))
            moments.append(self._image(self.quant_conv.forward_tokens(h), x.dtype))
        posterior = DiagonalGaussianDistribution(torch.cat(moments))
        return AutoencoderKLOutput(posterior) if return_dict else (posterior,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """z: [N, latent_channels, h, w] -> images [N, 3, 8h, 8w]."""
        chunks = z.split(1) if (self.use_slicing and z.shape[0] > 1) else [z]
        out = []
        for c in chunks:
            h = self.post_quant_conv.forward_tokens(self._tokens(c))
            out.append(self._image(self.decoder.forward_tokens(h), z.dtype))
        dec = torch.cat(out)
        return DecoderOutput(dec) if return_dict else (dec,)

    def forward(self, sample, sample_posterior=False, generator=None):
        post = self.encode(sample).latent_dist
        return self.decode(post.sample(generator) if sample_posterior else post.mode())
