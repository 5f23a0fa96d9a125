"""UNetPseudo3DConditionModel (reference: video_diffusion/models/unet_3d_condition.py), MI355X engine.

Public surface kept: constructor kwargs, `from_2d_model`, `load_2d_state_dict`, `forward(sample, timestep,
encoder_hidden_states).sample` on [B,4,F,H,W] latents, `.config`, state_dict key names.  Internally the forward
runs on token-major fp16 activations (`forward_tokens`) so the DDIM loops never convert layouts.
"""
import glob
import json
import math
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from .resnet import PseudoConv3d, Tokens, _LinearParams, _NormParams, group_norm_tokens
from .unet_3d_blocks import UNetMidBlockPseudo3DCrossAttn, get_down_block, get_up_block


@dataclass
class UNetPseudo3DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, k):  # the reference indexes ["sample"] (p2p_ddim_spatial_temporal.py:142)
        return getattr(self, k) if isinstance(k, str) else (self.sample,)[k]


class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = _LinearParams(cin, cout)
        self.linear_2 = _LinearParams(cout, cout)


TIME_EMBED_CACHE = True  # (switch for same-box A/B runs: False = the timestep enters as a device tensor at every forward, as before)


class UNetPseudo3DConditionModel(nn.Module):
    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 center_input_sample: bool = False, flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str] = ("CrossAttnDownBlockPseudo3D", "CrossAttnDownBlockPseudo3D",
                                                 "CrossAttnDownBlockPseudo3D", "DownBlockPseudo3D"),
                 mid_block_type: str = "UNetMidBlockPseudo3DCrossAttn",
                 up_block_types: Tuple[str] = ("UpBlockPseudo3D", "CrossAttnUpBlockPseudo3D", "CrossAttnUpBlockPseudo3D",
                                               "CrossAttnUpBlockPseudo3D"),
                 only_cross_attention=False, block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
                 layers_per_block: int = 2, downsample_padding: int = 1, mid_block_scale_factor: float = 1,
                 act_fn: str = "silu", norm_num_groups: int = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 1280,
                 attention_head_dim: Union[int, Tuple[int]] = 8, dual_cross_attention: bool = False,
                 use_linear_projection: bool = False, class_embed_type=None, num_class_embeds=None,
                 upcast_attention: bool = False, resnet_time_scale_shift: str = "default", **kwargs):
        super().__init__()
        if (center_input_sample or not flip_sin_to_cos or freq_shift != 0 or dual_cross_attention or use_linear_projection
                or class_embed_type is not None or num_class_embeds is not None or resnet_time_scale_shift != "default"
                or act_fn not in ("silu", "swish") or kwargs.get("temporal_downsample") or kwargs.get("temporal_downsample_time", 0)):
            raise NotImplementedError("only the SD-1.x configuration surface used by FateZero is implemented")
        cfg = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                   down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                   block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                   norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
                   attention_head_dim=attention_head_dim, center_input_sample=False, **kwargs)
        self.config = SimpleNamespace(**cfg)
        self.sample_size = sample_size
        model_config = dict(kwargs)
        time_embed_dim = block_out_channels[0] * 4
        self.conv_in = PseudoConv3d(in_channels, block_out_channels[0], kernel_size=3, padding=1, model_config=model_config)
        self.time_embedding = _TimestepEmbedding(block_out_channels[0], time_embed_dim)
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * len(down_block_types)
        self.down_blocks = nn.ModuleList()
        out_c = block_out_channels[0]
        for i, t in enumerate(down_block_types):
            in_c, out_c = out_c, block_out_channels[i]
            final = i == len(block_out_channels) - 1
            self.down_blocks.append(get_down_block(
                t, num_layers=layers_per_block, in_channels=in_c, out_channels=out_c, temb_channels=time_embed_dim,
                add_downsample=not final, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim, attn_num_head_channels=attention_head_dim[i],
                model_config=model_config))
        # the reference ignores `mid_block_type` (unet_3d_condition.py:167-181 always builds the cross-attention mid block), so a
        # 2-D config.json of a newer diffusers ("UNetMidBlock2DCrossAttn") loads there -- and here
        if only_cross_attention not in (False, None) or downsample_padding != 1:
            raise NotImplementedError("only_cross_attention / downsample_padding != 1 are not used by any SD-1.x checkpoint")
        self.mid_block = UNetMidBlockPseudo3DCrossAttn(
            in_channels=block_out_channels[-1], temb_channels=time_embed_dim, resnet_eps=norm_eps,
            output_scale_factor=mid_block_scale_factor, cross_attention_dim=cross_attention_dim,
            attn_num_head_channels=attention_head_dim[-1], resnet_groups=norm_num_groups, model_config=model_config)
        self.up_blocks = nn.ModuleList()
        rev_c = list(reversed(block_out_channels))
        rev_h = list(reversed(attention_head_dim))
        out_c = rev_c[0]
        self.num_upsamplers = 0
        for i, t in enumerate(up_block_types):
            final = i == len(block_out_channels) - 1
            prev, out_c = out_c, rev_c[i]
            in_c = rev_c[min(i + 1, len(block_out_channels) - 1)]
            if not final:
                self.num_upsamplers += 1
            self.up_blocks.append(get_up_block(
                t, num_layers=layers_per_block + 1, in_channels=in_c, out_channels=out_c, prev_output_channel=prev,
                temb_channels=time_embed_dim, add_upsample=not final, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim, attn_num_head_channels=rev_h[i], model_config=model_config))
        self.conv_norm_out = _NormParams(block_out_channels[0], norm_num_groups, norm_eps)
        self.conv_out = PseudoConv3d(block_out_channels[0], out_channels, kernel_size=3, padding=1, model_config=model_config)
        self._issuer = None  # fatezero_amd.issue.IssuePlans once enable_issue_plans() was called
        self._temb_cache, self._temb_freq = {}, {}

    # ------------------------------------------------------------------------------------------------------
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def invalidate_packed(self):
        self._temb_pack = None
        self._temb_cache, self._temb_freq = {}, {}
        if getattr(self, "_issuer", None) is not None:
            self._issuer.clear()  # recorded plans point at the packed weights
        for m in self.modules():
            for a in ("_packed", "_packed_up", "_qk", "_qkv", "_ctx_kv", "_ln_fold", "_chain"):
                if hasattr(m, a):
                    setattr(m, a, None)
            if hasattr(m, "_xchain"):
                m._xchain = {}

    def load_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def time_embed(self, timestep, batch, device):
        """Timesteps(C0, flip_sin_to_cos=True, shift 0) -> TimestepEmbedding [3P] (unet_3d_condition.py:338-362).
        Returns silu(emb) in fp16: every consumer (ResnetBlock.time_emb_proj) applies the non-linearity first."""
        c0 = self.config.block_out_channels[0]
        half = c0 // 2
        # The DDIM loops pass host integers (schedulers.py keeps `timesteps` on the host): a scalar that enters the device as a TENSOR is a
        # blocking host-to-device copy, i.e. a full stream synchronisation in front of every forward (the GPU drains, then idles until the
        # first launches of the forward are issued).  It enters as a kernel ARGUMENT instead (freq * float), and the result -- a function of
        # (timestep, weights) only -- is kept: the 50 timesteps of a job recur in every pass and every job.
        host_scalar = TIME_EMBED_CACHE and not (isinstance(timestep, torch.Tensor) and (timestep.is_cuda or timestep.numel() != 1))
        key = (float(timestep), batch, str(device)) if host_scalar else None
        if key is not None:
            hit = self._temb_cache.get(key)
            if hit is not None:
                return hit
        freq = self._temb_freq.get(str(device))
        if freq is None:
            freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=device) / half)
            self._temb_freq[str(device)] = freq
        if key is not None:
            e = (freq * key[0])[None].expand(batch, half)
        else:
            t = torch.as_tensor(timestep, dtype=torch.float32, device=device).reshape(-1).expand(batch)
            e = t[:, None] * freq[None]
        e = torch.cat([torch.cos(e), torch.sin(e)], dim=-1).to(torch.float16)
        e = self.time_embedding.linear_2.apply(F.silu(self.time_embedding.linear_1.apply(e)))
        e = F.silu(e)
        if key is not None:
            if len(self._temb_cache) >= 4096:
                self._temb_cache.clear()
            self._temb_cache[key] = e
        return e

    def _project_time_embeddings(self, temb_act):
        """All 22 `time_emb_proj` Linear layers of the ResNet blocks as ONE GEMM (they only depend on the timestep)."""
        from ... import kernels as K
        pk = getattr(self, "_temb_pack", None)
        if pk is None or pk[0].device != temb_act.device:
            from .resnet import ResnetBlockPseudo3D
            blocks = [m for m in self.modules() if isinstance(m, ResnetBlockPseudo3D)]
            w = torch.cat([b.time_emb_proj.weight.detach() for b in blocks], 0).to(device=temb_act.device, dtype=torch.float16)
            bias = torch.cat([b.time_emb_proj.bias.detach() for b in blocks], 0).to(device=temb_act.device, dtype=torch.float16)
            offs, o = [], 0
            for b in blocks:
                offs.append((o, b.out_channels))
                o += b.out_channels
            pk = (w.contiguous(), bias, blocks, offs)
            self._temb_pack = pk
        w, bias, blocks, offs = pk
        t_all = K.gemm(temb_act, w, bias)  # [B, sum Cout]
        for b, (o, c) in zip(blocks, offs):
            b._temb_cached = t_all[:, o:o + c]

    def forward_tokens(self, x: Tokens, timestep, ctx) -> Tokens:
        """x: latents as tokens [B*F, H*W, 4] fp16; ctx [B, 77, D] fp16 -> predicted noise, same layout."""
        from ... import kernels as K
        K.refresh_stream()
        temb_act = self.time_embed(timestep, x.b, x.data.device)
        ctx = ctx.to(torch.float16)
        issuer = self._issuer
        if issuer is None and os.environ.get("FZ_ISSUE_PLANS") == "1":
            issuer = self.enable_issue_plans()
        if issuer is not None:  # the launch list of this kind of forward, recorded earlier, re-issued from native code (fatezero_amd/issue.py)
            y = issuer.run(x, temb_act, ctx)
            if y is not None:
                return y
        return self._forward_body(x, temb_act, ctx)

    def enable_issue_plans(self, on=True, graph=None):
        """Native issue path: from its third occurrence on, a forward of a given kind (clip geometry, text context, controller kind) is
        replayed from a recorded launch plan instead of being walked in Python.  Off by default (FZ_ISSUE_PLANS=1 switches it on)."""
        if on and self._issuer is None:
            from ...issue import IssuePlans
            self._issuer = IssuePlans(self, graph=graph)
        elif not on:
            self._issuer = None
        return self._issuer

    def _forward_body(self, x: Tokens, temb_act, ctx) -> Tokens:
        """Everything of the forward that is the library's own launches: time-embedding projections, conv_in ... conv_out."""
        self._project_time_embeddings(temb_act)
        from .resnet import GN_FROM_EPILOGUE
        x = self.conv_in.forward_tokens(x, gn_groups=self.conv_norm_out.num_groups if GN_FROM_EPILOGUE else 0)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk.forward_tokens(x, temb_act, ctx)
            skips.extend(outs)
        x = self.mid_block.forward_tokens(x, temb_act, ctx)
        for blk in self.up_blocks:
            x = blk.forward_tokens(x, skips, temb_act, ctx)
        x = group_norm_tokens(self.conv_norm_out, x, span_frames=True, silu=True)
        return self.conv_out.forward_tokens(x)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                return_dict: bool = True):
        if class_labels is not None or attention_mask is not None:
            raise NotImplementedError
        factor = 2 ** self.num_upsamplers
        if any(s % factor != 0 for s in sample.shape[-2:]):
            raise ValueError(f"latent height/width must be multiples of {factor}")
        out_dtype = sample.dtype
        y = self.forward_tokens(Tokens.from_bcfhw(sample.to(torch.float16)), timestep, encoder_hidden_states)
        y = y.to_bcfhw().to(out_dtype)
        return UNetPseudo3DConditionOutput(sample=y) if return_dict else (y,)

    # ------------------------------------------------------------------------------------------------------
    _BLOCKS_2D_TO_3D = {"CrossAttnDownBlock2D": "CrossAttnDownBlockPseudo3D", "DownBlock2D": "DownBlockPseudo3D",
                        "UpBlock2D": "UpBlockPseudo3D", "CrossAttnUpBlock2D": "CrossAttnUpBlockPseudo3D"}

    @classmethod
    def from_2d_model(cls, model_path, model_config):
        """Build the pseudo-3D UNet from a diffusers 2-D UNet folder (reference: unet_3d_condition.py:449-483): the 2-D
        `config.json` with the block types renamed and `model_config` merged in, then the 2-D weights if the folder holds any.
        Extension: a `*.safetensors` file is accepted as well as the reference's `*.bin`."""
        cfg_file = os.path.join(model_path, "config.json")
        if not os.path.isfile(cfg_file):
            raise RuntimeError(f"{cfg_file} does not exist")
        with open(cfg_file, "r") as f:
            config = {k: v for k, v in json.load(f).items() if k not in ("_class_name", "_diffusers_version")}
        for key in ("down_block_types", "up_block_types"):
            config[key] = [cls._BLOCKS_2D_TO_3D.get(name, name) for name in config[key]]
        config.update(model_config or {})
        model = cls(**config)
        weights = sorted(glob.glob(os.path.join(model_path, "*.bin"))) or sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
        if weights:
            if weights[0].endswith(".safetensors"):
                from safetensors.torch import load_file
                state = load_file(weights[0], device="cpu")
            else:
                state = torch.load(weights[0], map_location="cpu")
            model.load_2d_state_dict(state_dict=state)
        return model

    def load_2d_state_dict(self, state_dict, **kwargs):
        """unet_3d_condition.py:485-501: every 2-D key must exist with the same shape; every non-temporal 3-D key must
        be provided."""
        sd3 = self.state_dict()
        for k, v in state_dict.items():
            if k not in sd3:
                raise KeyError(f"2d state_dict key {k} does not exist in 3d model")
            if v.shape != sd3[k].shape:
                raise ValueError(f"state_dict shape mismatch, 2d {v.shape}, 3d {sd3[k].shape}")
        for k in sd3:
            if "_temporal" in k:
                continue
            if k not in state_dict:
                raise KeyError(f"3d state_dict key {k} does not exist in 2d model")
        sd3.update(state_dict)
        self.load_state_dict(sd3, **kwargs)
