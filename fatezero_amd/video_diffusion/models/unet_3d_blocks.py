"""Down / mid / up blocks (reference: video_diffusion/models/unet_3d_blocks.py), token-major engine.
Module and parameter names follow the reference so state_dicts are interchangeable."""
import torch
from torch import nn

from .attention import SpatioTemporalTransformerModel
from .resnet import DownsamplePseudo3D, ResnetBlockPseudo3D, Tokens, UpsamplePseudo3D


def _resnet(cin, cout, temb, eps, groups, scale, mc):
    return ResnetBlockPseudo3D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups,
                               output_scale_factor=scale, model_config=mc)


def _transformer(heads, channels, cross_dim, groups, mc):
    # attn_num_head_channels is used as the head COUNT, dim_head = channels // heads (unet_3d_blocks.py:269-272)
    return SpatioTemporalTransformerModel(heads, channels // heads, in_channels=channels, num_layers=1,
                                          cross_attention_dim=cross_dim, norm_num_groups=groups, model_config=mc)


class CrossAttnDownBlockPseudo3D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0, add_downsample=True,
                 model_config: dict = {}, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([_resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels,
                                              resnet_eps, resnet_groups, output_scale_factor, model_config)
                                      for i in range(num_layers)])
        self.attentions = nn.ModuleList([_transformer(attn_num_head_channels, out_channels, cross_attention_dim,
                                                      resnet_groups, model_config) for _ in range(num_layers)])
        self.downsamplers = None
        if add_downsample:
            self.downsamplers = nn.ModuleList([DownsamplePseudo3D(out_channels, use_conv=True, out_channels=out_channels,
                                                                  padding=1, name="op", model_config=model_config)])

    def forward_tokens(self, x: Tokens, temb_act, ctx):
        outs = []
        for resnet, attn in zip(self.resnets, self.attentions):
            x = attn.forward_tokens(resnet.forward_tokens(x, temb_act), ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].forward_tokens(x)
            outs.append(x)
        return x, outs


class DownBlockPseudo3D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 output_scale_factor=1.0, add_downsample=True, model_config: dict = {}, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([_resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels,
                                              resnet_eps, resnet_groups, output_scale_factor, model_config)
                                      for i in range(num_layers)])
        self.downsamplers = None
        if add_downsample:
            self.downsamplers = nn.ModuleList([DownsamplePseudo3D(out_channels, use_conv=True, out_channels=out_channels,
                                                                  padding=1, name="op", model_config=model_config)])

    def forward_tokens(self, x: Tokens, temb_act, ctx=None):
        outs = []
        for resnet in self.resnets:
            x = resnet.forward_tokens(x, temb_act)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].forward_tokens(x)
            outs.append(x)
        return x, outs


class UNetMidBlockPseudo3DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1,
                 output_scale_factor=1.0, cross_attention_dim=1280, model_config: dict = {}, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([_resnet(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups,
                                              output_scale_factor, model_config) for _ in range(2)])
        self.attentions = nn.ModuleList([_transformer(attn_num_head_channels, in_channels, cross_attention_dim,
                                                      resnet_groups, model_config)])

    def forward_tokens(self, x: Tokens, temb_act, ctx):
        x = self.resnets[0].forward_tokens(x, temb_act)
        x = self.attentions[0].forward_tokens(x, ctx)
        return self.resnets[1].forward_tokens(x, temb_act)


def _cat_skip(x: Tokens, skip: Tokens) -> Tokens:
    return Tokens.cat(x, skip)  # lazy: the resnet reads the two halves in place (resnet.py: CatTokens)


class CrossAttnUpBlockPseudo3D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0,
                 add_upsample=True, model_config: dict = {}, **unused):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip_c = in_channels if (i == num_layers - 1) else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            res.append(_resnet(rin + skip_c, out_channels, temb_channels, resnet_eps, resnet_groups, output_scale_factor,
                               model_config))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList([_transformer(attn_num_head_channels, out_channels, cross_attention_dim,
                                                      resnet_groups, model_config) for _ in range(num_layers)])
        self.upsamplers = None
        if add_upsample:
            self.upsamplers = nn.ModuleList([UpsamplePseudo3D(out_channels, use_conv=True, out_channels=out_channels,
                                                              model_config=model_config)])

    def forward_tokens(self, x: Tokens, skips, temb_act, ctx):
        for resnet, attn in zip(self.resnets, self.attentions):
            x = resnet.forward_tokens(_cat_skip(x, skips.pop()), temb_act)
            x = attn.forward_tokens(x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0].forward_tokens(x)
        return x


class UpBlockPseudo3D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, output_scale_factor=1.0, add_upsample=True, model_config: dict = {}, **unused):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip_c = in_channels if (i == num_layers - 1) else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            res.append(_resnet(rin + skip_c, out_channels, temb_channels, resnet_eps, resnet_groups, output_scale_factor,
                               model_config))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = None
        if add_upsample:
            self.upsamplers = nn.ModuleList([UpsamplePseudo3D(out_channels, use_conv=True, out_channels=out_channels,
                                                              model_config=model_config)])

    def forward_tokens(self, x: Tokens, skips, temb_act, ctx=None):
        for resnet in self.resnets:
            x = resnet.forward_tokens(_cat_skip(x, skips.pop()), temb_act)
        if self.upsamplers is not None:
            x = self.upsamplers[0].forward_tokens(x)
        return x


_DOWN = {"CrossAttnDownBlockPseudo3D": CrossAttnDownBlockPseudo3D, "DownBlockPseudo3D": DownBlockPseudo3D}
_UP = {"CrossAttnUpBlockPseudo3D": CrossAttnUpBlockPseudo3D, "UpBlockPseudo3D": UpBlockPseudo3D}


def get_down_block(down_block_type, **kw):
    down_block_type = down_block_type[7:] if down_block_type.startswith("UNetRes") else down_block_type
    if down_block_type not in _DOWN:
        raise ValueError(f"{down_block_type} does not exist.")
    if down_block_type == "CrossAttnDownBlockPseudo3D" and kw.get("cross_attention_dim") is None:
        raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlockPseudo3D")
    return _DOWN[down_block_type](**kw)


def get_up_block(up_block_type, **kw):
    up_block_type = up_block_type[7:] if up_block_type.startswith("UNetRes") else up_block_type
    if up_block_type not in _UP:
        raise ValueError(f"{up_block_type} does not exist.")
    if up_block_type == "CrossAttnUpBlockPseudo3D" and kw.get("cross_attention_dim") is None:
        raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlockPseudo3D")
    return _UP[up_block_type](**kw)
