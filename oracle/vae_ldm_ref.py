"""TEST INFRASTRUCTURE -- a SECOND, independently structured statement of the Stable-Diffusion VAE: the ORIGINAL latent-diffusion
autoencoder (CompVis/latent-diffusion `ldm/modules/diffusionmodules/model.py`: Encoder / Decoder / ResnetBlock / AttnBlock / Downsample /
Upsample, `ldm/models/autoencoder.py`: AutoencoderKL with quant_conv / post_quant_conv), of which diffusers' AutoencoderKL is a port, on an
LDM-format state dict (`encoder.down.{i}.block.{j}`, `encoder.mid.attn_1.{q,k,v,proj_out}` as 1x1 convolutions, `decoder.up.{i}` indexed
from the LOWEST resolution upwards ...), plus the published key mapping between the two formats (diffusers
`scripts/convert_original_stable_diffusion_to_diffusers.py: convert_ldm_vae_checkpoint`, applied here in reverse).

Why it exists: neither diffusers nor an SD checkpoint can be had offline, so `oracle/vae_oracle.py` -- the restatement of
diffusers 0.11.1's AutoencoderKL that the native VAE is tested against -- had nothing to be checked against.  This file follows a DIFFERENT
published source with different module structure, key names, block order and attention arithmetic (convolutional q / k / v, one
C^-1/2 scale on the logits instead of C^-1/4 on q and on k); tests/test_vae_pin.py feeds both the same weights through the key mapping and
requires the same moments / images.  Still not a run of the third-party code: (f)-1 stays "parity unpinned" in that strict sense -- but a
transcription slip in either statement (a swapped block, a wrong pad, eps, scale, key) now fails a test.
"""
import torch
import torch.nn.functional as F

GN_GROUPS, GN_EPS = 32, 1e-6


def _norm(x, sd, p, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], GN_EPS)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(x, sd, p, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _resnet_block(x, sd, p, groups):
    """ldm ResnetBlock(temb_channels=0): norm1 -> swish -> conv1 -> norm2 -> swish -> (dropout 0) -> conv2; `nin_shortcut` 1x1 when the width changes."""
    h = _conv(_swish(_norm(x, sd, p + ".norm1", groups)), sd, p + ".conv1")
    h = _conv(_swish(_norm(h, sd, p + ".norm2", groups)), sd, p + ".conv2")
    if p + ".nin_shortcut.weight" in sd:
        x = _conv(x, sd, p + ".nin_shortcut", padding=0)
    return x + h


def _attn_block(x, sd, p, groups):
    """ldm AttnBlock: q, k, v, proj_out are 1x1 convolutions; w = softmax(q^T k * C^-1/2) over the key axis; h = v w^T."""
    h = _norm(x, sd, p + ".norm", groups)
    q, k, v = (_conv(h, sd, p + "." + n, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)          # b, hw, c
    k = k.reshape(b, c, hh * ww)                            # b, c, hw
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))               # b, hw(query), hw(key)
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(h, sd, p + ".proj_out", padding=0)


def encode_moments(sd, ch_mult_len, num_res_blocks, x, groups=GN_GROUPS):
    h = _conv(x, sd, "encoder.conv_in")
    for i in range(ch_mult_len):
        for j in range(num_res_blocks):
            h = _resnet_block(h, sd, f"encoder.down.{i}.block.{j}", groups)
        if i != ch_mult_len - 1:  # ldm Downsample(with_conv): pad (0, 1, 0, 1) with zeros, conv stride 2 padding 0
            h = _conv(F.pad(h, (0, 1, 0, 1), mode="constant", value=0), sd, f"encoder.down.{i}.downsample.conv", stride=2, padding=0)
    h = _resnet_block(h, sd, "encoder.mid.block_1", groups)
    h = _attn_block(h, sd, "encoder.mid.attn_1", groups)
    h = _resnet_block(h, sd, "encoder.mid.block_2", groups)
    h = _conv(_swish(_norm(h, sd, "encoder.norm_out", groups)), sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def decode(sd, ch_mult_len, num_res_blocks, z, groups=GN_GROUPS):
    h = _conv(_conv(z, sd, "post_quant_conv", padding=0), sd, "decoder.conv_in")
    h = _resnet_block(h, sd, "decoder.mid.block_1", groups)
    h = _attn_block(h, sd, "decoder.mid.attn_1", groups)
    h = _resnet_block(h, sd, "decoder.mid.block_2", groups)
    for i in reversed(range(ch_mult_len)):  # ldm walks `up` from the highest index (lowest resolution) down to 0
        for j in range(num_res_blocks + 1):
            h = _resnet_block(h, sd, f"decoder.up.{i}.block.{j}", groups)
        if i != 0:  # ldm Upsample(with_conv): nearest 2x, conv 3x3
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"decoder.up.{i}.upsample.conv")
    return _conv(_swish(_norm(h, sd, "decoder.norm_out", groups)), sd, "decoder.conv_out")


def diffusers_to_ldm(sd, n_blocks, layers_per_block):
    """The published diffusers <- ldm VAE key mapping (convert_ldm_vae_checkpoint), applied in reverse:
        encoder.down_blocks.{i}.resnets.{j}.X        <- encoder.down.{i}.block.{j}.X            (conv_shortcut <- nin_shortcut)
        encoder.down_blocks.{i}.downsamplers.0.conv  <- encoder.down.{i}.downsample.conv
        {enc,dec}.mid_block.resnets.{0,1}            <- {enc,dec}.mid.block_{1,2}
        {enc,dec}.mid_block.attentions.0.{group_norm, query, key, value, proj_attn} <- mid.attn_1.{norm, q, k, v, proj_out}
                                                        (Linear [C, C]  <-  1x1 convolution [C, C, 1, 1])
        decoder.up_blocks.{i}.resnets.{j}            <- decoder.up.{n - 1 - i}.block.{j}
        decoder.up_blocks.{i}.upsamplers.0.conv      <- decoder.up.{n - 1 - i}.upsample.conv
        {enc,dec}.conv_norm_out                      <- {enc,dec}.norm_out;  conv_in, conv_out, quant_conv, post_quant_conv unchanged."""
    out, used = {}, set()

    def take(src, dst, conv1x1=False):
        for suffix in (".weight", ".bias"):
            t = sd[src + suffix]
            used.add(src + suffix)
            out[dst + suffix] = t[:, :, None, None].clone() if (conv1x1 and suffix == ".weight") else t.clone()

    def take_resnet(src, dst):
        for n in ("norm1", "conv1", "norm2", "conv2"):
            take(f"{src}.{n}", f"{dst}.{n}")
        if src + ".conv_shortcut.weight" in sd:
            take(src + ".conv_shortcut", dst + ".nin_shortcut")

    for side in ("encoder", "decoder"):
        take(f"{side}.conv_in", f"{side}.conv_in")
        take(f"{side}.conv_out", f"{side}.conv_out")
        take(f"{side}.conv_norm_out", f"{side}.norm_out")
        take_resnet(f"{side}.mid_block.resnets.0", f"{side}.mid.block_1")
        take_resnet(f"{side}.mid_block.resnets.1", f"{side}.mid.block_2")
        a = f"{side}.mid_block.attentions.0"
        take(a + ".group_norm", f"{side}.mid.attn_1.norm")
        for d, l in (("query", "q"), ("key", "k"), ("value", "v"), ("proj_attn", "proj_out")):
            take(f"{a}.{d}", f"{side}.mid.attn_1.{l}", conv1x1=True)
    for i in range(n_blocks):
        for j in range(layers_per_block):
            take_resnet(f"encoder.down_blocks.{i}.resnets.{j}", f"encoder.down.{i}.block.{j}")
        if i != n_blocks - 1:
            take(f"encoder.down_blocks.{i}.downsamplers.0.conv", f"encoder.down.{i}.downsample.conv")
        for j in range(layers_per_block + 1):
            take_resnet(f"decoder.up_blocks.{i}.resnets.{j}", f"decoder.up.{n_blocks - 1 - i}.block.{j}")
        if i != n_blocks - 1:
            take(f"decoder.up_blocks.{i}.upsamplers.0.conv", f"decoder.up.{n_blocks - 1 - i}.upsample.conv")
    take("quant_conv", "quant_conv")
    take("post_quant_conv", "post_quant_conv")
    missing = sorted(set(sd) - used)
    assert not missing, f"diffusers-format keys the mapping does not know: {missing[:5]}"
    return out


def sd_v1_vae_key_shapes():
    """Every tensor of the Stable-Diffusion v1.x VAE checkpoint (`vae/diffusion_pytorch_model.bin`, diffusers format) by NAME and SHAPE,
    written out from the published architecture -- `vae/config.json`: block_out_channels [128, 256, 512, 512], layers_per_block 2,
    latent_channels 4, norm_num_groups 32 -- and diffusers 0.11.1's module tree, NOT read off the native model: 248 tensors."""
    ks = {}

    def conv(p, cin, cout, k):
        ks[p + ".weight"], ks[p + ".bias"] = (cout, cin, k, k), (cout,)

    def norm(p, c):
        ks[p + ".weight"], ks[p + ".bias"] = (c,), (c,)

    def lin(p, cin, cout):
        ks[p + ".weight"], ks[p + ".bias"] = (cout, cin), (cout,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    def mid(p, c):
        resnet(p + ".resnets.0", c, c)
        resnet(p + ".resnets.1", c, c)
        a = p + ".attentions.0"
        norm(a + ".group_norm", c)
        for n in ("query", "key", "value", "proj_attn"):
            lin(a + "." + n, c, c)

    ch = [128, 256, 512, 512]
    conv("encoder.conv_in", 3, 128, 3)
    cin = 128
    for i, c in enumerate(ch):
        for j in range(2):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i != 3:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
    mid("encoder.mid_block", 512)
    norm("encoder.conv_norm_out", 512)
    conv("encoder.conv_out", 512, 8, 3)       # 2 x latent_channels: mean | logvar
    conv("quant_conv", 8, 8, 1)
    conv("post_quant_conv", 4, 4, 1)
    conv("decoder.conv_in", 4, 512, 3)
    mid("decoder.mid_block", 512)
    cin = 512
    for i, c in enumerate(reversed(ch)):      # 512, 512, 256, 128
        for j in range(3):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i != 3:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    norm("decoder.conv_norm_out", 128)
    conv("decoder.conv_out", 128, 3, 3)
    return ks
