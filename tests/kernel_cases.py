"""Kernel-level parity cases shared by the CPU-emulation suite (tests/test_kernels_emu.py) and the MI355X suite
(tests/test_kernels_gpu.py).  References are plain fp32 torch on CPU.  Tolerances are stated per check."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from fatezero_amd import kernels as K


def _frame_indices(index_list, clip):
    out = []
    for index in index_list:
        if isinstance(index, str):
            fr = {"first": 0, "last": clip - 1, "mid": (clip - 1) // 2}[index]
            out.append([fr] * clip)
        else:
            out.append([min(max(f + index, 0), clip - 1) for f in range(clip)])
    return out


def ref_self_attention(q, k, v, heads, clip, index_list):
    """fp32 restatement of spatial_temporal_forward + _attention (attention_register.py:131-218, :23-59).
    q,k,v: [N, L, C] float. Returns (O [N,L,C], P [N,heads,L,Lk])."""
    n, l, c = q.shape
    b = n // clip
    d = c // heads
    idx = _frame_indices(index_list, clip)
    if idx:
        k5, v5 = k.reshape(b, clip, l, c), v.reshape(b, clip, l, c)
        k = torch.cat([k5[:, fi] for fi in idx], dim=2).reshape(n, -1, c)
        v = torch.cat([v5[:, fi] for fi in idx], dim=2).reshape(n, -1, c)
    qh = q.reshape(n, l, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(n, -1, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(n, -1, heads, d).permute(0, 2, 1, 3)
    p = (qh @ kh.transpose(-1, -2) * d ** -0.5).softmax(-1)
    return p, vh


def _mk(shape, g, device, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).half().to(device)


def _vt(v, lp):
    return K.transpose_pad(v, lp)


def case_attn_self(device, *, batch, clip, heads, d, lq, index_list, mode, mask_kind=None, seed=0, qk_scale=1.5,
                   fold=False, shape=None):
    """fold=True: q is handed over in the log2 domain (desc.q_log2_scaled), as the host does for d % 16 != 0; the
    reference then sees exactly the same fp16 numbers divided by the folded factor."""
    g = torch.Generator().manual_seed(seed)
    n, c = batch * clip, heads * d
    q = _mk((n, lq, c), g, device, qk_scale)
    k = _mk((n, lq, c), g, device, qk_scale)
    if shape == "ramp":      # logits grow along the key axis: the running max moves in (almost) every tile
        k = (k.float() * torch.linspace(0.05, 3.0, lq, device=k.device)[None, :, None]).half()
    elif shape == "negative":  # every logit far below zero: the running max has to come DOWN on the first tile
        q, k = q.abs(), -k.abs()
    q_in, fkw = q, {}
    if fold:
        cs = d ** -0.5 * 1.4426950408889634
        q_in = (q.float() * cs).half()
        q = q_in.float() / cs
        fkw = dict(q_log2_scaled=True)
    v = _mk((n, lq, c), g, device)
    n_kv = max(1, len(index_list))
    lk = n_kv * lq
    vt = _vt(v, K.pad64(lq))
    out = torch.full((n, lq, c), float("nan"), dtype=torch.float16, device=device)
    p_ref, vh = ref_self_attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), heads, clip, index_list)
    res = {}
    if mode == K.FZ_ATTN_FLASH:
        K.attn_self(q_in, k, vt, out, clip_len=clip, heads=heads, index_list=index_list, mode=mode, **fkw)
        o_ref = (p_ref @ vh).permute(0, 2, 1, 3).reshape(n, lq, c)
    elif mode == K.FZ_ATTN_CAPTURE:
        p = torch.full((n, heads, lq, lk), float("nan"), dtype=torch.float16, device=device)
        K.attn_self(q_in, k, vt, out, clip_len=clip, heads=heads, index_list=index_list, mode=mode, p=p, **fkw)
        # the reference casts P to fp16 before P.V (attention_register.py:45,55)
        o_ref = (p_ref.half().float() @ vh).permute(0, 2, 1, 3).reshape(n, lq, c)
        pe = (p.float().cpu() - p_ref).abs()
        # stored P: within 1 fp16 ulp of the fp32 softmax cast to fp16 (+ fast-exp slack)
        ulp = torch.maximum(p_ref.abs() * 2.0 ** -10, torch.full_like(p_ref, 2.0 ** -24))
        res["p_max_err_ulps"] = float((pe / ulp).max())
        assert torch.isfinite(p.float()).all()
        assert float((pe / ulp).max()) <= 1.6, float((pe / ulp).max())
        rows = p.float().sum(-1)
        assert float((rows - 1).abs().max()) < 3e-3
    else:  # INJECT: frames of the second batch half take stored maps (cond half of the CFG batch)
        fcond = clip
        base = torch.rand(clip, heads, lq, lk, generator=g).softmax(-1).half().to(device)
        mask = None
        if mask_kind == "random":
            mask = (torch.rand(clip, lq, generator=g) > 0.5).float().to(device)
        elif mask_kind == "rows":
            mask = torch.zeros(clip, lq)
            mask[:, : lq // 3] = 1.0
            mask = mask.to(device)
        K.attn_self(q_in, k, vt, out, clip_len=clip, heads=heads, index_list=index_list, mode=K.FZ_ATTN_FLASH,
                    frame0=0, n_frames=n - fcond, **fkw)
        K.attn_self(q_in, (k if mask is not None else None), vt, out, clip_len=clip, heads=heads, index_list=index_list,
                    mode=mode, frame0=n - fcond, n_frames=fcond, p=base, row_mask=mask, **fkw)
        p_new = p_ref.clone()
        bs = base.float().cpu()
        if mask is None:
            p_new[n - fcond:] = bs
        else:
            m = mask.cpu()[:, None, :, None]
            p_new[n - fcond:] = m * p_ref[n - fcond:] + (1 - m) * bs  # attention_util.py:87
        o_ref = (p_new @ vh).permute(0, 2, 1, 3).reshape(n, lq, c)
    err = (out.float().cpu() - o_ref).abs().max().item()
    res["o_max_err"] = err
    assert torch.isfinite(out.float()).all(), "non-finite output"
    assert err < 4e-3 * max(1.0, float(o_ref.abs().max())), err  # fp16 output of an fp32-accumulated product
    return res


def ref_cross_fuse(base, cur, mapper, coef_a, coef_b):
    """new = (base @ M) * A + cur * B with base/cur [F,h,L,77], mapper [77,77]."""
    return (base @ mapper) * coef_a + cur * coef_b


def case_attn_cross(device, *, batch, clip, heads, d, lq, mode, seed=0, lk=77):
    g = torch.Generator().manual_seed(seed)
    n, c = batch * clip, heads * d
    q = _mk((n, lq, c), g, device, 1.5)
    k = _mk((batch, lk, c), g, device, 1.5)
    v = _mk((batch, lk, c), g, device)
    vt = K.transpose_pad(v, K.CROSS_KEYS)
    out = torch.full((n, lq, c), float("nan"), dtype=torch.float16, device=device)
    qf, kf, vf = q.float().cpu(), k.float().cpu(), v.float().cpu()
    qh = qf.reshape(n, lq, heads, d).permute(0, 2, 1, 3)
    kh = kf.repeat_interleave(clip, 0).reshape(n, lk, heads, d).permute(0, 2, 1, 3)
    vh = vf.repeat_interleave(clip, 0).reshape(n, lk, heads, d).permute(0, 2, 1, 3)
    p_ref = (qh @ kh.transpose(-1, -2) * d ** -0.5).softmax(-1)
    res = {}
    if mode == K.FZ_ATTN_FLASH:
        K.attn_cross(q, k, vt, out, clip_len=clip, heads=heads, lk=lk, mode=mode)
        o_ref = (p_ref.half().float() @ vh)
    elif mode == K.FZ_ATTN_CAPTURE:
        p = torch.full((n, heads, lq, K.CROSS_P_STRIDE), float("nan"), dtype=torch.float16, device=device)
        K.attn_cross(q, k, vt, out, clip_len=clip, heads=heads, lk=lk, mode=mode, p=p)
        o_ref = (p_ref.half().float() @ vh)
        pc = p.float().cpu()
        assert (pc[..., lk:] == 0).all(), "pad columns must be zero"
        ulp = torch.maximum(p_ref.abs() * 2.0 ** -10, torch.full_like(p_ref, 2.0 ** -24))
        res["p_max_err_ulps"] = float(((pc[..., :lk] - p_ref).abs() / ulp).max())
        assert res["p_max_err_ulps"] <= 1.6
    else:
        fcond = clip
        base = torch.zeros(clip, heads, lq, K.CROSS_P_STRIDE)
        base[..., :lk] = torch.rand(clip, heads, lq, lk, generator=g).softmax(-1)
        base = base.half().to(device)
        mapper = torch.zeros(lk, lk)
        perm = torch.randperm(lk, generator=g)
        mapper[torch.arange(lk), perm] = 1.0
        mapper[3, :] = 0
        mapper[3, 5] = 0.5
        mapper[3, 6] = 0.5
        coef = torch.zeros(2, K.CROSS_KEYS)
        coef[0, :lk] = (torch.rand(lk, generator=g) > 0.4).float() * torch.tensor([1.0, 10.0])[torch.randint(0, 2, (lk,), generator=g)]
        coef[1, :lk] = 1.0 - (coef[0, :lk] > 0).float() * 0.75
        mt = torch.zeros(K.CROSS_KEYS, K.CROSS_KEYS)
        mt[:lk, :lk] = mapper.t()
        cur_out = torch.full((clip, heads, lq, K.CROSS_P_STRIDE), float("nan"), dtype=torch.float16, device=device)
        K.attn_cross(q, k, vt, out, clip_len=clip, heads=heads, lk=lk, mode=K.FZ_ATTN_FLASH, frame0=0, n_frames=n - fcond)
        K.attn_cross(q, k, vt, out, clip_len=clip, heads=heads, lk=lk, mode=mode, frame0=n - fcond, n_frames=fcond,
                     p=base, mapper_t=mt.half().to(device), coef=coef.to(device), cur_out=cur_out)
        p_new = p_ref.clone()
        p_new[n - fcond:] = ref_cross_fuse(base.float().cpu()[..., :lk], p_ref[n - fcond:], mapper, coef[0, :lk], coef[1, :lk])
        o_ref = p_new.half().float() @ vh
        cc = cur_out.float().cpu()
        ulp = torch.maximum(p_ref[n - fcond:].abs() * 2.0 ** -10, torch.full_like(p_ref[n - fcond:], 2.0 ** -24))
        assert float(((cc[..., :lk] - p_ref[n - fcond:]).abs() / ulp).max()) <= 1.6
    o_ref = o_ref.permute(0, 2, 1, 3).reshape(n, lq, c)
    err = (out.float().cpu() - o_ref).abs().max().item()
    res["o_max_err"] = err
    assert torch.isfinite(out.float()).all()
    assert err < 6e-3 * max(1.0, float(o_ref.abs().max())), err
    return res


def case_attn_temporal(device, *, batch, clip, heads, d, tokens, seed=0):
    g = torch.Generator().manual_seed(seed)
    c = heads * d
    qkv = _mk((batch * clip, tokens, 3 * c), g, device)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    out = torch.full((batch * clip, tokens, c), float("nan"), dtype=torch.float16, device=device)
    K.attn_temporal(q, k, v, out, batch=batch, clip_len=clip, heads=heads)

    def r(t):  # '(b f) d c -> (b d) f c' then heads
        t = t.float().cpu().reshape(batch, clip, tokens, heads, d).permute(0, 2, 3, 1, 4)
        return t  # [b, tok, h, f, d]
    qh, kh, vh = r(q), r(k), r(v)
    p = (qh @ kh.transpose(-1, -2) * d ** -0.5).softmax(-1).half().float()
    o = (p @ vh).permute(0, 3, 1, 2, 4).reshape(batch * clip, tokens, c)
    err = (out.float().cpu() - o).abs().max().item()
    assert err < 4e-3 * max(1.0, float(o.abs().max())), err
    return {"o_max_err": err}


def case_groupnorm(device, *, n, span, tokens, c, groups, silu, eps=1e-5, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = ((torch.randn(n, tokens, c, generator=g) * 2 + torch.randn(1, 1, c, generator=g) * 3).half()).to(device)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).half().to(device)
    beta = (0.3 * torch.randn(c, generator=g)).half().to(device)
    y = K.groupnorm(x, gamma, beta, span=span, groups=groups, eps=eps, silu=silu)
    xr = x.float().cpu().reshape(n // span, span * tokens, c).permute(0, 2, 1)  # [b, C, f*tok]
    yr = F.group_norm(xr, groups, gamma.float().cpu(), beta.float().cpu(), eps)
    if silu:
        yr = F.silu(yr)
    yr = yr.permute(0, 2, 1).reshape(n, tokens, c)
    err = (y.float().cpu() - yr).abs().max().item()
    assert err < 2e-3 * max(1.0, float(yr.abs().max())), err
    return {"max_err": err}


def case_sharded_pieces(device, *, batch, clip, lo, hi, heads, d, tokens, groups, seed=0):
    """The kernel forms a frame-sharded clip uses, driven the way a rank owning frames [lo, hi) of the clip would drive them
    (fatezero_amd/dist.py), checked against the single-GPU kernels on the whole clip: split GroupNorm, sparse-causal
    attention on the extended K/V frame axis [left halo | own | right halo | anchors], temporal attention with gathered K/V."""
    g = torch.Generator().manual_seed(seed)
    c, fl, res = heads * d, hi - lo, {}

    def own(t):  # frames [lo, hi) of every batch element of a [(b f), ...] tensor
        return t.reshape(batch, clip, *t.shape[1:])[:, lo:hi].reshape(batch * fl, *t.shape[1:]).contiguous()

    # -- GroupNorm: partials of all frames (here computed locally, in clip order) + apply on the own frames ------
    x = ((torch.randn(batch * clip, tokens, c, generator=g) * 2 + 1).half()).to(device)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).half().to(device)
    beta = (0.3 * torch.randn(c, generator=g)).half().to(device)
    y_full = K.groupnorm(x, gamma, beta, span=clip, groups=groups, eps=1e-5, silu=True)
    part = K.groupnorm_stats(x, groups=groups)
    y_own = K.groupnorm_apply(own(x), gamma, beta, part.view(batch, clip, *part.shape[1:]).contiguous(), span=fl,
                              groups=groups, eps=1e-5, silu=True)
    # (fz_groupnorm runs its one-launch form where a group fits one workgroup -- exact two-sweep statistics instead of merged chunk
    # partials: the same numbers to a rounding of the fp32 statistics, i.e. at most an fp16 ulp of the output here and there)
    gn_err = float((y_own.float() - own(y_full).float()).abs().max())
    assert gn_err <= 2e-3 * max(1.0, float(y_full.float().abs().max())), gn_err
    # -- sparse-causal attention, index [-1, 'first', +1]: halos of one frame on both sides + anchor frame 0 -----
    qk = _mk((batch * clip, tokens, 2 * c), g, device, 1.5)
    q, k = qk[..., :c], qk[..., c:]
    v = _mk((batch * clip, tokens, c), g, device)
    vt = _vt(v, K.pad64(tokens))
    idx = [-1, "first", 1]
    o_full = torch.empty(batch * clip, tokens, c, dtype=torch.float16, device=device)
    K.attn_self(q, k, vt, o_full, clip_len=clip, heads=heads, index_list=idx, mode=K.FZ_ATTN_FLASH)

    def ext(t):
        t4 = t.reshape(batch, clip, *t.shape[1:])
        sel = [max(lo - 1, 0)] + list(range(lo, hi)) + [min(hi, clip - 1)] + [0]
        return t4[:, sel].reshape(batch * len(sel), *t.shape[1:]).contiguous()
    o_own = torch.empty(batch * fl, tokens, c, dtype=torch.float16, device=device)
    K.attn_self(own(q), ext(k), ext(vt), o_own, clip_len=fl, heads=heads, index_list=idx, mode=K.FZ_ATTN_FLASH,
                kv_slots_override=([0, 1, 0], [-1, fl + 2, 1]), kv_clip_len=fl + 3, kv_frame_off=1)
    assert torch.equal(o_own, own(o_full)), "extended K/V frame axis must address the same frames"
    p_full = torch.empty(batch * clip, heads, tokens, 3 * tokens, dtype=torch.float16, device=device)
    K.attn_self(q, k, vt, o_full, clip_len=clip, heads=heads, index_list=idx, mode=K.FZ_ATTN_CAPTURE, p=p_full)
    p_own = torch.empty(batch * fl, heads, tokens, 3 * tokens, dtype=torch.float16, device=device)
    K.attn_self(own(q), ext(k), ext(vt), o_own, clip_len=fl, heads=heads, index_list=idx, mode=K.FZ_ATTN_CAPTURE, p=p_own,
                kv_slots_override=([0, 1, 0], [-1, fl + 2, 1]), kv_clip_len=fl + 3, kv_frame_off=1)
    assert torch.equal(p_own, own(p_full)) and torch.equal(o_own, own(o_full))
    # -- temporal attention: own query frames against all frames' K/V ------------------------------------------
    qkv = _mk((batch * clip, tokens, 3 * c), g, device)
    t_full = torch.empty(batch * clip, tokens, c, dtype=torch.float16, device=device)
    K.attn_temporal(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], t_full, batch=batch, clip_len=clip, heads=heads)
    kv = qkv[..., c:].contiguous()
    t_own = torch.empty(batch * fl, tokens, c, dtype=torch.float16, device=device)
    K.attn_temporal(own(qkv)[..., :c], kv[..., :c], kv[..., c:], t_own, batch=batch, clip_len=fl, kv_frames=clip, heads=heads)
    assert torch.equal(t_own, own(t_full))
    return res


def case_layernorm(device, *, rows, c, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(rows, c, generator=g) * 2 + 1).half().to(device)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).half().to(device)
    beta = (0.3 * torch.randn(c, generator=g)).half().to(device)
    y = K.layernorm(x, gamma, beta)
    yr = F.layer_norm(x.float().cpu(), (c,), gamma.float().cpu(), beta.float().cpu())
    err = (y.float().cpu() - yr).abs().max().item()
    assert err < 2e-3 * max(1.0, float(yr.abs().max())), err
    return {"max_err": err}


def case_geglu(device, *, rows, inner, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(rows, 2 * inner, generator=g) * 2).half().to(device)
    y = K.geglu(x)
    xf = x.float().cpu()
    yr = xf[:, :inner] * F.gelu(xf[:, inner:])
    err = (y.float().cpu() - yr).abs().max().item()
    assert err < 2e-3 * max(1.0, float(yr.abs().max())), err
    return {"max_err": err}


def case_transpose_pad(device, *, n, l, c, lp, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, l, 2 * c, generator=g).half().to(device)[..., c:]  # strided view
    y = K.transpose_pad(x, lp)
    yr = torch.zeros(n, c, lp)
    yr[:, :, :l] = x.float().cpu().transpose(1, 2)
    assert torch.equal(y.float().cpu(), yr)
    return {}


def case_latent_update(device, *, frames, hw, blend, cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(4, frames, hw, generator=g)
    eu = torch.randn(frames, hw, 4, generator=g).half()
    ec = torch.randn(frames, hw, 4, generator=g).half()
    inv = torch.randn(4, frames, hw, generator=g)
    mask = (torch.rand(frames, hw, generator=g) > 0.5).float()
    cz, ce, gs = 1.0123, -0.0456, 7.5
    zd = z.clone().to(device)
    nxt = torch.empty(frames, hw, 4, dtype=torch.float16, device=device)
    K.latent_update(zd, eu.to(device) if cfg else None, ec.to(device), gs, cz, ce,
                    inv=inv.to(device) if blend else None, mask=mask.to(device) if blend else None, next_in=nxt)
    e = ec.float().permute(2, 0, 1)
    if cfg:
        e = eu.float().permute(2, 0, 1) + gs * (ec.float().permute(2, 0, 1) - eu.float().permute(2, 0, 1))
    zr = cz * z + ce * e
    if blend:
        zr = inv + mask[None] * (zr - inv)
    assert (zd.cpu() - zr).abs().max() < 1e-5
    assert (nxt.float().cpu().permute(2, 0, 1) - zr).abs().max() < 2e-3 * max(1.0, float(zr.abs().max()))
    return {}


def ref_blend_mask(maps, alpha, th, h, w, or_first):
    """spatial_blend.py:24-56 in fp32 torch. maps: list of [P,F,heads,r*r,77] float."""
    rr = []
    for item in maps:
        p, c, heads, r2, wd = item.shape
        res = int(r2 ** 0.5)
        rr.append(item.reshape(p, c, heads, res, res, wd).permute(0, 2, 1, 3, 4, 5))
    m = torch.cat(rr, dim=1)
    m = (m * alpha[:, None, None, None, None, :]).sum(-1).mean(1)
    m = F.max_pool2d(m, (3, 3), (1, 1), padding=(1, 1))
    mask = F.interpolate(m, size=(h, w))
    mask = mask / mask.max(-2, keepdim=True)[0].max(-1, keepdim=True)[0]
    mask = mask.gt(th)
    if or_first:
        mask = mask[:1] + mask
    return mask


def blob_maps(P_, F_, heads, res, g, lk=77, n_maps=5):
    """Structured cross-attention-like maps (Gaussian blobs per token) so masks are not degenerate."""
    out = []
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    for _ in range(n_maps):
        cx = torch.rand(P_, F_, 1, 1, lk, generator=g) * res
        cy = torch.rand(P_, F_, 1, 1, lk, generator=g) * res
        d2 = (xx.reshape(1, 1, 1, res * res, 1) - cx) ** 2 + (yy.reshape(1, 1, 1, res * res, 1) - cy) ** 2
        logits = torch.randn(P_, F_, heads, res * res, lk, generator=g) * 0.5 - d2 / (2 * (res / 4) ** 2)
        out.append(logits.softmax(-1))
    return out


def case_blend_mask(device, *, prompts, frames, heads, res, out_hw, or_first, th=0.3, seed=0):
    g = torch.Generator().manual_seed(seed)
    maps32 = blob_maps(prompts, frames, heads, res, g)
    dev_maps = []
    for m in maps32:
        buf = torch.zeros(prompts, frames, heads, res * res, K.CROSS_P_STRIDE, dtype=torch.float16)
        buf[..., :77] = m.half()
        dev_maps.append(buf.to(device))
    alpha = torch.zeros(prompts, 80)
    alpha[:, [2, 3]] = 1.0
    out = K.blend_mask(dev_maps, alpha.to(device), th, out_hw, or_with_first=or_first)
    ref = ref_blend_mask([b.float().cpu()[..., :77] for b in dev_maps], alpha[:, :77], th, out_hw[0], out_hw[1], or_first)
    diff = int((out.cpu().bool() != ref).sum())
    frac = float(ref.float().mean())
    assert 0.02 < frac < 0.98, f"degenerate mask ({frac})"
    assert diff == 0, f"{diff} differing mask elements"
    return {"ones": frac}


def case_conv3x3(device, *, n, h, w, cin, cout, stride=1, upsample=False, with_temb=False, with_res=False, fpb=1, seed=0,
                 tile_cfg=0, split_k=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h * w, cin, generator=g).half().to(device)
    wgt = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half()
    bias = (torch.randn(cout, generator=g) * 0.1).half().to(device)
    temb = (torch.randn(n // fpb, cout, generator=g)).half().to(device) if with_temb else None
    wt = K.pack_conv3x3_weight(wgt).to(device)
    xi = x.float().cpu().reshape(n, h, w, cin).permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    yr = F.conv2d(xi, wgt.float(), bias.float().cpu(), stride=stride, padding=1)
    ho, wo = yr.shape[2], yr.shape[3]
    res = torch.randn(n, ho * wo, cout, generator=g).half().to(device) if with_res else None
    y, (ho2, wo2) = K.conv3x3(x, wt, bias, hw=(h, w), stride=stride, upsample=upsample, temb=temb, frames_per_batch=fpb, res=res,
                              tile_cfg=tile_cfg, split_k=split_k)
    assert (ho2, wo2) == (ho, wo)
    yr = yr.permute(0, 2, 3, 1).reshape(n, ho * wo, cout)
    if with_temb:
        yr = yr + temb.float().cpu().repeat_interleave(fpb, 0)[:, None, :]
    if with_res:
        yr = yr + res.float().cpu()
    err = (y.float().cpu() - yr).abs().max().item()
    assert torch.isfinite(y.float()).all()
    assert err < 4e-3 * max(1.0, float(yr.abs().max())), err
    return {"max_err": err}


def case_conv3x3_up2(device, *, n, h, w, cin, cout, seed=0):
    """fz_conv3x3_up2 (nearest-2x + 3x3 convolution as four 2x2 convolutions on summed weights) against torch's interpolate + conv2d in fp32,
    and against fz_conv3x3(upsample=1) on the same operands (differs only by the fp16 rounding of the summed weights)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h * w, cin, generator=g).half().to(device)
    wgt = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half()
    bias = (torch.randn(cout, generator=g) * 0.1).half().to(device)
    wt = K.pack_conv3x3_weight(wgt).to(device)
    wup = K.pack_conv3x3_up2_weight(wt)
    # the packed weights against their definition
    w9 = wgt.float()
    rows = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}
    for z in range(4):
        for t in range(4):
            ref = sum(w9[:, :, ky, kx] for ky in rows[(z >> 1, t >> 1)] for kx in rows[(z & 1, t & 1)])
            assert torch.equal(wup[z, :, t, :].cpu(), ref.half()), (z, t)
    assert K.conv3x3_up2_ok(n, h, w, cin, cout)
    y, (ho, wo) = K.conv3x3_up2(x, wup, bias, hw=(h, w))
    assert (ho, wo) == (2 * h, 2 * w)
    xi = F.interpolate(x.float().cpu().reshape(n, h, w, cin).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    yr = F.conv2d(xi, wgt.float(), bias.float().cpu(), padding=1).permute(0, 2, 3, 1).reshape(n, ho * wo, cout)
    err = (y.float().cpu() - yr).abs().max().item()
    assert torch.isfinite(y.float()).all()
    assert err < 4e-3 * max(1.0, float(yr.abs().max())), err
    y9, _ = K.conv3x3(x, wt, bias, hw=(h, w), upsample=True)
    d9 = (y.float() - y9.float()).abs().max().item()
    assert d9 < 4e-3 * max(1.0, float(yr.abs().max())), d9
    return {"max_err": err, "vs_nine_taps": d9}


def case_groupnorm_cat(device, *, n, span, tokens, c1, c2, groups, silu=True, seed=0):
    """fz_groupnorm_cat == fz_groupnorm on the materialised torch.cat([x1, x2], channel): same kernels, same arithmetic -> bit-equal;
    and against torch's GroupNorm on the 5-D view (resnet.py:338 after unet_3d_blocks.py:384-395)."""
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(n, tokens, c1, generator=g).half().to(device)
    x2 = (torch.randn(n, tokens, c2, generator=g) * 2 + 0.5).half().to(device)
    gm = (1 + 0.2 * torch.randn(c1 + c2, generator=g)).half().to(device)
    bt = (0.2 * torch.randn(c1 + c2, generator=g)).half().to(device)
    y = K.groupnorm_cat(x1, x2, gm, bt, span=span, groups=groups, eps=1e-5, silu=silu)
    xc = torch.cat([x1, x2], -1).contiguous()
    assert torch.equal(y, K.groupnorm(xc, gm, bt, span=span, groups=groups, eps=1e-5, silu=silu))
    xr = xc.float().cpu().reshape(n // span, span * tokens, c1 + c2).permute(0, 2, 1)
    ref = F.group_norm(xr, groups, gm.float().cpu(), bt.float().cpu(), 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(n, tokens, c1 + c2)
    err = float((y.float().cpu() - ref).abs().max())
    assert err < 2e-2, err
    return {"max_err": err}


def case_temporal_conv3(device, *, batch, clip, tokens, cin, cout, with_res, seed=0, with_rows=False):
    """k=3 convolution over the frame axis (lora.py:31-54, resnet.py:42-55); with_rows: one row of cout values per batch element
    added to every output (the time embedding / the Conv1d bias riding in fz_temporal_conv3's `temb`)."""
    g = torch.Generator().manual_seed(seed)
    n = batch * clip
    x = torch.randn(n, tokens, cin, generator=g).half().to(device)
    w = (torch.randn(cout, cin, 3, generator=g) * (3 * cin) ** -0.5).half()
    res = torch.randn(n, tokens, cout, generator=g).half().to(device) if with_res else None
    rows = torch.randn(batch, cout, generator=g).half().to(device) if with_rows else None
    y = K.temporal_conv3(x, w.permute(0, 2, 1).contiguous().to(device), clip_len=clip, res=res, temb=rows)
    xr = x.float().cpu().reshape(batch, clip, tokens, cin).permute(0, 2, 3, 1).reshape(batch * tokens, cin, clip)
    yr = F.conv1d(xr, w.float(), None, padding=1).reshape(batch, tokens, cout, clip).permute(0, 3, 1, 2).reshape(n, tokens, cout)
    if with_res:
        yr = yr + res.float().cpu()
    if with_rows:
        yr = yr + rows.float().cpu().repeat_interleave(clip, 0)[:, None, :]
    err = (y.float().cpu() - yr).abs().max().item()
    assert err < 4e-3 * max(1.0, float(yr.abs().max())), err
    return {"max_err": err}


def case_gemm(device, *, rows, k, o, bias=True, n_res=0, geglu=False, ldx_extra=0, ldy_extra=0, tile_cfg=0, split_k=0, seed=0,
              lead=None):
    """fz_gemm vs fp32 torch: y = x @ w^T + bias (+ res) (+ res2), GEGLU epilogue, strided x / y views."""
    g = torch.Generator().manual_seed(seed)
    xfull = torch.randn(rows, k + ldx_extra, generator=g).half().to(device)
    x = xfull[:, ldx_extra:] if ldx_extra else xfull
    w = (torch.randn(o, k, generator=g) * k ** -0.5).half()
    b = (torch.randn(o, generator=g) * 0.3).half() if bias else None
    ow = o // 2 if geglu else o
    res = [torch.randn(rows, ow, generator=g).half().to(device) for _ in range(n_res)]
    ref = x.float().cpu() @ w.float().t()
    if bias:
        ref = ref + b.float()
    if geglu:
        ref = ref[:, :ow] * F.gelu(ref[:, ow:])
        wd, bd = K.pack_geglu(w, b)
    else:
        wd, bd = w, b
    for r in res:
        ref = ref + r.float().cpu()
    outfull = torch.full((rows, ow + ldy_extra), 7.0, dtype=torch.float16, device=device)
    out = outfull[:, :ow] if ldy_extra else outfull
    xin = x if lead is None else x.reshape(*lead, k)
    y = K.gemm(xin, wd.to(device), None if bd is None else bd.to(device), res=res[0] if n_res > 0 else None,
               res2=res[1] if n_res > 1 else None, out=out, geglu=geglu, tile_cfg=tile_cfg, split_k=split_k)
    err = (y.float().cpu() - ref).abs().max().item()
    assert torch.isfinite(y.float()).all()
    assert err < 4e-3 * max(1.0, float(ref.abs().max())), (err, float(ref.abs().max()))
    if ldy_extra:
        assert bool((outfull[:, ow:] == 7.0).all()), "wrote outside the output columns"
    return {"max_err": err}


def case_gemm_ln(device, *, rows, c, o, geglu=False, n_res=1, tile_cfg=0, seed=0, mean_shift=0.0):
    """LayerNorm fused around fz_gemm (fz_gemm_ln): a producer GEMM (out-projection + residual) that also emits the row
    statistics of what it stores, and a consumer GEMM that reads the RAW rows with the LayerNorm folded into its weights and
    epilogue -- against fp32 torch  LN(y1) @ W^T + b  on the fp16 y1 the producer stored."""
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(rows, c, generator=g).half().to(device)
    w0 = (torch.randn(c, c, generator=g) * c ** -0.5).half()
    b0 = (torch.randn(c, generator=g) * 0.3 + mean_shift).half()
    res = [torch.randn(rows, c, generator=g).half().to(device) for _ in range(n_res)]
    y1, st = K.gemm(x0, w0.to(device), b0.to(device), res=res[0] if n_res > 0 else None, res2=res[1] if n_res > 1 else None,
                    want_stats=True, tile_cfg=tile_cfg)
    ref1 = x0.float().cpu() @ w0.float().t() + b0.float()
    for r in res:
        ref1 = ref1 + r.float().cpu()
    e1 = (y1.float().cpu() - ref1).abs().max().item()
    assert e1 < 4e-3 * max(1.0, float(ref1.abs().max())), e1
    # consumer
    gamma = 1.0 + 0.2 * torch.randn(c, generator=g)
    beta = 0.1 * torch.randn(c, generator=g)
    w1 = (torch.randn(o, c, generator=g) * c ** -0.5).half()
    b1 = (torch.randn(o, generator=g) * 0.3).half()
    eps = 1e-5
    ow = o // 2 if geglu else o
    if st is None:
        # the library split K for the producer (long K, few rows; only with a GPU workspace): no statistics, y1 complete --
        # the host then runs the LayerNorm kernel and the plain GEMM (models/attention.py does exactly this)
        assert c >= 1024, "statistics may only be dropped for split-K shapes"
        xn_dev = K.layernorm(y1, gamma.half().to(device), beta.half().to(device), eps=eps)
        wd, bd = K.pack_geglu(w1, b1) if geglu else (w1, b1)
        y2 = K.gemm(xn_dev, wd.to(device), bd.to(device), geglu=geglu)
    else:
        assert st.shape == (rows, c // 64, 2)
        blocks = y1.float().cpu().view(rows, c // 64, 64)
        assert torch.allclose(st[..., 0].cpu(), blocks.sum(-1), rtol=1e-5, atol=1e-3)
        assert torch.allclose(st[..., 1].cpu(), (blocks * blocks).sum(-1), rtol=1e-5, atol=1e-3)
        ln = K.LnFold(w1, b1, gamma, beta, eps, device, pack=K.pack_geglu if geglu else None)
        y2 = K.gemm(y1, None, None, geglu=geglu, ln=ln, ln_stats=st)
    xn = F.layer_norm(y1.float().cpu(), (c,), gamma, beta, eps)
    ref2 = xn @ w1.float().t() + b1.float()
    if geglu:
        ref2 = ref2[:, :ow] * F.gelu(ref2[:, ow:])
    e2 = (y2.float().cpu() - ref2).abs().max().item()
    assert y2.shape == (rows, ow) and torch.isfinite(y2.float()).all()
    # the reference rounds LN(x) to fp16 before the GEMM; here x and gamma*W are the fp16 operands: same size of error
    assert e2 < 6e-3 * max(1.0, float(ref2.abs().max())), (e2, float(ref2.abs().max()))
    return {"producer_err": e1, "consumer_err": e2, "fused": st is not None}


def case_gemm_vt(device, *, n, l, k, c, lp, tile_cfg=0, seed=0):
    """Transposed-output form: V^T[n][c][lp] = w @ x[n]^T with zero padding of columns [l, lp)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, l, k, generator=g).half().to(device)
    w = (torch.randn(c, k, generator=g) * k ** -0.5).half()
    out = torch.full((n, c, lp), 3.0, dtype=torch.float16, device=device)
    K.gemm_vt(x, w.to(device), lp, out=out, tile_cfg=tile_cfg)
    ref = torch.zeros(n, c, lp)
    ref[:, :, :l] = (x.float().cpu() @ w.float().t()).transpose(1, 2)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, float(ref.abs().max())), err
    return {"max_err": err}


def case_gemm_qkvt(device, *, n, l, k, c, tile_cfg=0, seed=0, ldx_extra=0):
    """fz_gemm_qkvt: the q | k | V^T projection of a self-attention in one launch -- q|k token-major [n, l, 2c] and V^T [n, c, l]
    against fp32 torch, AND bit for bit against the two launches it replaces (fz_gemm on the q|k rows with the same tile -- same K
    order -- and the operand-swapped fz_gemm transpose_out form on the v rows: the same products, fp32-accumulated in the same K
    order, rounded once)."""
    g = torch.Generator().manual_seed(seed)
    xfull = torch.randn(n, l, k + ldx_extra, generator=g).half().to(device)
    x = xfull[..., ldx_extra:] if ldx_extra else xfull
    w = (torch.randn(3 * c, k, generator=g) * k ** -0.5).half().to(device)
    assert K.gemm_qkvt_ok(x, w, 2 * c)
    qk, vt = K.gemm_qkvt(x, w, 2 * c, tile_cfg=tile_cfg)
    assert qk.shape == (n, l, 2 * c) and vt.shape == (n, c, l)
    ref = x.float().cpu() @ w.float().cpu().t()
    scale = max(1.0, float(ref.abs().max()))
    e_qk = float((qk.float().cpu() - ref[..., : 2 * c]).abs().max())
    e_vt = float((vt.float().cpu() - ref[..., 2 * c:].transpose(1, 2)).abs().max())
    assert e_qk < 4e-3 * scale and e_vt < 4e-3 * scale, (e_qk, e_vt, scale)
    qk2 = K.gemm(x, w[: 2 * c], tile_cfg=tile_cfg, split_k=1)  # (the fused form never splits K: same summation order)
    vt2 = K.gemm_vt(x, w[2 * c:], l)
    return {"qk_err": e_qk, "vt_err": e_vt, "qk_bit_equal": bool(torch.equal(qk, qk2)), "vt_bit_equal": bool(torch.equal(vt, vt2)),
            "vt_max_diff_vs_two_launches": float((vt.float() - vt2.float()).abs().max())}


def case_gemm_lnout(device, *, rows, k, n_res=1, bias=True, seed=0, mean_shift=0.0, expect=True, tile_cfg=0, o=320, split_k=0):
    """fz_gemm_lnout: fz_gemm whose epilogue also writes LayerNorm(y) for whole-row (320-wide) tiles.  y must be BIT-IDENTICAL to fz_gemm's;
    y_ln vs fp32 torch LayerNorm of the stored fp16 y, and vs fz_layernorm on it (a few fp16 ulp: other summation order)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, k, generator=g).half().to(device)
    w = (torch.randn(o, k, generator=g) * k ** -0.5).half().to(device)
    b = ((torch.randn(o, generator=g) * 0.3) + mean_shift).half().to(device) if bias else None
    res = [(torch.randn(rows, o, generator=g) * 1.5).half().to(device) for _ in range(n_res)]
    gamma = (1.0 + 0.2 * torch.randn(o, generator=g)).half().to(device)
    beta = (0.1 * torch.randn(o, generator=g)).half().to(device)
    kw = dict(res=res[0] if n_res > 0 else None, res2=res[1] if n_res > 1 else None)
    y0 = K.gemm(x, w, b, tile_cfg=tile_cfg, split_k=split_k, **kw)
    y, yln = K.gemm_lnout(x, w, b, (gamma, beta, 1e-5), tile_cfg=tile_cfg, split_k=split_k, **kw)
    assert torch.equal(y, y0), "the LayerNorm epilogue must not change what is stored"
    if yln is None:
        assert not expect, "this shape runs on a whole-row tile: the LayerNorm must come from the epilogue"
        return None
    ref = F.layer_norm(y.float().cpu(), (o,), gamma.float().cpu(), beta.float().cpu(), 1e-5)
    e_t = float((yln.float().cpu() - ref).abs().max())
    e_k = float((yln.float() - K.layernorm(y, gamma, beta, eps=1e-5).float()).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert e_t < 2e-3 * scale and e_k <= 3 * 2.0 ** -10 * scale, (e_t, e_k, scale)
    return {"vs_torch": e_t, "vs_layernorm_kernel": e_k}


def case_ff_chain(device, *, rows, inner=1280, bias=True, res=True, ln=True, seed=0, lead=None, exact=True):
    """fz_ff_chain (csrc/ff_chain.hip): GEGLU up-projection -> gate -> down-projection + residual + LayerNorm in ONE launch, C = 320.
    BIT-IDENTICAL to the two launches it replaces (fz_gemm with the GEGLU epilogue + fz_gemm_lnout: same MFMA, same k order, same roundings)
    and within fp16 rounding of fp32 torch (h rounded to fp16 as the two-launch form stores it)."""
    g = torch.Generator().manual_seed(seed)
    c = 320
    xn = (torch.randn(rows, c, generator=g) * 1.2).half()
    w1 = (torch.randn(2 * inner, c, generator=g) * c ** -0.5).half()
    b1 = (torch.randn(2 * inner, generator=g) * 0.3).half() if bias else None
    w2 = (torch.randn(c, inner, generator=g) * inner ** -0.5).half()
    b2 = (torch.randn(c, generator=g) * 0.3).half() if bias else None
    r = (torch.randn(rows, c, generator=g) * 1.5).half() if res else None
    gamma = (1.0 + 0.2 * torch.randn(c, generator=g)).half()
    beta = (0.1 * torch.randn(c, generator=g)).half()
    dev = lambda t: None if t is None else t.to(device)
    assert K.ff_chain_ok(rows, c, inner)
    packed = K.ff_chain_pack(dev(w1), dev(b1), dev(w2))
    xin = dev(xn) if lead is None else dev(xn).reshape(*lead, c)
    rin = None if r is None else (dev(r) if lead is None else dev(r).reshape(*lead, c))
    y, yln = K.ff_chain(xin, packed, dev(b2), inner, res=rin, ln=(dev(gamma), dev(beta), 1e-5) if ln else None)
    assert y.shape == xin.shape and (yln is None) == (not ln)
    y = y.reshape(rows, c)
    # fp32 torch with the one rounding the chain shares with the reference's fp16 autocast: h is an fp16 tensor
    u = xn.float() @ w1.float().t()
    if bias:
        u = u + b1.float()
    h = (u[:, :inner] * F.gelu(u[:, inner:])).half().float()
    ref = h @ w2.float().t()
    if bias:
        ref = ref + b2.float()
    if res:
        ref = ref + r.float()
    scale = max(1.0, float(ref.abs().max()))
    err = float((y.float().cpu() - ref).abs().max())
    assert torch.isfinite(y.float()).all() and err < 4e-3 * scale, (err, scale)
    out = {"max_err": err}
    if ln:
        lref = F.layer_norm(y.float().cpu(), (c,), gamma.float(), beta.float(), 1e-5)
        e_ln = float((yln.reshape(rows, c).float().cpu() - lref).abs().max())
        assert e_ln < 2e-3 * max(1.0, float(lref.abs().max())), e_ln
        out["ln_vs_torch"] = e_ln
    # the two launches it replaces
    wp, bp = K.pack_geglu(dev(w1), dev(b1))
    h2 = K.gemm(dev(xn), wp, bp, geglu=True, split_k=1)   # (split_k=1: one fp32 accumulation chain over k ascending, as the chain kernel's)
    if ln:
        y2, yln2 = K.gemm_lnout(h2, dev(w2), dev(b2), (dev(gamma), dev(beta), 1e-5), res=dev(r), split_k=1)
    else:
        y2, yln2 = K.gemm(h2, dev(w2), dev(b2), res=dev(r), split_k=1), None
    d = float((y.float() - y2.float()).abs().max())
    out["vs_two_launches"] = d
    if exact:
        assert torch.equal(y, y2), d
        if ln and yln2 is not None:
            assert torch.equal(yln.reshape(rows, c), yln2)
    else:
        assert d <= 2 * 2.0 ** -10 * scale, (d, scale)
    return out


def case_xattn_chain(device, *, n, tokens, clip=1, lk=77, front=False, bias=True, ln=True, seed=0, exact=True):
    """fz_xattn_chain (csrc/xattn_chain.hip): attn2 of a 320-channel block -- to_q -> 77-key cross-attention -> to_out + residual + LayerNorm --
    in ONE launch; `front`: attn1.to_out + residual + norm2 in front of it in the same launch.  BIT-IDENTICAL to the launches it replaces
    (fz_gemm + fz_attn_cross + fz_gemm_lnout, and fz_gemm_lnout in front) and within fp16 rounding of fp32 torch."""
    g = torch.Generator().manual_seed(seed)
    c, heads, dh = 320, 8, 40
    nb = (n + clip - 1) // clip
    x = (torch.randn(n, tokens, c, generator=g) * 1.2).half()
    res = (torch.randn(n, tokens, c, generator=g) * 1.5).half()
    wq = (torch.randn(c, c, generator=g) * c ** -0.5 * 2.0).half()
    wk = (torch.randn(c, 768, generator=g) * 768 ** -0.5 * 2.0).half()
    wv = (torch.randn(c, 768, generator=g) * 768 ** -0.5).half()
    wo = (torch.randn(c, c, generator=g) * c ** -0.5).half()
    bo = (torch.randn(c, generator=g) * 0.3).half() if bias else None
    wo1 = (torch.randn(c, c, generator=g) * c ** -0.5).half()
    bo1 = (torch.randn(c, generator=g) * 0.3).half() if bias else None
    ctx = torch.randn(nb, lk, 768, generator=g).half()
    gam = [(1.0 + 0.2 * torch.randn(c, generator=g)).half() for _ in range(2)]
    bet = [(0.1 * torch.randn(c, generator=g)).half() for _ in range(2)]
    dev = lambda t: None if t is None else t.to(device)
    scale = dh ** -0.5
    assert K.xattn_chain_ok(n * tokens, tokens, c, heads, lk)
    kk = K.gemm(dev(ctx), dev(wk))
    vt = K.gemm_vt(dev(ctx), dev(wv), K.CROSS_KEYS)
    kvp = K.xattn_chain_kv_pack(kk, vt, lk)
    packed = K.xattn_chain_pack(dev(wq), dev(wo), (dev(wo1), dev(bo1), dev(gam[0]), dev(bet[0])) if front else None)
    lnp = (dev(gam[1]), dev(bet[1]), 1e-5) if ln else None
    kw = dict(res=dev(res), frames_per_batch=clip, heads=heads, lk=lk, scale=scale, ln=lnp)
    if front:
        y, yln, y1 = K.xattn_chain(dev(x), packed, kvp, dev(bo), front_eps=1e-5, **kw)
    else:
        y, yln = K.xattn_chain(dev(x), packed, kvp, dev(bo), **kw)
        y1 = None
    assert (yln is None) == (not ln)
    # ---- the launches it replaces
    if front:
        # (tile 254122 = 320 x 128: the whole-row tile the 64x64 level takes; small test shapes would get another one and no LayerNorm)
        y1r, xnr = K.gemm_lnout(dev(x), dev(wo1), dev(bo1), (dev(gam[0]), dev(bet[0]), 1e-5), res=dev(res), split_k=1, tile_cfg=254122)
        assert xnr is not None
        resr = y1r
    else:
        xnr, resr = dev(x), dev(res)
    q = K.gemm(xnr, dev(wq), split_k=1)
    o = torch.empty_like(q)
    K.attn_cross(q, kk, vt, o, clip_len=clip, heads=heads, lk=lk, scale=scale)
    if ln:
        yr, ylnr = K.gemm_lnout(o, dev(wo), dev(bo), lnp, res=resr, split_k=1, tile_cfg=254122)
    else:
        yr, ylnr = K.gemm(o, dev(wo), dev(bo), res=resr, split_k=1), None
    out = {"vs_launches": float((y.float() - yr.float()).abs().max())}
    if front:
        out["y1_vs_launch"] = float((y1.float() - y1r.float()).abs().max())
    # ---- fp32 torch on the fp16 operands (q, o and the LayerNorm output rounded to fp16 as every fp16 pipeline stores them)
    xf, rf = x.float(), res.float()
    if front:
        h1 = (xf @ wo1.float().t() + (bo1.float() if bias else 0.0)).half().float() + rf
        h1 = h1.half().float()
        xn = F.layer_norm(h1, (c,), gam[0].float(), bet[0].float(), 1e-5).half().float()
        rf = h1
    else:
        xn = xf
    qf = (xn @ wq.float().t()).half().float().reshape(n, tokens, heads, dh).permute(0, 2, 1, 3)
    kf = kk.float().cpu().reshape(nb, lk, heads, dh).permute(0, 2, 1, 3)
    vf = vt.float().cpu()[:, :, :lk].reshape(nb, heads, dh, lk).permute(0, 1, 3, 2)
    bidx = torch.arange(n) // clip
    pr = (qf @ kf[bidx].transpose(-1, -2) * scale).softmax(-1)
    of = (pr @ vf[bidx]).permute(0, 2, 1, 3).reshape(n, tokens, c).half().float()
    ref = of @ wo.float().t() + (bo.float() if bias else 0.0) + rf
    sc = max(1.0, float(ref.abs().max()))
    err = float((y.float().cpu() - ref).abs().max())
    assert torch.isfinite(y.float()).all() and err < 6e-3 * sc, (err, sc)
    out["max_err"] = err
    if ln:
        lref = F.layer_norm(y.float().cpu(), (c,), gam[1].float(), bet[1].float(), 1e-5)
        e_ln = float((yln.float().cpu() - lref).abs().max())
        assert e_ln < 2e-3 * max(1.0, float(lref.abs().max())), e_ln
        out["ln_vs_torch"] = e_ln
    if exact:
        if front:
            assert torch.equal(y1, y1r), out
        assert torch.equal(y, yr), out
        if ln and ylnr is not None:
            assert torch.equal(yln, ylnr)
    else:
        assert out["vs_launches"] <= 2 * 2.0 ** -10 * sc, out
    return out


def case_gn_from_epilogue(device, *, n, clip, tokens, cin, cout, groups=32, producer="tconv", seed=0):
    """GroupNorm statistics out of the PRODUCING launch's epilogue (fz_temporal_conv3_gn / fz_gemm_gn -> fz_groupnorm_from_partials): the
    producer's output must be bit-identical to the plain launch, and GroupNorm(+SiLU) from the epilogue's partials must match both the
    three-kernel fz_groupnorm on the same tensor (to the rounding of the fp32 statistics: an fp16 ulp) and fp32 torch -- for statistics
    spanning one frame (the transformer's norm) and the whole clip (the resnet norms).  Returns None when the library's launch for this
    shape cannot carry the statistics (the caller asserts whether it expected that)."""
    g = torch.Generator().manual_seed(seed)
    gam = (1 + 0.1 * torch.randn(cout, generator=g)).half().to(device)
    bet = (0.1 * torch.randn(cout, generator=g)).half().to(device)
    if producer == "tconv":
        x = torch.randn(n, tokens, cin, generator=g).half().to(device)
        w = (torch.randn(cout, 3, cin, generator=g) * (3 * cin) ** -0.5).half().to(device)
        r1 = (torch.randn(n, tokens, cout, generator=g) * 3 + 2).half().to(device)   # a non-zero mean: the shifted sums must cope
        temb = torch.randn(n // clip, cout, generator=g).half().to(device)
        y0 = K.temporal_conv3(x, w, clip_len=clip, res=r1, temb=temb)
        y1, part = K.temporal_conv3(x, w, clip_len=clip, res=r1, temb=temb, gn_groups=groups)
    else:
        x = torch.randn(n, tokens, cin, generator=g).half().to(device)
        w = (torch.randn(cout, cin, generator=g) * cin ** -0.5).half().to(device)
        b = torch.randn(cout, generator=g).half().to(device)
        r1 = (torch.randn(n, tokens, cout, generator=g) * 2 - 1).half().to(device)
        y0 = K.gemm(x, w, b, res=r1)
        y1, part = K.gemm_gn(x, w, b, res=r1, gn_groups=groups, rows_per_frame=tokens)
    assert torch.equal(y0, y1), "the statistics epilogue must not change what is stored"
    if part is None:
        return None
    assert tuple(part.shape) == (n, groups, tokens // 128, 3)
    res = {}
    for span in (1, clip):
        ref = K.groupnorm(y1, gam, bet, span=span, groups=groups, eps=1e-5, silu=True)
        got = K.groupnorm_from_partial(y1, gam, bet, part, span=span, groups=groups, eps=1e-5, silu=True)
        yc = y1.float().cpu()
        t = F.silu(F.group_norm(yc.view(n // span, span, tokens, cout).permute(0, 3, 1, 2).reshape(n // span, cout, -1), groups,
                                gam.float().cpu(), bet.float().cpu(), 1e-5))
        t = t.reshape(n // span, cout, span, tokens).permute(0, 2, 3, 1).reshape(n, tokens, cout)
        e_ref = float((got.float() - ref.float()).abs().max())
        e_t = float((got.float().cpu() - t).abs().max())
        assert e_t < 4e-3 * max(1.0, float(t.abs().max())), (span, e_t)
        assert e_ref <= 2 * 2.0 ** -10 * max(1.0, float(t.abs().max())), (span, e_ref)  # both round the same fp32 statistics: an ulp or two
        res[f"span{span}"] = {"vs_three_kernel": e_ref, "vs_torch": e_t}
    return res


def case_gn_epilogue_ragged_last_tile(device, *, frames=3, cin=64, cout=320, groups=32, seed=0):
    """fz_gemm_gn pinned to the 320 x 256 tile (two statistics passes of 128 rows) on a launch with 128 * odd rows: the LAST tile's second
    pass lies wholly beyond the rows and must write no record (it used to write groups * chunks * 3 floats past the end of `partial`, from
    stale staging rows).  Canary floats behind the partials, records against fp64 statistics of the stored tensor."""
    from fatezero_amd import _native as N
    import ctypes as C
    g = torch.Generator().manual_seed(seed)
    tokens = 128
    rows = frames * tokens
    assert (rows // 128) % 2 == 1
    x = torch.randn(frames, tokens, cin, generator=g).half().to(device)
    w = (torch.randn(cout, cin, generator=g) * cin ** -0.5).half().to(device)
    b = torch.randn(cout, generator=g).half().to(device)
    y = torch.empty(frames, tokens, cout, dtype=torch.float16, device=device)
    n_rec = frames * groups * (tokens // 128) * 3
    buf = torch.full((n_rec + 4096,), -777.0, dtype=torch.float32, device=device)
    d = N.FzGemmDesc()
    d.rows, d.in_features, d.out_features = rows, cin, cout
    d.ldx, d.ldw, d.ldy, d.ldres = cin, cin, cout, cout
    d.batch, d.epilogue, d.tile_cfg = 1, N.FZ_GEMM_PLAIN, 254222
    rc = N.lib().fz_gemm_gn(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), buf.data_ptr(), groups, tokens,
                            K._stream(x))
    assert rc == 0, rc
    assert bool((buf[n_rec:] == -777.0).all()), "statistics records written beyond the last frame"
    part = buf[:n_rec].view(frames, groups, 1, 3).cpu().double()
    yc = y.float().cpu().double().view(frames, tokens, groups, cout // groups)
    cnt = float(tokens * (cout // groups))
    mean = yc.mean(dim=(1, 3))
    m2 = ((yc - mean[:, None, :, None]) ** 2).sum(dim=(1, 3))
    assert bool((part[..., 0, 0] == cnt).all())
    assert torch.allclose(part[..., 0, 1], mean, rtol=1e-4, atol=1e-4) and torch.allclose(part[..., 0, 2], m2, rtol=1e-3, atol=1e-2)
    ref = x.float().cpu() @ w.float().cpu().t() + b.float().cpu()
    assert float((y.float().cpu() - ref).abs().max()) < 4e-3 * max(1.0, float(ref.abs().max()))
    return {"records": n_rec // 3}


def case_lora_pair(device, *, batch, clip, tokens, c, with_temb=True, with_res2=True, seed=0, up_scale=1.0, gn_groups=0):
    """fz_lora_pair (up(down(x)) + x (+ temb) (+ res2) of the temporal LoRA in one launch, lora.py:31-54) against (1) fz_temporal_conv3
    called twice -- bit for bit: same fp16 rounding of the rank-160 intermediate, same K order, same epilogue order -- and (2) fp32 torch
    conv1d with the intermediate rounded to fp16."""
    g = torch.Generator().manual_seed(seed)
    n, rank = batch * clip, 160
    x = torch.randn(n, tokens, c, generator=g).half().to(device)
    wd = (torch.randn(rank, c, 3, generator=g) * (3 * c) ** -0.5).half()
    wu = (torch.randn(c, rank, 3, generator=g) * up_scale * (3 * rank) ** -0.5).half()
    temb = torch.randn(batch, c, generator=g).half().to(device) if with_temb else None
    res2 = torch.randn(n, tokens, c, generator=g).half().to(device) if with_res2 else None
    wdn, wun = wd.permute(0, 2, 1).contiguous().to(device), wu.permute(0, 2, 1).contiguous().to(device)
    assert K.lora_pair_ok(n, tokens, c, rank, clip)
    y = K.lora_pair(x, wdn, wun, clip_len=clip, res2=res2, temb=temb)

    def conv_no_split(xi, wt, res=None, r2=None, rows=None):  # fz_temporal_conv3 without a workspace: no split-K, one K order
        from fatezero_amd import _native as N
        out = torch.empty(n, tokens, wt.shape[0], dtype=torch.float16, device=device)
        rc = N.lib().fz_temporal_conv3(xi.data_ptr(), wt.data_ptr(), None if res is None else res.data_ptr(),
                                       None if r2 is None else r2.data_ptr(), None if rows is None else rows.data_ptr(),
                                       0 if rows is None else rows.stride(0), out.data_ptr(), n, tokens, xi.shape[2], wt.shape[0], clip,
                                       None, 0, K._stream(xi))
        assert rc == 0, rc
        return out

    y2 = conv_no_split(conv_no_split(x, wdn), wun, res=x, r2=res2, rows=temb)
    same = torch.equal(y, y2)
    assert same, float((y.float() - y2.float()).abs().max())
    xr = x.float().cpu().reshape(batch, clip, tokens, c).permute(0, 2, 3, 1).reshape(batch * tokens, c, clip)
    dr = F.conv1d(xr, wd.float(), None, padding=1).half().float()
    yr = F.conv1d(dr, wu.float(), None, padding=1).reshape(batch, tokens, c, clip).permute(0, 3, 1, 2).reshape(n, tokens, c)
    yr = yr + x.float().cpu()
    if with_temb:
        yr = yr + temb.float().cpu().repeat_interleave(clip, 0)[:, None, :]
    if with_res2:
        yr = yr + res2.float().cpu()
    err = (y.float().cpu() - yr).abs().max().item()
    assert err < 4e-3 * max(1.0, float(yr.abs().max())), err
    res = {"max_err": err, "bit_identical_to_two_launches": same}
    if gn_groups > 0:
        # fz_lora_pair_gn: the same y, plus Welford partials from which GroupNorm(+SiLU) must match the three-kernel fz_groupnorm on the same
        # tensor (to the rounding of the fp32 statistics) and fp32 torch, for statistics over one frame and over the clip
        y3, part = K.lora_pair(x, wdn, wun, clip_len=clip, res2=res2, temb=temb, gn_groups=gn_groups)
        assert torch.equal(y3, y), "the statistics epilogue must not change what is stored"
        res["partial"] = None if part is None else tuple(part.shape)
        if part is not None:
            gam = (1 + 0.1 * torch.randn(c, generator=g)).half().to(device)
            bet = (0.1 * torch.randn(c, generator=g)).half().to(device)
            for span in (1, clip):
                ref = K.groupnorm(y, gam, bet, span=span, groups=gn_groups, eps=1e-5, silu=True)
                got = K.groupnorm_from_partial(y, gam, bet, part, span=span, groups=gn_groups, eps=1e-5, silu=True)
                yc = y.float().cpu()
                t = F.silu(F.group_norm(yc.view(n // span, span, tokens, c).permute(0, 3, 1, 2).reshape(n // span, c, -1), gn_groups,
                                        gam.float().cpu(), bet.float().cpu(), 1e-5))
                t = t.reshape(n // span, c, span, tokens).permute(0, 2, 3, 1).reshape(n, tokens, c)
                e_ref = float((got.float() - ref.float()).abs().max())
                e_t = float((got.float().cpu() - t).abs().max())
                assert e_t < 4e-3 * max(1.0, float(t.abs().max())), (span, e_t)
                assert e_ref <= 2 * 2.0 ** -10 * max(1.0, float(t.abs().max())), (span, e_ref)
                res[f"gn_span{span}"] = {"vs_three_kernel": e_ref, "vs_torch": e_t}
    return res
