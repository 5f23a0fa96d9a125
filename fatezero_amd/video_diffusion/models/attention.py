"""Spatio-temporal transformer (reference: video_diffusion/models/attention.py and the patched forwards of
video_diffusion/prompt_attention/attention_register.py).

Differences from the reference, by design:
  * activations stay token-major [(b f), (h w), c] fp16 from proj_in to proj_out; the temporal attention reads
    the same buffer with strides instead of rearranging '(b f) d c -> (b d) f c' (attention.py:327-337);
  * the controller hook is not a tensor callback between softmax and P.V (attention_register.py:47-55) but an
    `AttnPlan` the controller hands to the fused HIP kernels: which frames run plain flash attention, which
    capture their probability map into the HBM arena, which take stored maps / masks / word mappers
    (fatezero_amd/video_diffusion/prompt_attention/attention_store.py).  Foreign controllers that only implement
    the reference's `__call__(attn, is_cross, place)` still work through a materialise-call-inject path;
  * cross-attention K/V are projected once per batch element, not once per frame (the reference repeats the
    text context F times before projecting, attention.py:104).
"""
import copy
import os
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from ... import dist as D
from ... import issue as _issue
from ... import kernels as K
from .resnet import Tokens, _LinearParams, _NormParams, group_norm_tokens

_LOG2E = 1.4426950408889634


@dataclass
class SpatioTemporalTransformerModelOutput:
    sample: torch.Tensor


class AttnPlan:
    """What the attention kernels do for one controlled layer call.

    Frames [0, n_plain) run plain flash attention; frames [n_plain, N) run `mode` with the given operands."""
    __slots__ = ("n_plain", "mode", "p", "row_mask", "mapper_t", "coef", "cur_out", "capture_first")

    def __init__(self, n_plain, mode=K.FZ_ATTN_FLASH, p=None, row_mask=None, mapper_t=None, coef=None, cur_out=None,
                 capture_first=None):
        self.n_plain, self.mode, self.p, self.row_mask = n_plain, mode, p, row_mask
        self.mapper_t, self.coef, self.cur_out, self.capture_first = mapper_t, coef, cur_out, capture_first


def _plan_for(controller, is_cross, place, n, clip, heads, lq, lk, device):
    if controller is None:
        return AttnPlan(n)
    planner = getattr(controller, "attention_plan", None)
    if planner is not None:
        rec = _issue.recording()
        if rec is not None:  # a forward being recorded into a native issue plan: the controller call is an event, not part of the plan
            return rec.controller_call(planner, (is_cross, place, n, clip, heads, lq, lk, device))
        return planner(is_cross, place, n, clip, heads, lq, lk, device)
    return None  # foreign controller: generic path


# LayerNorm folded into the GEMMs around it (fz_gemm_ln: row statistics from the producing out-projection, correction in the
# consuming projection's epilogue).  Correct and tested (kernel cases on MI355X, the pipeline with the switch on in
# tests/test_pipeline_emu.py) but OFF: same-box A/B (scripts/ab_bench.py, profiles/r02_ab_ln_fusion.txt) shows the job 1.1-1.3 %
# SLOWER with it -- the 30 LayerNorm launches it removes per forward (C <= 640 levels) cost less than the statistics loop and the
# per-row correction add to the 60 GEMM epilogues.
LN_FUSION = False
# q | k | V^T of a self-attention in ONE launch (fz_gemm_qkvt: the V columns leave transposed from the same GEMM; the LayerNorm
# output is read once instead of twice, one launch instead of two) wherever the frame's token count is a multiple of 64 -- every level
# of a 512^2 clip.  Bit-identical to the two launches (tests/kernel_cases.py: case_gemm_qkvt).  Switch kept for same-box A/B runs.
QKV_FUSION = os.environ.get("FZ_NO_QKV_FUSION") is None  # (the env switch: same-box A/B runs of bench.py)
LN_FUSION_MAX_C = 640  # wider rows (K = 1280) want split-K in the consuming GEMM, which the fused epilogue excludes
# LayerNorm out of the PRODUCING projection's epilogue (fz_gemm_lnout, round 5): every `x = f(norm(x)) + x` step ends in a Linear + residual
# whose output is the next LayerNorm's input, and at the 320-channel level the 320-wide GEMM tile holds whole rows -- the epilogue computes
# exact row statistics on the values it stores and writes LN(y) beside y: proj_in -> norm1, attn1.to_out -> norm2, attn2.to_out -> norm3,
# ff.net[2] -> norm_temporal.  Four LayerNorm launches per 64x64-level block gone.  (env switch: same-box A/B runs of bench.py)
# Only there: at 640 channels a row spans two column tiles, and handing the LayerNorm to the split-K tail kernel of the 1280-wide projections
# (one wave per row) was built and measured a LOSS for the job (profiles/r05_ln_from_producer_ab.txt).
LN_FROM_PRODUCER = os.environ.get("FZ_NO_LN_FROM_PRODUCER") is None
LN_FROM_PRODUCER_C = 320
# The whole feed-forward CHAIN of a 320-channel block in ONE launch (fz_ff_chain, round 6): GEGLU up-projection -> gate -> down-projection +
# residual + norm_temporal, the rows x 1280 intermediate in registers only, the weights streamed as pre-packed MFMA fragments.  Bit-identical
# to fz_gemm(GEGLU) + fz_gemm_lnout; used where fz_ff_chain_preferred says the one launch is the faster form.  (env switch: same-box A/B runs)
FF_CHAIN = os.environ.get("FZ_NO_FF_CHAIN") is None
# The cross-attention CHAIN of a 320-channel block in ONE launch (fz_xattn_chain, round 6): attn2.to_q -> 77-key cross-attention -> attn2.to_out +
# residual + norm3, and (XATTN_CHAIN_FRONT) attn1.to_out + residual + norm2 in front of it in the same launch: q, the attention output and the
# LayerNorm outputs never leave the CU, the weights and the text context's K / V^T are streamed as pre-packed MFMA fragments.  Bit-identical to
# fz_gemm + fz_attn_cross + fz_gemm_lnout; used where no controller touches the maps (more than 32 x 32 queries: attention_store.py:83) and
# fz_xattn_chain_preferred says the one launch is the faster form.  (env switches: same-box A/B runs)
XATTN_CHAIN = os.environ.get("FZ_NO_XATTN_CHAIN") is None
XATTN_CHAIN_FRONT = os.environ.get("FZ_XATTN_FRONT") is not None  # (the front form measured EQUAL for the job, profiles/r06_xattn_chain_job_ab.txt: opt-in)


class Prenormed:
    """LN(y) that came out of the projection which produced y (fz_gemm_lnout): travels where the row statistics of fz_gemm_ln would."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


def _out_proj(lin, out, residual, want_stats, ln_next):
    """to_out / ff.net[2]: Linear + residual; with `ln_next` (the LayerNorm that consumes the result) also LN(result) from the same launch."""
    if ln_next is not None and out.is_contiguous():
        g, b = ln_next.packed(out.device)
        w, bias = lin.packed(out.dtype, out.device)
        y, yln = K.gemm_lnout(out, w, bias, (g, b, ln_next.eps), res=residual)
        return y, (None if yln is None else Prenormed(yln))
    return lin.apply(out, res=residual, want_stats=want_stats)


def _ln_ready(norm, stats, x):
    """Can the LayerNorm `norm` of x be folded into the Linear that consumes it?  (statistics from the producing GEMM at hand,
    64-channel blocks, fp16 engine)"""
    return (LN_FUSION and norm is not None and stats is not None and x.shape[-1] % 64 == 0 and x.shape[-1] <= LN_FUSION_MAX_C
            and x.dtype == torch.float16)


def layer_norm_tokens(norm, x):
    g, b = norm.packed(x.device)
    return K.layernorm(x, g, b, eps=norm.eps)


class CrossAttention(nn.Module):
    """Parameter container + executor for diffusers' CrossAttention as used by the reference (to_q/to_k/to_v without
    bias, to_out = [Linear, Dropout]); head count = `heads`, scale = dim_head**-0.5."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, **unused):
        super().__init__()
        inner = dim_head * heads
        self.query_dim, self.inner_dim = query_dim, inner
        self.is_cross_module = cross_attention_dim is not None
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = _LinearParams(query_dim, inner, bias=bias)
        self.to_k = _LinearParams(cross_attention_dim, inner, bias=bias)
        self.to_v = _LinearParams(cross_attention_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([_LinearParams(inner, query_dim), nn.Identity()])
        # set by register_attention_control
        self.controller = None
        self.place_in_unet = None
        self._qk = None
        self._qk_fold = 1.0
        self._qkv = None
        self._qkv_self = None
        self._ctx_kv = None   # (ctx, its version, K, V^T, fz_xattn_chain's pack of them or None)
        self._xchain = {}     # fz_xattn_chain's packed weights by (device, id of attn1's to_out or None)
        self._ln_fold = None  # (id of the norm, LnFold): the consuming projection with its LayerNorm folded in

    # -- helpers -----------------------------------------------------------------------------------------
    def _qk_weight(self, dtype, device, q_fold=1.0):
        """Fused [Wq; Wk] projection weight; `q_fold` is multiplied into the Wq rows in fp32 before the cast."""
        if self._qk is None or self._qk.dtype != dtype or self._qk.device != device or self._qk_fold != q_fold:
            wq = self.to_q.weight.detach().float() * q_fold
            self._qk = torch.cat([wq, self.to_k.weight.detach().float()], 0).to(device=device, dtype=dtype).contiguous()
            self._qk_fold = q_fold
        return self._qk

    def project_context_into(self, ctx, kk, vt, kvp=None):
        """K / V^T of the text context `ctx` written into EXISTING buffers (those of an earlier context of the same shape) and made this
        module's cached projections: the launch records of a native issue plan keep pointing at valid data (fatezero_amd/issue.py).  `kvp`:
        the fz_xattn_chain pack of those projections the records point at (re-packed in place)."""
        if kk.shape[:-1] != ctx.shape[:-1] or kk.device != ctx.device or ctx.dtype != kk.dtype:
            raise RuntimeError(f"issue plan: text context {tuple(ctx.shape)} / {ctx.dtype} does not fit the recorded projections {tuple(kk.shape)}")
        w, b = self.to_k.packed(ctx.dtype, ctx.device)
        K.gemm(ctx, w, b, out=kk)
        K.gemm_vt(ctx, self.to_v.packed(ctx.dtype, ctx.device)[0], K.CROSS_KEYS, out=vt)
        if kvp is not None:
            K.xattn_chain_kv_pack(kk, vt, ctx.shape[1], out=kvp)
        self._ctx_kv = (ctx, ctx._version, kk, vt, kvp)

    def _context_kv(self, ctx, want_pack=False):
        """K / V^T of the text context (and their fz_xattn_chain pack): they depend only on (ctx, weights) and the DDIM loops pass the same
        embedding tensor at every step, so they are projected once per job instead of once per layer call (16 x 100 times per job)."""
        kvc = self._ctx_kv
        if kvc is None or kvc[0] is not ctx or kvc[1] != ctx._version:
            kk = self.to_k.apply(ctx)
            vt = K.gemm_vt(ctx, self.to_v.packed(ctx.dtype, ctx.device)[0], K.CROSS_KEYS)  # V^T straight out of the GEMM
            kvc = self._ctx_kv = (ctx, ctx._version, kk, vt, None)
        if want_pack and kvc[4] is None:
            kvc = self._ctx_kv = kvc[:4] + (K.xattn_chain_kv_pack(kvc[2], kvc[3], ctx.shape[1]),)
        return kvc[2], kvc[3], kvc[4]

    def _chain_applies(self, n, lq, c, dtype, ctx):
        """Does fz_xattn_chain carry this layer call (and is it the faster form)?  Only where no controller stores or edits the maps."""
        return (XATTN_CHAIN and lq > 32 ** 2 and dtype == torch.float16 and ctx.dtype == torch.float16 and self.inner_dim == c
                and self.to_q.bias is None and D.active_shard() is None and K.xattn_chain_preferred(n * lq, lq, c, self.heads, ctx.shape[1]))

    def _chain_weights(self, device, front=None):
        """fz_xattn_chain's packed weights; front = (attn1's to_out, the LayerNorm behind it) for the front form."""
        key = (str(device), None if front is None else (id(front[0]), id(front[1])))
        p = self._xchain.get(key)
        if p is None:
            wq = self.to_q.packed(torch.float16, device)[0]
            wo = self.to_out[0].packed(torch.float16, device)[0]
            fr = None
            if front is not None:
                w1, b1 = front[0].packed(torch.float16, device)
                fr = (w1, b1) + tuple(front[1].packed(device))
            p = self._xchain[key] = K.xattn_chain_pack(wq, wo, fr)
        return p

    def forward_cross_after(self, x: Tokens, o1, attn1_out, hs, norm_in, ctx, clip: int, ln_next):
        """attn1's output projection + residual + `norm_in` AND this cross-attention + residual + `ln_next` in ONE launch (fz_xattn_chain, front
        form): o1 = attn1's attention output [N, L, C], hs = attn1's residual.  Falls back to the separate launches when the controller wants
        the maps of this call.  Returns what forward_cross returns."""
        n, lq, c = o1.shape
        plan = _plan_for(self.controller, True, self.place_in_unet, n, clip, self.heads, lq, ctx.shape[1], o1.device)
        if plan is None or (plan.mode == K.FZ_ATTN_FLASH):
            _, _, kvp = self._context_kv(ctx, want_pack=True)
            packed = self._chain_weights(o1.device, (attn1_out, norm_in))
            g3, b3 = ln_next.packed(o1.device)
            bo = self.to_out[0].packed(o1.dtype, o1.device)[1]
            y, yln, _ = K.xattn_chain(o1, packed, kvp, bo, res=hs, frames_per_batch=clip, heads=self.heads, lk=ctx.shape[1], scale=self.scale,
                                      ln=(g3, b3, ln_next.eps), front_eps=norm_in.eps)
            return y, Prenormed(yln)
        hs1, st = _out_proj(attn1_out, o1, hs, False, norm_in)
        return self.forward_cross(x.like(hs1), ctx, clip, residual=hs1, norm=norm_in, stats=st, ln_next=ln_next, plan=plan)

    def _generic_controller_call(self, controller, is_cross, q, k, vt, out, clip, lq, lk_total, run_capture, run_inject):
        """Reference protocol for a controller that only has __call__: materialise P, call it, apply the result.
        Like the reference with xformers enabled (attention_register.py:112-116,198-204) maps larger than 32x32
        tokens bypass the controller."""
        n = q.shape[0]
        width = K.CROSS_P_STRIDE if is_cross else lk_total
        p = torch.empty(n, self.heads, lq, width, dtype=torch.float16, device=q.device)
        run_capture(p)
        view = p[..., :lk_total] if is_cross else p
        new = controller(view, is_cross, self.place_in_unet)
        if new is not view:
            view.copy_(new)
        run_inject(p)

    # -- cross attention (attention_register.py:71-128) ------------------------------------------------------
    def forward_cross(self, x: Tokens, ctx, clip: int, residual=None, norm=None, stats=None, want_stats=False, ln_next=None, plan=False):
        """x.data: hidden states [N, L, C] -- LayerNorm'ed, or RAW together with (`norm`, `stats` = the row sums their producer
        wrote): the LayerNorm then rides in the to_q GEMM (fz_gemm_ln); ctx: [B, 77, Dctx] fp16.  Returns residual +
        to_out(attention) (the block's `hidden_states = attn2(...) + hidden_states`, attention.py:303-311, fused into the GEMM
        epilogue), plus that result's row statistics when want_stats."""
        n, lq, c = x.data.shape
        ctrl = self.controller
        if plan is False:  # (a plan handed in was already taken from the controller for THIS call: forward_cross_after's fall-back)
            if (ln_next is not None and residual is not None and residual.is_contiguous() and not want_stats
                    and self._chain_applies(n, lq, c, x.data.dtype, ctx)):
                plan = _plan_for(ctrl, True, self.place_in_unet, n, clip, self.heads, lq, ctx.shape[1], x.data.device)
                if plan is None or plan.mode == K.FZ_ATTN_FLASH:  # to_q -> cross-attention -> to_out + residual + LayerNorm in one launch
                    xn = stats.t if isinstance(stats, Prenormed) else (x.data if norm is None else layer_norm_tokens(norm, x.data))
                    _, _, kvp = self._context_kv(ctx, want_pack=True)
                    g3, b3 = ln_next.packed(xn.device)
                    y, yln = K.xattn_chain(xn.contiguous(), self._chain_weights(xn.device), kvp, self.to_out[0].packed(xn.dtype, xn.device)[1],
                                           res=residual, frames_per_batch=clip, heads=self.heads, lk=ctx.shape[1], scale=self.scale,
                                           ln=(g3, b3, ln_next.eps))
                    return y, Prenormed(yln)
        if isinstance(stats, Prenormed):  # LN(x) came out of the producing projection's epilogue
            q = self.to_q.apply(stats.t)
        elif norm is not None:
            if _ln_ready(norm, stats, x.data):
                if self._ln_fold is None or self._ln_fold[0] is not norm or self._ln_fold[1].w.device != x.data.device:
                    self._ln_fold = (norm, K.LnFold(self.to_q.weight, self.to_q.bias, norm.weight, norm.bias, norm.eps,
                                                    x.data.device))
                q = K.gemm(x.data, None, None, ln=self._ln_fold[1], ln_stats=stats)
            else:
                q = self.to_q.apply(layer_norm_tokens(norm, x.data))
        else:
            q = self.to_q.apply(x.data)
        kk, vt, _ = self._context_kv(ctx)
        lk = ctx.shape[1]
        out = torch.empty(n, lq, self.inner_dim, dtype=q.dtype, device=q.device)
        kw = dict(clip_len=clip, heads=self.heads, lk=lk, scale=self.scale)
        if plan is False:
            plan = _plan_for(ctrl, True, self.place_in_unet, n, clip, self.heads, lq, lk, q.device)
        if plan is None:
            if lq > 32 ** 2:
                K.attn_cross(q, kk, vt, out, mode=K.FZ_ATTN_FLASH, **kw)
            else:
                ident = torch.eye(K.CROSS_KEYS, dtype=torch.float16, device=q.device)
                coef = torch.zeros(2, K.CROSS_KEYS, dtype=torch.float32, device=q.device)
                coef[0] = 1.0
                self._generic_controller_call(
                    ctrl, True, q, kk, vt, out, clip, lq, lk,
                    lambda p: K.attn_cross(q, kk, vt, out, mode=K.FZ_ATTN_CAPTURE, p=p, **kw),
                    lambda p: K.attn_cross(q, kk, vt, out, mode=K.FZ_ATTN_INJECT, p=p, mapper_t=ident, coef=coef, **kw))
        elif plan.mode == K.FZ_ATTN_FLASH:
            K.attn_cross(q, kk, vt, out, mode=K.FZ_ATTN_FLASH, **kw)
        else:
            if plan.n_plain > 0:
                K.attn_cross(q, kk, vt, out, mode=K.FZ_ATTN_FLASH, frame0=0, n_frames=plan.n_plain, **kw)
            if plan.n_plain < n:
                K.attn_cross(q, kk, vt, out, mode=plan.mode, frame0=plan.n_plain, n_frames=n - plan.n_plain, p=plan.p,
                             mapper_t=plan.mapper_t, coef=plan.coef, cur_out=plan.cur_out, **kw)
        return _out_proj(self.to_out[0], out, residual, want_stats, ln_next)

    # -- temporal attention (attention.py:327-337; never controlled, attention_register.py:242) ---------------
    def forward_temporal(self, x_norm, batch: int, clip: int, residual=None, norm=None, stats=None):
        """x_norm: [B*F, L, C] LayerNorm'ed -- or RAW with (`norm`, `stats`): the LayerNorm then rides in the fused q/k/v GEMM;
        attention over the F frames of every (b, token); + residual in the epilogue."""
        n, l, c = x_norm.shape
        if isinstance(stats, Prenormed):
            x_norm, norm, stats = stats.t, None, None
        if norm is not None and not _ln_ready(norm, stats, x_norm):
            x_norm, norm = layer_norm_tokens(norm, x_norm), None
        if norm is not None:
            if self._ln_fold is None or self._ln_fold[0] is not norm or self._ln_fold[1].w.device != x_norm.device:
                wcat = torch.cat([self.to_q.weight.detach(), self.to_k.weight.detach(), self.to_v.weight.detach()], 0)
                self._ln_fold = (norm, K.LnFold(wcat, None, norm.weight, norm.bias, norm.eps, x_norm.device))
            qkv = K.gemm(x_norm, None, None, ln=self._ln_fold[1], ln_stats=stats)
        else:
            if self._qkv is None or self._qkv.device != x_norm.device:
                self._qkv = torch.cat([self._qk_weight(x_norm.dtype, x_norm.device),
                                       self.to_v.packed(x_norm.dtype, x_norm.device)[0]], 0).contiguous()
            shard = D.active_shard()
            if shard is not None:
                # frames split over ranks: every pixel attends over ALL frames of the clip.  K | V first, their all-gather POSTED,
                # then the Q projection -- it runs while RCCL moves the other ranks' K / V (same weights, same arithmetic per
                # output element as the fused q|k|v GEMM: the rows of the weight matrix are merely projected in two launches)
                inner = self.inner_dim
                kv_loc = K.gemm(x_norm, self._qkv[inner:])
                pend = shard.all_gather_frames_async(kv_loc.reshape(batch, clip, l, 2 * inner), tag="temporal_attn")
                q = K.gemm(x_norm, self._qkv[:inner])
                kv = pend.wait().reshape(batch * shard.clip_len, l, 2 * inner)
                out = torch.empty(n, l, inner, dtype=x_norm.dtype, device=x_norm.device)
                K.attn_temporal(q, kv[..., :inner], kv[..., inner:], out, batch=batch, clip_len=clip,
                                kv_frames=shard.clip_len, heads=self.heads, scale=self.scale)
                return self.to_out[0].apply(out, res=residual)
            qkv = K.gemm(x_norm, self._qkv)
        inner = self.inner_dim
        out = torch.empty(n, l, inner, dtype=x_norm.dtype, device=x_norm.device)
        shard = D.active_shard()
        if shard is not None:  # every pixel attends over ALL frames of the clip: gather this layer's K and V
            kv = shard.all_gather_frames(qkv[..., inner:].reshape(batch, clip, l, 2 * inner))
            kv = kv.reshape(batch * shard.clip_len, l, 2 * inner)
            K.attn_temporal(qkv[..., :inner], kv[..., :inner], kv[..., inner:], out, batch=batch, clip_len=clip,
                            kv_frames=shard.clip_len, heads=self.heads, scale=self.scale)
            return self.to_out[0].apply(out, res=residual)
        K.attn_temporal(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], out, batch=batch, clip_len=clip,
                        heads=self.heads, scale=self.scale)
        return self.to_out[0].apply(out, res=residual)

    def load_state_dict(self, *a, **k):  # packed weights / cached projections must follow the parameters
        self._qk = None
        self._qkv = None
        self._qkv_self = None
        self._ctx_kv = None
        self._xchain = {}
        self._ln_fold = None
        return super().load_state_dict(*a, **k)


class _ShardedKV:
    """K / V^T of a frame-sharded clip on an extended frame axis [left halo | own frames | right halo | anchors]: the
    neighbour frames a relative index reaches and the 'first' / 'mid' / 'last' anchors are fetched from their owners
    (attention.py:376-386 semantics: relative indices clamp to the CLIP ends, anchors are global frame numbers).
    `post(t)` starts the fetch for one operand and returns a handle; `extend(handle)` waits and assembles -- the caller puts
    independent work (the other projections) in between."""

    def __init__(self, shard, batch, clip, index_list):
        self.shard, self.batch, self.clip = shard, batch, clip
        rel = [i for i in index_list if not isinstance(i, str)]
        self.left, self.right = max([0] + [-i for i in rel]), max([0] + [i for i in rel])
        self.anchors = [a for a in dict.fromkeys(K.kv_slots([i], shard.clip_len)[1][0] for i in index_list if isinstance(i, str))]
        kabs, kval = [], []
        for i in index_list:
            if isinstance(i, str):
                kabs.append(1)
                kval.append(self.left + clip + self.right + self.anchors.index(K.kv_slots([i], shard.clip_len)[1][0]))
            else:
                kabs.append(0)
                kval.append(int(i))
        self.ext = dict(kv_slots_override=(kabs, kval), kv_clip_len=self.left + clip + self.right + len(self.anchors),
                        kv_frame_off=self.left)

    def _wanted(self, r):
        fr = self.shard.frames_of(r)
        return list(range(fr.start - self.left, fr.start)) + list(range(fr.stop, fr.stop + self.right)) + self.anchors

    def post(self, t):  # [B*clip, ...]
        t4 = t.reshape(self.batch, self.clip, *t.shape[1:])
        return t4, self.shard.fetch_frames_async(t4, self._wanted, tag="kv")

    def extend(self, handle):  # -> [B*(left+clip+right+len(anchors)), ...]
        t4, pend = handle
        got = pend.wait()
        e = torch.cat([got[:, :self.left], t4, got[:, self.left:]], dim=1)
        return e.reshape(self.batch * e.shape[1], *t4.shape[2:])


class SparseCausalAttention(CrossAttention):
    """attention.py:340-422 / attention_register.py:131-218: frame f attends the K/V of frames idx_j(f)."""

    def forward_self(self, x: Tokens, clip: int, index_list, residual=None, want_stats=False, ln_next=None, raw_out=False):
        n, lq, c = x.data.shape
        xn = x.data
        # head dims with a free MFMA contraction slot (SD-1.x: 40): the softmax scale and log2(e) go into Wq, q comes out of
        # the projection GEMM in the log2 domain and the flash kernel gets its running max for free (csrc/attn_flash.hip)
        d_head = self.inner_dim // self.heads
        folded = d_head % 16 != 0 and d_head % 8 == 0
        wqk = self._qk_weight(xn.dtype, xn.device, self.scale * _LOG2E if folded else 1.0)
        kw = dict(clip_len=clip, heads=self.heads, index_list=index_list, scale=self.scale, q_log2_scaled=folded)
        shard = D.active_shard()
        if shard is not None and len(index_list) > 0:
            # frames split over ranks: K first and its neighbour / anchor fetch POSTED, then V^T and its fetch, then the Q projection
            # -- both transfers run under it; the waits come last (the rows of the fused q|k weight are projected in two launches)
            skv = _ShardedKV(shard, n // clip, clip, index_list)
            kk = K.gemm(xn, wqk[self.inner_dim:])
            hk = skv.post(kk)
            vt = K.gemm_vt(xn, self.to_v.packed(xn.dtype, xn.device)[0], K.pad64(lq))
            hv = skv.post(vt)
            q = K.gemm(xn, wqk[: self.inner_dim])
            kk, vt = skv.extend(hk), skv.extend(hv)
            kw.update(skv.ext)
        else:
            wv = self.to_v.packed(xn.dtype, xn.device)[0]
            if QKV_FUSION and lq % 64 == 0 and (2 * self.inner_dim) % 64 == 0 and xn.is_contiguous():
                # ONE launch: [Wq ; Wk ; Wv] against the LayerNorm output, the V columns leaving transposed (fz_gemm_qkvt)
                c = self._qkv_self
                if c is None or c[0] is not wqk or c[1] is not wv:
                    c = self._qkv_self = (wqk, wv, torch.cat([wqk, wv], 0).contiguous())
                qk, vt = K.gemm_qkvt(xn, c[2], 2 * self.inner_dim)
            else:
                qk = K.gemm(xn, wqk)
                # V^T straight out of the projection GEMM (operands swapped: [N, C, L], rows zero-padded to a multiple of 64 keys)
                vt = K.gemm_vt(xn, wv, K.pad64(lq))
            q, kk = qk[..., : self.inner_dim], qk[..., self.inner_dim:]
        out = torch.empty(n, lq, self.inner_dim, dtype=xn.dtype, device=xn.device)
        n_kv = max(1, len(index_list))
        ctrl = self.controller
        plan = _plan_for(ctrl, False, self.place_in_unet, n, clip, self.heads, lq, n_kv * lq, xn.device)
        if plan is None:
            if lq > 32 ** 2:
                K.attn_self(q, kk, vt, out, mode=K.FZ_ATTN_FLASH, **kw)
            else:
                self._generic_controller_call(
                    ctrl, False, q, kk, vt, out, clip, lq, n_kv * lq,
                    lambda p: K.attn_self(q, kk, vt, out, mode=K.FZ_ATTN_CAPTURE, p=p, **kw),
                    lambda p: K.attn_self(q, None, vt, out, mode=K.FZ_ATTN_INJECT, p=p, **kw))
        elif plan.mode == K.FZ_ATTN_FLASH and plan.capture_first is None:
            K.attn_self(q, kk, vt, out, mode=K.FZ_ATTN_FLASH, **kw)  # nothing to capture or inject: one launch
        else:
            if plan.n_plain > 0:
                K.attn_self(q, kk, vt, out, mode=K.FZ_ATTN_FLASH, frame0=0, n_frames=plan.n_plain, **kw)
            if plan.n_plain < n:
                rest = dict(frame0=plan.n_plain, n_frames=n - plan.n_plain)
                if plan.capture_first is not None:  # edit pass with save_self_attention: keep the live map too
                    K.attn_self(q, kk, vt, out, mode=K.FZ_ATTN_CAPTURE, p=plan.capture_first, **rest, **kw)
                if plan.mode == K.FZ_ATTN_INJECT:
                    K.attn_self(q, kk if plan.row_mask is not None else None, vt, out, mode=K.FZ_ATTN_INJECT, p=plan.p,
                                row_mask=plan.row_mask, **rest, **kw)
                elif not (plan.mode == K.FZ_ATTN_FLASH and plan.capture_first is not None):
                    K.attn_self(q, kk, vt, out, mode=plan.mode, p=plan.p, **rest, **kw)
        if raw_out:  # the output projection rides in the launch that follows (forward_cross_after)
            return out
        return _out_proj(self.to_out[0], out, residual, want_stats, ln_next)


class _GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = _LinearParams(dim, inner * 2)
        self._packed = None

    def packed(self, dtype, device):
        """The projection's rows regrouped (32 h rows | 32 gate rows) for the GEGLU epilogue of fz_gemm."""
        if self._packed is None or self._packed[0].dtype != dtype or self._packed[0].device != device:
            w = self.proj.weight.detach().to(device=device, dtype=dtype)
            b = None if self.proj.bias is None else self.proj.bias.detach().to(device=device, dtype=dtype)
            self._packed = K.pack_geglu(w, b)
        return self._packed


class FeedForward(nn.Module):
    """diffusers FeedForward(activation_fn='geglu') [3P]: net = [GEGLU(dim, 4 dim), Dropout, Linear(4 dim, dim)]."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim, dim * mult), nn.Identity(), _LinearParams(dim * mult, dim)])
        self._ln_fold = None
        self._chain = None  # (device, fz_ff_chain's packed weight stream, b2)

    def _chain_pack(self, device):
        if self._chain is None or self._chain[0] != device:
            g, lin = self.net[0], self.net[2]
            w1 = g.proj.weight.detach().to(device=device, dtype=torch.float16)
            b1 = None if g.proj.bias is None else g.proj.bias.detach().to(device=device, dtype=torch.float16)
            w2 = lin.weight.detach().to(device=device, dtype=torch.float16)
            b2 = None if lin.bias is None else lin.bias.detach().to(device=device, dtype=torch.float16).contiguous()
            self._chain = (device, K.ff_chain_pack(w1, b1, w2), b2)
        return self._chain[1], self._chain[2]

    def load_state_dict(self, *a, **k):  # the packed stream must follow the parameters
        self._chain = None
        self._ln_fold = None
        return super().load_state_dict(*a, **k)

    def apply(self, x, res=None, norm=None, stats=None, want_stats=False, ln_next=None):
        """res + Linear(h * gelu(gate)): the 8C-wide GEGLU intermediate is never written (gate applied in the epilogue of the
        projection GEMM), the residual add rides in the epilogue of the output GEMM.  x: LayerNorm'ed, or RAW with
        (`norm`, `stats`) -- the LayerNorm then rides in the projection GEMM as well (fz_gemm_ln)."""
        g = self.net[0]
        fused = g.proj.weight.shape[0] % 64 == 0
        if isinstance(stats, Prenormed):
            x, norm, stats = stats.t, None, None
        if norm is not None and not (fused and _ln_ready(norm, stats, x)):
            x, norm = layer_norm_tokens(norm, x), None
        inner = g.proj.weight.shape[0] // 2
        if (FF_CHAIN and norm is None and not want_stats and x.dtype == torch.float16 and x.is_contiguous() and (res is None or res.is_contiguous())
                and D.active_shard() is None and K.ff_chain_preferred(x.numel() // x.shape[-1], x.shape[-1], inner)):
            packed, b2 = self._chain_pack(x.device)
            ln = None
            if ln_next is not None:
                gam, bet = ln_next.packed(x.device)
                ln = (gam, bet, ln_next.eps)
            y, yln = K.ff_chain(x, packed, b2, inner, res=res, ln=ln)
            return y, (None if yln is None else Prenormed(yln))
        if norm is not None:
            if getattr(self, "_ln_fold", None) is None or self._ln_fold[0] is not norm or self._ln_fold[1].w.device != x.device:
                self._ln_fold = (norm, K.LnFold(g.proj.weight, g.proj.bias, norm.weight, norm.bias, norm.eps, x.device,
                                                pack=K.pack_geglu))
            h = K.gemm(x, None, None, geglu=True, ln=self._ln_fold[1], ln_stats=stats)
        elif fused:
            w, b = g.packed(x.dtype, x.device)
            h = K.gemm(x, w, b, geglu=True)
        else:
            h = K.geglu(g.proj.apply(x))
        return _out_proj(self.net[2], h, res, want_stats, ln_next)


class SpatioTemporalTransformerBlock(nn.Module):
    """attention.py:147-337 with temporal_attention_position='after_feedforward' (the only one used)."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None, model_config: dict = {},
                 **unused):
        super().__init__()
        self.model_config = copy.deepcopy(model_config)
        if "least_sc_channel" in model_config and dim < model_config["least_sc_channel"]:
            self.model_config["SparseCausalAttention_index"] = []  # attention.py:171-173
        self.attn1 = SparseCausalAttention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm1 = _NormParams(dim)
        self.attn2 = CrossAttention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                                    dim_head=attention_head_dim)
        self.norm2 = _NormParams(dim)
        self.attn_temporal = CrossAttention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim)
        nn.init.zeros_(self.attn_temporal.to_out[0].weight.data)  # attention.py:224 (the bias stays random!)
        self.norm_temporal = _NormParams(dim)
        self.ff = FeedForward(dim)
        self.norm3 = _NormParams(dim)

    @property
    def sc_index(self):
        return self.model_config.get("SparseCausalAttention_index", [-1, "first"])  # attention.py:347 default

    def forward_tokens(self, x: Tokens, ctx, prenorm1=None):
        hs = x.data
        clip = x.f
        lnp = LN_FROM_PRODUCER and hs.shape[-1] == LN_FROM_PRODUCER_C and hs.dtype == torch.float16 and D.active_shard() is None
        # every `x = f(norm(x)) + x` of attention.py:295-337 ends in a GEMM: the residual add is that GEMM's epilogue -- and
        # so are the row statistics of the result, which let norm2 / norm3 / norm_temporal ride inside the GEMM that consumes
        # them (fz_gemm_ln; `st` is None where that form does not apply and the LayerNorm kernel runs instead).  norm1 stays a
        # kernel: its output also feeds the transposed V projection.
        n1 = prenorm1 if prenorm1 is not None else layer_norm_tokens(self.norm1, hs)
        want = LN_FUSION and hs.shape[-1] <= LN_FUSION_MAX_C
        if (lnp and XATTN_CHAIN_FRONT and not want and hs.is_contiguous()
                and self.attn2._chain_applies(hs.shape[0], hs.shape[1], hs.shape[2], hs.dtype, ctx)):
            # attn1.to_out + residual + norm2 -> attn2 (to_q, cross-attention, to_out) + residual + norm3: ONE launch behind the self-attention
            o1 = self.attn1.forward_self(x.like(n1), clip, self.sc_index, raw_out=True)
            hs, st = self.attn2.forward_cross_after(x, o1, self.attn1.to_out[0], hs, self.norm2, ctx, clip, self.norm3)
        else:
            hs, st = self.attn1.forward_self(x.like(n1), clip, self.sc_index, residual=hs, want_stats=want, ln_next=self.norm2 if lnp else None)
            hs, st = self.attn2.forward_cross(x.like(hs), ctx, clip, residual=hs, norm=self.norm2, stats=st, want_stats=want,
                                              ln_next=self.norm3 if lnp else None)
        hs, st = self.ff.apply(hs, res=hs, norm=self.norm3, stats=st, want_stats=want, ln_next=self.norm_temporal if lnp else None)
        hs = self.attn_temporal.forward_temporal(hs, x.b, clip, residual=hs, norm=self.norm_temporal, stats=st)
        return x.like(hs)


class SpatioTemporalTransformerModel(nn.Module):
    """attention.py:31-144: per-frame GroupNorm(eps 1e-6) -> 1x1 proj_in -> block -> 1x1 proj_out -> + residual."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 norm_num_groups=32, cross_attention_dim=None, model_config: dict = {}, **unused):
        super().__init__()
        assert num_layers == 1
        inner = num_attention_heads * attention_head_dim
        self.norm = _NormParams(in_channels, norm_num_groups, 1e-6)
        self.proj_in = _Conv1x1Params(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([SpatioTemporalTransformerBlock(
            inner, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim,
            model_config=model_config)])
        self.proj_out = _Conv1x1Params(inner, in_channels)

    def forward_tokens(self, x: Tokens, ctx) -> Tokens:
        h = group_norm_tokens(self.norm, x, span_frames=False, silu=False)
        blk = self.transformer_blocks[0]
        pre1 = None
        if (LN_FROM_PRODUCER and self.proj_in.weight.shape[0] == LN_FROM_PRODUCER_C and h.data.dtype == torch.float16 and h.data.is_contiguous()
                and D.active_shard() is None):
            y, pre1 = self.proj_in.apply_lnout(h.data, blk.norm1)  # proj_in + norm1 of the block in one launch
            h = h.like(y)
        else:
            h = h.like(self.proj_in.apply(h.data))
        h = blk.forward_tokens(h, ctx, prenorm1=pre1)
        from .resnet import GN_FROM_EPILOGUE
        if GN_FROM_EPILOGUE and D.active_shard() is None:
            # the next consumer is a GroupNorm (the following resnet's norm1 / conv_norm_out) with the UNet's group count: where the
            # projection runs on a 320-wide tile its epilogue writes that norm's statistics partials (fz_gemm_gn)
            y, part = self.proj_out.apply_gn(h.data, res=x.data, groups=self.norm.num_groups, rows_per_frame=x.data.shape[1])
            out = x.like(y)
            out.gn = None if part is None else (part, self.norm.num_groups)
            return out
        return x.like(self.proj_out.apply(h.data, res=x.data))


class _Conv1x1Params(nn.Module):
    """nn.Conv2d(k=1) parameters ([Cout, Cin, 1, 1]) applied as a GEMM on token-major data."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = nn.Parameter(torch.zeros(cout))
        self._packed = None

    def _pack(self, x):
        if self._packed is None or self._packed[0].dtype != x.dtype or self._packed[0].device != x.device:
            self._packed = (self.weight.detach().reshape(self.weight.shape[0], -1).to(device=x.device, dtype=x.dtype).contiguous(),
                            self.bias.detach().to(device=x.device, dtype=x.dtype))
        return self._packed

    def apply(self, x, res=None):
        w, b = self._pack(x)
        return K.gemm(x, w, b, res=res)

    def apply_lnout(self, x, norm):
        """apply() + LayerNorm `norm` of the result out of the same launch: (y, LN(y) or None)."""
        w, b = self._pack(x)
        g, be = norm.packed(x.device)
        return K.gemm_lnout(x, w, b, (g, be, norm.eps))

    def apply_gn(self, x, res, groups, rows_per_frame):
        """apply() + the GroupNorm(groups) statistics partials of the result out of the same launch: (y, partial or None)."""
        w, b = self._pack(x)
        return K.gemm_gn(x, w, b, res=res, gn_groups=groups, rows_per_frame=rows_per_frame)
