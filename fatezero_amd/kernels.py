"""Tensor-level wrappers over the C ABI (include/fatezero_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function below hands raw pointers and
sizes to libfatezero_hip.so.  Nothing in this file computes with torch ops -- if the native library is missing
the call raises (fatezero_amd._native.NativeLibraryError).
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as N

FZ_ATTN_FLASH, FZ_ATTN_CAPTURE, FZ_ATTN_INJECT = N.FZ_ATTN_FLASH, N.FZ_ATTN_CAPTURE, N.FZ_ATTN_INJECT
CROSS_P_STRIDE = N.FZ_CROSS_P_STRIDE
CROSS_KEYS = N.FZ_CROSS_MAX_KEYS
SUPPORTED_HEAD_DIMS = (16, 32, 40, 64, 80, 128, 160)


def _raw_stream(dev_index: int) -> int:
    """PyTorch's CURRENT HIP stream on that device as a raw hipStream_t (no Stream object is built: ~0.2 us per call)."""
    return torch._C._cuda_getCurrentRawStream(dev_index)


def refresh_stream():
    """Kept for callers of earlier revisions: the launch stream is read from PyTorch at EVERY launch now (per device), so there is
    nothing to refresh."""


_launches = [0]


def launch_count() -> int:
    """Kernel launches issued through this module so far (every wrapper asks `_stream` for the launch stream exactly once per
    launch): dist.Pending uses it to tell an exchange that had compute queued behind it from one that was waited for at once."""
    return _launches[0]


def _stream(t: torch.Tensor):
    """The stream a launch on `t` goes to: PyTorch's current stream OF t's DEVICE, read per launch -- a call under
    `torch.cuda.stream(s)` launches on s, a tensor on cuda:1 launches on cuda:1's stream whatever ran before.  The split-K /
    GroupNorm scratch is keyed on this same handle (_scratch_key)."""
    _launches[0] += 1
    if t.is_cuda:
        return C.c_void_p(_raw_stream(t.device.index))
    if not N.is_test_backend():
        raise RuntimeError("fatezero_amd kernels run on the GPU only (CPU tensors are accepted only by the "
                           "emulation backend used in tests)")
    return C.c_void_p(0)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk16(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.float16, t.dtype
            assert t.data_ptr() % 16 == 0, "16-byte alignment required"


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


# ------------------------------------------------------------------------------------------------------------
# sparse-causal self attention
# ------------------------------------------------------------------------------------------------------------
def kv_slots(index_list: Sequence, clip_len: int) -> Tuple[List[int], List[int]]:
    """SparseCausalAttention_index -> per kv slot (is_absolute, frame or offset), attention_register.py:162-183.
    An empty list means 'own frame' (attention.py:171-173 sets it for dim < least_sc_channel)."""
    if len(index_list) == 0:
        return [0], [0]
    kabs, kval = [], []
    for index in index_list:
        if isinstance(index, str):
            if index == "first":
                fr = 0
            elif index == "last":
                fr = clip_len - 1
            elif index in ("mid", "middle"):
                fr = int((clip_len - 1) // 2)
            else:
                raise ValueError(f"unknown SparseCausalAttention_index entry {index!r}")
            kabs.append(1)
            kval.append(fr)
        else:
            assert isinstance(index, int), "relative index must be int"
            kabs.append(0)
            kval.append(int(index))
    return kabs, kval


def attn_self(q: torch.Tensor, k: Optional[torch.Tensor], vt: torch.Tensor, out: torch.Tensor, *, clip_len: int,
              heads: int, index_list: Sequence, mode: int = FZ_ATTN_FLASH, frame0: int = 0, n_frames: Optional[int] = None,
              p: Optional[torch.Tensor] = None, p_frame_off: int = 0, row_mask: Optional[torch.Tensor] = None,
              mask_frame_off: int = 0, scale: Optional[float] = None, k_head_major: Optional[torch.Tensor] = None,
              q_log2_scaled: bool = False, kv_slots_override: Optional[Tuple[List[int], List[int]]] = None,
              kv_clip_len: int = 0, kv_frame_off: int = 0):
    """q,k,out: [N, L, >=C] views with row stride (token-major); vt: [N, C, Lpad]; p: [Fp, heads, Lq, Lk] fp16.
    k_head_major (optional): K as a contiguous [N, heads, L, d] tensor (fully coalesced key tiles) instead of `k`.

    q_log2_scaled: q already carries scale*log2(e) (folded into the projection weight), see include/fatezero_hip.h.
    kv_clip_len / kv_frame_off / kv_slots_override: frame-sharded clips -- k / vt carry kv_clip_len frames per batch
    element (halos + own frames + anchors, fatezero_amd/dist.py) and the slots are given on that extended axis.

    Frames frame0 .. frame0+n_frames-1 of q/out are processed; k/vt are indexed by source frame.
    """
    N_, lq, c = q.shape
    d_head = c // heads
    assert d_head in SUPPORTED_HEAD_DIMS, d_head
    _chk16(q, k, vt, out, p)
    n_frames = N_ - frame0 if n_frames is None else n_frames
    kabs, kval = kv_slots(index_list, clip_len) if kv_slots_override is None else kv_slots_override
    d = N.FzAttnSelfDesc()
    d.kv_clip_len, d.kv_frame_off = kv_clip_len, kv_frame_off
    d.n_frames, d.frame0, d.clip_len, d.heads, d.head_dim = n_frames, frame0, clip_len, heads, d_head
    d.lq, d.lkf, d.n_kv = lq, (lq if k is None else k.shape[1]), len(kabs)
    for j in range(len(kabs)):
        d.kv_abs[j], d.kv_val[j] = kabs[j], kval[j]
    d.scale = float(scale if scale is not None else d_head ** -0.5)
    d.mode = mode
    d.q_log2_scaled = 1 if q_log2_scaled else 0
    assert q.stride(2) == 1 and out.stride(2) == 1 and vt.stride(2) == 1
    d.q_frame_stride, d.q_row_stride = q.stride(0), q.stride(1)
    if k_head_major is not None:
        assert k_head_major.is_contiguous() and k_head_major.shape[1] == heads and k_head_major.shape[3] == d_head
        d.lkf = k_head_major.shape[2]
        d.k_frame_stride, d.k_row_stride, d.k_head_stride = k_head_major.stride(0), d_head, k_head_major.stride(1)
        k = k_head_major
    elif k is not None:
        assert k.stride(2) == 1
        d.k_frame_stride, d.k_row_stride = k.stride(0), k.stride(1)
    d.vt_frame_stride, d.vt_chan_stride = vt.stride(0), vt.stride(1)
    assert vt.shape[2] >= pad64(d.lkf), (vt.shape, d.lkf)
    d.o_frame_stride, d.o_row_stride = out.stride(0), out.stride(1)
    if p is not None:
        assert p.stride(3) == 1 and p.shape[3] == d.n_kv * d.lkf, (p.shape, d.n_kv, d.lkf)
        d.p_frame_stride, d.p_head_stride, d.p_row_stride = p.stride(0), p.stride(1), p.stride(2)
    d.p_frame_off, d.mask_frame_off = p_frame_off, mask_frame_off
    if row_mask is not None:
        assert row_mask.dtype == torch.float32 and row_mask.is_contiguous() and row_mask.shape[-1] == lq
    N.check(N.lib().fz_attn_self(C.byref(d), _ptr(q), _ptr(k), _ptr(vt), _ptr(out), _ptr(p), _ptr(row_mask), _stream(q)),
            "fz_attn_self")
    return out


# ------------------------------------------------------------------------------------------------------------
# cross attention
# ------------------------------------------------------------------------------------------------------------
def attn_cross(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, *, clip_len: int, heads: int,
               lk: int, mode: int = FZ_ATTN_FLASH, frame0: int = 0, n_frames: Optional[int] = None,
               p: Optional[torch.Tensor] = None, p_frame_off: int = 0, mapper_t: Optional[torch.Tensor] = None,
               coef: Optional[torch.Tensor] = None, cur_out: Optional[torch.Tensor] = None, scale: Optional[float] = None):
    """q,out: [N, L, >=C]; k: [B, >=lk, >=C]; vt: [B, C, 96]; p / cur_out: [Fp, heads, Lq, 80] fp16 (row stride 80)."""
    N_, lq, c = q.shape
    d_head = c // heads
    assert d_head in SUPPORTED_HEAD_DIMS, d_head
    _chk16(q, k, vt, out, p, mapper_t, cur_out)
    d = N.FzAttnCrossDesc()
    d.n_frames = N_ - frame0 if n_frames is None else n_frames
    d.frame0, d.clip_len, d.heads, d.head_dim, d.lq, d.lk = frame0, clip_len, heads, d_head, lq, lk
    d.scale = float(scale if scale is not None else d_head ** -0.5)
    d.mode = mode
    d.q_frame_stride, d.q_row_stride = q.stride(0), q.stride(1)
    d.k_batch_stride, d.k_row_stride = k.stride(0), k.stride(1)
    d.vt_batch_stride, d.vt_chan_stride = vt.stride(0), vt.stride(1)
    assert vt.shape[2] >= CROSS_KEYS and vt.stride(2) == 1
    d.o_frame_stride, d.o_row_stride = out.stride(0), out.stride(1)
    if p is not None:
        assert p.stride(3) == 1 and p.stride(2) >= CROSS_P_STRIDE
        d.p_frame_stride, d.p_head_stride, d.p_row_stride = p.stride(0), p.stride(1), p.stride(2)
        if cur_out is not None:
            assert cur_out.stride() == p.stride()
    d.p_frame_off = p_frame_off
    d.store_cur = 1 if cur_out is not None else 0
    if coef is not None:
        assert coef.dtype == torch.float32 and coef.is_contiguous() and coef.numel() == 2 * CROSS_KEYS
    if mapper_t is not None:
        assert mapper_t.is_contiguous() and tuple(mapper_t.shape) == (CROSS_KEYS, CROSS_KEYS)
    N.check(N.lib().fz_attn_cross(C.byref(d), _ptr(q), _ptr(k), _ptr(vt), _ptr(out), _ptr(p), _ptr(mapper_t), _ptr(coef),
                                  _ptr(cur_out), _stream(q)), "fz_attn_cross")
    return out


def attn_temporal(q, k, v, out, *, batch: int, clip_len: int, heads: int, scale: Optional[float] = None,
                  kv_frames: Optional[int] = None):
    """q,out: [B*clip_len, tokens, >=C]; k,v: [B*kv_frames, tokens, >=C] token-major views (kv_frames defaults to clip_len;
    it is larger when the clip is frame-sharded and k / v were all-gathered)."""
    _, tokens, c = q.shape
    d_head = c // heads
    kv_frames = clip_len if kv_frames is None else kv_frames
    _chk16(q, k, v, out)
    assert k.stride(1) == v.stride(1) and q.stride(0) == tokens * q.stride(1) and k.stride(0) == tokens * k.stride(1)
    assert v.stride(0) == tokens * v.stride(1) and out.stride(0) == tokens * out.stride(1)
    assert q.shape[0] == batch * clip_len and k.shape[0] == batch * kv_frames
    N.check(N.lib().fz_attn_temporal_ex(_ptr(q), _ptr(k), _ptr(v), _ptr(out), batch, clip_len, kv_frames, tokens, heads,
                                        d_head, q.stride(1), k.stride(1), out.stride(1),
                                        float(scale if scale is not None else d_head ** -0.5), _stream(q)),
            "fz_attn_temporal_ex")
    return out


# ------------------------------------------------------------------------------------------------------------
def blend_mask(maps: List[torch.Tensor], alpha: torch.Tensor, th: float, out_hw: Tuple[int, int], *, or_with_first: bool,
               out: Optional[torch.Tensor] = None):
    """maps: list of fp16 [P, F, heads, r*r, >=80] views (row stride >= 80); alpha float [P, 80] -> float [P,F,h,w]."""
    m0 = maps[0]
    P_, F_, heads, npix, _ = m0.shape
    res = int(round(npix ** 0.5))
    assert res * res == npix, "the shape of attention map must be a square"
    for m in maps:
        assert m.shape[:4] == m0.shape[:4] and m.stride() == m0.stride() and m.dtype == torch.float16
        assert m.stride(4) == 1 and m.stride(2) == npix * m.stride(3) and m.stride(1) == heads * m.stride(2)
    h, w = out_hw
    if out is None:
        out = torch.empty(P_, F_, h, w, dtype=torch.float32, device=m0.device)
    arr = (C.c_void_p * len(maps))(*[m.data_ptr() for m in maps])
    assert alpha.dtype == torch.float32 and alpha.is_contiguous() and tuple(alpha.shape) == (P_, 80)
    N.check(N.lib().fz_blend_mask(arr, len(maps), P_, m0.stride(0), F_, heads, res, m0.stride(3), _ptr(alpha), float(th),
                                  h, w, 1 if or_with_first else 0, _ptr(out), None, _stream(m0)), "fz_blend_mask")
    return out


_gn_scratch = {}
_gn_plans = {}
_scratch_gen = [0]  # bumped whenever a scratch buffer (GroupNorm partials, split-K slabs) is allocated, regrown or evicted


def scratch_generation() -> int:
    """Changes whenever a process-wide scratch buffer a launch may point at was (re)allocated or released: a recorded issue plan
    (fatezero_amd/issue.py) holds raw pointers into those buffers and is only valid for the generation it was recorded under."""
    return _scratch_gen[0]


def release_scratch():
    """Give every scratch buffer (all devices, all streams) back to the allocator, e.g. between jobs of very different sizes.  Launch
    descriptors and issue plans that point into them are dropped / recorded again (scratch_generation changes)."""
    if _ws or _gn_scratch:
        _scratch_gen[0] += 1
    _ws.clear()
    _gn_scratch.clear()
    _gn_plans.clear()
    del _scratch_lru[:]


def scratch_buffers(t_or_device):
    """The scratch tensors of the caller's (device, stream): whoever keeps raw pointers into them keeps these alive."""
    key = _scratch_key(t_or_device)
    return [b for b in (_ws.get(key), _gn_scratch.get(key)) if b is not None]


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, span: int, groups: int, eps: float,
              silu: bool, out: Optional[torch.Tensor] = None):
    """x: [N, tokens, C] contiguous fp16; statistics shared by `span` consecutive frames."""
    key = (x.shape, span, groups, _scratch_key(x))
    plan = _gn_plans.get(key)
    if plan is None:  # first use of this signature: validate, size the scratch
        n, tokens, c = x.shape
        assert x.is_contiguous()
        _chk16(x, gamma, beta)
        chunks = N.lib().fz_groupnorm_chunks(tokens, c)
        need = n * chunks * groups * 3 + (n // span) * groups * 2
        skey = _scratch_key(x)
        buf = _gn_scratch.get(skey)
        if buf is None or buf.numel() < need:
            buf = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=x.device)
            _gn_scratch[skey] = buf
            _gn_plans.clear()  # plans hold the scratch pointer
            _scratch_gen[0] += 1
        plan = _gn_plans[key] = (n, tokens, c, buf.data_ptr(), buf)
    n, tokens, c, scratch, _ = plan
    # cheap per-call checks (the plan is keyed on the shape only: a later non-contiguous / non-fp16 / misaligned tensor of the same
    # shape must not reach the kernel)
    if not x.is_contiguous() or x.dtype != torch.float16 or gamma.dtype != torch.float16 or beta.dtype != torch.float16 or \
            ((x.data_ptr() | gamma.data_ptr() | beta.data_ptr()) & 15):
        raise ValueError("fz_groupnorm: x must be contiguous fp16, gamma / beta fp16, all 16-byte aligned")
    if out is None:
        out = torch.empty_like(x)
    elif not out.is_contiguous() or out.dtype != torch.float16 or out.shape != x.shape or (out.data_ptr() & 15):
        raise ValueError("fz_groupnorm: out must be a contiguous fp16 tensor of x's shape")
    rc = N.lib().fz_groupnorm(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), n, span, tokens, c, groups, eps,
                              1 if silu else 0, scratch, _stream(x))
    if rc:
        N.check(rc, "fz_groupnorm")
    return out


def groupnorm_cat(x1: torch.Tensor, x2: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, span: int, groups: int, eps: float,
                  silu: bool) -> torch.Tensor:
    """GroupNorm(+SiLU) of torch.cat([x1, x2], -1) without building the concatenation (fz_groupnorm_cat): x1 [N, tokens, C1],
    x2 [N, tokens, C2] contiguous fp16 -> [N, tokens, C1 + C2]."""
    n, tokens, c1 = x1.shape
    c2 = x2.shape[2]
    if not (x1.is_contiguous() and x2.is_contiguous() and x2.shape[:2] == x1.shape[:2] and x1.dtype == x2.dtype == torch.float16
            and gamma.dtype == beta.dtype == torch.float16 and gamma.numel() == c1 + c2):
        raise ValueError("fz_groupnorm_cat: two contiguous fp16 tensors [N, tokens, C1] / [N, tokens, C2], gamma / beta over C1 + C2")
    _chk16(x1, x2, gamma, beta)
    c = c1 + c2
    chunks = N.lib().fz_groupnorm_chunks(tokens, c)
    need = n * chunks * groups * 3 + (n // span) * groups * 2
    skey = _scratch_key(x1)
    buf = _gn_scratch.get(skey)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=x1.device)
        _gn_scratch[skey] = buf
        _gn_plans.clear()  # plans hold the scratch pointer
        _scratch_gen[0] += 1
    out = torch.empty(n, tokens, c, dtype=torch.float16, device=x1.device)
    N.check(N.lib().fz_groupnorm_cat(x1.data_ptr(), c1, x2.data_ptr(), c2, out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), n, span,
                                     tokens, groups, eps, 1 if silu else 0, buf.data_ptr(), _stream(x1)), "fz_groupnorm_cat")
    return out


def groupnorm_stats(x: torch.Tensor, *, groups: int) -> torch.Tensor:
    """Welford partials (count, mean, M2) of this rank's frames: float [N, G, chunks, 3] (fz_groupnorm_stats)."""
    n, tokens, c = x.shape
    assert x.is_contiguous()
    _chk16(x)
    chunks = N.lib().fz_groupnorm_chunks(tokens, c)
    partial = torch.empty(n, groups, chunks, 3, dtype=torch.float32, device=x.device)
    N.check(N.lib().fz_groupnorm_stats(_ptr(x), n, tokens, c, groups, _ptr(partial), _stream(x)), "fz_groupnorm_stats")
    return partial


def groupnorm_apply(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, partial_all: torch.Tensor, *, span: int,
                    groups: int, eps: float, silu: bool, out: Optional[torch.Tensor] = None):
    """Normalise the local frames x [N, tokens, C] with statistics merged from partial_all
    [stat_sets, frames_per_set, G, chunks, 3] (the partials of ALL ranks' frames); frame n uses stat set n // span."""
    n, tokens, c = x.shape
    assert x.is_contiguous() and partial_all.is_contiguous() and partial_all.dtype == torch.float32
    sets, per_set = partial_all.shape[0], partial_all.shape[1]
    assert n // span == sets and partial_all.shape[3] == N.lib().fz_groupnorm_chunks(tokens, c)
    _chk16(x, gamma, beta)
    if out is None:
        out = torch.empty_like(x)
    stats = torch.empty(sets, groups, 2, dtype=torch.float32, device=x.device)
    N.check(N.lib().fz_groupnorm_apply(_ptr(x), _ptr(out), _ptr(gamma), _ptr(beta), n, span, tokens, c, groups, float(eps),
                                       1 if silu else 0, _ptr(partial_all), sets, per_set, _ptr(stats), _stream(x)),
            "fz_groupnorm_apply")
    return out


def groupnorm_from_partial(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, partial: torch.Tensor, *, span: int, groups: int,
                           eps: float, silu: bool) -> torch.Tensor:
    """GroupNorm(+SiLU) of x [N, tokens, C] from Welford partials [N, groups, chunks, 3] a producer's epilogue wrote (temporal_conv3 /
    gemm_gn with gn_groups): merge + normalise, no statistics pass over x (fz_groupnorm_from_partials)."""
    n, tokens, c = x.shape
    if not (x.is_contiguous() and x.dtype == torch.float16 and partial.is_contiguous() and partial.dtype == torch.float32
            and partial.dim() == 4 and partial.shape[0] == n and partial.shape[1] == groups and partial.shape[3] == 3 and n % span == 0):
        raise ValueError("fz_groupnorm_from_partials: x [N, tokens, C] contiguous fp16, partial [N, groups, chunks, 3] fp32")
    _chk16(x, gamma, beta)
    out = torch.empty_like(x)
    stats = torch.empty(n // span, groups, 2, dtype=torch.float32, device=x.device)
    N.check(N.lib().fz_groupnorm_from_partials(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), n, span, tokens, c, groups,
                                               float(eps), 1 if silu else 0, partial.data_ptr(), partial.shape[2], stats.data_ptr(),
                                               _stream(x)), "fz_groupnorm_from_partials")
    return out


def gemm_gn(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, res: Optional[torch.Tensor] = None, gn_groups: int,
            rows_per_frame: int):
    """y = x @ w^T + bias (+ res) as K.gemm, plus the GroupNorm(gn_groups) Welford partials of y [rows / rows_per_frame, gn_groups,
    rows_per_frame / 128, 3] out of the same launch (fz_gemm_gn); returns (y, partial or None)."""
    k, o = x.shape[-1], w.shape[0]
    rows = x.numel() // k
    if not gn_epilogue_ok(rows_per_frame, o, gn_groups) or rows % rows_per_frame or not x.is_contiguous():
        return _gemm_unwrapped(x, w, bias, res=res), None
    _chk16(x, w, bias, res)
    y = torch.empty(tuple(x.shape[:-1]) + (o,), dtype=torch.float16, device=x.device)
    if res is not None and (not res.is_contiguous() or res.shape != y.shape):
        return _gemm_unwrapped(x, w, bias, res=res), None
    d = N.FzGemmDesc()
    d.rows, d.in_features, d.out_features = rows, k, o
    d.ldx, d.ldw, d.ldy, d.ldres = k, w.stride(0), o, o
    d.batch, d.epilogue = 1, N.FZ_GEMM_PLAIN
    partial = torch.empty(rows // rows_per_frame, gn_groups, rows_per_frame // 128, 3, dtype=torch.float32, device=x.device)
    rc = N.lib().fz_gemm_gn(C.byref(d), x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                            None if res is None else res.data_ptr(), None, y.data_ptr(), partial.data_ptr(), gn_groups, rows_per_frame,
                            _stream(x))
    if rc == N.FZ_GEMM_NO_STATS:
        return y, None
    if rc:
        N.check(rc, "fz_gemm_gn")
    return y, partial


def pack_conv3x3_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight [Cout, Cin, 3, 3] -> [Cout, 9, Cin] fp16 (k = tap*Cin + ci contiguous per cout)."""
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1]).to(torch.float16).contiguous()


_ws = {}


def pack_geglu(w: torch.Tensor, b: Optional[torch.Tensor]):
    """GEGLU projection weight [2*inner, K] (rows [h ; gate], diffusers GEGLU.proj) -> rows regrouped so that every 64-row
    group is 32 h rows followed by the 32 matching gate rows (include/fatezero_hip.h, FZ_GEMM_GEGLU)."""
    two, k = w.shape
    inner = two // 2
    assert inner % 32 == 0, inner
    wp = torch.stack([w[:inner].reshape(inner // 32, 32, k), w[inner:].reshape(inner // 32, 32, k)], 1).reshape(two, k)
    bp = None
    if b is not None:
        bp = torch.stack([b[:inner].reshape(inner // 32, 32), b[inner:].reshape(inner // 32, 32)], 1).reshape(two)
    return wp.contiguous(), (None if bp is None else bp.contiguous())


_WS_FLOATS = 64 << 20  # 256 MB of fp32 split-K scratch per device, allocated on first need


_SCRATCH_STREAMS_MAX = 4  # per-stream scratch sets kept alive (least recently used beyond that are released)
_scratch_lru = []


def _scratch_key(t_or_device):
    """Scratch buffers (split-K slabs, GroupNorm partials) are per (device, launch stream) -- the SAME raw handle `_stream` hands to
    the launch: two streams that both take a split-K path must not share the fp32 slabs.  At most _SCRATCH_STREAMS_MAX stream sets
    stay allocated; an evicted set goes back to PyTorch's caching allocator, which hands a block to another stream only after the
    work queued on the stream that allocated it -- the stream the scratch was used on -- has run."""
    dev = t_or_device.device if isinstance(t_or_device, torch.Tensor) else torch.device(t_or_device)
    if dev.type != "cuda":
        return (dev, 0)
    key = (dev, _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device()))
    if not _scratch_lru or _scratch_lru[-1] != key:
        if key in _scratch_lru:
            _scratch_lru.remove(key)
        _scratch_lru.append(key)
        while len(_scratch_lru) > _SCRATCH_STREAMS_MAX:
            old = _scratch_lru.pop(0)
            if _ws.pop(old, None) is not None:
                _scratch_gen[0] += 1
            if _gn_scratch.pop(old, None) is not None:
                _gn_plans.clear()  # plans hold the scratch pointer
                _scratch_gen[0] += 1
    return key


def _ws_ptr(device):
    key = _scratch_key(device)
    buf = _ws.get(key)
    if buf is None:
        buf = _ws[key] = torch.empty(_WS_FLOATS, dtype=torch.float32, device=device)
        _scratch_gen[0] += 1
    return buf.data_ptr()


_gemm_plans = {}


class LnFold:
    """A LayerNorm folded into the Linear that consumes it (fz_gemm_ln): w = gamma (.) W in fp16, c1 = row sums of that w,
    c0 = W beta + bias (fp32), eps.  `pack` (optional) reorders rows / per-row vectors the way the GEMM epilogue expects them
    (pack_geglu for the GEGLU projection)."""
    __slots__ = ("w", "c1", "c0", "eps")

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                 device, pack=None):
        w32 = weight.detach().float().cpu()
        wg = (w32 * gamma.detach().float().cpu()[None, :]).half()
        c0 = w32 @ beta.detach().float().cpu()
        if bias is not None:
            c0 = c0 + bias.detach().float().cpu()
        c1 = wg.float().sum(1)  # of the ROUNDED weights: what the MFMA actually sums
        if pack is not None:
            _, c0 = pack(wg, c0)
            wg, c1 = pack(wg, c1)
        self.w = wg.to(device).contiguous()
        self.c1 = c1.float().to(device).contiguous()
        self.c0 = c0.float().to(device).contiguous()
        self.eps = float(eps)


def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, res: Optional[torch.Tensor] = None,
         res2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, geglu: bool = False, tile_cfg: int = 0,
         split_k: int = 0, ln: Optional[LnFold] = None, ln_stats: Optional[torch.Tensor] = None, want_stats: bool = False):
    """y[..., o] = x[..., :] @ w[o, :] + bias (+ res) (+ res2); x: [..., K] with unit channel stride and ONE row stride
    (a channel slice of a token-major tensor is fine); w: [O, K] fp16 (GEGLU: packed by pack_geglu); res / out: [..., O'].
    The descriptor of a call signature (shapes / strides / flags) is validated and built once and then reused: per call only
    the pointers change.

    LayerNorm fusion (fz_gemm_ln):  ln + ln_stats -> x is the RAW LayerNorm input, `ln` the folded weights (w is ignored) and
    ln_stats the per-row block sums [rows, K / 64, 2] its producer wrote;  want_stats=True -> returns (y, stats) where stats are
    the block sums of y's rows [rows, O / 64, 2] for the next LayerNorm, or None when the library split K for this shape."""
    if ln is not None:
        w = ln.w
    key = (x.shape, x.stride(), w.shape, w.stride(0), geglu, tile_cfg, split_k, x.device,
           None if res is None else res.stride(), res2 is not None, None if out is None else (out.shape, out.stride()))
    plan = _gemm_plans.get(key)
    if plan is None:
        plan = _gemm_plans[key] = _gemm_plan(x, w, bias, res, res2, out, geglu, tile_cfg, split_k)
    d_ref, out_shape, want_ws, _keep = plan
    if out is None:
        out = torch.empty(out_shape, dtype=torch.float16, device=x.device)
    ptrs = x.data_ptr() | w.data_ptr() | out.data_ptr()
    for t in (bias, res, res2):  # the plan is built once per signature: pointers and dtypes of the optional operands change per call
        if t is not None:
            if t.dtype != torch.float16:
                raise ValueError("fz_gemm: bias / res / res2 must be fp16")
            ptrs |= t.data_ptr()
    if (ptrs & 15) or x.dtype != torch.float16 or w.dtype != torch.float16:
        raise ValueError("fz_gemm operands must be fp16 and 16-byte aligned")
    if res is not None and res2 is not None and res2.stride() != res.stride():
        raise ValueError("fz_gemm: res2 must have the layout of res")
    if ln is None and not want_stats:
        rc = N.lib().fz_gemm(d_ref, x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                             None if res is None else res.data_ptr(), None if res2 is None else res2.data_ptr(), out.data_ptr(),
                             _ws_ptr(x.device) if want_ws else None, _stream(x))
        if rc:
            N.check(rc, "fz_gemm")
        return out
    lnd = N.FzGemmLn()
    stats = None
    if ln is not None:
        rows = x.numel() // x.shape[-1]
        assert ln_stats is not None and ln_stats.dtype == torch.float32 and ln_stats.is_contiguous()
        assert ln_stats.numel() == rows * (x.shape[-1] // 64) * 2, (ln_stats.shape, x.shape)
        lnd.stats_in, lnd.c1, lnd.c0, lnd.eps = ln_stats.data_ptr(), ln.c1.data_ptr(), ln.c0.data_ptr(), ln.eps
    if want_stats:
        rows = out.numel() // out.shape[-1]
        stats = torch.empty(rows, out.shape[-1] // 64, 2, dtype=torch.float32, device=x.device)
        lnd.stats_out = stats.data_ptr()
    rc = N.lib().fz_gemm_ln(d_ref, C.byref(lnd), x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                            None if res is None else res.data_ptr(), None if res2 is None else res2.data_ptr(), out.data_ptr(),
                            _ws_ptr(x.device) if want_ws else None, _stream(x))
    if rc == N.FZ_GEMM_NO_STATS:
        stats = None
    elif rc:
        N.check(rc, "fz_gemm_ln")
    return (out, stats) if want_stats else out


def gemm_lnout(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], ln, *, res: Optional[torch.Tensor] = None,
               res2: Optional[torch.Tensor] = None, tile_cfg: int = 0, split_k: int = 0):
    """y = x @ w^T + bias (+ res) (+ res2) as K.gemm, plus LN(y) out of the same launch where the library's launch for the shape holds whole rows
    (fz_gemm_lnout: 320 output channels on a 320-wide tile, no split-K); ln = (gamma, beta, eps) fp16.  Returns (y, y_ln) -- y_ln is None where
    the epilogue form does not apply (y is complete: run K.layernorm)."""
    k, o = x.shape[-1], w.shape[0]
    rows = x.numel() // k
    gam, bet, eps = ln
    if (o != 320 or x.dtype != torch.float16 or not x.is_contiguous() or (res is not None and (not res.is_contiguous() or res.shape[-1] != o))
            or (res2 is not None and (not res2.is_contiguous() or res2.shape[-1] != o)) or gam.dtype != torch.float16):
        return _gemm_unwrapped(x, w, bias, res=res, res2=res2), None
    _chk16(x, w, bias, res, res2, gam, bet)
    y = torch.empty(tuple(x.shape[:-1]) + (o,), dtype=torch.float16, device=x.device)
    yln = torch.empty_like(y)
    d = N.FzGemmDesc()
    d.rows, d.in_features, d.out_features = rows, k, o
    d.ldx, d.ldw, d.ldy, d.ldres = k, w.stride(0), o, o
    d.batch, d.epilogue, d.tile_cfg, d.split_k = 1, N.FZ_GEMM_PLAIN, tile_cfg, split_k
    want_ws = o % 4 == 0 and (k >= 1024 or split_k > 1) and 2 * rows * o <= _WS_FLOATS
    if want_ws:
        d.workspace_floats = _WS_FLOATS
    rc = N.lib().fz_gemm_lnout(C.byref(d), x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(res), _ptr(res2), y.data_ptr(), gam.data_ptr(),
                               bet.data_ptr(), float(eps), yln.data_ptr(), o, _ws_ptr(x.device) if want_ws else None, _stream(x))
    if rc == N.FZ_GEMM_NO_STATS:
        return y, None
    if rc:
        N.check(rc, "fz_gemm_lnout")
    return y, yln


_gemm_unwrapped = gemm  # (gemm_gn's fallback: a harness that wraps K.gemm -- bench.py's timers / launch log -- must see ONE call, gemm_gn's)


def gemm_batched(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None):
    """y[b] = x[b] @ w[b]^T for every batch element in ONE launch: x [B, rows, K], w [B, O, K], y [B, rows, O]; fp16, unit
    innermost strides, one row stride and one batch stride each (views are fine).  The two products of a single-head attention
    block (scores = q k^T, out = P (V^T)^T): FzGemmDesc.w_batch_stride."""
    b, rows, k = x.shape
    o = w.shape[1]
    assert w.shape == (b, o, k) and x.stride(2) == 1 and w.stride(2) == 1 and x.dtype == w.dtype == torch.float16
    if out is None:
        out = torch.empty(b, rows, o, dtype=torch.float16, device=x.device)
    assert out.shape == (b, rows, o) and out.stride(2) == 1 and out.dtype == torch.float16
    for t in (x, w, out):
        if (t.data_ptr() & 15) or (t.stride(0) % 8) or (t.stride(1) % 8):
            raise ValueError("fz_gemm (batched): 16-byte aligned operands, row / batch strides multiples of 8 halves")
    d = N.FzGemmDesc()
    d.rows, d.in_features, d.out_features = rows, k, o
    d.ldx, d.ldw, d.ldy = x.stride(1), w.stride(1), out.stride(1)
    d.batch, d.epilogue = b, N.FZ_GEMM_PLAIN
    d.x_batch_stride, d.w_batch_stride, d.y_batch_stride = x.stride(0), w.stride(0), out.stride(0)
    want_ws = o % 4 == 0 and 2 * b * rows * o <= _WS_FLOATS
    d.workspace_floats = _WS_FLOATS if want_ws else 0
    rc = N.lib().fz_gemm(C.byref(d), x.data_ptr(), w.data_ptr(), None, None, None, out.data_ptr(),
                         _ws_ptr(x.device) if want_ws else None, _stream(x))
    if rc:
        N.check(rc, "fz_gemm (batched)")
    return out


def _gemm_plan(x, w, bias, res, res2, out, geglu, tile_cfg, split_k):
    k = x.shape[-1]
    o = w.shape[0]
    ow = o // 2 if geglu else o
    rows = x.numel() // k
    assert x.stride(-1) == 1 and w.stride(1) == 1 and w.shape[1] == k
    ldx = x.stride(-2) if x.dim() > 1 else k
    for d in range(x.dim() - 2):  # leading dims must collapse onto the row stride
        assert x.stride(d) == x.stride(d + 1) * x.shape[d + 1], "x must have a single row stride"
    _chk16(x, w, bias, res, res2)
    out_shape = tuple(x.shape[:-1]) + (ow,)
    ldy = ow
    if out is not None:
        assert out.stride(-1) == 1 and out.shape[-1] == ow and out.numel() // ow == rows and out.dtype == torch.float16
        ldy = out.stride(-2) if out.dim() > 1 else ow
    d = N.FzGemmDesc()
    d.rows, d.in_features, d.out_features = rows, k, o
    d.ldx, d.ldw, d.ldy = ldx, w.stride(0), ldy
    d.batch, d.epilogue = 1, (N.FZ_GEMM_GEGLU if geglu else N.FZ_GEMM_PLAIN)
    d.tile_cfg, d.split_k = tile_cfg, split_k
    for r in (res, res2):
        if r is not None:
            assert r.shape[-1] == ow and r.stride(-1) == 1 and r.numel() // ow == rows and r.dtype == torch.float16
    if res is not None:
        d.ldres = res.stride(-2) if res.dim() > 1 else ow
        if res2 is not None:
            assert (res2.stride(-2) if res2.dim() > 1 else ow) == d.ldres
    # split-K scratch only where it can pay: long K and few enough outputs that the fp32 slabs stay small
    want_ws = (not geglu) and o % 4 == 0 and (k >= 1024 or split_k > 1) and 2 * rows * o <= _WS_FLOATS
    if want_ws:
        d.workspace_floats = _WS_FLOATS
    return (C.byref(d), out_shape, want_ws, d)


_vt_plans = {}


def gemm_vt(x: torch.Tensor, w: torch.Tensor, lp: int, out: Optional[torch.Tensor] = None, tile_cfg: int = 0):
    """x: [N, L, K] (unit channel stride), w: [C, K] -> V^T [N, C, lp] = w @ x[n]^T, columns [L, lp) zero."""
    key = (x.shape, x.stride(), w.shape, w.stride(0), lp, tile_cfg, x.device, None if out is None else out.stride())
    plan = _vt_plans.get(key)
    n, l, k = x.shape
    c = w.shape[0]
    if out is None:
        out = torch.empty(n, c, lp, dtype=torch.float16, device=x.device)
    if plan is None:
        assert x.stride(2) == 1 and w.stride(1) == 1 and lp >= l and lp % 8 == 0
        _chk16(x, w, out)
        d = N.FzGemmDesc()
        d.rows, d.rows_store, d.in_features, d.out_features = l, lp, k, c
        d.ldx, d.ldw, d.ldy = x.stride(1), w.stride(0), out.stride(1)
        d.batch, d.x_batch_stride, d.y_batch_stride = n, x.stride(0), out.stride(0)
        d.transpose_out, d.tile_cfg = 1, tile_cfg
        plan = _vt_plans[key] = (C.byref(d), d)
    rc = N.lib().fz_gemm(plan[0], x.data_ptr(), w.data_ptr(), None, None, None, out.data_ptr(), None, _stream(x))
    if rc:
        N.check(rc, "fz_gemm(vt)")
    return out


# ------------------------------------------------------------------------------------------------------------
# The FeedForward chain of the 64x64 level in one launch (csrc/ff_chain.hip)
# ------------------------------------------------------------------------------------------------------------
def ff_chain_ok(rows: int, channels: int, inner: int) -> bool:
    return bool(N.lib().fz_ff_chain_ok(rows, channels, inner))


def ff_chain_preferred(rows: int, channels: int, inner: int) -> bool:
    """Is the one launch the faster form on MI355X for this shape (fz_ff_chain_preferred)?"""
    return bool(N.lib().fz_ff_chain_preferred(rows, channels, inner))


def ff_chain_pack(w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor) -> torch.Tensor:
    """GEGLU projection weight w1 [2 inner, C] (rows [val ; gate], diffusers GEGLU.proj), its bias b1 [2 inner] (or None) and the output
    projection w2 [C, inner] -> the operand-fragment stream fz_ff_chain reads (uint8, on w1's device).  Pack once per weight set."""
    two, c = w1.shape
    inner = two // 2
    if two % 2 or tuple(w2.shape) != (c, inner) or not ff_chain_ok(1, c, inner):
        raise ValueError(f"fz_ff_chain_pack: w1 [2 inner, 320] / w2 [320, inner] with inner % 32 == 0, got {tuple(w1.shape)} / {tuple(w2.shape)}")
    w1, w2 = w1.contiguous(), w2.contiguous()
    b1 = None if b1 is None else b1.contiguous()
    for t in (w1, b1, w2):
        if t is not None and t.dtype != torch.float16:
            raise ValueError("fz_ff_chain_pack: fp16 operands")
    _chk16(w1, b1, w2)
    out = torch.empty(N.lib().fz_ff_chain_pack_bytes(c, inner), dtype=torch.uint8, device=w1.device)
    rc = N.lib().fz_ff_chain_pack(w1.data_ptr(), _ptr(b1), w2.data_ptr(), out.data_ptr(), c, inner, _stream(w1))
    if rc:
        N.check(rc, "fz_ff_chain_pack")
    return out


def ff_chain(xn: torch.Tensor, packed: torch.Tensor, b2: Optional[torch.Tensor], inner: int, *, res: Optional[torch.Tensor] = None, ln=None):
    """y = Linear2(val * gelu(gate)) + b2 (+ res) with val | gate = xn W1^T + b1, and LayerNorm(y) when ln = (gamma, beta, eps) is given, in
    ONE launch (fz_ff_chain); xn / res: [..., 320] fp16 contiguous, `packed` from ff_chain_pack.  Returns (y, y_ln or None)."""
    c = xn.shape[-1]
    rows = xn.numel() // c
    if xn.dtype != torch.float16 or not xn.is_contiguous() or (res is not None and (res.dtype != torch.float16 or not res.is_contiguous()
                                                                                    or res.shape != xn.shape)):
        raise ValueError("fz_ff_chain: xn / res must be contiguous fp16 tensors of one shape")
    if packed.numel() != N.lib().fz_ff_chain_pack_bytes(c, inner):
        raise ValueError("fz_ff_chain: `packed` is not the stream of ff_chain_pack for these sizes")
    gam = bet = None
    eps = 0.0
    if ln is not None:
        gam, bet, eps = ln
        if gam.dtype != torch.float16 or bet.dtype != torch.float16:
            raise ValueError("fz_ff_chain: LayerNorm weight / bias must be fp16")
    _chk16(xn, b2, res, gam, bet)
    if packed.data_ptr() % 16 or packed.dtype != torch.uint8:
        raise ValueError("fz_ff_chain: `packed` must be the 16-byte aligned uint8 stream of ff_chain_pack")
    y = torch.empty_like(xn)
    yln = torch.empty_like(xn) if ln is not None else None
    rc = N.lib().fz_ff_chain(xn.data_ptr(), packed.data_ptr(), _ptr(b2), _ptr(res), y.data_ptr(), _ptr(gam), _ptr(bet), float(eps), _ptr(yln),
                             rows, c, inner, _stream(xn))
    if rc:
        N.check(rc, "fz_ff_chain")
    return y, yln


# ------------------------------------------------------------------------------------------------------------
# The cross-attention chain of the 64x64 level in one launch (csrc/xattn_chain.hip)
# ------------------------------------------------------------------------------------------------------------
def xattn_chain_ok(rows: int, rows_per_frame: int, channels: int, heads: int, lk: int) -> bool:
    return bool(N.lib().fz_xattn_chain_ok(rows, rows_per_frame, channels, heads, lk))


def xattn_chain_preferred(rows: int, rows_per_frame: int, channels: int, heads: int, lk: int) -> bool:
    """Is the one launch the faster form on MI355X for this shape (fz_xattn_chain_preferred)?"""
    return bool(N.lib().fz_xattn_chain_preferred(rows, rows_per_frame, channels, heads, lk))


def xattn_chain_pack(wq: torch.Tensor, wo: torch.Tensor, front=None) -> torch.Tensor:
    """attn2.to_q / attn2.to_out weights [320, 320] -> the operand-fragment stream fz_xattn_chain reads (uint8, on wq's device); front =
    (wo1 [320, 320], bias1 [320] or None, gamma1, beta1): attn1.to_out and the LayerNorm behind it, for the `front` form.  Pack once per
    weight set."""
    wo1, bo1, g1, b1 = front if front is not None else (None, None, None, None)
    ws = [wq.contiguous(), wo.contiguous()] + ([] if wo1 is None else [wo1.contiguous()])
    for t in ws:
        if t.dtype != torch.float16 or tuple(t.shape) != (320, 320):
            raise ValueError(f"fz_xattn_chain_pack: fp16 [320, 320] weights, got {t.dtype} {tuple(t.shape)}")
    vs = [None if t is None else t.contiguous() for t in (bo1, g1, b1)]
    for t in vs:
        if t is not None and (t.dtype != torch.float16 or t.numel() != 320):
            raise ValueError("fz_xattn_chain_pack: fp16 [320] bias / LayerNorm parameters")
    if wo1 is not None and (vs[1] is None or vs[2] is None):
        raise ValueError("fz_xattn_chain_pack: the front form needs the LayerNorm's gamma and beta")
    _chk16(*ws, *vs)
    out = torch.empty(N.lib().fz_xattn_chain_pack_bytes(int(wo1 is not None)), dtype=torch.uint8, device=wq.device)
    rc = N.lib().fz_xattn_chain_pack(ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr() if wo1 is not None else None, _ptr(vs[0]), _ptr(vs[1]),
                                     _ptr(vs[2]), out.data_ptr(), _stream(wq))
    if rc:
        N.check(rc, "fz_xattn_chain_pack")
    return out


def xattn_chain_kv_pack(k: torch.Tensor, vt: torch.Tensor, lk: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K [B, >= lk, 320] and V^T [B, 320, >= 96] of the text context (what fz_attn_cross takes) -> the per-batch fragment stream of
    fz_xattn_chain.  Pack once per context (into `out` when given: an issue plan's records keep pointing at it)."""
    b = k.shape[0]
    if k.dtype != torch.float16 or vt.dtype != torch.float16 or vt.shape[0] != b or k.shape[2] < 320 or vt.shape[1] != 320 or \
            vt.shape[2] < CROSS_KEYS or k.stride(2) != 1 or vt.stride(2) != 1 or k.shape[1] < lk:
        raise ValueError(f"fz_xattn_chain_kv_pack: k [B, >= lk, 320] / vt [B, 320, >= 96] fp16, got {tuple(k.shape)} / {tuple(vt.shape)}")
    _chk16(k, vt)
    nbytes = N.lib().fz_xattn_chain_kv_pack_bytes(b)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=k.device)
    elif out.numel() != nbytes or out.dtype != torch.uint8 or out.device != k.device:
        raise ValueError("fz_xattn_chain_kv_pack: `out` does not fit this context")
    rc = N.lib().fz_xattn_chain_kv_pack(k.data_ptr(), k.stride(0), k.stride(1), vt.data_ptr(), vt.stride(0), vt.stride(1), b, lk,
                                        out.data_ptr(), _stream(k))
    if rc:
        N.check(rc, "fz_xattn_chain_kv_pack")
    return out


def xattn_chain(x: torch.Tensor, packed: torch.Tensor, kv_packed: torch.Tensor, bias_out: Optional[torch.Tensor], *, res: Optional[torch.Tensor],
                frames_per_batch: int, heads: int, lk: int, scale: float, ln=None, front_eps: Optional[float] = None):
    """x: [N, L, 320] fp16 contiguous.  front_eps is None: x = LayerNorm'ed hidden states, returns (y, y_ln or None) with
    y = attn2(x, context) + res.  front_eps = eps of the LayerNorm in front (`packed` then holds attn1.to_out and that LayerNorm): x = attn1's
    attention output, `res` its residual; returns (y, y_ln or None, y1) with y1 = to_out1(x) + res and y = attn2(LayerNorm1(y1), context) + y1.
    (fz_xattn_chain)"""
    n, l, c = x.shape
    rows = n * l
    front = front_eps is not None
    ts = [x] + ([] if res is None else [res])
    for t in ts:
        if t.dtype != torch.float16 or not t.is_contiguous() or t.shape != x.shape:
            raise ValueError("fz_xattn_chain: x / res must be contiguous fp16 tensors of one shape")
    if not xattn_chain_ok(rows, l, c, heads, lk):
        raise ValueError(f"fz_xattn_chain: unsupported shape {tuple(x.shape)} heads {heads} lk {lk}")
    if packed.numel() != N.lib().fz_xattn_chain_pack_bytes(int(front)) or packed.dtype != torch.uint8 or packed.data_ptr() % 16:
        raise ValueError("fz_xattn_chain: `packed` is not the stream of xattn_chain_pack for this form")
    nb = (n + frames_per_batch - 1) // frames_per_batch
    if kv_packed.numel() < N.lib().fz_xattn_chain_kv_pack_bytes(nb) or kv_packed.dtype != torch.uint8 or kv_packed.data_ptr() % 16:
        raise ValueError("fz_xattn_chain: `kv_packed` does not cover the batch")
    d = N.FzXattnChain()
    keep = [x, res, bias_out]
    y = torch.empty_like(x)
    yln = None
    d.x, d.res, d.packed, d.kv_packed, d.bias_out, d.y = x.data_ptr(), _ptr(res), packed.data_ptr(), kv_packed.data_ptr(), _ptr(bias_out), y.data_ptr()
    if ln is not None:
        gam, bet, eps = ln
        if gam.dtype != torch.float16 or bet.dtype != torch.float16:
            raise ValueError("fz_xattn_chain: LayerNorm weight / bias must be fp16")
        yln = torch.empty_like(x)
        d.y_ln, d.ln_gamma, d.ln_beta, d.ln_eps = yln.data_ptr(), gam.data_ptr(), bet.data_ptr(), float(eps)
        keep += [gam, bet]
    y1 = None
    if front:
        if res is None:
            raise ValueError("fz_xattn_chain: the front form needs the residual")
        y1 = torch.empty_like(x)
        d.front, d.y1, d.ln1_eps = 1, y1.data_ptr(), float(front_eps)
    _chk16(*[t for t in keep if t is not None])
    d.rows, d.rows_per_frame, d.frames_per_batch, d.channels, d.heads, d.lk, d.scale = rows, l, frames_per_batch, c, heads, lk, float(scale)
    rc = N.lib().fz_xattn_chain(C.byref(d), _stream(x))
    if rc:
        N.check(rc, "fz_xattn_chain")
    return (y, yln, y1) if front else (y, yln)


_qkvt_plans = {}


def gemm_qkvt_ok(x: torch.Tensor, w: torch.Tensor, split: int) -> bool:
    """Shapes fz_gemm_qkvt carries: token counts per frame that are multiples of 64 (V^T rows need no zero padding then), a k | v
    boundary on a multiple of 64 output columns."""
    return x.dim() == 3 and x.shape[1] % 64 == 0 and split % 64 == 0 and 0 < split < w.shape[0] and x.shape[2] % 8 == 0


def gemm_qkvt(x: torch.Tensor, w: torch.Tensor, split: int, tile_cfg: int = 0):
    """The q | k | V^T projection of a self-attention in ONE launch (fz_gemm_qkvt): x [N, L, K] (unit channel stride, one row
    stride), w [split + Cv, K] = rows [Wq ; Wk ; Wv] -> (y [N, L, split] = x @ w[:split]^T,  vt [N, Cv, L] = w[split:] @ x[n]^T)."""
    n, l, k = x.shape
    o = w.shape[0]
    cv = o - split
    key = (x.shape, x.stride(), w.shape, w.stride(0), split, tile_cfg, x.device)
    plan = _qkvt_plans.get(key)
    if plan is None:
        if not gemm_qkvt_ok(x, w, split) or x.stride(2) != 1 or w.stride(1) != 1 or w.shape[1] != k or x.stride(0) != l * x.stride(1):
            raise ValueError("fz_gemm_qkvt: x [N, L, K] with L % 64 == 0 and a single row stride, w [split + Cv, K], split % 64 == 0")
        d = N.FzGemmDesc()
        d.rows, d.in_features, d.out_features = n * l, k, o
        d.ldx, d.ldw, d.ldy = x.stride(1), w.stride(0), split
        d.batch, d.epilogue, d.tile_cfg = 1, N.FZ_GEMM_PLAIN, tile_cfg
        plan = _qkvt_plans[key] = (C.byref(d), d)
    if x.dtype != torch.float16 or w.dtype != torch.float16 or ((x.data_ptr() | w.data_ptr()) & 15):
        raise ValueError("fz_gemm_qkvt operands must be fp16 and 16-byte aligned")
    y = torch.empty(n, l, split, dtype=torch.float16, device=x.device)
    vt = torch.empty(n, cv, l, dtype=torch.float16, device=x.device)
    rc = N.lib().fz_gemm_qkvt(plan[0], x.data_ptr(), w.data_ptr(), y.data_ptr(), vt.data_ptr(), split, l, cv * l, l, _stream(x))
    if rc:
        N.check(rc, "fz_gemm_qkvt")
    return y, vt


def conv3x3(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor], *, hw: Tuple[int, int], stride: int = 1,
            upsample: bool = False, temb: Optional[torch.Tensor] = None, frames_per_batch: int = 1,
            res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, tile_cfg: int = 0, split_k: int = 0):
    """x: [N, H*W, Cin] token-major fp16 -> [N, Ho*Wo, Cout]."""
    n, _, cin = x.shape
    h, w = hw
    cout = wt.shape[0]
    hu, wu = (2 * h, 2 * w) if upsample else (h, w)
    ho, wo = (hu - 1) // stride + 1, (wu - 1) // stride + 1
    if not (x.is_contiguous() and wt.is_contiguous() and wt.shape[1] == 9 and wt.shape[2] == cin and x.dtype == torch.float16):
        raise ValueError("fz_conv3x3: x [N, H*W, Cin] and wt [Cout, 9, Cin] must be contiguous fp16")
    if out is None:
        out = torch.empty(n, ho * wo, cout, dtype=torch.float16, device=x.device)
    if res is not None:
        assert res.is_contiguous() and res.shape == out.shape
    ts = 0
    if temb is not None:
        assert temb.shape == (n // frames_per_batch, cout) and temb.stride(1) == 1
        ts = temb.stride(0)
    use_ws = cout % 4 == 0 and cin % 8 == 0 and 2 * n * ho * wo * cout <= _WS_FLOATS
    rc = N.lib().fz_conv3x3(x.data_ptr(), wt.data_ptr(), None if bias is None else bias.data_ptr(),
                            None if temb is None else temb.data_ptr(), ts, None if res is None else res.data_ptr(),
                            out.data_ptr(), n, h, w, cin, cout, stride, 1 if upsample else 0, frames_per_batch,
                            _ws_ptr(x.device) if use_ws else None, _WS_FLOATS if use_ws else 0, tile_cfg, split_k, _stream(x))
    if rc:
        N.check(rc, "fz_conv3x3")
    return out, (ho, wo)


def conv3x3_up2_ok(n: int, h: int, w: int, cin: int, cout: int) -> bool:
    """Does the sub-pixel form of nearest-2x + 3x3 convolution (fz_conv3x3_up2: four 2x2 convolutions of the input) carry this shape?"""
    return bool(N.lib().fz_conv3x3_up2_ok(n, h, w, cin, cout))


def conv3x3_up2_preferred(n: int, h: int, w: int, cin: int, cout: int) -> bool:
    """... and is it the faster path there (measured rule of the library)?"""
    return bool(N.lib().fz_conv3x3_up2_preferred(n, h, w, cin, cout))


def pack_conv3x3_up2_weight(wt: torch.Tensor) -> torch.Tensor:
    """wt: fz_conv3x3's packed weights [Cout, 9, Cin] -> [4 parities, Cout, 4 taps, Cin], the taps summed per output parity (fz_conv3x3_up2_pack)."""
    cout, nine, cin = wt.shape
    assert nine == 9 and wt.is_contiguous() and wt.dtype == torch.float16
    out = torch.empty(4, cout, 4, cin, dtype=torch.float16, device=wt.device)
    assert out.numel() == N.lib().fz_conv3x3_up2_pack_halves(cin, cout)
    rc = N.lib().fz_conv3x3_up2_pack(wt.data_ptr(), out.data_ptr(), cin, cout, _stream(wt))
    if rc:
        N.check(rc, "fz_conv3x3_up2_pack")
    return out


def conv3x3_up2(x: torch.Tensor, wt_up: torch.Tensor, bias: Optional[torch.Tensor], *, hw: Tuple[int, int], out: Optional[torch.Tensor] = None):
    """x: [N, H*W, Cin] -> nearest-2x upsampling + 3x3 convolution -> [N, 4 H*W, Cout] on the summed weights of pack_conv3x3_up2_weight."""
    n, _, cin = x.shape
    h, w = hw
    cout = wt_up.shape[1]
    if not (x.is_contiguous() and wt_up.is_contiguous() and tuple(wt_up.shape) == (4, cout, 4, cin) and x.dtype == torch.float16):
        raise ValueError("fz_conv3x3_up2: x [N, H*W, Cin] and wt_up [4, Cout, 4, Cin] must be contiguous fp16")
    if out is None:
        out = torch.empty(n, 4 * h * w, cout, dtype=torch.float16, device=x.device)
    rc = N.lib().fz_conv3x3_up2(x.data_ptr(), wt_up.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), n, h, w, cin, cout, _stream(x))
    if rc:
        N.check(rc, "fz_conv3x3_up2")
    return out, (2 * h, 2 * w)


def gn_epilogue_ok(tokens: int, cout: int, groups: int) -> bool:
    """Can a producer of a [N, tokens, cout] tensor emit its GroupNorm(groups) statistics from its epilogue (fz_gemm_gn /
    fz_temporal_conv3_gn)?  (whole 128-row chunks per frame, 320-wide tiles holding whole groups of an even width)"""
    if groups <= 0 or cout % groups or cout % 320 or tokens % 128:
        return False
    return cout // groups in (10, 20)  # the instantiated group widths (SD-1.x: 320 / 640 channels over 32 groups)


def temporal_conv3(x: torch.Tensor, wt: torch.Tensor, *, clip_len: int, res: Optional[torch.Tensor] = None,
                   res2: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None, out=None, gn_groups: int = 0):
    """x: [N, tokens, Cin]; wt: [Cout, 3, Cin] (nn.Conv1d weight [Cout, Cin, 3] permuted); -> [N, tokens, Cout] (+res).
    gn_groups > 0: returns (y, partial) -- partial = the Welford partials [N, gn_groups, tokens / 128, 3] of y's GroupNorm statistics
    written by the launch's own epilogue (fz_temporal_conv3_gn), or None where the launch the library picks cannot produce them."""
    n, tokens, cin = x.shape
    cout = wt.shape[0]
    if not (x.is_contiguous() and wt.is_contiguous() and wt.shape[1] == 3 and wt.shape[2] == cin and x.dtype == torch.float16):
        raise ValueError("fz_temporal_conv3: x [N, tokens, Cin] and wt [Cout, 3, Cin] must be contiguous fp16")
    if out is None:
        out = torch.empty(n, tokens, cout, dtype=torch.float16, device=x.device)
    for r in (res, res2):
        if r is not None:
            assert r.is_contiguous() and r.shape == out.shape and r.dtype == torch.float16
    ts = 0
    if temb is not None:
        assert temb.shape == (n // clip_len, cout) and temb.stride(1) == 1 and temb.dtype == torch.float16
        ts = temb.stride(0)
    use_ws = cout % 4 == 0 and 2 * n * tokens * cout <= _WS_FLOATS
    if gn_groups > 0:
        partial = None
        if gn_epilogue_ok(tokens, cout, gn_groups):
            partial = torch.empty(n, gn_groups, tokens // 128, 3, dtype=torch.float32, device=x.device)
            rc = N.lib().fz_temporal_conv3_gn(x.data_ptr(), wt.data_ptr(), None if res is None else res.data_ptr(),
                                              None if res2 is None else res2.data_ptr(), None if temb is None else temb.data_ptr(), ts,
                                              out.data_ptr(), n, tokens, cin, cout, clip_len,
                                              _ws_ptr(x.device) if use_ws else None, _WS_FLOATS if use_ws else 0,
                                              partial.data_ptr(), gn_groups, _stream(x))
            if rc == N.FZ_GEMM_NO_STATS:
                partial = None
            elif rc:
                N.check(rc, "fz_temporal_conv3_gn")
            return out, partial
    rc = N.lib().fz_temporal_conv3(x.data_ptr(), wt.data_ptr(), None if res is None else res.data_ptr(),
                                   None if res2 is None else res2.data_ptr(), None if temb is None else temb.data_ptr(), ts,
                                   out.data_ptr(), n, tokens, cin, cout, clip_len,
                                   _ws_ptr(x.device) if use_ws else None, _WS_FLOATS if use_ws else 0, _stream(x))
    if rc:
        N.check(rc, "fz_temporal_conv3")
    return (out, None) if gn_groups > 0 else out


def lora_pair_ok(n: int, tokens: int, channels: int, rank: int, clip_len: int) -> bool:
    """Shapes the one-launch temporal LoRA pair carries (fz_lora_pair_ok: rank 160, channels % 320 == 0, clip_len divides 128 ...)."""
    return bool(N.lib().fz_lora_pair_ok(n, tokens, channels, rank, clip_len))


def lora_pair_preferred(n: int, tokens: int, channels: int, rank: int, clip_len: int) -> bool:
    """Where the one launch is also faster than temporal_conv3 twice (fz_lora_pair_preferred: launches of >= 256 workgroups)."""
    return bool(N.lib().fz_lora_pair_preferred(n, tokens, channels, rank, clip_len))


def lora_pair(x: torch.Tensor, w_down: torch.Tensor, w_up: torch.Tensor, *, clip_len: int, res2: Optional[torch.Tensor] = None,
              temb: Optional[torch.Tensor] = None, out=None, gn_groups: int = 0):
    """up(down(x)) + x (+ temb per clip) (+ res2) of the temporal LoRA (lora.py:31-54) in one launch: x [N, tokens, C],
    w_down [rank, 3, C], w_up [C, 3, rank] (the packing of temporal_conv3).  Bit-identical to temporal_conv3 twice.
    gn_groups > 0: returns (y, partial) -- the Welford partials [N, gn_groups, chunks, 3] of y's GroupNorm statistics out of the same launch
    (fz_lora_pair_gn), or (y, None) where that form does not exist."""
    n, tokens, c = x.shape
    rank = w_down.shape[0]
    if not (x.is_contiguous() and w_down.is_contiguous() and w_up.is_contiguous() and x.dtype == torch.float16
            and w_down.shape == (rank, 3, c) and w_up.shape == (c, 3, rank)):
        raise ValueError("fz_lora_pair: x [N, tokens, C], w_down [rank, 3, C], w_up [C, 3, rank] must be contiguous fp16")
    if out is None:
        out = torch.empty_like(x)
    if res2 is not None:
        assert res2.is_contiguous() and res2.shape == out.shape and res2.dtype == torch.float16
    ts = 0
    if temb is not None:
        assert temb.shape == (n // clip_len, c) and temb.stride(1) == 1 and temb.dtype == torch.float16
        ts = temb.stride(0)
    args = (x.data_ptr(), w_down.data_ptr(), w_up.data_ptr(), None if temb is None else temb.data_ptr(), ts,
            None if res2 is None else res2.data_ptr(), out.data_ptr(), n, tokens, c, rank, clip_len)
    if gn_groups > 0:
        chunks = N.lib().fz_lora_pair_gn_chunks(n, tokens, c, rank, clip_len, gn_groups)
        if chunks > 0:
            partial = torch.empty(n, gn_groups, chunks, 3, dtype=torch.float32, device=x.device)
            N.check(N.lib().fz_lora_pair_gn(*args, partial.data_ptr(), gn_groups, _stream(x)), "fz_lora_pair_gn")
            return out, partial
    rc = N.lib().fz_lora_pair(*args, _stream(x))
    if rc:
        N.check(rc, "fz_lora_pair")
    return (out, None) if gn_groups > 0 else out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, out=None):
    c = x.shape[-1]
    assert x.is_contiguous() and x.dtype == torch.float16
    if gamma.dtype != torch.float16 or beta.dtype != torch.float16 or ((x.data_ptr() | gamma.data_ptr() | beta.data_ptr()) & 15):
        raise ValueError("fz_layernorm: fp16, 16-byte aligned x / gamma / beta")
    if out is None:
        out = torch.empty_like(x)
    rc = N.lib().fz_layernorm(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.numel() // c, c, eps, _stream(x))
    if rc:
        N.check(rc, "fz_layernorm")
    return out


def geglu(x: torch.Tensor, out=None):
    inner = x.shape[-1] // 2
    assert x.is_contiguous()
    _chk16(x)
    if out is None:
        out = torch.empty(*x.shape[:-1], inner, dtype=x.dtype, device=x.device)
    N.check(N.lib().fz_geglu(_ptr(x), _ptr(out), x.numel() // (2 * inner), inner, _stream(x)), "fz_geglu")
    return out


def softmax_rows(x: torch.Tensor, scale: float = 1.0, out=None):
    """softmax(scale * x) over the last dim of an fp16 matrix [..., cols] (one row stride), fp32 arithmetic (fz_softmax_rows)."""
    cols = x.shape[-1]
    assert x.stride(-1) == 1 and x.dtype == torch.float16
    rows = x.numel() // cols
    ldx = x.stride(-2) if x.dim() > 1 else cols
    _chk16(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    N.check(N.lib().fz_softmax_rows(_ptr(x), _ptr(out), rows, cols, ldx, cols, float(scale), _stream(x)), "fz_softmax_rows")
    return out


def transpose_pad(x: torch.Tensor, lp: int, out=None):
    """x: [N, L, >=C view] (unit channel stride) -> [N, C, lp] with zero padding."""
    n, l, c = x.shape
    assert x.stride(2) == 1 and x.dtype == torch.float16
    if out is None:
        out = torch.empty(n, c, lp, dtype=x.dtype, device=x.device)
    N.check(N.lib().fz_transpose_pad(_ptr(x), _ptr(out), n, l, c, x.stride(0), x.stride(1), lp, _stream(x)),
            "fz_transpose_pad")
    return out


def latent_update(z: torch.Tensor, eps_u: Optional[torch.Tensor], eps_c: torch.Tensor, guidance: float, cz: float,
                  ce: float, *, inv: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                  next_in: Optional[torch.Tensor] = None):
    """z: float [4, F, hw] (in place); eps_*: fp16 [F, hw, 4]; inv like z; mask float [F, hw]; next_in fp16 [F, hw, 4]."""
    _, frames, hw = z.shape
    assert z.dtype == torch.float32 and z.is_contiguous() and eps_c.is_contiguous() and eps_c.dtype == torch.float16
    N.check(N.lib().fz_latent_update(_ptr(z), _ptr(eps_u), _ptr(eps_c), float(guidance), float(cz), float(ce), _ptr(inv),
                                     _ptr(mask), _ptr(next_in), frames, hw, _stream(z)), "fz_latent_update")
    return z


def accumulate(acc: torch.Tensor, x: torch.Tensor):
    assert acc.dtype == torch.float32 and x.dtype == torch.float16 and acc.numel() == x.numel()
    assert acc.is_contiguous() and x.is_contiguous()
    N.check(N.lib().fz_accumulate(_ptr(acc), _ptr(x), x.numel(), _stream(x)), "fz_accumulate")
    return acc


def version() -> str:
    return N.lib().fz_version().decode()
