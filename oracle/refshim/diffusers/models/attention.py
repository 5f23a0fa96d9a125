"""diffusers/models/attention.py (0.11.1): CrossAttention, FeedForward, GEGLU, AdaLayerNorm (restated)."""
import torch
import torch.nn.functional as F
from torch import nn


class CrossAttention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, added_kv_proj_dim=None, norm_num_groups=None):
        super().__init__()
        inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.sliceable_head_dim = heads
        self._slice_size = None
        self._use_memory_efficient_attention_xformers = False
        self.added_kv_proj_dim = added_kv_proj_dim
        self.group_norm = (nn.GroupNorm(num_channels=inner_dim, num_groups=norm_num_groups, eps=1e-5, affine=True)
                           if norm_num_groups is not None else None)
        self.to_q = nn.Linear(query_dim, inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        if self.added_kv_proj_dim is not None:
            self.add_k_proj = nn.Linear(added_kv_proj_dim, cross_attention_dim)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, cross_attention_dim)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim), nn.Dropout(dropout)])

    def reshape_heads_to_batch_dim(self, tensor):
        b, s, dim = tensor.shape
        h = self.heads
        tensor = tensor.reshape(b, s, h, dim // h)
        return tensor.permute(0, 2, 1, 3).reshape(b * h, s, dim // h)

    def reshape_batch_dim_to_heads(self, tensor):
        bh, s, dim = tensor.shape
        h = self.heads
        tensor = tensor.reshape(bh // h, h, s, dim)
        return tensor.permute(0, 2, 1, 3).reshape(bh // h, s, dim * h)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        encoder_hidden_states = encoder_hidden_states if encoder_hidden_states is not None else hidden_states
        query = self.reshape_heads_to_batch_dim(self.to_q(hidden_states))
        key = self.reshape_heads_to_batch_dim(self.to_k(encoder_hidden_states))
        value = self.reshape_heads_to_batch_dim(self.to_v(encoder_hidden_states))
        hidden_states = self._attention(query, key, value, attention_mask)
        hidden_states = self.to_out[0](hidden_states)
        return self.to_out[1](hidden_states)

    def _attention(self, query, key, value, attention_mask=None):
        if self.upcast_attention:
            query, key = query.float(), key.float()
        scores = torch.baddbmm(
            torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device),
            query, key.transpose(-1, -2), beta=0, alpha=self.scale)
        if attention_mask is not None:
            scores = scores + attention_mask
        if self.upcast_softmax:
            scores = scores.float()
        probs = scores.softmax(dim=-1).to(value.dtype)
        return self.reshape_batch_dim_to_heads(torch.bmm(probs, value))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, hidden_states):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu"):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        assert activation_fn == "geglu"
        self.net = nn.ModuleList([GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out)])

    def forward(self, hidden_states):
        for module in self.net:
            hidden_states = module(hidden_states)
        return hidden_states


class AdaLayerNorm(nn.Module):  # never instantiated by SD-1.x configs (num_embeds_ada_norm=None)
    def __init__(self, *a, **k):
        raise NotImplementedError
