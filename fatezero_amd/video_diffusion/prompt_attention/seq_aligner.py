"""Host-side prompt algebra, part 2 (reference: video_diffusion/prompt_attention/seq_aligner.py).

Token-level alignment of the source and target prompts: a 77x77 replacement mapper for same-length word swaps
(:152-196) and, for refinement edits, a global (Needleman-Wunsch, gap 0 / match 1 / mismatch -1) alignment giving
a gather index + a 0/1 'token existed in the source' vector (:61-129).  Outputs are exact integers / 0-1 floats."""
import numpy as np
import torch

from .ptp_utils import get_word_inds

GAP, MATCH, MISMATCH = 0, 1, -1
_LEFT, _UP, _DIAG, _STOP = 1, 2, 3, 4


def global_align(x, y):
    """Score + trace-back tables with the reference's tie-breaking order: left, then up, then diagonal."""
    nx, ny = len(x), len(y)
    score = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    score[0, 1:] = (np.arange(ny) + 1) * GAP
    score[1:, 0] = (np.arange(nx) + 1) * GAP
    trace = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    trace[0, 1:], trace[1:, 0], trace[0, 0] = _LEFT, _UP, _STOP
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            left, up = score[i, j - 1] + GAP, score[i - 1, j] + GAP
            diag = score[i - 1, j - 1] + (MATCH if x[i - 1] == y[j - 1] else MISMATCH)
            best = max(left, up, diag)
            score[i, j] = best
            trace[i, j] = _LEFT if best == left else (_UP if best == up else _DIAG)
    return score, trace


def aligned_target_to_source(x, y, trace):
    """For every target token j (in order) the aligned source token i, or -1 for an inserted token."""
    i, j = len(x), len(y)
    pairs = []
    while i > 0 or j > 0:
        step = trace[i, j]
        if step == _DIAG:
            i, j = i - 1, j - 1
            pairs.append((j, i))
        elif step == _LEFT:
            j -= 1
            pairs.append((j, -1))
        elif step == _UP:
            i -= 1
        else:
            break
    pairs.reverse()
    return torch.tensor(pairs, dtype=torch.int64)


def get_mapper(x: str, y: str, tokenizer, max_len=77):
    x_seq, y_seq = tokenizer.encode(x), tokenizer.encode(y)
    _, trace = global_align(x_seq, y_seq)
    base = aligned_target_to_source(x_seq, y_seq, trace)
    n = base.shape[0]
    alphas = torch.ones(max_len)
    alphas[:n] = base[:, 1].ne(-1).float()
    mapper = torch.zeros(max_len, dtype=torch.int64)
    mapper[:n] = base[:, 1]
    mapper[n:] = len(y_seq) + torch.arange(max_len - len(y_seq))
    return mapper, alphas


def get_refinement_mapper(prompts, tokenizer, max_len=77):
    pairs = [get_mapper(prompts[0], prompts[i], tokenizer, max_len) for i in range(1, len(prompts))]
    return torch.stack([p[0] for p in pairs]), torch.stack([p[1] for p in pairs])


def get_replacement_mapper_(x: str, y: str, tokenizer, max_len=77):
    words_x, words_y = x.split(" "), y.split(" ")
    if len(words_x) != len(words_y):
        raise ValueError(f"attention replacement edit can only be applied on prompts with the same length"
                         f" but prompt A has {len(words_x)} words and prompt B has {len(words_y)} words.")
    changed = [i for i in range(len(words_y)) if words_y[i] != words_x[i]]
    src = [get_word_inds(x, i, tokenizer) for i in changed]
    tgt = [get_word_inds(y, i, tokenizer) for i in changed]
    mapper = np.zeros((max_len, max_len))
    i = j = cur = 0
    while i < max_len and j < max_len:
        if cur < len(src) and src[cur][0] == i:
            s, t = src[cur], tgt[cur]
            if len(s) == len(t):
                mapper[s, t] = 1
            else:
                for it in t:
                    mapper[s, it] = 1 / len(t)
            cur += 1
            i += len(s)
            j += len(t)
        elif cur < len(src):
            mapper[i, j] = 1
            i += 1
            j += 1
        else:
            mapper[j, j] = 1
            i += 1
            j += 1
    return torch.from_numpy(mapper).float()


def get_replacement_mapper(prompts, tokenizer, max_len=77):
    return torch.stack([get_replacement_mapper_(prompts[0], prompts[i], tokenizer, max_len) for i in range(1, len(prompts))])
