"""Host-side prompt algebra, part 1 (reference: video_diffusion/prompt_attention/ptp_utils.py:144-199).

Runs once per prompt pair on the CPU; its outputs (token index sets, the per-step per-word 0/1 gate
`cross_replace_alpha`) are kernel constants and must be exact.  Visualisation helpers of the reference file are
out of scope."""
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch


def get_word_inds(text: str, word_place, tokenizer) -> np.ndarray:
    """Token positions (1-based, after BOS) of a word given by value or by index (ptp_utils.py:144-162)."""
    words = text.split(" ")
    if isinstance(word_place, str):
        wanted = {i for i, w in enumerate(words) if w == word_place}
    elif isinstance(word_place, int):
        wanted = {word_place}
    else:
        wanted = set(word_place)
    hits = []
    if wanted:
        pieces = [tokenizer.decode([tok]).strip("#") for tok in tokenizer.encode(text)][1:-1]
        word_idx, consumed = 0, 0
        for pos, piece in enumerate(pieces):
            consumed += len(piece)
            if word_idx in wanted:
                hits.append(pos + 1)
            if consumed >= len(words[word_idx]):
                word_idx, consumed = word_idx + 1, 0
    return np.array(hits)


def update_alpha_time_word(alpha, bounds: Union[float, Tuple[float, float]], prompt_ind: int,
                           word_inds: Optional[torch.Tensor] = None):
    """ptp_utils.py:165-176."""
    if isinstance(bounds, float):
        bounds = 0, bounds
    n = alpha.shape[0]
    start, end = int(bounds[0] * n), int(bounds[1] * n)
    if word_inds is None:
        word_inds = torch.arange(alpha.shape[2])
    alpha[:, prompt_ind, word_inds] = 0
    alpha[start:end, prompt_ind, word_inds] = 1
    return alpha


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=77):
    """[num_steps+1, len(prompts)-1, 1, 1, 77] float 0/1 (ptp_utils.py:179-199)."""
    if not hasattr(cross_replace_steps, "items"):
        cross_replace_steps = {"default_": cross_replace_steps}
    steps = {k: (tuple(v) if isinstance(v, (list, tuple)) or type(v).__name__ == "ListConfig" else v)
             for k, v in cross_replace_steps.items()}
    steps.setdefault("default_", (0.0, 1.0))
    alpha = torch.zeros(num_steps + 1, len(prompts) - 1, max_num_words)
    for i in range(len(prompts) - 1):
        alpha = update_alpha_time_word(alpha, steps["default_"], i)
    for key, item in steps.items():
        if key == "default_":
            continue
        for i in range(1, len(prompts)):
            ind = get_word_inds(prompts[i], key, tokenizer)
            if len(ind) > 0:
                alpha = update_alpha_time_word(alpha, item, i - 1, ind)
    return alpha.reshape(num_steps + 1, len(prompts) - 1, 1, 1, max_num_words)
