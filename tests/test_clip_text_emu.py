"""Native CLIP text encoder + BPE tokenizer against the third-party originals that ARE installed
here: transformers' CLIPTextModel (same random weights) and CLIPTokenizer (same synthetic vocab / merges files).  The
encoder's kernels run on the CPU emulation (test infrastructure; tests/test_clip_text_gpu.py repeats the model case on MI355X)."""
import collections
import json
import os

import pytest
import torch

from fatezero_amd import _native, build
from fatezero_amd.video_diffusion.models.clip_text import CLIPTextModel, CLIPTokenizer, _bytes_to_unicode


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    _native.use_test_backend(build.build_emu())
    yield
    _native.reset_backend()


def synthetic_bpe(folder, n_merges=80):
    """A small but real BPE vocabulary in the file format of the SD checkpoints (vocab.json + merges.txt)."""
    corpus = ("a silver jeep driving down a curvy road in the countryside watercolor painting of porsche car moving on the road "
              "squirrel rabbit eating carrot van gogh style swan swarovski crystal cat tiger flamingo").split()
    words = collections.Counter(tuple(w[:-1]) + (w[-1] + "</w>",) for w in corpus)
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for p in zip(w[:-1], w[1:]):
                pairs[p] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = {}
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] = new.get(tuple(out), 0) + c
        words = new
    base = list(_bytes_to_unicode().values())
    vocab = base + [b + "</w>" for b in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
    os.makedirs(folder, exist_ok=True)
    json.dump({t: i for i, t in enumerate(vocab)}, open(os.path.join(folder, "vocab.json"), "w"))
    with open(os.path.join(folder, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    json.dump({"model_max_length": 77, "pad_token": "<|endoftext|>"}, open(os.path.join(folder, "tokenizer_config.json"), "w"))
    return len(vocab)


PROMPTS = ["a silver jeep driving down a curvy road in the countryside", "A Porsche car, moving on the ROAD!",
           "watercolor painting of a silver jeep's road", "  swan   with\ttwo   spaces ", "rabbit & carrot: 3 cats, 12 tigers",
           "", "café naïve"]


def test_tokenizer_matches_transformers(tmp_path):
    transformers = pytest.importorskip("transformers")
    folder = str(tmp_path / "tokenizer")
    synthetic_bpe(folder)
    ours = CLIPTokenizer.from_pretrained(str(tmp_path), subfolder="tokenizer")
    ref = transformers.CLIPTokenizer(os.path.join(folder, "vocab.json"), os.path.join(folder, "merges.txt"))
    assert ours.model_max_length == 77 and ours.pad_token_id == ours.eos_token_id == ref.eos_token_id
    for p in PROMPTS:
        a, b = ours.encode(p), ref.encode(p)
        assert a == b, (p, a, b)
        pa = ours(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        pb = ref(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        assert torch.equal(pa, pb), p
        assert [ours.decode([i]) for i in a] == [ref.decode([i]) for i in b], p  # what ptp_utils.get_word_inds consumes
    long = " ".join(["jeep road"] * 60)
    assert torch.equal(ours(long, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids,
                       ref(long, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids)


def clip_model_case(device, tol=1e-2):
    transformers = pytest.importorskip("transformers")
    torch.manual_seed(0)
    cfg = transformers.CLIPTextConfig(vocab_size=600, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                      num_attention_heads=4, max_position_embeddings=77, hidden_act="quick_gelu",
                                      eos_token_id=2, bos_token_id=0, pad_token_id=1)
    ref = transformers.CLIPTextModel(cfg).eval()
    sd = {k: v.half().float() for k, v in ref.state_dict().items()}
    ref.load_state_dict(sd)
    ours = CLIPTextModel(dict(vocab_size=600, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4,
                              max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=2))
    # the SD checkpoints carry the transformers-4 layout: `text_model.` prefix + a position_ids buffer
    legacy = {"text_model." + k: v for k, v in sd.items()}
    legacy["text_model.embeddings.position_ids"] = torch.arange(77)[None]
    ours.load_state_dict(legacy)
    ours = ours.to(device).half().eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 598, (2, 77), generator=g)
    ids[:, 0] = 598
    ids[0, 9:] = 599
    ids[1, 30:] = 599
    with torch.no_grad():
        want = ref(ids)
    got = ours(ids.to(device))
    assert got[0].shape == (2, 77, 128) and got[0].dtype == torch.float16
    err = float((got[0].float().cpu() - want[0]).abs().max() / want[0].abs().max())
    perr = float((got.pooler_output.float().cpu() - want.pooler_output).abs().max() / want.pooler_output.abs().max())
    assert err < tol and perr < tol, (err, perr)
    return err, perr


def test_text_encoder_matches_transformers():
    print(clip_model_case("cpu"))


def test_from_pretrained_layout(tmp_path):
    root = tmp_path / "ckpt" / "text_encoder"
    os.makedirs(root)
    m = CLIPTextModel(dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2))
    json.dump(dict(vars(m.config), architectures=["CLIPTextModel"], model_type="clip_text_model"), open(root / "config.json", "w"))
    torch.save(m.state_dict(), root / "pytorch_model.bin")
    loaded = CLIPTextModel.from_pretrained(str(tmp_path / "ckpt"), subfolder="text_encoder")
    assert loaded.config.hidden_size == 64
    assert torch.equal(loaded.state_dict()["text_model.final_layer_norm.weight"], m.state_dict()["text_model.final_layer_norm.weight"])
