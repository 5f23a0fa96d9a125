"""OpenAI CLIP on the native kernels (CPU emulation backend) vs the reference's own model; preprocessing and tokenizer checks."""
import gzip
import os

import numpy as np
import pytest
import torch
from PIL import Image

from fatezero_amd import _native, build

import clip_cases as CC


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    _native.use_test_backend(build.build_emu())
    yield
    _native.reset_backend()


def test_clip_tiny_matches_reference_model():
    res = CC.run_clip_case("clip_tiny", "cpu")
    print(res)
    CC.check(res)


def test_preprocess_matches_the_reference_transform():
    """clip.py:79-86 restated without torchvision: shorter side -> n_px (PIL bicubic), centre crop, RGB, [0, 1], normalise."""
    from fatezero_amd import clip
    rng = np.random.RandomState(0)
    t = clip._Transform(32)
    for (w, h) in [(48, 80), (80, 48), (32, 32), (33, 64)]:
        img = Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8))
        x = t(img)
        assert x.shape == (3, 32, 32) and x.dtype == torch.float32
        if w <= h:
            r = img.resize((32, int(32 * h / w)), Image.BICUBIC)
        else:
            r = img.resize((int(32 * w / h), 32), Image.BICUBIC)
        rw, rh = r.size
        left, top = int(round((rw - 32) / 2.0)), int(round((rh - 32) / 2.0))
        ref = np.asarray(r.crop((left, top, left + 32, top + 32)), dtype=np.float32) / 255.0
        ref = (ref - np.array(clip._MEAN, dtype=np.float32)) / np.array(clip._STD, dtype=np.float32)
        assert np.allclose(x.permute(1, 2, 0).numpy(), ref, atol=1e-6)
    grey = t(Image.fromarray(rng.randint(0, 256, (40, 40), dtype=np.uint8)))  # mode L -> RGB
    assert grey.shape == (3, 32, 32)


def _write_bpe(path):
    from test_clip_text_emu import synthetic_bpe
    import json
    folder = os.path.join(os.path.dirname(path), "hf")
    synthetic_bpe(folder)
    merges = open(os.path.join(folder, "merges.txt"), encoding="utf-8").read().split("\n")[1:]
    with gzip.open(path, "wt", encoding="utf-8") as f:  # the OpenAI file: a header line, then one merge per line
        f.write('"bpe_simple_vocab_16e6.txt#version: 0.2\n' + "\n".join(m for m in merges if m) + "\n")
    return json.load(open(os.path.join(folder, "vocab.json")))


def test_tokenize_layout(tmp_path):
    """clip.py:195-235 on a synthetic merges file in the OpenAI format: [SOT] ids [EOT] then zeros; EOT is the largest id."""
    from fatezero_amd import clip
    path = str(tmp_path / "bpe_simple_vocab_16e6.txt.gz")
    vocab = _write_bpe(path)
    toks = clip.tokenize(["a silver jeep driving down a curvy road", "A Porsche car &amp; a road"], bpe_path=path)
    assert toks.shape == (2, 77) and toks.dtype == torch.long
    sot, eot = vocab["<|startoftext|>"], vocab["<|endoftext|>"]
    assert eot == len(vocab) - 1 and sot == eot - 1  # the vocabulary derived from the merges file has the reference's layout
    for row in toks:
        n = int((row != 0).sum())
        assert row[0] == sot and row[n - 1] == eot and int(row.argmax()) == n - 1 and bool((row[n:] == 0).all())
    amp = clip.tokenize("a &amp; b", bpe_path=path)[0]
    assert torch.equal(amp, clip.tokenize("a & b", bpe_path=path)[0])  # html.unescape, simple_tokenizer.py basic_clean
    with pytest.raises(RuntimeError):
        clip.tokenize("road " * 100, bpe_path=path)
    cut = clip.tokenize("road " * 100, bpe_path=path, truncate=True)[0]
    assert cut[-1] == eot and int((cut != 0).sum()) == 77


REF_BPE = "/root/reference/CLIP/clip/bpe_simple_vocab_16e6.txt.gz"


@pytest.mark.skipif(not os.path.exists(REF_BPE), reason="the reference tree only exists in the authoring container")
def test_tokenize_equals_the_reference_tokenizer():
    """Same ids as the reference's SimpleTokenizer (clip/simple_tokenizer.py, imported by file path with a pass-through ftfy stub --
    ftfy is not installed here and only repairs mojibake) on the metric's prompts and some awkward ones."""
    import importlib.util
    import sys
    import types
    from fatezero_amd import clip
    sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda s: s))
    spec = importlib.util.spec_from_file_location("ref_simple_tokenizer", "/root/reference/CLIP/clip/simple_tokenizer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref = mod.SimpleTokenizer(REF_BPE)
    sot, eot = ref.encoder["<|startoftext|>"], ref.encoder["<|endoftext|>"]
    prompts = ["a silver jeep driving down a curvy road in the countryside", "a Porsche car driving down a curvy road in the countryside",
               "watercolor painting of a silver jeep driving down a curvy road", "A squirrel, eating a carrot!", "van gogh style: rabbit's jump",
               "  two   spaces\tand tabs ", "swarovski crystal swan swimming in a river near a wall and bushes", "3 cats & 12 tigers (cartoon)"]
    got = clip.tokenize(prompts, bpe_path=REF_BPE)
    for p, row in zip(prompts, got):
        want = [sot] + ref.encode(p) + [eot]
        assert row[:len(want)].tolist() == want and bool((row[len(want):] == 0).all()), p


def test_metric_cli_with_the_native_encoder(tmp_path, capsys):
    """`python -m fatezero_amd.metrics` end to end on the emulator: a tiny OpenAI-format checkpoint file (state dict of the tiny case),
    a synthetic merges file, two result folders of PNG frames and a prompt YAML in the layout of CLIP/bench_clean_prompt.yaml; the
    numbers must equal frame_metrics on features computed step by step with the same model."""
    import json
    from fatezero_amd import clip, metrics
    meta = json.load(open(os.path.join(CC.GOLD, "clip_meta.json")))["clip_tiny"]
    sd = CC.clip_weights([(k, tuple(s)) for k, s in meta["state_dict_shapes"]])
    # the tokenizer's vocabulary must fit the checkpoint's embedding table: rebuild the table for the synthetic vocabulary size
    bpe = str(tmp_path / "bpe_simple_vocab_16e6.txt.gz")
    vocab = _write_bpe(bpe)
    g = torch.Generator().manual_seed(3)
    sd["token_embedding.weight"] = torch.randn(len(vocab), sd["token_embedding.weight"].shape[1], generator=g) * 0.02
    ckpt = str(tmp_path / "tiny_clip.pt")
    torch.save(sd, ckpt)
    rng = np.random.RandomState(5)
    prompts = {"car_a": {"source": "a silver jeep driving down a road", "target": "a porsche car driving down a road"},
               "car_b": {"source": "a rabbit eating a carrot", "target": "a squirrel eating a carrot"}}
    for name, n in (("car_a", 3), ("car_b", 2)):
        os.makedirs(tmp_path / "results" / name)
        for i in range(n):
            Image.fromarray(rng.randint(0, 256, (40 + 8 * i, 36, 3), dtype=np.uint8)).save(str(tmp_path / "results" / name / f"{i:05d}.png"))
    import yaml
    yaml.safe_dump(prompts, open(tmp_path / "prompts.yaml", "w"))
    out = metrics.main(["--results", str(tmp_path / "results"), "--prompts", str(tmp_path / "prompts.yaml"), "--clip", ckpt, "--bpe", bpe,
                        "--device", "cpu"])
    text = capsys.readouterr().out
    assert "dataset_average_rate" in text and "folder_temporal_consistency" in text
    model, pre = clip.load(ckpt, device="cpu")
    want_rate, want_con = [], []
    for name in ("car_a", "car_b"):
        files = sorted(os.listdir(tmp_path / "results" / name))
        imgs = torch.stack([pre(metrics.crop_read_image_path(str(tmp_path / "results" / name / f))) for f in files])
        fi = model.encode_image(imgs)
        ft = model.encode_text(clip.tokenize([prompts[name]["source"], prompts[name]["target"]], bpe_path=bpe))
        a, c = metrics.frame_metrics(fi, ft, float(model.logit_scale.exp()))
        want_rate.append(a)
        want_con.append(c)
    assert abs(out["dataset_average_rate"] - sum(want_rate) / 2) < 1e-6
    assert abs(out["dataset_average_tempconst"] - sum(want_con) / 2) < 1e-5


def test_clip_vitb32_matches_reference_model_emu():
    # the real ViT-B/32 dimensions (12 + 12 layers, 151 M parameters): ~12 s on the emulator
    res = CC.run_clip_case("clip_vitb32", "cpu")
    print(res)
    CC.check(res)


REF_PROMPTS = "/root/reference/CLIP/bench_clean_prompt.yaml"


@pytest.mark.skipif(not os.path.exists(REF_PROMPTS), reason="the reference tree only exists in the authoring container")
def test_reference_prompt_yaml_loads():
    from fatezero_amd.config_driver import load_config
    cfg = load_config(REF_PROMPTS)
    assert cfg["swan_duck"]["source"].startswith("a black swan") and "target" in cfg["swan_cartoon"]


def test_reference_import_paths_resolve():
    """`import clip` with CLIP/ on the path (CLIP/frame_acc_tem_con.py:2) and the script's own file name."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "CLIP"))
    try:
        sys.modules.pop("clip", None)
        clip = importlib.import_module("clip")
        assert clip.load.__module__ == "fatezero_amd.clip" and "ViT-B/32" in clip.available_models()
        assert callable(clip.tokenize) and hasattr(clip, "CLIP")
    finally:
        sys.path.remove(os.path.join(root, "CLIP"))
        sys.modules.pop("clip", None)
    src = open(os.path.join(root, "CLIP", "frame_acc_tem_con.py")).read()
    assert "fatezero_amd.metrics" in src
    with pytest.raises(FileNotFoundError):
        from fatezero_amd import clip as native_clip
        native_clip.load("ViT-B/32", device="cpu", download_root=os.path.join(root, "no_such_dir"))  # nothing is downloaded here
