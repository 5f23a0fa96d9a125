"""Command line / YAML entry point (reference: test_fatezero.py:46-262).

    python test_fatezero.py --config config/teaser/jeep_posche.yaml

`test(config=..., **yaml)` keeps the reference's keyword signature and control flow -- load tokenizer / text encoder / VAE /
inflated UNet / DDIM scheduler from `pretrained_model_path`, build the pipeline named by `test_pipeline_config.target`, read the
clip with `ImageSequenceDataset`, DDIM-invert it with the source prompt (`editing_config.use_invertion_latents`, capturing
attention when `use_inversion_attention`), then let `P2pSampleLogger` run every editing prompt and write the gif / png / mp4
results under `logdir` -- on one MI355X, without accelerate / OmegaConf / diffusers / transformers: the YAML loader is
`fatezero_amd.config_driver.load_config`, the models are the native ones of `fatezero_amd.video_diffusion.models`.
"""
import copy
import datetime
import logging
import os
from glob import glob
from typing import Dict, Optional

import torch

from . import config_driver
from .video_diffusion.common.image_util import log_train_samples
from .video_diffusion.common.instantiate_from_config import instantiate_from_config
from .video_diffusion.data.dataset import ImageSequenceDataset
from .video_diffusion.models.clip_text import CLIPTextModel, CLIPTokenizer
from .video_diffusion.models.unet_3d_condition import UNetPseudo3DConditionModel
from .video_diffusion.models.vae import AutoencoderKL
from .video_diffusion.pipelines.p2p_validation_loop import P2pSampleLogger
from .video_diffusion.schedulers import DDIMScheduler

DEFAULT_PIPELINE = "video_diffusion.pipelines.stable_diffusion.SpatioTemporalStableDiffusionPipeline"


def get_time_string() -> str:
    return datetime.datetime.now().strftime("%y%m%d-%H%M%S")


def _plain(node):
    """config_driver.Config / nested containers -> plain dict / list (for the config.yml dump)."""
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_plain(v) for v in node]
    return node


def _logger(logdir: str) -> logging.Logger:
    log = logging.getLogger("fatezero")
    log.setLevel(logging.INFO)
    for h in list(log.handlers):
        log.removeHandler(h)
    fh = logging.FileHandler(os.path.join(logdir, "log.txt"))
    fh.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
    log.addHandler(fh)
    log.addHandler(logging.StreamHandler())
    return log


def set_seed(seed: int):
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def collate_fn(examples):
    """One batch from dataset items (test_fatezero.py:34-42)."""
    return {"prompt_ids": torch.cat([e["prompt_ids"] for e in examples], dim=0),
            "images": torch.stack([e["images"] for e in examples])}


def test(config: str, pretrained_model_path: str, dataset_config: Dict, logdir: str = None, editing_config: Optional[Dict] = None,
         test_pipeline_config: Optional[Dict] = None, gradient_accumulation_steps: int = 1, seed: Optional[int] = None,
         mixed_precision: Optional[str] = "fp16", batch_size: int = 1, model_config: dict = {}, verbose: bool = True,
         device: Optional[str] = None, **kwargs):
    """Returns the dict {"logdir", "samples", "latents_all_step"} (the reference returns nothing; its outputs are the files)."""
    args = dict(config=config, pretrained_model_path=pretrained_model_path, dataset_config=dataset_config, logdir=logdir,
                editing_config=editing_config, test_pipeline_config=test_pipeline_config, seed=seed, mixed_precision=mixed_precision,
                batch_size=batch_size, model_config=model_config, verbose=verbose, **kwargs)
    if logdir is None:
        logdir = config.replace("config", "result").replace(".yml", "").replace(".yaml", "")
    logdir += f"_{get_time_string()}"
    os.makedirs(logdir, exist_ok=True)
    import yaml
    with open(os.path.join(logdir, "config.yml"), "w") as f:
        yaml.safe_dump(_plain(args), f)
    logger = _logger(logdir)
    if mixed_precision not in (None, "no", "fp16"):
        raise NotImplementedError(f"mixed_precision={mixed_precision}: the MI355X engine stores fp16 and accumulates in fp32")
    if seed is not None:
        set_seed(seed)
    device = torch.device(device if device is not None else ("cuda:0" if torch.cuda.is_available() else "cpu"))

    tokenizer = CLIPTokenizer.from_pretrained(pretrained_model_path, subfolder="tokenizer")
    text_encoder = CLIPTextModel.from_pretrained(pretrained_model_path, subfolder="text_encoder")
    vae = AutoencoderKL.from_pretrained(pretrained_model_path, subfolder="vae")
    unet = UNetPseudo3DConditionModel.from_2d_model(os.path.join(pretrained_model_path, "unet"), model_config=model_config)

    test_pipeline_config = dict(test_pipeline_config or {})
    test_pipeline_config.setdefault("target", DEFAULT_PIPELINE)
    pipeline = instantiate_from_config(test_pipeline_config, vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet,
                                       scheduler=DDIMScheduler.from_pretrained(pretrained_model_path, subfolder="scheduler"),
                                       disk_store=kwargs.get("disk_store", False))
    pipeline.scheduler.set_timesteps(editing_config["num_inference_steps"])
    pipeline.set_progress_bar_config(disable=True)
    pipeline.print_pipeline(logger)
    for m in (vae, unet, text_encoder):
        m.requires_grad_(False)
        m.eval()
        m.to(device=device, dtype=torch.float16)

    prompt_ids = tokenizer(dataset_config["prompt"], truncation=True, padding="max_length", max_length=tokenizer.model_max_length,
                           return_tensors="pt").input_ids
    video_dataset = ImageSequenceDataset(**dataset_config, prompt_ids=prompt_ids)
    loader = torch.utils.data.DataLoader(video_dataset, batch_size=batch_size, shuffle=True, num_workers=0, collate_fn=collate_fn)
    log_train_samples(save_path=os.path.join(logdir, "train_samples.gif"), train_dataloader=loader)
    batch = next(iter(loader))
    assert batch["images"].shape[0] == 1, "Only support, overfiting on a single video"
    b, c, f, h, w = batch["images"].shape
    images = batch["images"].to(device=device, dtype=torch.float16).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)  # (b f) c h w

    latents_all_step = None
    init_latents = None
    sample_logger = None
    if editing_config is not None:
        sample_logger = P2pSampleLogger(**editing_config, logdir=logdir, source_prompt=dataset_config["prompt"])
    if editing_config.get("use_invertion_latents", False):  # (sic) precompute the inverted latents of this video
        text_embeddings = pipeline._encode_prompt(dataset_config["prompt"], device=device, num_images_per_prompt=1,
                                                  do_classifier_free_guidance=True, negative_prompt=None)
        latents_all_step = pipeline.prepare_latents_ddim_inverted(
            images, batch_size=1, num_images_per_prompt=1, text_embeddings=text_embeddings, prompt=dataset_config["prompt"],
            store_attention=editing_config.get("use_inversion_attention", False), LOW_RESOURCE=True,
            save_path=logdir if verbose else None)
        init_latents = latents_all_step[-1]
    samples = None
    if sample_logger is not None:
        samples = sample_logger.log_sample_images(image=images, pipeline=pipeline, device=device, step=0, latents=init_latents,
                                                  save_dir=logdir if verbose else None)
    logger.info("results in %s", logdir)
    return {"logdir": logdir, "samples": samples, "latents_all_step": latents_all_step}


def run_config_file(config: str, **overrides):
    """test_fatezero.py:254-276: one run for a checkpoint folder that holds `unet/`, otherwise one per `checkpoint_*` child."""
    cfg = config_driver.load_config(config)
    cfg.update(overrides)
    root = cfg["pretrained_model_path"]
    if "unet" in os.listdir(root):
        return [test(config=config, **cfg)]
    results = []
    for checkpoint in sorted(glob(os.path.join(root, "checkpoint_*"))):
        epoch = checkpoint.split("_")[-1]
        if "pretrained_epoch_list" not in cfg or int(epoch) in cfg["pretrained_epoch_list"]:
            one = copy.deepcopy(dict(cfg))
            one["pretrained_model_path"] = checkpoint
            if "logdir" not in one:
                logdir = config.replace("config", "result").replace(".yml", "").replace(".yaml", "")
                one["logdir"] = logdir + f"/{os.path.basename(checkpoint)}"
            results.append(test(config=config, **one))
    return results


def run():
    import click

    @click.command()
    @click.option("--config", type=str, default="config/sample.yml")
    def _main(config):
        run_config_file(config)

    _main()


if __name__ == "__main__":
    run()
