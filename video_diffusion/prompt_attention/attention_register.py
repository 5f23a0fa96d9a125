from fatezero_amd.video_diffusion.prompt_attention.attention_register import register_attention_control  # noqa: F401
