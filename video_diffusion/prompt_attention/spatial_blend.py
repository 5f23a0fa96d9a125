from fatezero_amd.video_diffusion.prompt_attention.spatial_blend import SpatialBlender  # noqa: F401
