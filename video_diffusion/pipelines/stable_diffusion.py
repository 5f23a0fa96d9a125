from fatezero_amd.video_diffusion.pipelines.stable_diffusion import SpatioTemporalStableDiffusionPipeline  # noqa: F401
