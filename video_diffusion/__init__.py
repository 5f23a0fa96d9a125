"""Alias of fatezero_amd.video_diffusion so that the reference's import paths and YAML `target:` strings
(`video_diffusion.pipelines.p2p_ddim_spatial_temporal.P2pDDIMSpatioTemporalPipeline`, ...) resolve unchanged."""
