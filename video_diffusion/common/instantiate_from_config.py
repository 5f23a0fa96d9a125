from fatezero_amd.video_diffusion.common.instantiate_from_config import get_obj_from_str, instantiate_from_config  # noqa: F401
