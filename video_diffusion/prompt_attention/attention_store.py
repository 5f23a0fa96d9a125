from fatezero_amd.video_diffusion.prompt_attention.attention_store import *  # noqa: F401,F403
from fatezero_amd.video_diffusion.prompt_attention.attention_store import AttentionControl, AttentionStore  # noqa: F401
