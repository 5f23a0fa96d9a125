from fatezero_amd.video_diffusion.prompt_attention.attention_util import *  # noqa: F401,F403
from fatezero_amd.video_diffusion.prompt_attention.attention_util import (AttentionStore, EmptyControl, make_controller,  # noqa: F401
    register_attention_control, AttentionReplace, AttentionRefine, AttentionReweight, get_equalizer, show_cross_attention)
