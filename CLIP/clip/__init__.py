"""`import clip` as the reference's metric script does it (CLIP/frame_acc_tem_con.py:2 with CLIP/ on the path): an alias of
fatezero_amd.clip -- OpenAI CLIP on the native MI355X kernels (load, tokenize, available_models, the CLIP class)."""
from fatezero_amd.clip import CLIP, available_models, build_model, load, tokenize  # noqa: F401
