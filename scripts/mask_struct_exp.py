"""Round 4: how the full-width blend mask can be made non-degenerate with procedural weights (CPU oracle only, F = 1): quantiles of the
normalised 16^2 blend-word score for scaled blend-word context rows / smooth latents -- scaling saturates the softmax; the split is moved by
blend_th (tests/pipeline_cases.py: FULL_BLEND_TH)."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
from oracle import fatezero_oracle as O
from oracle.weights import procedural_state_dict
from helpers import ReplayTokenizer
import pipeline_cases as PC
from fatezero_amd.video_diffusion.models import UNetPseudo3DConditionModel
torch.set_num_threads(8)
F = 1
mc = {"lora": 160}
unet = UNetPseudo3DConditionModel(sample_size=64, **PC.SD15, **mc)
shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
sd = procedural_state_dict(shapes)
ounet = O.OracleUNet(sd, O.UNetConfig(**PC.SD15, model_config=mc))
tok = ReplayTokenizer()
al = O.blend_alpha_layers([PC.FULL_SRC, PC.FULL_TGT], [["silver", "jeep"], ["Porsche", "car"]], tok)
word_pos = [2, 3]
def smooth_latent(g, F, amp, n=8):
    base = torch.randn(1, 4, F, n, n, generator=g)
    up = torch.nn.functional.interpolate(base.reshape(1, 4 * F, n, n), size=(64, 64), mode="bicubic", align_corners=False)
    return up.reshape(1, 4, F, 64, 64) * amp
for (scale, amp, noise, n) in [(1, 0, 1.0, 8), (1, 2.0, 0.3, 4), (2, 2.0, 0.3, 4), (3, 3.0, 0.2, 4), (0.5, 3.0, 0.2, 4)]:
    g = torch.Generator().manual_seed(11)
    z0 = torch.randn(1, 4, F, 64, 64, generator=g) * noise + smooth_latent(g, F, amp, n)
    emb = torch.randn(2, 77, 768, generator=g) * 0.5
    emb[:, word_pos] *= scale
    st = O.StoreController()
    lat = O.ddim_inversion(ounet, O.DDIMSchedule(1), z0, emb[1:], st)
    sd0 = st.attention_store_all_step[0]
    maps = sd0["down_cross"][2:4] + sd0["up_cross"][:3]
    rr = []
    for item in maps:
        item = item[None]
        p, c, heads, r, w = item.shape
        rr.append(item.reshape(p, c, heads, 16, 16, w).permute(0, 2, 1, 3, 4, 5).float())
    m = (torch.cat(rr, 1) * al[0:1]).sum(-1).mean(1)   # [1,F,16,16]
    mp = torch.nn.functional.max_pool2d(m, 3, 1, 1)
    nm = mp / mp.amax(dim=(-2, -1), keepdim=True)
    raw = m / m.amax(dim=(-2, -1), keepdim=True)
    q = torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9])
    print(f"scale {scale} amp {amp} noise {noise} n {n}: raw map mean {float(m.mean()):.4f}; pooled/max quantiles",
          [round(float(x), 3) for x in torch.quantile(nm.flatten(), q)], "unpooled", [round(float(x), 3) for x in torch.quantile(raw.flatten(), q)], flush=True)
