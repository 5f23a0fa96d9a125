import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pipeline_cases as PC
from helpers import load_json, load_npz
from fatezero_amd.video_diffusion.models import resnet as R, attention as A, unet_3d_condition as U
from fatezero_amd.video_diffusion.prompt_attention import attention_util as AU
name = "pipe_refine_reweight_latentblend"
dev = "cuda"
meta = load_json("pipeline_meta.json")[name]; gz = load_npz(name + ".npz")
trace = []
def wrap(cls, label):
    orig = cls.forward_tokens
    def f(self, x, *a, **k):
        out = orig(self, x, *a, **k)
        o = out[0] if isinstance(out, tuple) else out
        trace.append((label, tuple(o.data.shape), float(o.data.float().abs().sum())))
        return out
    cls.forward_tokens = f
wrap(R.ResnetBlockPseudo3D, "resnet"); wrap(A.SpatioTemporalTransformerModel, "transformer"); wrap(R.PseudoConv3d, "conv")
o_sc = AU.AttentionControlEdit.step_callback
def sc(self, x_t):
    xin = float(x_t.float().abs().sum())
    y = o_sc(self, x_t)
    trace.append(("cb_input", 0, xin))
    for k in ("down_cross", "mid_cross", "up_cross"):
        for i, a in enumerate(self._sum_storage.get(k, [])):
            trace.append((f"sum_{k}_{i}", tuple(a.shape), float(a.double().sum())))
        for i, cm in enumerate(self._all_step_maps[-1][k]):
            trace.append((f"cur_{k}_{i}", tuple(cm.storage.shape), float(cm.storage.double().sum())))
    if self.latent_blend is not None and self.latent_blend.mask_list:
        trace.append(("latent_mask", 0, float(self.latent_blend.mask_list[-1].double().sum())))
    sis = len(self.additional_attention_store.latents_store) - self.cur_step
    trace.append(("inv_latent", sis, float(self.additional_attention_store.latents_store[sis].double().abs().sum())))
    trace.append(("step_callback", tuple(y.shape), float(y.float().abs().sum()))); return y
AU.AttentionControlEdit.step_callback = sc
res, pipe = PC.run_pipeline_case(name, dev, return_pipe=True)
print({k: res[k] for k in ("inv_err", "edit_err", "edit_scale")})
emb_tgt = torch.from_numpy(gz["emb_tgt"]).to(dev)
kw = dict(meta["kwargs"]); kw.pop("save_path", None)
runs = []
for r in range(5):
    trace.clear()
    out = pipe(latents=torch.from_numpy(gz["zT"]).to(dev), output_type="latent", **kw)
    ed = out["sdimage_output"].images.float().cpu()
    print("edit", r, "vs golden", float((ed - torch.from_numpy(gz["edited"])).abs().max()), "trace len", len(trace))
    runs.append(list(trace))
for r in (1, 2, 3, 4):
    for i, (a, b) in enumerate(zip(runs[0], runs[r])):
        if abs(a[2] - b[2]) > 1e-3 * max(1.0, abs(a[2])):
            print("run", r, "first divergence at", i, a, b, "prev", runs[0][i - 1] if i else None)
            break
    else:
        print("run", r, "identical traces")
