import sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import pipeline_cases as PC
from helpers import ReplayTokenizer, load_json, load_npz
from oracle import fatezero_oracle as O
name = "pipe_refine_reweight_latentblend"
dev = "cuda"
meta = load_json("pipeline_meta.json")[name]; consts = load_json("host_constants.json")[meta["prompt_case"]]; gz = load_npz(name + ".npz")
res, pipe = PC.run_pipeline_case(name, dev, return_pipe=True)
print({k: res[k] for k in ("inv_err", "edit_err", "edit_scale")})
# native edit again, recording per-step latents
emb_tgt = torch.from_numpy(gz["emb_tgt"]).to(dev)
kw = dict(meta["kwargs"]); kw.pop("save_path", None)
steps = []
out = pipe(latents=torch.from_numpy(gz["zT"]).to(dev), output_type="latent", callback=lambda i, t, l: steps.append(l.float().cpu().clone()), **kw)
ed2 = out["sdimage_output"].images.float().cpu()
print("second native edit vs golden", float((ed2 - torch.from_numpy(gz["edited"])).abs().max()))
# oracle per-step
store = pipe.store_controller
shapes = load_json("unet_meta.json")["unet_tiny16_default"]["state_dict_shapes"]
from oracle.weights import procedural_state_dict
cfg = O.UNetConfig(**PC.TINY["tiny16"], model_config=meta["model_config"])
unet = O.OracleUNet(procedural_state_dict([(n, tuple(s)) for n, s in shapes]), cfg)
ost = O.StoreController()
ost.attention_store_all_step = [{k: [t.float().cpu() for t in v] for k, v in d.items()} for d in store.attention_store_all_step]
ost.latents_store = [t.float().cpu() for t in store.latents_store]
ctrl = O.make_edit_controller(ReplayTokenizer(), consts["prompts"], ost, meta["T"], kw["is_replace_controller"], dict(kw["cross_replace_steps"]),
    kw["self_replace_steps"], blend_words=kw.get("blend_words"), eq_params=kw.get("eq_params"), blend_th=tuple(kw["blend_th"]),
    blend_self_attention=kw.get("blend_self_attention", False), blend_latents=kw.get("blend_latents", False), save_self_attention=kw["save_self_attention"])
sched = O.DDIMSchedule(meta["T"])
lat = torch.from_numpy(gz["zT"]); emb = torch.from_numpy(gz["emb_tgt"])
for i, t in enumerate(sched.timesteps):
    t = int(t)
    eps2 = unet(torch.cat([lat] * 2), t, emb, ctrl); eu, ec = eps2.chunk(2)
    lat_pre = sched.step(eu + kw["guidance_scale"] * (ec - eu), t, lat)
    lat = ctrl.step_callback(lat_pre)
    print(f"step {i}: native-vs-oracle {float((steps[i] - lat).abs().max()):.4f}  (oracle pre-blend vs native {float((steps[i] - lat_pre).abs().max()):.4f}) |lat| {float(lat.abs().max()):.2f}")
