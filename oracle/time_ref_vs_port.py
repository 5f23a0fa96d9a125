#!/usr/bin/env python3
"""TEST / BENCH INFRASTRUCTURE (authoring container only: /root/reference does not exist on the GPU box).

bench.py's `cpu_baseline` times the fp32 CPU PORT of the loop (oracle/fatezero_oracle.py, kind = "port"); BASELINE.md section 3 speaks of
"the reference's own video_diffusion code".  This script times BOTH on the same host, same thread pool, same weights, same inputs: one
UNet forward of a 3-frame 512x512 clip at full SD-1.x width (the unit bench.py's sample is made of), inversion mode (B = 1) and CFG
edit mode (B = 2), through
  * the UNMODIFIED reference UNetPseudo3DConditionModel (imported from /root/reference exactly as oracle/gen_golden.py does:
    diffusers 0.11.1 classes restated in oracle/refshim), fp32, eager PyTorch, no controller registered, and
  * oracle.OracleUNet (the port), no controller,
and prints the ratio, so that the port-timed baseline can be read as a reference-timed one.

    python oracle/time_ref_vs_port.py [threads] > profiles/r04_cpu_ref_vs_port.txt
"""
import importlib.util
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "refshim"))
from stubs import install  # noqa: E402

install()
sys.path.insert(0, REF)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ROOT]
import torch  # noqa: E402

torch.cuda.get_device_name = lambda *a, **k: "cpu"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


weights = _load("_oracle_weights", os.path.join(HERE, "weights.py"))
from video_diffusion.models.unet_3d_condition import UNetPseudo3DConditionModel  # noqa: E402
import video_diffusion as _vd  # noqa: E402
assert list(_vd.__path__)[0].startswith(REF), "the timed module must be the unmodified reference"
O = _load("fatezero_oracle_port", os.path.join(HERE, "fatezero_oracle.py"))

threads = int(sys.argv[1]) if len(sys.argv) > 1 else torch.get_num_threads()
torch.set_num_threads(threads)
SD15 = dict(block_out_channels=(320, 640, 1280, 1280), norm_num_groups=32, cross_attention_dim=768, attention_head_dim=8)
mc = {"lora": 160}
ref = UNetPseudo3DConditionModel(sample_size=64, model_config=mc, **SD15).eval()
shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
sd = weights.procedural_state_dict(shapes)
ref.load_state_dict(sd)
port = O.OracleUNet(sd, O.UNetConfig(**SD15, model_config=mc))
g = torch.Generator().manual_seed(1)
F = 3
print(f"host: {os.cpu_count()} logical CPUs, torch threads {threads}; one UNet forward, {F} frames x 512^2 (latents {F}x64x64x4), fp32, full SD-1.x width, lora 160")
for mode, B in (("inversion (B=1)", 1), ("CFG edit (B=2)", 2)):
    z = torch.randn(B, 4, F, 64, 64, generator=g)
    ctx = torch.randn(B, 77, 768, generator=g)
    res = {}
    with torch.no_grad():
        for name, fn in (("reference", lambda: ref(z, 481, ctx).sample), ("port", lambda: port(z, 481, ctx))):
            fn()  # warm-up
            ts = []
            for _ in range(2):
                t0 = time.time()
                y = fn()
                ts.append(time.time() - t0)
            res[name] = (min(ts), y)
    err = float((res["reference"][1] - res["port"][1]).abs().max())
    print(f"{mode:16s} reference {res['reference'][0]:7.2f} s   port {res['port'][0]:7.2f} s   reference / port = "
          f"{res['reference'][0] / res['port'][0]:.3f}   max |reference - port| = {err:.2e} (scale {float(res['reference'][1].abs().max()):.2f})")
