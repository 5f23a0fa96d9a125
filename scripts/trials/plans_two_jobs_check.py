# Two full-width jobs in ONE process with the native issue plans on, against the walked pipeline: bit-identical outputs of both jobs -- in particular the
# second job's inversion replays the 8-frame plan AFTER the 16-frame kind made its lazily cached buffers (weight packs of the upsampler form among them).
#   python scripts/trials/plans_two_jobs_check.py [ddim steps]
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
z0 = torch.randn(1, 4, 8, 64, 64, generator=torch.Generator().manual_seed(1234)).to(dev)
outs = {}
for plans in (False, True):
    pipe = bench.build_pipeline(dev)
    if plans:
        pipe.unet.enable_issue_plans()
    outs[plans] = [bench.run_job(pipe, z0, T, dev).clone() for _ in range(2)]
    if plans:
        print("plan statistics:", {k: v for k, v in pipe.unet._issuer.stats.items() if not isinstance(v, list)})
    del pipe
    torch.cuda.empty_cache()
for j in range(2):
    a, b = outs[False][j], outs[True][j]
    print(f"job {j}: finite {bool(torch.isfinite(b.float()).all())}, bit-identical to the walked pipeline: {torch.equal(a, b)}, max |diff| {float((a.float() - b.float()).abs().max()):.3e}")
print("walked job 0 == walked job 1:", torch.equal(outs[False][0], outs[False][1]))
