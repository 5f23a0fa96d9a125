#!/usr/bin/env python3
"""Per-phase wall times (inversion / edit) of consecutive jobs on one pipeline."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda")
pipe = bench.build_pipeline(dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
z0 = torch.randn(1, 4, 8, 64, 64, device=dev)
for job in range(3):
    pipe.scheduler.set_timesteps(T)
    pipe.release_attention_maps()
    pipe.store_controller = type(pipe.store_controller)()
    emb_src = pipe._encode_prompt(bench.SRC_PROMPT, dev, 1, True, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1, text_embeddings=emb_src,
                                             store_attention=True, LOW_RESOURCE=True, latents=z0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = pipe(prompt=bench.TGT_PROMPT, source_prompt=bench.SRC_PROMPT, edit_type="swap", num_inference_steps=T,
               latents=lat[-1], output_type="latent", **bench.EDIT_KW)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"job {job}: inversion {1e3 * (t1 - t0) / T:.1f} ms/step, edit {1e3 * (t2 - t1) / T:.1f} ms/step, "
          f"mem {torch.cuda.memory_allocated() / 1e9:.1f} GB reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB")
