#!/usr/bin/env python3
"""GroupNorm(+SiLU) on the SD-1.x pyramid shapes: fz_groupnorm (one launch where a group fits a workgroup) against the three-kernel
form driven through its split entry points (fz_groupnorm_stats + fz_groupnorm_apply = stats, finalize, apply).  GPU time per call from
the kernel trace (launches this small are host-bound under HIP events):
    rocprofv3 --kernel-trace --output-format csv -d DIR -o gn -- python scripts/gn_ab.py run
    python scripts/gn_ab.py report DIR/**/gn_kernel_trace.csv
Segments are separated by the fill kernel of a float64 marker tensor in the trace."""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(f, t, c, 8) for f in (8, 16) for (t, c) in ((64, 1280), (64, 2560), (256, 1280), (256, 1920), (256, 2560), (1024, 640),
                                                        (1024, 320), (1024, 960), (4096, 320))]
SHAPES += [(f, t, c, 1) for f in (8, 16) for (t, c) in ((1024, 640), (256, 1280), (4096, 320))]  # per-frame statistics
REP = 10


def run():
    import torch
    from fatezero_amd import _native
    if os.environ.get("FZ_VARIANT_LIB"):  # a trial build of the kernels (scripts/build_variant.sh), never the product's library
        _native.HIP_LIB = os.environ["FZ_VARIANT_LIB"]
    from fatezero_amd import kernels as K
    dev = "cuda"
    marker = torch.empty(64, device=dev, dtype=torch.float64)  # its fill kernel (FillFunctor<double>) is the segment separator
    for (f, t, c, span) in SHAPES:
        marker.fill_(0.0)  # segment 3 i: set-up and warm-up calls of shape i
        x = torch.randn(f, t, c, device=dev).half()
        g = torch.ones(c, device=dev).half(); b = torch.zeros(c, device=dev).half()
        out = torch.empty_like(x)
        K.groupnorm(x, g, b, span=span, groups=32, eps=1e-5, silu=True, out=out)  # plans / scratch
        p = K.groupnorm_stats(x, groups=32)
        torch.cuda.synchronize()
        marker.fill_(1.0)
        for _ in range(REP):
            K.groupnorm(x, g, b, span=span, groups=32, eps=1e-5, silu=True, out=out)
        marker.fill_(2.0)
        for _ in range(REP):
            p = K.groupnorm_stats(x, groups=32)
            K.groupnorm_apply(x, g, b, p.view(f // span, span, *p.shape[1:]), span=span, groups=32, eps=1e-5, silu=True, out=out)
        torch.cuda.synchronize()
    marker.fill_(3.0)
    torch.cuda.synchronize()


def report(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    segs, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "FillFunctor<double>" in name:
            cur = []
            segs.append(cur)
        elif cur is not None and name.startswith(("gn_", "void gn_")):
            cur.append((name.split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("frames tokens     C span | fz_groupnorm: us per call (kernels per call) | stats + finalize + apply: us per call")
    for i, sh in enumerate(SHAPES):
        one, three = segs[3 * i + 1], segs[3 * i + 2]
        print(f"{sh[0]:4d} {sh[1]:6d} {sh[2]:6d} {sh[3]:3d}   | {sum(d for _, d in one) / REP:8.1f} ({len(one) // REP}) | {sum(d for _, d in three) / REP:8.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
