"""Attribute the per-dispatch counter rows of scripts/pmc_job.sh to the kernel classes bench.py reports (by kernel name; the MODE-0
igemm dispatches by the launch log bench.py wrote in the same run) and write bytes per launch per class."""
import collections
import csv
import glob
import json
import re
import sys

out_dir, out_json, ddim = sys.argv[1], sys.argv[2], int(sys.argv[3])


def klass(name):
    if "attn_flash_kernel" in name:
        return "flash" if re.search(r"ILi40E|<40,", name) else "flash_other"
    if "attn_self_kernel" in name:  # <D, MODE, ABL>: MODE 1 = FZ_ATTN_CAPTURE, 2 = FZ_ATTN_INJECT (csrc/attn_self.hip)
        m = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"<(\d+), ?(\d+), ?(\d+)>", name)
        return ("capture" if m.group(2) == "1" else "inject") if m else "attn_self"
    if "ff_chain_kernel" in name:
        return "ff_chain"
    if "xattn_chain_kernel" in name:
        return "xattn_chain"
    if "conv_halo_kernel" in name:  # the halo form of the 3x3 convolution, whole / in K slices / as the upsampler's four 2x2 convolutions (csrc/conv_halo.hip)
        return "conv3x3"
    if "igemm_reduce" in name:
        return "splitk_reduce"
    if "lora_pair_kernel" in name:  # both temporal LoRA convolutions in one launch (csrc/lora_pair.hip)
        return "temporal_conv"
    if "igemm_kernel" in name:
        m = re.search(r"igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
        mode = int(m.group(7))
        return {0: "MODE0", 1: "conv3x3", 3: "conv3x3", 2: "temporal_conv"}[mode]
    for key, c in (("gn_", "groupnorm"), ("layernorm", "layernorm"), ("attn_cross", "cross_attn"), ("attn_temporal", "temporal_attn"),
                   ("conv3x3_small", "conv_in")):
        if key in name:
            return c
    return "other"


res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"{out_dir}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0)))
    rows.sort()
    log = json.load(open(f"{out_dir}/launchlog_{ctr}.json"))
    mode0 = [r for r in rows if klass(r[1]) == "MODE0"]
    assert len(mode0) == len(log), (ctr, len(mode0), len(log))
    it = iter(log)
    agg = collections.defaultdict(lambda: [0, 0.0])
    shapes = collections.defaultdict(lambda: [0, 0.0])  # per (class, kernel template, grid size): which launch shapes over-fetch
    for did, name, val, grid in rows:
        c = klass(name)
        if c == "MODE0":
            c = next(it)
        agg[c][0] += 1
        agg[c][1] += val
        if "igemm_kernel" in name:
            m = re.search(r"igemm_kernel<([^>]*)>", name)
            shapes[(c, m.group(1) if m else name[:40], grid)][0] += 1
            shapes[(c, m.group(1) if m else name[:40], grid)][1] += val
    res[ctr] = agg
    res[ctr + "_shapes"] = shapes
classes = {}
for c in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    nf, f = res["FETCH_SIZE"].get(c, [0, 0.0])
    nw, w = res["WRITE_SIZE"].get(c, [0, 0.0])
    n = max(nf, nw, 1)
    classes[c] = {"launches": n, "FETCH_SIZE_KiB_per_launch": f / max(nf, 1), "WRITE_SIZE_KiB_per_launch": w / max(nw, 1),
                  "traffic_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
                  "traffic_GB_per_job": (2.0 * f + w) * 1024.0 / 1e9}
shape_rows = []
for key in sorted(set(res["FETCH_SIZE_shapes"]) | set(res["WRITE_SIZE_shapes"])):
    nf, f = res["FETCH_SIZE_shapes"].get(key, [0, 0.0])
    nw, w = res["WRITE_SIZE_shapes"].get(key, [0, 0.0])
    shape_rows.append({"class": key[0], "igemm_kernel": key[1], "grid_threads": key[2], "launches": max(nf, nw),
                       "fetch_MB_per_launch": 2.0 * f / max(nf, 1) * 1024 / 1e6, "write_MB_per_launch": w / max(nw, 1) * 1024 / 1e6})
shape_rows.sort(key=lambda r: -(r["fetch_MB_per_launch"] + r["write_MB_per_launch"]) * r["launches"])
json.dump({"igemm_launch_shapes": shape_rows[:80], "what": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two runs) over `bench.py --steps 1 --warmup 0 --ddim-steps {ddim}`: "
                   "every dispatch of the job, attributed by kernel name (projection GEMMs: by bench.py's launch log); "
                   "traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; gfx950 tallies a 128-B read request as 64 B)",
           "classes": classes}, open(out_json, "w"), indent=1)
print(json.dumps({k: round(v["traffic_GB_per_job"], 2) for k, v in classes.items()}))
