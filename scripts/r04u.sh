# Round 4: own kernels vs hipBLASLt / MIOpen on the SD-1.x shapes with the final launch geometry (the library's own tile choice).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04u; mkdir -p $O
(timeout 300 python scripts/kbench.py --gemm --nosweep) > $O/kbench_gemm.json 2> $O/gemm.err
(timeout 300 python scripts/kbench.py --conv --nosweep) > $O/kbench_conv.json 2> $O/conv.err
python - <<PY
import json
g = json.load(open("$O/kbench_gemm.json")); c = json.load(open("$O/kbench_conv.json"))
rg = sorted((v["fz_TF"] / v["lib_TF"], k) for k, v in g.items() if isinstance(v, dict) and v.get("lib_TF"))
print("GEMM: %d shapes, behind hipBLASLt on %d; worst five:" % (len(rg), sum(1 for r, k in rg if r < 1.0)), [(round(r, 2), k) for r, k in rg[:5]], "median", round(rg[len(rg) // 2][0], 2))
rc = sorted((v["fz_TF"] / v["miopen_TF"], k) for k, v in c.items() if isinstance(v, dict) and v.get("miopen_TF"))
print("conv: %d shapes, behind MIOpen on %d; min %.2f median %.2f" % (len(rc), sum(1 for r, k in rc if r < 1.0), rc[0][0], rc[len(rc) // 2][0]))
PY
