# round-3 GPU call q: own kernels vs hipBLASLt / MIOpen on the SD-1.x shapes (scripts/kbench.py, library's own tile choice)
O=gpurun_out/r03q; mkdir -p $O
(timeout 300 python scripts/kbench.py --gemm --nosweep) > $O/kbench_gemm.json 2> $O/gemm.err
(timeout 300 python scripts/kbench.py --conv --nosweep) > $O/kbench_conv.json 2> $O/conv.err
python - <<PY
import json
for n in ("gemm","conv"):
    d=json.load(open("$O/kbench_%s.json"%n))
    behind=[]
    for k,v in d.items():
        lib=v.get("hipblaslt_TF", v.get("miopen_TF")); fz=v.get("fz_TF")
        if lib and fz and fz<lib: behind.append((k, round(fz), round(lib)))
    print(n, len(d), "shapes; behind the library on", len(behind), behind)
PY
