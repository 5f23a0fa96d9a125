R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
(timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "temporal or gemm") > $O/k.log 2>&1; tail -2 $O/k.log
for i in 1 2; do
  FZ_IGEMM_NO_154122=1 timeout 200 python scripts/tile154_ab.py > $O/off_$i.json 2>> $O/err
  timeout 200 python scripts/tile154_ab.py > $O/on_$i.json 2>> $O/err
done
python - <<PY
import json
off=[json.load(open("$O/off_%d.json"%i))["us"] for i in (1,2)]
on=[json.load(open("$O/on_%d.json"%i))["us"] for i in (1,2)]
for k in off[0]:
    print(f"{k:52s} without {off[0][k]:7.1f} {off[1][k]:7.1f}   with 154122 {on[0][k]:7.1f} {on[1][k]:7.1f}  {min(off[0][k],off[1][k])/min(on[0][k],on[1][k]):5.2f}x")
PY
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
for i in 1 2; do
  (FZ_IGEMM_NO_154122=1 timeout 200 $B | python -c "import sys,json; print('A without 154122', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
  (timeout 200 $B | python -c "import sys,json; print('B with    154122', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
done
