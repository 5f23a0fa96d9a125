# Round 4, second GPU call: parity of the new forms (fz_gemm_qkvt, flat split-K grid), then same-box A/B: kernel level (xcd_ks_ab.py,
# two interleaved pairs) and job level (bench.py: old mapping + two launches | new mapping | new mapping + fused q|k|V^T), interleaved.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py -x -q) > $O/ktests.log 2>&1; tail -3 $O/ktests.log
for i in 1 2; do
  FZ_IGEMM_NO_XCD_KS=1 timeout 200 python scripts/xcd_ks_ab.py > $O/ks_off_$i.json 2>> $O/ks.err
  timeout 200 python scripts/xcd_ks_ab.py > $O/ks_on_$i.json 2>> $O/ks.err
done
python - <<PY
import json
off=[json.load(open("$O/ks_off_%d.json"%i))["us"] for i in (1,2)]
on=[json.load(open("$O/ks_on_%d.json"%i))["us"] for i in (1,2)]
for k in off[0]:
    a=min(o[k] for o in off); b=min(o[k] for o in on)
    print(f"{k:34s} old {a:8.1f} us  new {b:8.1f} us  {a/b:5.2f}x")
PY
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
for i in 1 2; do
  (FZ_IGEMM_NO_XCD_KS=1 FZ_NO_QKV_FUSION=1 timeout 200 $B | python -c "import sys,json; print('A old mapping, two launches ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
  (FZ_NO_QKV_FUSION=1 timeout 200 $B | python -c "import sys,json; print('B xcd-ks,      two launches ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
  (timeout 200 $B | python -c "import sys,json; print('C xcd-ks,      fused qkvt   ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
done
