# Round 4: flash segment timeline (finer split), then the grouped (2-D) tile order A/B: kernel level and job level.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O $R/build_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFZ_FLASH_TIMING -o $R/build_tmp/flash_timing $R/scripts/flash_timing.hip > $O/ft_build.log 2>&1
(timeout 60 $R/build_tmp/flash_timing) > $O/flash_timing.txt 2>&1; cat $O/flash_timing.txt
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py -x -q) > $O/ktests.log 2>&1; tail -2 $O/ktests.log
for i in 1 2; do
  FZ_IGEMM_NO_TILE_ORDER=1 timeout 200 python scripts/tile_order_ab.py > $O/to_off_$i.json 2>> $O/to.err
  timeout 200 python scripts/tile_order_ab.py > $O/to_on_$i.json 2>> $O/to.err
done
python - <<PY
import json
off=[json.load(open("$O/to_off_%d.json"%i))["us"] for i in (1,2)]
on=[json.load(open("$O/to_on_%d.json"%i))["us"] for i in (1,2)]
for k in off[0]:
    print(f"{k:34s} a-fastest {off[0][k]:8.1f} {off[1][k]:8.1f} us   grouped order {on[0][k]:8.1f} {on[1][k]:8.1f} us  {min(off[0][k],off[1][k])/min(on[0][k],on[1][k]):5.2f}x")
PY
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
for i in 1 2 3; do
  (FZ_IGEMM_NO_TILE_ORDER=1 timeout 200 $B | python -c "import sys,json; print('A a-fastest always ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
  (timeout 200 $B | python -c "import sys,json; print('B grouped tile order', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
done
