# Round 4: fz_lora_pair with the deeper rings and the >= 256-workgroup rule: parity, kernel-level A/B, job-level A/B.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04w; mkdir -p $O
(timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora_pair") > $O/ktests.log 2>&1; tail -3 $O/ktests.log
timeout 300 python scripts/lora_pair_ab.py > $O/lora_pair_ab.json 2> $O/lora_pair_ab.txt; cat $O/lora_pair_ab.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
for i in 1 2 3; do
  (FZ_NO_LORA_PAIR=1 timeout 200 $B | python -c "import sys,json; print('A two launches ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
  (timeout 200 $B | python -c "import sys,json; print('B fz_lora_pair  ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
done
