# (Ran against a TRIAL build whose fz_lora_pair_gn wrote chunk-major records; measured no better than the shipped layout and not kept:
# profiles/r04_lora_pair_gn_ab.txt.  On the shipped tree this script repeats scripts/r04y.sh.)
# Round 4 (chunk-major record layout): GroupNorm partials out of fz_lora_pair's epilogue (fz_lora_pair_gn): parity, kernel-level cost, job-level A/B.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z; mkdir -p $O
(timeout 100 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora_pair or gn_from or groupnorm") > $O/ktests.log 2>&1; tail -2 $O/ktests.log
timeout 60 python scripts/lora_pair_gn_ab.py > $O/lora_pair_gn_ab.json 2> $O/lora_pair_gn_ab.txt; cat $O/lora_pair_gn_ab.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
for i in 1 2; do
  (FZ_NO_LORA_PAIR_GN=1 timeout 100 $B | python -c "import sys,json; print('A pair, statistics kernel  ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
  (timeout 100 $B | python -c "import sys,json; print('B pair + partials (default)', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
done
