# round-3 GPU call j: same-box, process-interleaved A/B of the whole job: the round-2 tree (build_tmp/r02_tree = commit 4f583c6) against this tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
(timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -q -x -k "lazy_concatenation or png or unet_vs" ) > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2 3; do
  (cd $R/build_tmp/r02_tree && timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r02 tree  ms/job', round(d['ms_per_step'],1), 'flash', round(d['roofline']['achieved']))") | tee -a $O/ab.txt
  (cd $R && timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('this tree ms/job', round(d['ms_per_step'],1), 'flash', round(d['roofline']['achieved']))") | tee -a $O/ab.txt
done
