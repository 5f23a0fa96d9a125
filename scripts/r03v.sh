# round-3 GPU call v: cfg5-shaped job (32 f x 576^2, 10 steps), round-2 tree vs this tree, process-interleaved on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O
X="--frames 32 --latent-size 72 --ddim-steps 10 --warmup 1 --steps 2 --no-cpu-baseline --no-kernel-breakdown"
for i in 1 2; do
  (cd $R/build_tmp/r02_tree && timeout 200 python bench.py $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r02 tree  ms/job', round(d['ms_per_step'],1))") | tee -a $O/ab.txt
  (cd $R && timeout 200 python bench.py $X --no-n-edit2-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('this tree ms/job', round(d['ms_per_step'],1))") | tee -a $O/ab.txt
done
