# In-situ HBM traffic of ONE bench job per kernel class: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
# pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") over `python bench.py` itself, nothing but --kernel-trace beside --pmc.
#   bash scripts/pmc_job.sh [tag] [ddim steps] [extra bench.py args]  -> gpurun_out/<tag>.json  (copy to profiles/r05_pmc_job.json)
# Round 5 passes `--blend-th <the split threshold bench.py bisects>`: the job whose blend mask splits the rows, so that the masked-inject
# launches the pass attributes really read stored rows (with the config's 0.3 and procedural weights they read none).
TAG=${1:-pmc_job}; T=${2:-50}; EXTRA="${@:3}"; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG.d; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  FZ_BENCH_LAUNCHLOG=$OUT/launchlog_$c.json timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- \
    python $R/bench.py --steps 1 --warmup 0 --ddim-steps $T --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-box $EXTRA > $OUT/$c.log 2>&1
  tail -2 $OUT/$c.log
done
python $R/scripts/pmc_job_summary.py $OUT $R/gpurun_out/$TAG.json $T
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
