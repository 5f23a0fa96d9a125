# Round 4: the chooser's split-K launch cost (FZ_SPLITK_LAUNCH_US = 3 us, fitted on kernel-level sweeps) judged IN SITU: 3 / 8 / 16 us.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in 3 8 16; do
  export FZ_IGEMM_SPLITK_US=$v
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v_$rep -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  f=$(ls $O/prof_$v_$rep/*/bench_kernel_stats.csv $O/prof_$v_$rep/bench_kernel_stats.csv 2>/dev/null | head -1)
  cp "$f" $O/kernel_stats_${v}_$rep.csv 2>/dev/null; rm -rf $O/prof_$v_$rep
done; done
cd $R
python - <<PY
import csv, re, collections
def load(v, r):
    agg = collections.defaultdict(float); calls = collections.defaultdict(int)
    for row in csv.DictReader(open("$O/kernel_stats_%s_%d.csv" % (v, r))):
        n = row["Name"]; m = re.search(r"igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", n)
        k = ("igemm mode %s" % m.group(7)) if m else ("reduce" if "reduce" in n else "other")
        agg[k] += float(row["TotalDurationNs"]) / 3e6; calls[k] += int(row["Calls"]) // 3
    return agg, calls
for v in (3, 8, 16):
    for r in (1, 2):
        a, c = load(v, r)
        print("splitk_us %2d run %d: total %.1f ms/job | " % (v, r, sum(a.values())) + " | ".join("%s %.1f ms (%d)" % (k, a[k], c[k]) for k in sorted(a)))
PY
