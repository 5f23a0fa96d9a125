# Round 4: GroupNorm statistics from the producing epilogue -- parity, then the job-level A/B and the per-kernel in-situ attribution.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "producing_epilogue or groupnorm or temporal or gemm") > $O/k.log 2>&1; grep "gn from epilogue" $O/k.log | cut -c1-300; tail -2 $O/k.log
(timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -k "fullwidth_sd15 or test_pipeline or unet_vs") > $O/p.log 2>&1; tail -2 $O/p.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
for i in 1 2 3; do
  (FZ_NO_GN_EPILOGUE=1 timeout 200 $B | python -c "import sys,json; print('A gn_stats kernel     ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
  (timeout 200 $B | python -c "import sys,json; print('B stats from epilogue ', json.loads(sys.stdin.read())['ms_per_step'])") 2>> $O/job.err
done
cd /tmp; export TMPDIR=/tmp
for v in off on off2 on2; do
  if [ "${v:0:3}" = "off" ]; then export FZ_NO_GN_EPILOGUE=1; else unset FZ_NO_GN_EPILOGUE; fi
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_$v.json 2> $O/bench_$v.err
  f=$(ls $O/prof_$v/*/bench_kernel_stats.csv $O/prof_$v/bench_kernel_stats.csv 2>/dev/null | head -1)
  cp "$f" $O/kernel_stats_$v.csv 2>/dev/null; rm -rf $O/prof_$v
done
cd $R
python - <<PY
import csv, re, collections
def load(v):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open("$O/kernel_stats_%s.csv" % v)):
        n = r["Name"]; m = re.search(r"igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (\d+), (true|false)(?:, (true|false))?", n)
        if m:
            k = "igemm mode %s%s" % (m.group(7), " GS" if m.group(12) == "true" else "")
        elif "gn_stats" in n: k = "gn_stats"
        elif "gn_" in n: k = "gn other"
        else: k = "other"
        agg[k][0] += int(r["Calls"]) // 3; agg[k][1] += float(r["TotalDurationNs"]) / 3e6
    return agg
for v in ("off", "off2", "on", "on2"):
    a = load(v)
    print("%-4s total %.1f ms/job | " % (v, sum(t for c, t in a.values())) + " | ".join("%s %.1f ms (%d)" % (k, a[k][1], a[k][0]) for k in sorted(a)))
PY
