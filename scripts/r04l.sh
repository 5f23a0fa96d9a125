# Round 4: the flat K-slice grid for ALL split-K launches, judged IN SITU (per-kernel totals of two profiles per setting) -- the kernel-level
# A/B repeats one launch with its weights resident in Infinity Cache; inside the job the weights come from HBM.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in off on off2 on2; do
  if [ "${v:0:2}" = "on" ]; then export FZ_IGEMM_XCD_KS_ALL=1; else unset FZ_IGEMM_XCD_KS_ALL; fi
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench_$v.json 2> $O/bench_$v.err
  f=$(ls $O/prof_$v/*/bench_kernel_stats.csv $O/prof_$v/bench_kernel_stats.csv 2>/dev/null | head -1)
  cp "$f" $O/kernel_stats_$v.csv 2>/dev/null; rm -rf $O/prof_$v
done
cd $R
python - <<PY
import csv
def load(v):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open("$O/kernel_stats_%s.csv" % v))}
a1,a2,b1,b2 = load("off"),load("off2"),load("on"),load("on2")
tot=lambda d: sum(t for c,t in d.values())/3e6
print("kernel sum per job ms: convs only %.1f %.1f   all split-K launches %.1f %.1f" % (tot(a1),tot(a2),tot(b1),tot(b2)))
rows=[]
for k in a1:
    if k in b1 and k in a2 and k in b2:
        a=(a1[k][1]+a2[k][1])/2; b=(b1[k][1]+b2[k][1])/2
        rows.append(((b-a)/3e6, k[:90], a1[k][0], a/3e6, b/3e6, abs(a1[k][1]-a2[k][1])/3e6))
rows.sort()
for d,k,c,a,b,noise in rows[:8]+rows[-8:]:
    print(f"{d:+7.2f} ms/job  {k:90s} calls {c:6d}  {a:8.2f} -> {b:8.2f}  (run-to-run {noise:.2f})")
PY
