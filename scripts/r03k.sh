# round-3 GPU call k: socket power and shader clock sampled while the job runs (is the job power-capped?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O
rocm-smi --showmaxpower --showclocks --showpower > $O/smi_idle.txt 2>&1
(while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.4; done) > $O/smi_samples.txt &
SMI=$!
(timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe) > $O/bench.json 2>/dev/null
kill $SMI
head -c 250 $O/bench.json; echo; grep -E "Max|max" $O/smi_idle.txt | head -5; sort $O/smi_samples.txt | uniq -c | sort -rn | head -12; wc -l $O/smi_samples.txt
