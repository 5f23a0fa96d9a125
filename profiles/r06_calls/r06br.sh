timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or conv or tconv or temporal or split" 2>&1 | tail -2
python scripts/ab_two_libs_gemm.py build_tmp/libfz_before_reduce_batch.so fatezero_amd/libfatezero_hip.so 2>&1 | grep -v amdgpu | grep "lib\| 1024 \|  512 " > $O/reduce_batch_ab.txt; cat $O/reduce_batch_ab.txt
python scripts/ab_two_libs_conv.py build_tmp/libfz_before_reduce_batch.so fatezero_amd/libfatezero_hip.so 2>&1 | grep -v amdgpu > $O/reduce_batch_conv_ab.txt; cat $O/reduce_batch_conv_ab.txt
R=$GRAFT_REPO_ROOT; A=$R/$O
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $A/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-split-mask --no-box > $A/bench_prof.json 2> $A/bench_prof.err; cd $R
f=$(ls $A/prof/*/bench_kernel_stats.csv $A/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $A/kernel_stats.csv; rm -rf $A/prof; python scripts/kstats.py $A/kernel_stats.csv 3 70 > $A/kstats.txt; grep "total\|reduce" $A/kstats.txt
