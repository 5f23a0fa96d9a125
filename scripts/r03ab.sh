# round-3 GPU call ab: one-launch GroupNorm: parity + per-shape GPU time from the kernel trace against the three-kernel form
O=$GRAFT_REPO_ROOT/gpurun_out/r03ab; mkdir -p $O
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm or sharded") > $O/tests.log 2>&1; tail -2 $O/tests.log
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o gn -- python $GRAFT_REPO_ROOT/scripts/gn_ab.py run > $O/gn_run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/gn_kernel_trace.csv $O/prof/gn_kernel_trace.csv 2>/dev/null | head -1)
python scripts/gn_ab.py report "$f" > $O/gn_ab.txt 2>&1; cat $O/gn_ab.txt
rm -rf $O/prof
