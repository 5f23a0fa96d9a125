# round-3 GPU call ah: second thread layout of the one-launch GroupNorm (trial build, every group that fits) against the three-kernel form
O=$GRAFT_REPO_ROOT/gpurun_out/r03ah; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FZ_VARIANT_LIB=$GRAFT_REPO_ROOT/build_tmp/libfz_gn2.so timeout 100 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o gn -- python $GRAFT_REPO_ROOT/scripts/gn_ab.py run > $O/gn_run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/gn_kernel_trace.csv $O/prof/gn_kernel_trace.csv 2>/dev/null | head -1)
python scripts/gn_ab.py report "$f" > $O/gn_ab_layout2.txt 2>&1; cat $O/gn_ab_layout2.txt
rm -rf $O/prof
