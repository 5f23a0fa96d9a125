# round-3 GPU call h: whole-kernel / K-loop / set-up ticks against the wall clock of the launch (what is the shader clock under this load,
# and how much of a launch is outside the K loop)
O=gpurun_out/r03h; mkdir -p $O
(timeout 100 build_tmp/igemm_timeline 254222 254218 254122) > $O/timeline.txt 2>&1
grep "wave 0 of" $O/timeline.txt
(rocm-smi --showclocks --showpower 2>/dev/null | head -30) > $O/smi.txt; cat $O/smi.txt | head -20
