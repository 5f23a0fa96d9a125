timeout 300 ./build_tmp/ubench_launch_gap > $O/ubench_launch_gap.txt 2>&1; cat $O/ubench_launch_gap.txt
