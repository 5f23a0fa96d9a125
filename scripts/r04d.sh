# Round 4, fourth GPU call: the one-sided peer transport over HIP IPC (two ranks on the box's one GPU), the qkvt cases again.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
(timeout 500 python -m pytest tests/test_dist_gpu.py -x -q -s) > $O/dist_gpu.log 2>&1; tail -25 $O/dist_gpu.log
(timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "qkvt or split") > $O/k.log 2>&1; tail -3 $O/k.log
