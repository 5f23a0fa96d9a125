# round-3 GPU call c: instruction-segment timeline of the igemm K loops + interleaved within-process A/B of the loop variants;
# the kernel tests the previous call did not reach
O=gpurun_out/r03c; mkdir -p $O
(timeout 120 build_tmp/igemm_timeline 254222 254218 1254218 3254218 5254218 244222 1244218) > $O/timeline.txt 2>&1
(timeout 200 build_tmp/igemm_ab 254222 254218 1254218 2254218 3254218 4254218 5254218 244222 244218 1244218 5244218) > $O/ab.txt 2>&1
(timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "pingpong or temporal or every_tile_shape") > $O/ktests.log 2>&1; tail -3 $O/ktests.log
cat $O/timeline.txt $O/ab.txt
