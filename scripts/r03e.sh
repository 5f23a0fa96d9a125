# round-3 GPU call e: ping-pong loop v3 (scalar-base LDS-DMA): timeline + interleaved A/B + parity
O=gpurun_out/r03e; mkdir -p $O
(timeout 120 build_tmp/igemm_timeline 254222 254218 1254218 4254218 244222 244218) > $O/timeline.txt 2>&1
(timeout 200 build_tmp/igemm_ab 254222 254218 1254218 2254218 4254218 5254218 244222 244218 1244218) > $O/ab.txt 2>&1
cat $O/timeline.txt | grep -v "5254218\|gemm  4096\|gemm  8192\|conv  8f" ; cat $O/ab.txt
(timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "pingpong or every_tile_shape") > $O/ktests.log 2>&1; tail -3 $O/ktests.log
