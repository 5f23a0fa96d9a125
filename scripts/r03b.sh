# round-3 GPU call b: ping-pong igemm loop -- parity (bit-equality with the ring loop) + same-box A/B against the ring tiles and the trial variants
O=gpurun_out/r03b; mkdir -p $O
(timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "every_tile_shape or pingpong") > $O/ktests.log 2>&1; tail -3 $O/ktests.log
(timeout 300 python scripts/tile_trial.py 0,254222,254218,244222,244218) > $O/trial_main.txt 2>&1
for v in nostag noprio k32; do (FZ_TRIAL_LIB=build_tmp/libfz_$v.so timeout 200 python scripts/tile_trial.py 254218,244218) > $O/trial_$v.txt 2>&1; done
(timeout 300 python scripts/tile_trial.py 0,254222,254218,244222,244218) > $O/trial_main2.txt 2>&1
tail -n 40 $O/trial_main.txt
