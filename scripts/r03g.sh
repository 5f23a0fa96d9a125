# round-3 GPU call g: K-32 phases (one barrier pair per K tile) against the K-16-phase ping-pong loop and the ring loop, per tile shape
O=gpurun_out/r03g; mkdir -p $O
(timeout 200 build_tmp/igemm_ab 254222 254218 4254218 244222 244218 4244218 254122 254118 4254118 158122 158118 4158118) > $O/ab.txt 2>&1
cat $O/ab.txt
