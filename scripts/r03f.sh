# round-3 GPU call f: the library's own (tile, split-K) choice with ring tiles only (-1) vs with the ping-pong substitution (-2) on every
# conv / GEMM / temporal-conv shape of the job; new ping-pong tiles 320x128 / 160x256 against their ring twins
O=gpurun_out/r03f; mkdir -p $O
(timeout 300 build_tmp/igemm_ab prod) > $O/prod.txt 2>&1
(timeout 200 build_tmp/igemm_ab 254122 254118 1254118 254222 254218 158122 158118) > $O/ab.txt 2>&1
cat $O/prod.txt $O/ab.txt
