# round-3 GPU call z: two-workgroups-per-CU tiles (K step 32, 2-deep ring) on the epilogue-dominated short-K projections, A/B vs the library's choice
O=gpurun_out/r03z; mkdir -p $O
(timeout 200 build_tmp/igemm_ab shortk -2 244112 224212 224112 254112) > $O/igemm_shortk.txt 2>&1; cat $O/igemm_shortk.txt
