# Round 4: in-situ PMC passes on the final build (the launch log now covers fz_gemm_gn), the peer-transport exchange latency again.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s; mkdir -p $O
bash scripts/pmc_job.sh r04s_pmc_job 50 2>&1 | tail -2
(timeout 300 python -m pytest tests/test_dist_gpu.py -q -s -k "latency") > $O/lat.log 2>&1; grep "peer transport, us per" $O/lat.log; tail -1 $O/lat.log
