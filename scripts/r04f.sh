# Round 4, sixth GPU call: 10-wave 320 x 128 tile A/B, the 16-frame full-width forward vs the oracle.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
(timeout 200 python scripts/tile10_ab.py) > $O/tile10_ab.txt 2>&1; cat $O/tile10_ab.txt
(timeout 400 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "long_clips") > $O/long.log 2>&1; grep "fullwidth forward\|passed\|failed" $O/long.log | cut -c1-400
