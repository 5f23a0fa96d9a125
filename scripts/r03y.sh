# round-3 GPU call y: flash kernel reads a frame's coinciding kv slots once (frames 0 and 1 under [-1, 'first']): A/B harness, parity, short bench
O=gpurun_out/r03y; mkdir -p $O
(timeout 120 build_tmp/flash_ab) > $O/flash_ab.txt 2>&1; cat $O/flash_ab.txt
(timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash or self_") > $O/tests.log 2>&1; tail -2 $O/tests.log
(timeout 300 python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-n-edit2-probe) > $O/bench.json 2>$O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(round(d['ms_per_step']), 'ms', round(d['value'],3), 'frames/s', d['roofline'])"
