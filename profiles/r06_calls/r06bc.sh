timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -3
python scripts/conv_up2_ab.py > $O/conv_up2_ab.txt 2>&1; cat $O/conv_up2_ab.txt
