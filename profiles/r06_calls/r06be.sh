timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -3
python scripts/halo_split_ab.py 8x8x1280x1280 16x8x1280x1280 8x8x2560x1280 16x8x2560x1280 8x16x1280x1280 16x16x1280x1280 8x32x640x640 > $O/halo_split_ab_8x8.txt 2>&1; cat $O/halo_split_ab_8x8.txt
python scripts/conv_up2_ab.py > $O/conv_up2_ab.txt 2>&1; cat $O/conv_up2_ab.txt
