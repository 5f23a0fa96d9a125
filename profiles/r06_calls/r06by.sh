timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -2
python scripts/halo_split_ab.py --narrow 8x8x1280x1280 16x8x1280x1280 8x8x2560x1280 16x8x2560x1280 8x16x1280x1280 16x16x1280x1280 8x16x2560x1280 8x32x320x640 8x32x640x640 2>&1 | grep -v amdgpu > $O/halo_narrow_ab.txt; cat $O/halo_narrow_ab.txt
