timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "pixel_halo" 2>&1 | tail -2
python scripts/halo_split_ab.py > $O/halo_split_ab.txt 2>&1; cat $O/halo_split_ab.txt
