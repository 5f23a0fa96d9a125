FZ_CH_LIB=build_tmp/libfz_ch_timing.so python scripts/trials/ch_timing.py 2>&1 | grep -v amdgpu.ids > $O/ch_timing.txt; cat $O/ch_timing.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "pixel_halo" 2>&1 | tail -2; python scripts/kg_tile_ab.py 0,154299 > $O/halo_ab.txt 2>&1; cat $O/halo_ab.txt
