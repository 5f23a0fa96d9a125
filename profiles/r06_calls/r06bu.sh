python scripts/halo_split_ab.py 24x64x320x320 24x64x640x320 24x64x960x320 32x64x320x320 32x64x640x320 2>&1 | grep -v amdgpu | cut -c1-75 > $O/halo_large_ab.txt; cat $O/halo_large_ab.txt
