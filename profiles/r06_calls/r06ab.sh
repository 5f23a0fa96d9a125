python scripts/kg_tile_ab.py > $O/kg_tile_ab.txt 2>&1; cat $O/kg_tile_ab.txt; timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "every_tile or tile_shapes" 2>&1 | tail -2
