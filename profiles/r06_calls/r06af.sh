timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "every_tile" 2>&1 | tail -2; python scripts/kg_tile_ab.py 0,254122,252222,252218 > $O/kg_tile_ab.txt 2>&1; cat $O/kg_tile_ab.txt
