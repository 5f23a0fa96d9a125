python scripts/ff_chain_ab.py 4 8 16 > $O/ff_chain_timing.txt 2>&1; tail -16 $O/ff_chain_timing.txt
