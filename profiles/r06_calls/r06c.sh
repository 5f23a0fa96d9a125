python scripts/ff_chain_ab.py 4 8 16 > $O/ff_chain_ab_trials.txt 2>&1; tail -30 $O/ff_chain_ab_trials.txt
