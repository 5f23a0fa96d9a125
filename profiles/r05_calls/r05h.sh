timeout 300 python scripts/gelu_cost_trial.py > $O/gelu.txt 2>&1; cat $O/gelu.txt
