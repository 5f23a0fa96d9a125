# round-3 GPU call o: flash with the job's real operands vs random operands, back to back in one process
O=gpurun_out/r03o; mkdir -p $O
(timeout 300 python scripts/flash_insitu_probe.py) > $O/probe.txt 2>&1; tail -4 $O/probe.txt
