cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O; rm -f $O/probe.txt
export TMPDIR=/tmp
for d in 16 20 32 0; do
  FZ_RG_DBG=$d timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$d -o p -- python scripts/rowgemm_probe.py 8 > /tmp/out$d.txt 2>&1
  echo "dbg=$d 8 frames" >> $O/probe.txt
  python scripts/kstats.py /tmp/prof$d/p_kernel_stats.csv 1 3 >> $O/probe.txt
done
FZ_RG_DBG=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof16 -o p -- python scripts/rowgemm_probe.py 16 > /dev/null 2>&1
echo "16 frames" >> $O/probe.txt
python scripts/kstats.py /tmp/prof16/p_kernel_stats.csv 1 3 >> $O/probe.txt
grep "avg" $O/probe.txt | grep -v torch
