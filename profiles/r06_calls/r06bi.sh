R=$GRAFT_REPO_ROOT; A=$R/$O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $A/$c -o p -- python $R/scripts/trials/pmc_conv_shapes.py > $A/$c.log 2>&1; done
cd $R; python - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/"+os.environ["O"]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[]
    for f in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c: rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:70], float(r["Counter_Value"]), r.get("Grid_Size")))
    rows.sort()
    print(c)
    for r in rows:
        if "conv_halo" in r[1] or "igemm" in r[1]: print("  ", r[0], r[1], r[3], f"{r[2]/1024:.1f} MiB (raw KiB counter)")
PY
