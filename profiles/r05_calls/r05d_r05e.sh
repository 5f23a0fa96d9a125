cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
export PYTHONUNBUFFERED=1
cat >> tests/test_kernels_gpu.py <<'PY'


def test_tmp_rowgemm():
    KC.case_ln_gemm(DEV, rows=32768, o=320, n_res=1)
    KC.case_ln_gemm(DEV, rows=4096 * 3 + 200, o=640, n_res=2, seed=1)
    KC.case_ln_gemm(DEV, rows=65536, o=320, ln=False, bias=True, n_res=1, seed=2)
    KC.case_ln_gemm(DEV, rows=32768, o=960, bias=False, mean_shift=6.0, seed=3)
    print(KC.case_ln_gemm_qkvt(DEV, n=8, l=4096))
    print(KC.case_ln_gemm_qkvt(DEV, n=16, l=4096, ln=False, seed=1))
PY
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "tmp_rowgemm" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log
timeout 600 python scripts/rowgemm_ab.py > $O/ab.txt 2>&1
tail -5 $O/test.log; cat $O/ab.txt
