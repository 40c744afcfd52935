export PYTHONPATH=.
run() { timeout 300 python scripts/kbench.py --only gemm_qk,gemm_ff1,gemm_ff2 2>&1 | grep -v -E "amdgpu.ids|^\{"; }
for p in 6 7; do for a in 0 1; do echo "== PIPE $p ABLATE $a"; ALG_GEMM_PIPE=$p ALG_GEMM_ABLATE=$a run; done; done
