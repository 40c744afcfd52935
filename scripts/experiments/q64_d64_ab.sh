export PYTHONPATH=.
python scripts/experiments/dbg64.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_dit_kernels.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or attn" 2>&1 | tail -5
for f in 0 1; do echo "== ALG_ATTN64_Q64=$f"; ALG_ATTN64_Q64=$f timeout 300 python scripts/kbench.py --only attn_prescaled --iters 5 2>&1 | grep -v -E "amdgpu.ids|^\{"; done
