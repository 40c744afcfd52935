export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_wan_kernels.py -m gpu -q --no-header -p no:cacheprovider -x -k "d128" 2>&1 | tail -15
for f in 0 1; do echo "== ALG_ATTN128_Q64=$f"; ALG_ATTN128_Q64=$f timeout 300 python scripts/kbench.py --only attn128 --iters 5 2>&1 | grep -v -E "amdgpu.ids|^\{"; done
