export PYTHONPATH=.
run() { ALG_ATTN128_Q64=1 timeout 300 python scripts/kbench.py --only attn128 --iters 5 2>&1 | grep -v -E "amdgpu.ids|^\{"; }
echo "== q64 as built"; run
touch alg_amd/csrc/attention128_q64.hip; make -C alg_amd/csrc EXTRA="-DALG_Q64_NO_FMA" -j8 > /dev/null 2>&1
echo "== q64 NO_FMA"; run
