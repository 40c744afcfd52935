export PYTHONPATH=.
run() { ALG_ATTN128_Q64=1 timeout 300 python scripts/kbench.py --only attn128 --iters 5 2>&1 | grep -v -E "amdgpu.ids|^\{"; }
echo "== q64 as built"; run
for abl in "-DALG_Q64_DUMMY_VALU" "-DALG_Q64_NO_SOFTMAX"; do
  touch alg_amd/csrc/attention128_q64.hip; make -C alg_amd/csrc EXTRA="$abl" -j8 > /dev/null 2>&1
  echo "== q64 $abl"; run
done
