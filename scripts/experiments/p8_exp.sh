export PYTHONPATH=.
run() { ALG_GEMM_PIPE=$1 timeout 300 python scripts/kbench.py --only gemm_qk,gemm_out,gemm_ff1,gemm_ff2 2>&1 | grep -v -E "amdgpu.ids|^\{"; }
echo "== PIPE 6"; run 6
echo "== PIPE 8 (prefetch 3 steps)"; run 8
for abl in ALG_P8_NO_DMA ALG_P8_NO_READS; do
  touch alg_amd/csrc/gemm_p8.hip; make -C alg_amd/csrc EXTRA=-D$abl -j8 > /dev/null 2>&1
  echo "== PIPE 8 $abl"; run 8
done
touch alg_amd/csrc/gemm_p8.hip; make -C alg_amd/csrc EXTRA="-DALG_P8_NO_DMA -DALG_P8_NO_READS" -j8 > /dev/null 2>&1
echo "== PIPE 8 no DMA no reads"; run 8
