// bf16 GEMM, schedule 8 (see gemm_kernel.h): one translation unit per schedule keeps the build parallel.
#include "gemm_kernel.h"

namespace alg {
int launch_gemm_p8(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s) {
  return launch_gemm<8, 2>(a, m_tiles, n_tiles, nwg, s);
}
}  // namespace alg
