// bf16 ping-pong GEMM with convolution addressing of the A operand (see gemm_kernel.h, CONV): plain and residual epilogues.
#include "gemm_kernel.h"

namespace alg {
int launch_gemm_p6_conv(const alg_gemm_args* a, int m_tiles, int n_tiles, int64_t nwg, hipStream_t s) {
  static PerDeviceOnce attr_set;  // idempotent one-time setup per device; racing first calls both succeed
  const int dev_slot = current_device_slot();
  if (!device_done(attr_set, dev_slot)) {
    const void* fns[2] = {(const void*)gemm_bf16_kernel<ALG_ACT_NONE, true, 6, 4, false, true>,
                          (const void*)gemm_bf16_kernel<ALG_ACT_NONE, false, 6, 4, false, true>};
    for (const void* fn : fns) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
      if (e != hipSuccess) {
        set_error("alg_conv_cl_bf16: hipFuncSetAttribute(%d B LDS): %s", GEMM_LDS, hipGetErrorString(e));
        return ALG_ELAUNCH;
      }
    }
    device_mark(attr_set, dev_slot);
  }
  const dim3 grid(gemm_grid(nwg)), block(512);
  const int gm = gemm_group_m(6);
  if (a->R)
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_NONE, true, 6, 4, false, true>), grid, block, GEMM_LDS, s, *a, m_tiles,
                       n_tiles, gm, GemmNoPair{});
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<ALG_ACT_NONE, false, 6, 4, false, true>), grid, block, GEMM_LDS, s, *a, m_tiles,
                       n_tiles, gm, GemmNoPair{});
  return check_launch("alg_conv_cl_bf16");
}
}  // namespace alg
