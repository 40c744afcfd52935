// The one kernel the Llama-3 tower of HunyuanVideo's prompt encoder (LlavaForConditionalGeneration; reference
// pipeline_hunyuan_video_image2video_lowpass.py:282-420) needs beyond the GEMM / attention / RMSNorm / SiLU / multiply
// kernels the other encoders already use: the rotary embedding in its "rotate_half" form,
//     x' = x * cos + rotate_half(x) * sin,   rotate_half(x) = [-x[d/2:], x[:d/2]],
// over [rows][heads][128] bf16 in place, with the rounding points of the eager bf16 graph (each product and the sum are
// bf16 tensors; cos / sin arrive as the bf16 tables transformers builds, widened to fp32).
#include "common.h"

namespace alg {
namespace llama {

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(u[k] << 16);
    f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
  }
}

// 8 lanes own one head vector: lane j holds elements [8j, 8j + 8) of the first half AND of the second half
__global__ __launch_bounds__(256) void rope_half_kernel(bf16_t* __restrict__ x, const float* __restrict__ cos_tab,
                                                        const float* __restrict__ sin_tab, const int* __restrict__ pos,
                                                        int64_t rows, int heads, int64_t x_rs) {
  const int64_t vec = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  if (vec >= rows * heads) return;
  const int64_t row = vec / heads;
  const int head = (int)(vec - row * heads);
  bf16_t* p = x + row * x_rs + head * 128 + sub * 8;
  const int ps = pos[row];
  const float* c = cos_tab + (int64_t)ps * 128 + sub * 8;
  const float* s = sin_tab + (int64_t)ps * 128 + sub * 8;
  float a[8], b[8], o1[8], o2[8];
  unpack8(*(const uint4*)p, a);          // x[d],      d in the first half
  unpack8(*(const uint4*)(p + 64), b);   // x[d + 64]
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    o1[k] = rbf(rbf(a[k] * c[k]) + rbf(-b[k] * s[k]));
    o2[k] = rbf(rbf(b[k] * c[64 + k]) + rbf(a[k] * s[64 + k]));
  }
  uint4 r;
  r.x = pack_bf2(o1[0], o1[1]); r.y = pack_bf2(o1[2], o1[3]); r.z = pack_bf2(o1[4], o1[5]); r.w = pack_bf2(o1[6], o1[7]);
  *(uint4*)p = r;
  r.x = pack_bf2(o2[0], o2[1]); r.y = pack_bf2(o2[2], o2[3]); r.z = pack_bf2(o2[4], o2[5]); r.w = pack_bf2(o2[6], o2[7]);
  *(uint4*)(p + 64) = r;
}

}  // namespace llama
}  // namespace alg

using namespace alg;

extern "C" int alg_rope_half(void* x, const float* cos_tab, const float* sin_tab, const int* pos, int64_t rows, int heads,
                             int64_t x_rstride, void* stream) {
  if (rows < 0 || heads <= 0 || x_rstride < (int64_t)heads * 128 || (x_rstride & 7)) {
    set_error("alg_rope_half: bad shape rows=%lld heads=%d x_rstride=%lld", (long long)rows, heads, (long long)x_rstride);
    return ALG_EINVAL;
  }
  if (rows == 0) return ALG_OK;
  if (!x || !cos_tab || !sin_tab || !pos || ((uintptr_t)x & 15)) {
    set_error("alg_rope_half: null or misaligned pointer");
    return ALG_EINVAL;
  }
  const int64_t threads = rows * heads * 8;
  hipLaunchKernelGGL(llama::rope_half_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)x, cos_tab, sin_tab, pos, rows, heads, x_rstride);
  return check_launch("alg_rope_half");
}
