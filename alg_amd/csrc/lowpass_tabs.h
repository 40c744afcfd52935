// Antialias tap tables of the down_up filter as a device blob (built by lowpass_v2.hip's build_tables_kernel into a buffer
// the caller owns, read by the kernels of lowpass_v2.hip and lowpass_v3.hip).  Blob layout per table: xmin[n_out] | xsize[n_out] | w[n_out][taps] (ints / floats, 4 bytes each).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace alg {
namespace v2 {

__host__ __device__ inline int aa_taps(int in_size, int out_size) {
  float scale = (float)in_size / (float)out_size;
  float support = scale >= 1.0f ? scale : 1.0f;
  return (int)ceilf(support) * 2 + 1;
}

// ---- tap tables: blob layout per table = xmin[n_out] | xsize[n_out] | w[n_out][taps] (ints / floats, 4 bytes each) --------
struct Tab {
  int off;   // word offset of the table inside the blob
  int n_out, taps;
  __host__ __device__ int words() const { return n_out * (2 + taps); }
};

struct Tabs {
  Tab dw, dh, uw, uh;
  int words;
};

inline Tabs layout(int H, int W, int h1, int w1) {
  Tabs t;
  int o = 0;
  auto mk = [&](int n_out, int in_size) {
    Tab x;
    x.off = o, x.n_out = n_out, x.taps = aa_taps(in_size, n_out);
    o += x.words();
    return x;
  };
  t.dw = mk(w1, W), t.dh = mk(h1, H), t.uw = mk(W, w1), t.uh = mk(H, h1);
  t.words = (o + 3) & ~3;
  return t;
}

}  // namespace v2
}  // namespace alg
