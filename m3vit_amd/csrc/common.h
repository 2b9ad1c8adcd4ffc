// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of libm3vit_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/m3vit_hip.h"

namespace m3 {

typedef _Float16 half_t;
typedef __bf16 bf16_t;                   // the third activation dtype (M3_BF16)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// ----------------------------------------------------------------- error plumbing
void set_error(const char *fmt, ...);
int check_launch(const char *what);

#define M3_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      m3::set_error(__VA_ARGS__);             \
      return M3_ERR_ARG;                      \
    }                                         \
  } while (0)

// ------------------------------------------------------------------ MFMA traits
// One "chunk" is what a single 16-byte operand fragment per lane contracts over:
//   f16: 8 halves/lane  -> one v_mfma_f32_16x16x32_f16, KC = 32
//   f32: 4 floats/lane  -> four v_mfma_f32_16x16x4_f32 (k permuted identically on both
//                          operands, which a contraction allows), KC = 16
// Lane l = 16*g + i supplies rows/cols i = l & 15 and contraction slots of group g.
// Element j of the fragment is contraction index  16*(j/4) + 4*g + (j%4)  when the
// fragment is assembled from C-layout tiles (f16: two tiles), or simply
// KC/4*g + j when read from a K-contiguous row.  Both operands of one MFMA must use
// the same convention.
// C/D layout of every 16x16 tile: D[row = 4*g + r][col = i], r = 0..3.
template <typename T> struct Mma;

template <> struct Mma<float> {
  typedef f32x4 frag;                  // 16 bytes
  static constexpr int KC = 16;        // contraction elements per fragment
  static constexpr int EPL = 4;        // elements per lane per fragment
  static constexpr int CT = 1;         // C tiles (16 wide) that make one fragment
  static __device__ __forceinline__ f32x4 mma(const frag &a, const frag &b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ frag zero() { return frag{0.f, 0.f, 0.f, 0.f}; }
  // fragment from C-layout tile(s): f32 needs one tile
  static __device__ __forceinline__ frag from_tiles(const f32x4 *t) { return t[0]; }
  static __device__ __forceinline__ float to_float(float v) { return v; }
  static __device__ __forceinline__ float from_float(float v) { return v; }
};

template <> struct Mma<half_t> {
  typedef f16x8 frag;
  static constexpr int KC = 32;
  static constexpr int EPL = 8;
  static constexpr int CT = 2;
  static __device__ __forceinline__ f32x4 mma(const frag &a, const frag &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag zero() {
    return frag{(half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0};
  }
  static __device__ __forceinline__ frag from_tiles(const f32x4 *t) {
    frag f;
    f[0] = (half_t)t[0][0]; f[1] = (half_t)t[0][1]; f[2] = (half_t)t[0][2]; f[3] = (half_t)t[0][3];
    f[4] = (half_t)t[1][0]; f[5] = (half_t)t[1][1]; f[6] = (half_t)t[1][2]; f[7] = (half_t)t[1][3];
    return f;
  }
  static __device__ __forceinline__ float to_float(half_t v) { return (float)v; }
  static __device__ __forceinline__ half_t from_float(float v) { return (half_t)v; }
};

template <> struct Mma<bf16_t> {             // same fragment geometry as f16: one v_mfma_f32_16x16x32_bf16 per fragment pair
  typedef bf16x8 frag;
  static constexpr int KC = 32;
  static constexpr int EPL = 8;
  static constexpr int CT = 2;
  static __device__ __forceinline__ f32x4 mma(const frag &a, const frag &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag zero() {
    return frag{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  }
  static __device__ __forceinline__ frag from_tiles(const f32x4 *t) {
    frag f;
    f[0] = (bf16_t)t[0][0]; f[1] = (bf16_t)t[0][1]; f[2] = (bf16_t)t[0][2]; f[3] = (bf16_t)t[0][3];
    f[4] = (bf16_t)t[1][0]; f[5] = (bf16_t)t[1][1]; f[6] = (bf16_t)t[1][2]; f[7] = (bf16_t)t[1][3];
    return f;
  }
  static __device__ __forceinline__ float to_float(bf16_t v) { return (float)v; }
  static __device__ __forceinline__ bf16_t from_float(float v) { return (bf16_t)v; }
};

// 4 consecutive elements of T <-> f32x4
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  typedef f32x4 type;
  static __device__ __forceinline__ f32x4 load(const float *p) { return *(const f32x4 *)p; }
  static __device__ __forceinline__ void store(float *p, f32x4 v) { *(f32x4 *)p = v; }
};
template <> struct Vec4<half_t> {
  typedef f16x4 type;
  static __device__ __forceinline__ f32x4 load(const half_t *p) {
    f16x4 h = *(const f16x4 *)p;
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  static __device__ __forceinline__ void store(half_t *p, f32x4 v) {
    f16x4 h;
    h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
    *(f16x4 *)p = h;
  }
};

template <> struct Vec4<bf16_t> {
  typedef bf16x4 type;
  static __device__ __forceinline__ f32x4 load(const bf16_t *p) {
    bf16x4 h = *(const bf16x4 *)p;
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  static __device__ __forceinline__ void store(bf16_t *p, f32x4 v) {
    bf16x4 h;
    h[0] = (bf16_t)v[0]; h[1] = (bf16_t)v[1]; h[2] = (bf16_t)v[2]; h[3] = (bf16_t)v[3];
    *(bf16x4 *)p = h;
  }
};

// 8 consecutive elements of T <-> two f32x4: one 16-byte access for f16 (store-issue-bound epilogues:
// a wave64 dwordx4 store moves twice the bytes of a dwordx2 at the same issue cost), two for f32.
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float *p, f32x4 &a, f32x4 &b) { a = *(const f32x4 *)p; b = *(const f32x4 *)(p + 4); }
  static __device__ __forceinline__ void store(float *p, f32x4 a, f32x4 b) { *(f32x4 *)p = a; *(f32x4 *)(p + 4) = b; }
};
template <> struct Vec8<half_t> {
  static __device__ __forceinline__ void load(const half_t *p, f32x4 &a, f32x4 &b) {
    const f16x8 h = *(const f16x8 *)p;
    a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    b = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
  }
  static __device__ __forceinline__ void store(half_t *p, f32x4 a, f32x4 b) {
    f16x8 h;
    h[0] = (half_t)a[0]; h[1] = (half_t)a[1]; h[2] = (half_t)a[2]; h[3] = (half_t)a[3];
    h[4] = (half_t)b[0]; h[5] = (half_t)b[1]; h[6] = (half_t)b[2]; h[7] = (half_t)b[3];
    *(f16x8 *)p = h;
  }
};

template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t *p, f32x4 &a, f32x4 &b) {
    const bf16x8 h = *(const bf16x8 *)p;
    a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    b = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
  }
  static __device__ __forceinline__ void store(bf16_t *p, f32x4 a, f32x4 b) {
    bf16x8 h;
    h[0] = (bf16_t)a[0]; h[1] = (bf16_t)a[1]; h[2] = (bf16_t)a[2]; h[3] = (bf16_t)a[3];
    h[4] = (bf16_t)b[0]; h[5] = (bf16_t)b[1]; h[6] = (bf16_t)b[2]; h[7] = (bf16_t)b[3];
    *(bf16x8 *)p = h;
  }
};

// erf-GELU (nn.GELU default) and its derivative.  erfc is evaluated with Abramowitz-Stegun 7.1.26
// (|abs error| <= 1.5e-7, i.e. at the fp32 rounding level of the surrounding MFMA sums) instead
// of the ~40-instruction libm erff: in the GEMM epilogues the libm version cost as much as the
// whole K loop.  q = 0.5*erfc(|x|/sqrt2) is formed without cancellation; e = exp(-x^2/2) is
// shared with the derivative's pdf term.
__device__ __forceinline__ void gelu_parts(float x, float &cdf, float &e) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);
  e = __expf(-u * u);
  float poly = 1.061405429f;
  poly = __builtin_fmaf(poly, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float q = 0.5f * poly * t * e;
  cdf = (x < 0.f) ? q : 1.0f - q;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return __builtin_fmaf(x * 0.39894228040143267794f, e, cdf);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware block remap (8 XCDs, blocks dealt round-robin): logical tile ids
// that are adjacent (share operand panels) land on the same XCD's L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// row index / divisor with the divisor a kernel argument: a shift when it is a power of two (top-k = 2, 4, 8 ...: the
// routed-entry -> token map of every gathered operand), else the (~30 instruction) integer division
static inline int div_shift(int d) { return (d >= 1 && (d & (d - 1)) == 0) ? __builtin_ctz((unsigned)d) : -1; }
__device__ __forceinline__ int32_t div_by(int32_t v, int32_t d, int32_t sh) { return sh >= 0 ? (v >> sh) : (v / d); }

int launch_reduce_rows_f32(const float *part, int nrows, int N, int G, int64_t gstride, float *out, int beta,
                           hipStream_t s);
int launch_reduce_rows2_f32(const float *part, int nrows, int N, float *out0, float *out1, int beta, hipStream_t s);
int launch_reduce_rows2_batch_f32(const float *part, int64_t layer_stride, int nrows, int N, const m3_ln_param_grads *outs,
                                  int first, int count, int beta, hipStream_t s);
int launch_reduce_rows_i32(const int32_t *part, int nrows, int N, int G, int64_t gstride, int64_t *out, int beta,
                           hipStream_t s);

static inline int dtype_size(int dt) { return (dt == M3_F16 || dt == M3_BF16) ? 2 : 4; }
static inline bool dtype_ok(int dt) { return dt == M3_F32 || dt == M3_F16 || dt == M3_BF16; }

}  // namespace m3
