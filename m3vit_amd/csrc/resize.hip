// ReLU + bilinear x2 up-sampling (align_corners = False) in ONE pass, and its backward, for channels-last activations:
// the element-wise half of a decoder-head stage of the reference, models/heads/vit_up_head.py:181-214
//     x = F.relu(syncbn_fc_i(conv_i(x)), inplace=True) ; x = F.interpolate(x, size=x.shape[-1]*2, mode='bilinear')
// ("MIOpen first, custom later").  Under fp16 autocast torch runs the resize in fp32 (its output
// and the gradient that comes back are fp32 tensors 4x the stage's input, with a cast kernel each way) and its backward as a
// generic scatter: at 8 x 480 x 640 the resizes and the casts around them are ~6.5 of the head's 13.7 ms forward + backward.
// Here the 16-bit activations go in and come out (fp32 output optional, for the classifier stage), HBM-bound by design:
//
//   forward : thread = 8 (16-bit) / 4 (fp32) channels of one INPUT pixel: 3 x 3 neighbourhood (replicate-clamped: the
//             align_corners=False source coordinate (o + 0.5) / 2 - 0.5 clamped at 0 is exactly "weights 0.25 / 0.75 on the
//             clamped neighbours"), ReLU on the way in, the 2 x 2 output pixels it owns written as 16-byte vectors;
//   backward: thread = the same channels of one input pixel: the 4 x 4 window of output gradients it fed (rows 2i-1 .. 2i+2
//             with weights .25 .75 .75 .25, indices clamped - the clamped neighbours' share comes back to the border pixel),
//             times the ReLU mask recomputed from x.  A gather: deterministic, no atomics.
// Channels-last ([N, H, W, C], C a multiple of the vector width): consecutive lanes walk C, so every access of a wave is a
// run of whole cache lines.
#include "common.h"

namespace m3 {

template <typename T> struct RzVec;                 // one 16-byte vector of T <-> NV fp32 lanes
template <> struct RzVec<float> {
  static constexpr int NV = 4;
  static __device__ __forceinline__ void load(const float *p, float (&v)[4]) {
    const f32x4 t = *(const f32x4 *)p;
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void store(float *p, const float (&v)[4]) { *(f32x4 *)p = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct RzVec<half_t> {
  static constexpr int NV = 8;
  static __device__ __forceinline__ void load(const half_t *p, float (&v)[8]) {
    const f16x8 h = *(const f16x8 *)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  }
  static __device__ __forceinline__ void store(half_t *p, const float (&v)[8]) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
    *(f16x8 *)p = h;
  }
};
template <> struct RzVec<bf16_t> {
  static constexpr int NV = 8;
  static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[8]) {
    const bf16x8 h = *(const bf16x8 *)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  }
  static __device__ __forceinline__ void store(bf16_t *p, const float (&v)[8]) {
    bf16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (bf16_t)v[j];
    *(bf16x8 *)p = h;
  }
};

// store NV fp32 lanes as TO (NV is the INPUT's vector width: an fp32 output of a 16-bit input takes two 16-byte stores)
template <typename TO, int NV> __device__ __forceinline__ void rz_store(TO *p, const float (&v)[NV]) {
  if constexpr (sizeof(TO) == 4 && NV == 8) {
    *(f32x4 *)p = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4 *)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else if constexpr (sizeof(TO) == 2 && NV == 4) {
    typename Vec4<TO>::type h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = (TO)v[j];
    *(typename Vec4<TO>::type *)p = h;
  } else {
    RzVec<TO>::store(p, v);
  }
}
template <typename TG, int NV> __device__ __forceinline__ void rz_load(const TG *p, float (&v)[NV]) {
  if constexpr (sizeof(TG) == 4 && NV == 8) {
    const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else if constexpr (sizeof(TG) == 2 && NV == 4) {
    const typename Vec4<TG>::type h = *(const typename Vec4<TG>::type *)p;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)h[j];
  } else {
    RzVec<TG>::load(p, v);
  }
}

template <typename T, typename TO, bool RELU>
__global__ __launch_bounds__(256) void relu_up2x_fwd_kernel(const T *__restrict__ x, int64_t total, int H, int W, int C,
                                                            TO *__restrict__ y) {
  constexpr int NV = RzVec<T>::NV;
  const int CV = C / NV;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int cv = (int)(t % CV);
  int64_t r = t / CV;
  const int ix = (int)(r % W); r /= W;
  const int iy = (int)(r % H);
  const int64_t n = r / H;
  const int ym = iy > 0 ? iy - 1 : 0, yp = iy + 1 < H ? iy + 1 : H - 1;
  const int xm = ix > 0 ? ix - 1 : 0, xp = ix + 1 < W ? ix + 1 : W - 1;
  const T *base = x + (n * H * (int64_t)W) * C + cv * NV;
  const int rows[3] = {ym, iy, yp}, cols[3] = {xm, ix, xp};
  float z[3][3][NV];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      RzVec<T>::load(base + ((int64_t)rows[a] * W + cols[b]) * C, z[a][b]);
      if (RELU) {
#pragma unroll
        for (int j = 0; j < NV; ++j) z[a][b][j] = z[a][b][j] > 0.f ? z[a][b][j] : 0.f;
      }
    }
  // horizontal blends of every row: h[a][0] -> output column 2*ix, h[a][1] -> 2*ix + 1
  float h[3][2][NV];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      h[a][0][j] = 0.25f * z[a][0][j] + 0.75f * z[a][1][j];
      h[a][1][j] = 0.75f * z[a][1][j] + 0.25f * z[a][2][j];
    }
  TO *out = y + ((n * 2 * H + 2 * iy) * (2 * (int64_t)W) + 2 * ix) * C + cv * NV;
  const int64_t orow = 2 * (int64_t)W * C;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float o0[NV], o1[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      o0[j] = 0.25f * h[0][b][j] + 0.75f * h[1][b][j];
      o1[j] = 0.75f * h[1][b][j] + 0.25f * h[2][b][j];
    }
    rz_store<TO, NV>(out + b * C, o0);
    rz_store<TO, NV>(out + orow + b * C, o1);
  }
}

template <typename T, typename TG, bool RELU>
__global__ __launch_bounds__(256) void relu_up2x_bwd_kernel(const TG *__restrict__ dy, const T *__restrict__ x, int64_t total,
                                                            int H, int W, int C, T *__restrict__ dx) {
  constexpr int NV = RzVec<T>::NV;
  const int CV = C / NV;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int cv = (int)(t % CV);
  int64_t r = t / CV;
  const int ix = (int)(r % W); r /= W;
  const int iy = (int)(r % H);
  const int64_t n = r / H;
  const int H2 = 2 * H, W2 = 2 * W;
  const float wgt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  int oy[4], ox[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int v = 2 * iy - 1 + a;
    oy[a] = v < 0 ? 0 : (v >= H2 ? H2 - 1 : v);
    v = 2 * ix - 1 + a;
    ox[a] = v < 0 ? 0 : (v >= W2 ? W2 - 1 : v);
  }
  const TG *gbase = dy + (n * H2 * (int64_t)W2) * C + cv * NV;
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float rowacc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) rowacc[j] = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float g[NV];
      rz_load<TG, NV>(gbase + ((int64_t)oy[a] * W2 + ox[b]) * C, g);
#pragma unroll
      for (int j = 0; j < NV; ++j) rowacc[j] += wgt[b] * g[j];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += wgt[a] * rowacc[j];
  }
  const int64_t off = ((n * H + iy) * (int64_t)W + ix) * C + cv * NV;
  if (RELU) {
    float xv[NV];
    RzVec<T>::load(x + off, xv);
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = xv[j] > 0.f ? acc[j] : 0.f;
  }
  RzVec<T>::store(dx + off, acc);
}

}  // namespace m3

using namespace m3;

static bool rz_shape_ok(int dtype, int64_t N, int H, int W, int C) {
  const int nv = dtype == M3_F32 ? 4 : 8;
  return N >= 0 && H >= 1 && W >= 1 && C >= nv && C % nv == 0 && N * H * (int64_t)W * C < ((int64_t)1 << 40);
}

template <typename T, typename TO>
static void rz_launch_fwd(const void *x, int64_t N, int H, int W, int C, int relu, void *y, hipStream_t s) {
  const int64_t total = N * H * (int64_t)W * (C / RzVec<T>::NV);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (relu) hipLaunchKernelGGL((relu_up2x_fwd_kernel<T, TO, true>), grid, block, 0, s, (const T *)x, total, H, W, C, (TO *)y);
  else hipLaunchKernelGGL((relu_up2x_fwd_kernel<T, TO, false>), grid, block, 0, s, (const T *)x, total, H, W, C, (TO *)y);
}

template <typename T, typename TG>
static void rz_launch_bwd(const void *dy, const void *x, int64_t N, int H, int W, int C, int relu, void *dx, hipStream_t s) {
  const int64_t total = N * H * (int64_t)W * (C / RzVec<T>::NV);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (relu) hipLaunchKernelGGL((relu_up2x_bwd_kernel<T, TG, true>), grid, block, 0, s, (const TG *)dy, (const T *)x, total, H, W, C, (T *)dx);
  else hipLaunchKernelGGL((relu_up2x_bwd_kernel<T, TG, false>), grid, block, 0, s, (const TG *)dy, (const T *)x, total, H, W, C, (T *)dx);
}

extern "C" int m3_relu_up2x_fwd(const void *x, int x_dtype, int64_t N, int H, int W, int C, int relu, void *y, int y_dtype,
                                void *stream) {
  M3_REQUIRE(x && y, "m3_relu_up2x_fwd: null operand");
  M3_REQUIRE(dtype_ok(x_dtype) && (y_dtype == x_dtype || y_dtype == M3_F32), "m3_relu_up2x_fwd: output dtype is the input's or fp32");
  M3_REQUIRE(rz_shape_ok(x_dtype, N, H, W, C), "m3_relu_up2x_fwd: C must be a multiple of 8 (16-bit) / 4 (fp32)");
  M3_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "m3_relu_up2x_fwd: 16-byte aligned tensors");
  if (N == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == M3_F32) rz_launch_fwd<float, float>(x, N, H, W, C, relu, y, s);
  else if (x_dtype == M3_F16) { if (y_dtype == M3_F32) rz_launch_fwd<half_t, float>(x, N, H, W, C, relu, y, s); else rz_launch_fwd<half_t, half_t>(x, N, H, W, C, relu, y, s); }
  else { if (y_dtype == M3_F32) rz_launch_fwd<bf16_t, float>(x, N, H, W, C, relu, y, s); else rz_launch_fwd<bf16_t, bf16_t>(x, N, H, W, C, relu, y, s); }
  return check_launch("m3_relu_up2x_fwd");
}

extern "C" int m3_relu_up2x_bwd(const void *dy, int dy_dtype, const void *x, int x_dtype, int64_t N, int H, int W, int C,
                                int relu, void *dx, void *stream) {
  M3_REQUIRE(dy && dx && (x || !relu), "m3_relu_up2x_bwd: null operand");
  M3_REQUIRE(dtype_ok(x_dtype) && (dy_dtype == x_dtype || dy_dtype == M3_F32), "m3_relu_up2x_bwd: gradient dtype is the input's or fp32");
  M3_REQUIRE(rz_shape_ok(x_dtype, N, H, W, C), "m3_relu_up2x_bwd: C must be a multiple of 8 (16-bit) / 4 (fp32)");
  M3_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)x % 16) == 0, "m3_relu_up2x_bwd: 16-byte aligned tensors");
  if (N == 0) return M3_OK;
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == M3_F32) rz_launch_bwd<float, float>(dy, x, N, H, W, C, relu, dx, s);
  else if (x_dtype == M3_F16) { if (dy_dtype == M3_F32) rz_launch_bwd<half_t, float>(dy, x, N, H, W, C, relu, dx, s); else rz_launch_bwd<half_t, half_t>(dy, x, N, H, W, C, relu, dx, s); }
  else { if (dy_dtype == M3_F32) rz_launch_bwd<bf16_t, float>(dy, x, N, H, W, C, relu, dx, s); else rz_launch_bwd<bf16_t, bf16_t>(dy, x, N, H, W, C, relu, dx, s); }
  return check_launch("m3_relu_up2x_bwd");
}
