// Deterministic second-stage reductions shared by the bias-grad, LayerNorm-parameter and
// gate-summary paths: out[g][n] = (beta ? out : 0) + sum_r part[g][r][n], rows added in a
// fixed order (8 interleaved row lanes, then lanes 0..7), so results are run-to-run identical.
#include "common.h"

namespace m3 {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const TI *__restrict__ part, int nrows, int N,
                                                          int64_t gstride, TO *__restrict__ out, int beta) {
  __shared__ TO s[8][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c, g = blockIdx.y;
  TO acc = 0;
  if (n < N) {
    const TI *p = part + (int64_t)g * gstride + n;
#pragma unroll 4
    for (int r = rl; r < nrows; r += 8) acc += (TO)p[(int64_t)r * N];
  }
  s[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && n < N) {
    TO t = s[0][c];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += s[i][c];
    TO *o = out + (int64_t)g * N + n;
    *o = beta ? (*o + t) : t;
  }
}

// two outputs (LayerNorm dgamma / dbeta): part is [2][nrows][N]
__global__ __launch_bounds__(256) void reduce_rows2_kernel(const float *__restrict__ part, int nrows, int N,
                                                           float *__restrict__ out0, float *__restrict__ out1, int beta) {
  __shared__ float s[8][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c, g = blockIdx.y;
  float acc = 0.f;
  if (n < N) {
    const float *p = part + (int64_t)g * nrows * N + n;
#pragma unroll 4
    for (int r = rl; r < nrows; r += 8) acc += p[(int64_t)r * N];
  }
  s[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && n < N) {
    float t = s[0][c];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += s[i][c];
    float *o = (g ? out1 : out0) + n;
    *o = beta ? (*o + t) : t;
  }
}

// the same for `count` layers in one launch (blockIdx.z = layer first + z): layer j's partials at part + j * layer_stride,
// its outputs from a device-resident pointer table
__global__ __launch_bounds__(256) void reduce_rows2_batch_kernel(const float *__restrict__ part, int64_t layer_stride, int nrows,
                                                                 int N, const m3_ln_param_grads *__restrict__ outs, int first,
                                                                 int beta) {
  __shared__ float s[8][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c, g = blockIdx.y, layer = first + blockIdx.z;
  float acc = 0.f;
  if (n < N) {
    const float *p = part + (int64_t)layer * layer_stride + (int64_t)g * nrows * N + n;
#pragma unroll 4
    for (int r = rl; r < nrows; r += 8) acc += p[(int64_t)r * N];
  }
  s[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && n < N) {
    float t = s[0][c];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += s[i][c];
    float *o = (g ? outs[layer].dbeta : outs[layer].dgamma) + n;
    *o = beta ? (*o + t) : t;
  }
}

int launch_reduce_rows2_batch_f32(const float *part, int64_t layer_stride, int nrows, int N, const m3_ln_param_grads *outs,
                                  int first, int count, int beta, hipStream_t s) {
  hipLaunchKernelGGL(reduce_rows2_batch_kernel, dim3((N + 31) / 32, 2, count), dim3(256), 0, s, part, layer_stride, nrows, N,
                     outs, first, beta);
  return check_launch("reduce_rows2_batch_f32");
}

int launch_reduce_rows2_f32(const float *part, int nrows, int N, float *out0, float *out1, int beta, hipStream_t s) {
  hipLaunchKernelGGL(reduce_rows2_kernel, dim3((N + 31) / 32, 2), dim3(256), 0, s, part, nrows, N, out0, out1, beta);
  return check_launch("reduce_rows2_f32");
}

int launch_reduce_rows_f32(const float *part, int nrows, int N, int G, int64_t gstride, float *out, int beta,
                           hipStream_t s) {
  hipLaunchKernelGGL((reduce_rows_kernel<float, float>), dim3((N + 31) / 32, G), dim3(256), 0, s, part, nrows, N,
                     gstride, out, beta);
  return check_launch("reduce_rows_f32");
}

int launch_reduce_rows_i32(const int32_t *part, int nrows, int N, int G, int64_t gstride, int64_t *out, int beta,
                           hipStream_t s) {
  hipLaunchKernelGGL((reduce_rows_kernel<int32_t, int64_t>), dim3((N + 31) / 32, G), dim3(256), 0, s, part, nrows, N,
                     gstride, out, beta);
  return check_launch("reduce_rows_i32");
}

}  // namespace m3
