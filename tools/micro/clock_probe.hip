// What do s_memtime ticks mean, and what does the shader clock do under an MFMA load?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o build/clock_probe && build/clock_probe
// A chain of dependent v_mfma_f32_16x16x32_f16 (back-to-back issue on one SIMD) timed by s_memtime, by s_memrealtime
// (constant 100 MHz) and by HIP events, with 1 wave on the chip and with 4 waves on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void chain(int n, unsigned long long *out, float *sink) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(0.5f - j * 0.01f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  unsigned long long *out; float *sink;
  hipMalloc(&out, 16); hipMalloc(&sink, 4);
  const int n = 200000;                      // 800k MFMAs per wave
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int grid = cfg == 0 ? 1 : 256 * (cfg == 1 ? 1 : 2), block = cfg == 0 ? 64 : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<<<grid, block>>>(n / 10, out, sink); hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<<<grid, block>>>(n, out, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double mf = 4.0 * n;
    printf("grid %4d x %3d threads: %.3f ms by events | s_memtime %llu ticks (%.2f per MFMA, %.3f ticks/ns) | s_memrealtime %llu ticks (%.1f MHz)"
           " | %.1f ns per MFMA -> %.2f GHz if an MFMA is 16 cycles; chip %.0f TFLOP/s\n",
           grid, block, ms, h[0], h[0] / mf, h[0] / (ms * 1e6), h[1], h[1] / (ms * 1e3), ms * 1e6 / mf, 16.0 / (ms * 1e6 / mf),
           2.0 * 16 * 16 * 32 * mf * grid * (block / 64) / (ms * 1e-3) / 1e12);
  }
  return 0;
}
