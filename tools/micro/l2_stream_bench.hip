// What one CU can pull from an L2-RESIDENT buffer, with every CU of the chip pulling at once: plain 16-byte loads into
// registers against LDS-DMA (global_load_lds, 16 bytes per lane) - the operand path of the GEMM / weight-gradient kernels.
// Every workgroup streams the same 2 MiB region (each XCD's L2 keeps its own copy), 8 loads per lane in flight.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_stream_bench tools/micro/l2_stream_bench.hip && /tmp/l2_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int REGION = 2 << 20;

template <int MODE>
__global__ __launch_bounds__(256) void stream(const char *__restrict__ buf, int64_t region, int passes, uint32_t *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, wave = tid >> 6;
  // workgroup b starts at its own offset so that the CUs of an XCD do not all hit one channel at the same moment
  int64_t off = ((int64_t)blockIdx.x * 65536) % region;
  u32x4 acc = u32x4{0u, 0u, 0u, 0u};
  const int64_t per_iter = 256 * 16 * 8;                         // 32 KiB per workgroup and iteration
  const int iters = (int)(region / per_iter) * passes;
  for (int it = 0; it < iters; ++it) {
    const char *p = buf + off + tid * 16;
    if (MODE == 0) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const u32x4 *)(p + j * 4096);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds((glb_void *)(buf + off + j * 4096 + wave * 1024 + (tid & 63) * 16),
                                         (lds_void *)(smem + ((it & 1) * 8 + j) * 4096 + wave * 1024), 16, 0, 0);
      if (it & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // keep one iteration's pieces in flight behind this one
    }
    off += per_iter;
    if (off >= region) off -= region;
  }
  if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
static void run(const char *name, const char *buf, uint32_t *sink, int wgs, int64_t region) {
  const int passes = 16;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(stream<MODE>, dim3(wgs), dim3(256), 65536, 0, buf, region, 2, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(stream<MODE>, dim3(wgs), dim3(256), 65536, 0, buf, region, passes, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * (double)(region / (256 * 16 * 8)) * passes * 32768.0;
  printf("%-28s region %5.1f MiB  workgroups %5d  %8.1f us  %7.2f TB/s chip  %6.1f GB/s per CU\n", name, region / 1048576.0, wgs, ms * 1e3,
         bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256);
}

int main() {
  char *buf; uint32_t *sink;
  hipMalloc(&buf, 1 << 30); hipMemset(buf, 1, 1 << 30); hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void *)stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int64_t region : {(int64_t)REGION, (int64_t)64 << 20, (int64_t)1 << 30}) {
    for (int wgs : {256, 512, 1024}) {
      run<0>("16-byte loads to registers", buf, sink, wgs, region);
      if (wgs <= 512) run<1>("LDS-DMA 16 bytes per lane", buf, sink, wgs, region);
    }
  }
  return 0;
}
