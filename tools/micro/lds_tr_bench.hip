// LDS read throughput on gfx950, per instruction kind: cycles per wave-instruction with 1 .. 8 waves of a workgroup issuing
// the same stream of reads (one CU: all waves of a workgroup share its LDS).  What the weight-gradient kernels need to know:
// is a transposed 16-bit fragment read (ds_read_b64_tr_b16, 512 bytes per wave) as cheap as its bytes (4 cycles at 128 B/clk)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_tr_bench tools/micro/lds_tr_bench.hip && /tmp/lds_tr_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void bench(int iters, int rs, unsigned long long *out, uint32_t *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 32768 / 4; i += blockDim.x) ((uint32_t *)smem)[i] = i;
  __syncthreads();
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void *)smem;
  uint32_t addr;
  if (KIND == 0) {            // the kernels' transposed read: row 4 lg + (li >> 2), 8 bytes at 4 (li & 3), XOR swizzle, row stride rs
    const int row = 4 * lg + (li >> 2), s3 = row & 7;
    addr = lds0 + row * rs + ((((li & 3) >> 1) ^ (s3 << 1)) * 16) + 8 * (li & 1);
  } else if (KIND == 1) {     // same without the swizzle (all rows at the same column)
    const int row = 4 * lg + (li >> 2);
    addr = lds0 + row * rs + ((li & 3) >> 1) * 16 + 8 * (li & 1);
  } else if (KIND == 2) {     // plain ds_read_b64, lane-linear
    addr = lds0 + lane * 8;
  } else {                    // ds_read_b128, lane-linear
    addr = lds0 + lane * 16;
  }
  u32x2 a0, a1, a2, a3, a4, a5, a6, a7;
  u32x4 b0, b1, b2, b3;
  uint32_t acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND <= 1) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(a0) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:32" : "=v"(a1) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:64" : "=v"(a2) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:96" : "=v"(a3) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(a4) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8224" : "=v"(a5) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8256" : "=v"(a6) : "v"(addr));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8288" : "=v"(a7) : "v"(addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += a0[0] + a1[1] + a2[0] + a3[1] + a4[0] + a5[1] + a6[0] + a7[1];
    } else if (KIND == 2) {
      asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(a0) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(a1) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(a2) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(a3) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(a4) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:2560" : "=v"(a5) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:3072" : "=v"(a6) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:3584" : "=v"(a7) : "v"(addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += a0[0] + a1[1] + a2[0] + a3[1] + a4[0] + a5[1] + a6[0] + a7[1];
    } else {
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(b0) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(b1) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b2) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(b3) : "v"(addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += b0[0] + b1[1] + b2[2] + b3[3];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int KIND>
static void run(const char *name, int rs, int per_iter_bytes, int instr_per_iter) {
  unsigned long long *out; uint32_t *sink;
  hipMalloc(&out, 8 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
  const int iters = 4096;
  for (int waves : {1, 2, 4, 8}) {
    hipLaunchKernelGGL(bench<KIND>, dim3(1), dim3(64 * waves), 32768, 0, iters, rs, out, sink);
    hipLaunchKernelGGL(bench<KIND>, dim3(1), dim3(64 * waves), 32768, 0, iters, rs, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int w = 0; w < waves; ++w) cyc = cyc > (double)h[w] ? cyc : (double)h[w];
    // s_memtime counts at 100 MHz-ish constant rate on some parts: report raw ticks and per-instruction ratios
    printf("%-44s rs=%4d waves=%d  ticks/iter %8.2f  ticks per wave-instr %6.2f  bytes/tick (CU) %7.1f\n", name, rs, waves, cyc / iters,
           cyc / iters / instr_per_iter, (double)per_iter_bytes * waves / (cyc / iters));
  }
  hipFree(out); hipFree(sink);
}

int main() {
  run<0>("ds_read_b64_tr_b16, kernel addressing", 256, 8 * 512, 8);
  run<0>("ds_read_b64_tr_b16, kernel addressing", 512, 8 * 512, 8);
  run<1>("ds_read_b64_tr_b16, no swizzle", 256, 8 * 512, 8);
  run<1>("ds_read_b64_tr_b16, no swizzle", 288, 8 * 512, 8);
  run<2>("ds_read_b64 lane-linear", 0, 8 * 512, 8);
  run<3>("ds_read_b128 lane-linear", 0, 4 * 1024, 4);
  return 0;
}
