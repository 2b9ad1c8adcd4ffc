// Groups of S workgroups ON ONE XCD stream the SAME fresh region (each group its own: HBM traffic is the sum of the regions),
// LDS-DMA, two iterations of 32 KiB per workgroup in flight - the way the 3-9 workgroups of a row slab share a GEMM operand.
// lockstep: all members ask for the same lines at the same moment; staggered: member m runs m x STG bytes behind member m - 1.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/shared_stream_bench tools/micro/shared_stream_bench.hip && /tmp/shared_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int DUMMY>
__global__ __launch_bounds__(256) void stream(const char *__restrict__ buf, int64_t region, int S, int64_t stagger, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int b = blockIdx.x;
  const int lid = (b % 8) * (nwg / 8) + b / 8;           // consecutive logical ids sit on one XCD
  const int grp = lid / S, mem = lid % S;
  const char *base = buf + (int64_t)grp * region;
  const int64_t per_iter = 32768;
  const int iters = (int)(region / per_iter);
  // member m starts m * stagger bytes BEHIND (it reads the region's tail first, then follows member 0 around)
  int64_t off = (region - (int64_t)mem * stagger) % region;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((glb_void *)(base + off + j * 4096 + wave * 1024 + (tid & 63) * 16),
                                       (lds_void *)(smem + ((it & 1) * 8 + j) * 4096 + wave * 1024), 16, 0, 0);
    if (it > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    off += per_iter;
    if (off >= region) off -= region;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
  const int nwg = 256;
  const int64_t total = (int64_t)3 << 30;
  char *buf;
  if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, total);
  hipFuncSetAttribute((const void *)stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int S : {1, 3, 9, 27}) {
    const int groups = nwg / S + 1;
    int64_t region = (total / groups) / 32768 * 32768;
    if (region > ((int64_t)96 << 20)) region = (int64_t)96 << 20;
    for (int64_t stg : {(int64_t)0, (int64_t)32768, (int64_t)65536, (int64_t)262144}) {
      if (S == 1 && stg) continue;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(stream<0>, dim3(nwg), dim3(256), 65536, 0, buf, region, S, stg, nwg);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      const double per_cu = (double)region / (best * 1e-3) / 1e9;
      printf("sharers %2d  stagger %7lld B  region %5.1f MiB  %8.1f us  %6.1f GB/s per CU  delivered %6.2f TB/s  unique (HBM) %5.2f TB/s\n", S,
             (long long)stg, region / 1048576.0, best * 1e3, per_cu, per_cu * nwg / 1e3, per_cu * nwg / 1e3 / S);
    }
  }
  return 0;
}
