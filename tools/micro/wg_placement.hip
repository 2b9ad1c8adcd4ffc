// Where does workgroup i of a launch run?  (XCC, SE, CU) of each of 2048 workgroups of 256 threads with 32 KiB of LDS
// (4 resident per CU, as gemm_nt_dma_kernel), so that a tile order can put workgroups that share an operand on one CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wg_placement.hip -o build/wg_placement && build/wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
  extern __shared__ char smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  // stay resident for a while so that the first 1024 workgroups really are co-resident
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { smem[threadIdx.x] = (char)threadIdx.x; }
}
int main() {
  const int n = 2048;
  unsigned *d; hipMalloc(&d, n * 8);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  probe<<<n, 256, 32768>>>(d, 40000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  auto cu_of = [&](int i) { unsigned hw = h[2 * i], x = h[2 * i + 1] & 0xf; return (x << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf); };
  printf("first 40 workgroups: (xcc, se, sh, cu)\n");
  for (int i = 0; i < 40; ++i) { unsigned hw = h[2 * i]; printf("  wg %3d -> xcc %u se %u sh %u cu %2u\n", i, h[2 * i + 1] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf); }
  std::map<unsigned, std::vector<int>> by;
  for (int i = 0; i < 1024; ++i) by[cu_of(i)].push_back(i);
  printf("first 1024 workgroups occupy %zu distinct CUs; residents of the first CUs:\n", by.size());
  int shown = 0;
  for (auto &kv : by) { if (shown++ >= 12) break; printf("  cu %05x:", kv.first); for (int i : kv.second) printf(" %d", i); printf("\n"); }
  // how regular is it: distance pattern between workgroups sharing a CU
  std::map<int, int> hist;
  for (auto &kv : by) for (size_t a = 1; a < kv.second.size(); ++a) hist[kv.second[a] - kv.second[a - 1]]++;
  printf("id distance between consecutive residents of one CU (first 1024): ");
  for (auto &kv : hist) if (kv.second > 8) printf("%d x%d  ", kv.first, kv.second);
  printf("\n");
  return 0;
}
