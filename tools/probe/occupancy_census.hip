// occupancy_census.hip - how many 256-thread workgroups REALLY share a CU, by LDS size and register count: every workgroup
// notes when and where it lived (s_memrealtime, HW_REG_HW_ID, HW_REG_XCC_ID) and idles ~20 us; the host counts the
// workgroups alive per CU at mid-launch.  (The occupancy API is advisory; the adaLN producer never showed more than four
// workgroups on a CU where its 32 KiB and 74 registers admit five.)
//   hipcc -O3 --offload-arch=gfx950 -o occupancy_census occupancy_census.hip && ./occupancy_census
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

template <int VGPRS>
__global__ __launch_bounds__(256) void census(unsigned long long* rec, int spin) {
  extern __shared__ char lds[];
  if (VGPRS > 64) asm volatile("v_mov_b32 v79, 0" ::: "v79");     // makes the kernel allocate 80 registers
  if (VGPRS > 80) asm volatile("v_mov_b32 v95, 0" ::: "v95");     // ... 96
  if (VGPRS > 96) asm volatile("v_mov_b32 v103, 0" ::: "v103");   // ... 104
  lds[threadIdx.x] = 1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    rec[blockIdx.x * 4 + 0] = t0;
    rec[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    rec[blockIdx.x * 4 + 2] = hw;
    rec[blockIdx.x * 4 + 3] = xcc & 0xF;
  }
}

template <int VGPRS>
void run(int lds_bytes, unsigned long long* d) {
  const int n = 256 * 12;
  std::vector<unsigned long long> h(n * 4);
  hipFuncSetAttribute((const void*)census<VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(census<VGPRS>, dim3(n), dim3(256), lds_bytes, 0, d, 400);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int i = 0; i < n; ++i) { lo = std::min(lo, h[i * 4]); hi = std::max(hi, h[i * 4 + 1]); }
  // the peak number of workgroups alive on one CU at any workgroup's start
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
  for (int i = 0; i < n; ++i) {
    const unsigned long long hw = h[i * 4 + 2], key = (h[i * 4 + 3] << 16) | ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5);
    per_cu[key].push_back({h[i * 4], h[i * 4 + 1]});
  }
  int peak_min = 1 << 30, peak_max = 0;
  for (auto& kv : per_cu) {
    int peak = 0;
    for (auto& a : kv.second) {
      int alive = 0;
      for (auto& b : kv.second) alive += (b.first <= a.first && b.second > a.first);
      peak = std::max(peak, alive);
    }
    peak_min = std::min(peak_min, peak);
    peak_max = std::max(peak_max, peak);
  }
  int api = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census<VGPRS>, 256, lds_bytes);
  printf("LDS %6d B, %3d registers: occupancy API %d workgroups per CU; observed peak per CU: %d .. %d (%zu CUs), launch %.1f us\n", lds_bytes,
         VGPRS <= 64 ? 64 : VGPRS, api, peak_min, peak_max, per_cu.size(), (hi - lo) * 0.01);
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 256 * 12 * 32);
  for (int lds : {1024, 16384, 25600, 26624, 27136, 27648, 30720, 31232, 31744, 32256, 32512, 32768, 34816, 40960, 41472}) run<64>(lds, d);
  for (int lds : {16384, 25600, 32768}) run<80>(lds, d);
  for (int lds : {16384, 25600, 32768}) run<96>(lds, d);
  return 0;
}
