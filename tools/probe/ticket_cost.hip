// ticket_cost.hip - what does "the last workgroup to finish does the fix-up" cost a streaming kernel?
// The dual-format quantizer needs a tensor-wide decision (any NaN => everything zero) and takes a second launch for it.
// One launch would do if every workgroup, after its stores, released them (agent-scope fence = L2 write-back on a chip
// whose eight L2s are not coherent with each other) and drew a ticket from ONE device-scope counter - the workgroup with
// the last ticket then knows that all stores of the launch have reached memory.  This probe times a plain 16-byte copy
// (the quantizer's access pattern: 512 vectors per workgroup) in four forms over the grid sizes of a generation:
//   0 plain   1 + relaxed ticket   2 + release fence and ticket   3 + fence and ticket by ONE workgroup in 8 (blockIdx % 8 == 0
//   after an XCD-local count is NOT attempted: the mapping of workgroups to XCDs is no contract)
//   hipcc -O3 --offload-arch=gfx950 -o ticket_cost ticket_cost.hip && ./ticket_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void copy_ticket(const u32x4* __restrict__ x, u32x4* __restrict__ out, long n_vec,
                                                   unsigned* ctr, unsigned* sink) {
  const long base = (long)blockIdx.x * 512 + threadIdx.x;
  u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (base < n_vec) a = __builtin_nontemporal_load(x + base);
  if (base + 256 < n_vec) b = __builtin_nontemporal_load(x + base + 256);
  if (base < n_vec) __builtin_nontemporal_store(a, out + base);
  if (base + 256 < n_vec) __builtin_nontemporal_store(b, out + base + 256);
  if (MODE == 0) return;
  if (MODE >= 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (MODE == 2 || MODE == 3) __syncthreads();      // every wavefront's stores released before the workgroup's ticket
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 3) *sink = __hip_atomic_load(sink + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the flag read
    }
  }
}

template <int MODE>
float run(const u32x4* x, u32x4* out, long n_vec, unsigned* ctr, int iters) {
  const unsigned grid = (unsigned)((n_vec + 511) / 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(copy_ticket<MODE>, dim3(grid), dim3(256), 0, 0, x, out, n_vec, ctr, ctr + 2);
  std::vector<float> ts;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(copy_ticket<MODE>, dim3(grid), dim3(256), 0, 0, x, out, n_vec, ctr, ctr + 2);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms * 1e3f / iters);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main() {
  const long max_vec = 25600L * 7680 / 8;
  u32x4 *x, *out;
  unsigned* ctr;
  hipMalloc(&x, max_vec * 16);
  hipMalloc(&out, max_vec * 16);
  hipMalloc(&ctr, 64);
  hipMemset(x, 1, max_vec * 16);
  hipMemset(ctr, 0, 64);
  printf("# us per launch (median of 5 bursts of 200 back-to-back launches), 16-byte copy, 512 vectors per workgroup\n");
  printf("# rows x 7680 fp16   workgroups   plain   +relaxed ticket   +release fence, barrier, ticket   (same + flag read by the last)\n");
  const int rows[] = {100, 400, 900, 1600, 2500, 3600, 6400, 10000, 16900, 25600};
  for (int r : rows) {
    const long n_vec = (long)r * 7680 / 8;
    const float t0 = run<0>(x, out, n_vec, ctr, 200), t1 = run<1>(x, out, n_vec, ctr, 200), t2 = run<2>(x, out, n_vec, ctr, 200),
                t3 = run<3>(x, out, n_vec, ctr, 200);
    printf("%6d %10ld %8.2f %8.2f %8.2f %8.2f\n", r, (n_vec + 511) / 512, t0, t1, t2, t3);
    unsigned h[4];
    hipMemcpy(h, ctr, 16, hipMemcpyDeviceToHost);
    if (h[0] != 0) printf("  counter not back at zero: %u\n", h[0]);
  }
  return 0;
}
