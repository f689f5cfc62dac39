// copy_persistent.hip - does a persistent, software-pipelined 16-byte copy beat the one-tile-per-workgroup copy that
// sets this repo's "practical ceiling" (0.80 of 8 TB/s)?  Wavefront-private tiles, buffer-addressed, next tile's loads
// issued before the current tile's stores, first pass peeled so that the loop's wait for the prefetched tile leaves
// the stores in flight (vmcnt(U + ..) instead of vmcnt(0), see fpq_rotate_mfma.h).
// build: hipcc -O3 --offload-arch=gfx950 -o copy_persistent copy_persistent.hip ; run: ./copy_persistent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_big_grid(const u32x4* __restrict__ x, u32x4* __restrict__ o, int64_t n) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < n) __builtin_nontemporal_store(__builtin_nontemporal_load(x + v), o + v);
}

template <int U>
__global__ __launch_bounds__(256) void copy_tile(const u32x4* __restrict__ x, u32x4* __restrict__ o, int64_t n) {
  const int64_t v0 = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
  u32x4 r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const int64_t v = v0 + u * 256; r[u] = v < n ? __builtin_nontemporal_load(x + v) : u32x4{0, 0, 0, 0}; }
#pragma unroll
  for (int u = 0; u < U; ++u) { const int64_t v = v0 + u * 256; if (v < n) __builtin_nontemporal_store(r[u], o + v); }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes < 0 ? 0 : (bytes > 0x7FFFFFFF ? 0x7FFFFFFF : (int)bytes), 0x00020000);
}

// wavefront w of the grid copies tiles w, w + W, ... of U * 64 vectors
template <int U, bool PEEL, int WAVES_PER_EU>
__global__ __launch_bounds__(256, WAVES_PER_EU) void copy_persistent(const u32x4* __restrict__ x, u32x4* __restrict__ o, int64_t n) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tiles = (n + U * 64 - 1) / (U * 64);
  const int64_t step = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  auto load = [&](int64_t tile, u32x4 (&r)[U]) {
    const __amdgpu_buffer_rsrc_t s = rsrc(x + tile * (U * 64), (n - tile * (U * 64)) * 16);
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = __builtin_amdgcn_raw_buffer_load_b128(s, lane * 16 + u * 1024, 0, 2);
  };
  auto store = [&](int64_t tile, const u32x4 (&r)[U]) {
    const __amdgpu_buffer_rsrc_t d = rsrc(o + tile * (U * 64), (n - tile * (U * 64)) * 16);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(r[u], d, lane * 16 + u * 1024, 0, 2);
  };
  u32x4 cur[U], nxt[U];
  if (t >= tiles) return;
  load(t, cur);
  auto pass = [&]() {
    load(t + step < tiles ? t + step : tiles, nxt);   // beyond the end: empty range, zeros
    store(t, cur);
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
  };
  if (PEEL) {
    pass();
    for (t += step; t < tiles; t += step) pass();
  } else {
    for (; t < tiles; t += step) pass();
  }
}

int main() {
  const int64_t n = 65536ll * 1920 / 8;   // 16-byte vectors of the headline tensor
  constexpr int NB = 4;
  u32x4 *xs[NB], *os[NB];
  for (int b = 0; b < NB; ++b) { CK(hipMalloc(&xs[b], n * 16)); CK(hipMalloc(&os[b], n * 16)); CK(hipMemset(xs[b], b + 1, n * 16)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int rot = 0;
  auto time_it = [&](const char* name, auto launch) {
    for (int i = 0; i < 40; ++i) { launch(xs[rot % NB], os[rot % NB]); ++rot; }
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) { launch(xs[rot % NB], os[rot % NB]); ++rot; }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t / 20);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-44s med %7.2f us  min %7.2f us  %6.0f GB/s  frac %.3f\n", name, ms[3] * 1e3, ms[0] * 1e3, n * 32 / ms[3] / 1e6, n * 32 / ms[3] / 1e6 / 8000);
  };
  time_it("copy, one vector per lane, big grid", [&](u32x4* x, u32x4* o) { hipLaunchKernelGGL(copy_big_grid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, o, n); });
  time_it("copy tile U2 (the quantizer's shape)", [&](u32x4* x, u32x4* o) { hipLaunchKernelGGL((copy_tile<2>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, x, o, n); });
#define PERSIST(U, PEEL, W, G) time_it("persistent U" #U " peel=" #PEEL " waves/EU " #W " grid " #G, [&](u32x4* x, u32x4* o) { \
    hipLaunchKernelGGL((copy_persistent<U, PEEL, W>), dim3(G), dim3(256), 0, 0, x, o, n); });
  PERSIST(4, true, 8, 2048) PERSIST(4, false, 8, 2048)
  PERSIST(4, true, 8, 4096) PERSIST(4, true, 8, 1024)
  PERSIST(2, true, 8, 2048) PERSIST(2, true, 8, 4096)
  PERSIST(8, true, 4, 1024) PERSIST(8, true, 4, 2048) PERSIST(8, false, 4, 1024)
  PERSIST(1, true, 8, 2048) PERSIST(1, true, 8, 4096)
  for (int b = 0; b < NB; ++b) { // verify the last variant's copy
    std::vector<uint32_t> h(4); hipMemcpy(h.data(), os[b] + n - 1, 16, hipMemcpyDeviceToHost);
    if (h[0] != (uint32_t)(0x01010101u * (b + 1))) { printf("verify failed buffer %d: %08x\n", b, h[0]); return 1; }
  }
  printf("verified\n");
  return 0;
}
