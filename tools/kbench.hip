// kbench.hip - standalone kernel microbenchmark / cross-check (development tool).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude -o tools/kbench tools/kbench.hip
// Compares the fast fp16 path with the generic kernels bit for bit on several
// input distributions, then times variants interleaved in one process next to a
// plain 16-byte copy (the practical HBM ceiling on the same device).
#include "../fpqvar_amd/csrc/fpq_kernels.hip"

#include <cstdio>
#include <functional>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

namespace {
__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: ~gaussian, mode 1: raw random bit patterns, mode 2: gelu-like (mostly small negatives + positives), mode 3: heavy tails
__global__ void fill_kernel(uint16_t* x, int64_t n, int mode, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = hash32((uint32_t)i * 2654435761u + seed);
    uint32_t h2 = hash32(h + 0x9e3779b9u);
    float u = 0.f;
    for (int k = 0; k < 4; ++k) { u += (float)((h >> (8 * k)) & 0xFF) / 255.f; }
    float g = (u - 2.0f) * 1.732f;   // ~N(0,1)
    float v;
    if (mode == 0) v = g;
    else if (mode == 2) { float t = 1.5f * g; v = 0.5f * t * (1.f + tanhf(0.79788456f * (t + 0.044715f * t * t * t))); }
    else if (mode == 3) v = g * expf(0.7f * ((float)(h2 & 0xFFFF) / 65535.f * 4.f - 2.f));
    else v = 0.f;
    uint16_t b = (mode == 1) ? (uint16_t)(h2 & 0xFFFF) : (uint16_t)f2h(v);
    if (mode != 1) {
      if ((h2 & 0xFFFFF) == 7) b = 0;                       // exact zeros
      if (((i >> 7) & 0x3FF) == 5) b = 0;                   // an all-zero group now and then
      if (((i >> 7) & 0x3FF) == 9) b &= 0x7FFF;             // single-sign groups
      if (((i >> 7) & 0x3FF) == 11) b |= 0x8000;
      if (((i >> 7) & 0x3FF) == 13) b = (uint16_t)(h2 & 0x83FF);  // subnormal-only group
    }
    x[i] = b;
  }
}

__global__ void diff_kernel(const uint16_t* a, const uint16_t* b, int64_t n, unsigned long long* cnt, long long* first) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint16_t p = a[i], q = b[i];
    bool pn = (p & 0x7FFF) > 0x7C00, qn = (q & 0x7FFF) > 0x7C00;
    bool same = (pn && qn) || (p == q);
    if (!same) { atomicAdd(cnt, 1ULL); atomicMin((unsigned long long*)first, (unsigned long long)i); }
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ o, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t v0 = (int64_t)blockIdx.x * 256 + threadIdx.x; v0 < n; v0 += stride * U) {
    u32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t v = v0 + u * stride; r[u] = v < n ? (NT ? __builtin_nontemporal_load(x + v) : x[v]) : u32x4{0,0,0,0}; }
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t v = v0 + u * stride; if (v < n) { if (NT) __builtin_nontemporal_store(r[u], o + v); else o[v] = r[u]; } }
  }
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_tile_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ o, int64_t n) {
  const int64_t tiles = (n + 256 * U - 1) / (256 * U);
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t v0 = tile * (256 * U) + threadIdx.x;
    u32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t v = v0 + u * 256; r[u] = v < n ? (NTL ? __builtin_nontemporal_load(x + v) : x[v]) : u32x4{0,0,0,0}; }
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t v = v0 + u * 256; if (v < n) { if (NTS) __builtin_nontemporal_store(r[u], o + v); else o[v] = r[u]; } }
  }
}

struct Variant { std::string name; std::function<void()> run; std::vector<float> ms; };

int fpq_quant_rows_generic(const void* x, void* out, int64_t rows, int64_t cols, int id, int, int, hipStream_t st) {
  DualArgs d = {};
  return dispatch_rows<false>(x, out, rows, cols, FPQ_F16, FPQ_F16, make_fmt(id), d, st);
}
int fpq_quant_rows_dual_generic(const void* x, void* out, int64_t rows, int64_t cols, int neg, int pos, int, int,
                                const void*, float, void*, hipStream_t st) {
  DualArgs d;
  d.fneg = make_fmt(neg); d.fpos = make_fmt(pos); d.clip_absmax = nullptr; d.clip_strength = 1.f; d.nan_flag = nullptr;
  return dispatch_rows<true>(x, out, rows, cols, FPQ_F16, FPQ_F16, d.fneg, d, st);
}
}  // namespace

int main(int argc, char** argv) {
  int64_t rows = 65536, cols = 1920;
  if (argc > 2) { rows = atoll(argv[1]); cols = atoll(argv[2]); }
  const int64_t n = rows * cols, n_vec = n / 8;
  uint16_t *x, *o1, *o2;
  constexpr int NB = 4;   // rotate over NB input/output pairs (2 GB) so that nothing is re-served by the 256 MiB MALL
  uint16_t *xs[NB], *os[NB];
  for (int b = 0; b < NB; ++b) { CK(hipMalloc(&xs[b], n * 2)); CK(hipMalloc(&os[b], n * 2)); }
  x = xs[0]; o1 = os[0]; o2 = os[1];
  int rot = 0;
  unsigned long long* cnt; long long* first;
  CK(hipMalloc(&cnt, 8)); CK(hipMalloc(&first, 8));
  hipStream_t st = 0;

  struct Case { const char* name; int sym_table; int neg, pos; };
  const Case cases[] = {{"e2m1", FPQ_E2M1, -1, -1}, {"e1m2", FPQ_E1M2, -1, -1}, {"e3m0", FPQ_E3M0, -1, -1},
                        {"e2m3", FPQ_E2M3, -1, -1}, {"e3m2", FPQ_E3M2, -1, -1},
                        {"dual fp4", -1, FPQ_E1M2_NEG, FPQ_E2M1_POS}, {"dual fp6", -1, FPQ_INT_NEG, FPQ_E2M3_POS}};
  int bad_total = 0;
  const int64_t chk_rows = 8192;   // cross-check on a slice (the generic kernels are the slow part)
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st, x, n, mode, 1234u + mode);
    for (const Case& c : cases) {
      for (int g : {128, 64, 1920, 7680, 15360}) {
        int64_t nr = chk_rows * cols / g;
        int r1, r2;
        if (c.sym_table >= 0) {
          r1 = fpq_quant_rows_generic(x, o1, nr, g, c.sym_table, FPQ_F16, FPQ_F16, st);
          r2 = fpq_quant_rows(x, o2, nr, g, c.sym_table, FPQ_F16, FPQ_F16, st);
        } else {
          r1 = fpq_quant_rows_dual_generic(x, o1, nr, g, c.neg, c.pos, FPQ_F16, FPQ_F16, nullptr, 1.0f, nullptr, st);
          r2 = fpq_quant_rows_dual(x, o2, nr, g, c.neg, c.pos, FPQ_F16, FPQ_F16, nullptr, 1.0f, nullptr, st);
        }
        if (r1 || r2) { printf("launch error %d %d\n", r1, r2); return 1; }
        CK(hipMemsetAsync(cnt, 0, 8, st)); CK(hipMemsetAsync(first, 0x7f, 8, st));
        hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, st, o1, o2, nr * g, cnt, first);
        unsigned long long hc; long long hf;
        CK(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, first, 8, hipMemcpyDeviceToHost));
        printf("check mode %d %-9s g=%-5d: %llu mismatches", mode, c.name, g, hc);
        if (hc) {
          static uint16_t hx[16384]; uint16_t ha, hb; int64_t g0 = hf / g * g;
          CK(hipMemcpy(&ha, o1 + hf, 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, o2 + hf, 2, hipMemcpyDeviceToHost));
          CK(hipMemcpy(hx, x + g0, 2 * g, hipMemcpyDeviceToHost));
          uint16_t am = 0; for (int i = 0; i < g; ++i) am = std::max<uint16_t>(am, hx[i] & 0x7FFF);
          printf("  first @%lld x=0x%04x generic=0x%04x fast=0x%04x group absmax=0x%04x", hf, hx[hf - g0], ha, hb, am);
          bad_total++;
        }
        printf("\n");
      }
    }
  }

  // ---- timing ----
  for (int b = 0; b < NB; ++b) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st, xs[b], n, 0, 99u + b);
  CK(hipDeviceSynchronize());
  const bool rotate = !(argc > 3 && atoi(argv[3]) == 0);
  auto X = [&]() { return rotate ? xs[rot % NB] : xs[0]; };
  auto O = [&]() { uint16_t* p = rotate ? os[rot % NB] : os[0]; ++rot; return p; };
  std::vector<Variant> vs;
  auto add = [&](std::string name, std::function<void()> f) { vs.push_back({name, f, {}}); };
  int gcopy = 2048;
  add("copy nt U4", [&] { hipLaunchKernelGGL((copy_kernel<4, true>), dim3(gcopy), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy plain U4", [&] { hipLaunchKernelGGL((copy_kernel<4, false>), dim3(gcopy), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy nt U1 big grid", [&] { hipLaunchKernelGGL((copy_kernel<1, true>), dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  auto tiles = [&](int U) { return (unsigned)((n_vec + 256 * U - 1) / (256 * U)); };
  add("copy tile U2 nt/nt", [&] { hipLaunchKernelGGL((copy_tile_kernel<2, true, true>), dim3(tiles(2)), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy tile U4 nt/nt", [&] { hipLaunchKernelGGL((copy_tile_kernel<4, true, true>), dim3(tiles(4)), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy tile U8 nt/nt", [&] { hipLaunchKernelGGL((copy_tile_kernel<8, true, true>), dim3(tiles(8)), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy tile U4 nt/plain", [&] { hipLaunchKernelGGL((copy_tile_kernel<4, true, false>), dim3(tiles(4)), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy tile U4 plain/nt", [&] { hipLaunchKernelGGL((copy_tile_kernel<4, false, true>), dim3(tiles(4)), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy tile U4 pl/pl", [&] { hipLaunchKernelGGL((copy_tile_kernel<4, false, false>), dim3(tiles(4)), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("copy tile U4 cap4096", [&] { hipLaunchKernelGGL((copy_tile_kernel<4, true, true>), dim3(4096), dim3(256), 0, st, (const u32x4*)X(), (u32x4*)O(), n_vec); });
  add("fast e2m1 U1", [&] { launch_fast16<false, 1>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U2", [&] { launch_fast16<false, 2>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U4", [&] { launch_fast16<false, 4>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U2 pl/nt", [&] { launch_fast16<false, 2, false, true>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U2 nt/pl", [&] { launch_fast16<false, 2, true, false>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U2 pl/pl", [&] { launch_fast16<false, 2, false, false>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U4 pl/nt", [&] { launch_fast16<false, 4, false, true>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U8", [&] { launch_fast16<false, 8>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("fast e2m1 U4 cap2048", [&] { launch_fast16<false, 4>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st, 2048); });
  add("fast e2m1 U4 cap8192", [&] { launch_fast16<false, 4>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st, 8192); });
  add("dual fp4 U1", [&] { launch_fast16<true, 1>(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st); });
  add("dual fp4 U2", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st); });
  add("dual fp4 U4", [&] { launch_fast16<true, 4>(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st); });
  add("dual fp4 U2 cap4096", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st, 4096); });
  add("dual fp4 U2 cap8192", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st, 8192); });
  add("dual fp4 U2 cap16384", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st, 16384); });
  add("dual fp4 U4 cap8192", [&] { launch_fast16<true, 4>(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st, 8192); });
  add("sym e2m1 U2 cap16384", [&] { launch_fast16<false, 2>(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_E2M1, st, 16384); });
  add("dualfp6 U2 cap2048", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_INT_NEG, FPQ_E2M3_POS, st, 2048); });
  add("dualfp6 U2 cap4096", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_INT_NEG, FPQ_E2M3_POS, st, 4096); });
  add("dualfp6 U4 cap4096", [&] { launch_fast16<true, 4>(X(), O(), n / 128, 128, FPQ_INT_NEG, FPQ_E2M3_POS, st, 4096); });
  add("dualfp6 U2 cap8192", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_INT_NEG, FPQ_E2M3_POS, st, 8192); });
  add("dualfp6 U2 full", [&] { launch_fast16<true, 2>(X(), O(), n / 128, 128, FPQ_INT_NEG, FPQ_E2M3_POS, st); });
  add("PAIR8 sym e2m1", [&] { launch_fast16_pair8<false>(X(), O(), n / 128, FPQ_E2M1, FPQ_E2M1, st); });
  add("PAIR8 dual fp4", [&] { launch_fast16_pair8<true>(X(), O(), n / 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, st); });
  add("PAIR8 dual fp6 (fill)", [&] { launch_fast16_pair8<true>(X(), O(), n / 128, FPQ_INT_NEG, FPQ_E2M3_POS, st); });
  static uint32_t* dflag = nullptr; if (!dflag) CK(hipMalloc(&dflag, 4));
  add("dual fp4 U2 +nanflag", [&] { fpq_quant_rows_dual(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, FPQ_F16, FPQ_F16, nullptr, 1.f, dflag, st); });
  add("fast dualfp6 U4 (fill)", [&] { launch_fast16<true, 4>(X(), O(), n / 128, 128, FPQ_INT_NEG, FPQ_E2M3_POS, st); });
  add("fast e2m3 token1920", [&] { fpq_quant_rows(X(), O(), n / 1920, 1920, FPQ_E2M3, FPQ_F16, FPQ_F16, st); });
  add("fast e2m3 token7680", [&] { fpq_quant_rows(X(), O(), n / 7680, 7680, FPQ_E2M3, FPQ_F16, FPQ_F16, st); });
  add("fast dualfp6 tok7680", [&] { fpq_quant_rows_dual(X(), O(), n / 7680, 7680, FPQ_INT_NEG, FPQ_E2M3_POS, FPQ_F16, FPQ_F16, nullptr, 1.f, nullptr, st); });
  add("generic e2m3 tok1920", [&] { fpq_quant_rows_generic(X(), O(), n / 1920, 1920, FPQ_E2M3, FPQ_F16, FPQ_F16, st); });
  // fp32 weights path: the same buffers seen as n/2 floats (bit patterns of two random halves = wild floats,
  // fine for timing; correctness of this path is covered by the pytest suite)
  auto add32 = [&](std::string name, std::function<void()> f) { vs.push_back({name + " [8B/el]", f, {}}); };
  add32("generic f32 e2m1 g128", [&] { DualArgs d = {}; dispatch_rows<false>(X(), O(), n / 2 / 128, 128, FPQ_F32, FPQ_F32, make_fmt(FPQ_E2M1), d, st); });
  add32("generic f32->f16 e2m3 row1920", [&] { DualArgs d = {}; dispatch_rows<false>(X(), O(), n / 2 / 1920, 1920, FPQ_F32, FPQ_F16, make_fmt(FPQ_E2M3), d, st); });
  static const uint32_t kSign[4] = {0x5a5ac3c3u, 0x0ff0a55au, 0x12345678u, 0x9abcdef0u};
  add("rotate+quant e2m1 f16", [&] { fpq_rotate_quant_rows(X(), O(), nullptr, n / 1920, 1920, FPQ_F16, nullptr, kSign, FPQ_E2M1, st); });
  add("generic e2m1 g128", [&] { fpq_quant_rows_generic(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_F16, FPQ_F16, st); });
  add("fast e2m1 g128", [&] { fpq_quant_rows(X(), O(), n / 128, 128, FPQ_E2M1, FPQ_F16, FPQ_F16, st); });
  add("fast e2m3 g128", [&] { fpq_quant_rows(X(), O(), n / 128, 128, FPQ_E2M3, FPQ_F16, FPQ_F16, st); });
  add("fast dual fp4 g128", [&] { fpq_quant_rows_dual(X(), O(), n / 128, 128, FPQ_E1M2_NEG, FPQ_E2M1_POS, FPQ_F16, FPQ_F16, nullptr, 1.f, nullptr, st); });
  add("fast e2m3 c=64", [&] { fpq_quant_rows(X(), O(), n / 64, 64, FPQ_E2M3, FPQ_F16, FPQ_F16, st); });
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 7, iters = 10;
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) {
      v.run();   // warm
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) v.run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / iters);
    }
  printf("\nrotate=%d\n", (int)rotate);
  printf("\n%-24s %10s %10s %10s %8s\n", "variant", "med us", "min us", "GB/s(med)", "frac8T");
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    double bytes = (v.name.find("[8B/el]") != std::string::npos) ? (double)n / 2 * 8 : (double)n * 4;
    if (v.name.find("f32->f16") != std::string::npos) bytes = (double)n / 2 * 6;
    double gbs = bytes / (med * 1e-3) / 1e9;
    printf("%-24s %10.1f %10.1f %10.0f %8.3f\n", v.name.c_str(), med * 1e3, mn * 1e3, gbs, gbs / 8000.0);
  }
  printf("\nbad=%d\n", bad_total);
  return bad_total ? 2 : 0;
}
