// lds_dma_issue.hip - what does ISSUING one 1 KiB LDS-DMA piece (global_load_lds_dwordx4) cost a wavefront, by address pattern?
// profiles/r05_gemm6_stamps.txt puts 29 % of the FP6 GEMM's wavefront time into issuing nine pieces per K step (100 cycles each).
// The GEMM's operands are row-major code rows: a piece gathers 96-byte (FP6) or 64-byte (FP4) segments of ~11 / 16 different rows,
// row_bytes apart, none of them aligned to a 128-byte line.  This probe runs the GEMM's feed loop WITHOUT the matrix work - two
// workgroups of four wavefronts per CU, two LDS stages, per step: vmcnt(0), barrier, nine (six) pieces - over three patterns:
//   rows96   the FP6 GEMM's pattern (256 + 128 rows x 96 bytes per step out of 1440-byte rows)
//   rows64   the FP4 GEMM's pattern (256 + 128 rows x 64 bytes per step out of 960-byte rows)
//   tiled    the same bytes from a pre-tiled operand: every piece is 1 KiB contiguous and 1 KiB aligned
//   k-major  planes [step][rows][seg]: a piece is 1 KiB contiguous too (16 / 10.7 consecutive rows' segments), with or without the
//            GEMM's chunk permutation inside a row's segment (does the lane ORDER inside whole lines matter?)
// and prints s_memtime cycles per piece (median over wavefronts) and the loop's wall time.  MFMA_PER_STEP > 0 adds that many
// dependent-free MFMAs per step and wavefront after the issue (the GEMM has 32) to see the feed under matrix load.
//   hipcc -O3 --offload-arch=gfx950 -o lds_dma_issue lds_dma_issue.hip && ./lds_dma_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

struct Args {
  const uint8_t* a;      // [T, row_bytes] or tiled image of the same size
  const uint8_t* w;      // [O, row_bytes]
  unsigned long long* stamps;   // [workgroups * 4][2]: issue cycles, wait cycles
  float* sink;
  int T, O, row_bytes, steps, seg;   // seg: bytes of a row per step (96 / 64); 0 = tiled
  int pieces;            // per wavefront and step
  int mfma;              // MFMAs per step and wavefront
  int swizzle;           // the lane -> chunk permutation of the GEMM's LDS image inside a row's segment
};

__global__ __launch_bounds__(256, 2) void feed_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_col = g.O / 128;
  const int cpx = (n_col + 7) >> 3;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int col_blk = xcd * cpx + local % cpx, row_blk = local / cpx;
  if (col_blk >= n_col || row_blk >= g.T / 256) return;
  const int t0 = row_blk * 256, o0 = col_blk * 128;
  const int stage = g.pieces * 4 * 1024;
  const uint8_t* gbase[2];
  uint32_t voff[12];
  const int apieces = g.pieces * 2 / 3;   // 256 of the 384 rows are A's
  if (g.seg) {
    gbase[0] = g.a + (int64_t)t0 * g.row_bytes;
    gbase[1] = g.w + (int64_t)o0 * g.row_bytes;
    const int lanes_per_row = g.seg / 16;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i >= g.pieces) break;
      const int piece = wave + 4 * i;
      const int ci = piece * 64 + lane;                    // 16-byte chunk index inside the operand's part of the stage
      const bool is_a = i < apieces;
      const int cj = is_a ? ci : ci - apieces * 4 * 64;
      const int r = cj / lanes_per_row;
      int c = cj - r * lanes_per_row;
      if (g.swizzle) c = g.seg == 64 ? (c ^ ((0x78 >> (((r & 15) >> 2) << 1)) & 3)) : (c + 6 - ((r >> 3) & 1)) % 6;   // the GEMMs' LDS images
      voff[i] = (uint32_t)r * (uint32_t)g.row_bytes + (uint32_t)c * 16;
    }
  } else {
    // tiled: the tile's K step s is one contiguous block of 256 (128) x seg bytes: [row_blk][s][piece][1024]
    gbase[0] = g.a + (int64_t)row_blk * g.steps * apieces * 4 * 1024;
    gbase[1] = g.w + (int64_t)col_blk * g.steps * (g.pieces - apieces) * 4 * 1024;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i >= g.pieces) break;
      const bool is_a = i < apieces;
      const int piece = is_a ? wave + 4 * i : wave + 4 * (i - apieces);
      voff[i] = (uint32_t)piece * 1024 + (uint32_t)lane * 16;
    }
  }
  // k-major planes [step][rows][seg] (row_bytes == seg): the rows' segments of one step are contiguous, the step stride is rows * seg
  const bool kmajor = g.seg && g.row_bytes == g.seg;
  const int64_t step_a = kmajor ? (int64_t)g.T * g.seg : g.seg ? g.seg : apieces * 4 * 1024;
  const int64_t step_w = kmajor ? (int64_t)g.O * g.seg : g.seg ? g.seg : (g.pieces - apieces) * 4 * 1024;
  unsigned long long issue = 0, wait = 0;
  v4f_t acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = v4f_t{0, 0, 0, 0};
  v8i_t af = v8i_t{lane, 1, 2, 3, 4, 5, 0, 0}, bf = v8i_t{7, lane, 5, 4, 3, 2, 0, 0};
#define ONE(s, buf, i_)                                                                                              \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                      \
               :                                                                                                     \
               : "v"(voff[i_]), "s"(gbase[(i_) < apieces ? 0 : 1] + (s) * ((i_) < apieces ? step_a : step_w)), \
                 "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + (buf) * stage +           \
                                                                                   (wave + 4 * (i_)) * 1024))       \
               : "m0")
#pragma unroll
  for (int i = 0; i < 12; ++i)
    if (i < g.pieces) ONE(0, 0, i);
  for (int s = 0; s < g.steps; ++s) {
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
    if (s + 1 < g.steps) {
#pragma unroll
      for (int i = 0; i < 12; ++i)
        if (i < g.pieces) ONE(s + 1, (s + 1) & 1, i);
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long tc = __builtin_amdgcn_s_memtime();
    wait += tb - ta;
    issue += tc - tb;
    if (g.mfma) {
#pragma unroll
      for (int k = 0; k < 32; ++k)
        acc[k & 7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, acc[k & 7], 2, 2, 0, 0, 0, 0);
    }
  }
  float sum = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) sum += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  sum += (float)smem[(tid * 16) % (2 * stage)];
  if (sum == 12345.678f) g.sink[0] = sum;
  if (lane == 0) {
    g.stamps[((int64_t)blockIdx.x * 4 + wave) * 2 + 0] = issue;
    g.stamps[((int64_t)blockIdx.x * 4 + wave) * 2 + 1] = wait;
  }
}

int main() {
  const int T = 65536, O = 5760, K = 1920, steps = K / 128;
  struct Case { const char* name; int seg, row_bytes, pieces, swizzle; } cases[] = {
      {"rows96 (FP6 GEMM)", 96, 1440, 9, 1}, {"tiled, 9 pieces  ", 0, 1440, 9, 0}, {"k-major 96       ", 96, 96, 9, 0}, {"k-major 96 swizzl", 96, 96, 9, 1},
      {"rows64 (FP4 GEMM)", 64, 960, 6, 1},  {"tiled, 6 pieces  ", 0, 960, 6, 0},  {"k-major 64       ", 64, 64, 6, 0}, {"k-major 64 swizzl", 64, 64, 6, 1}};
  const int n_wg = 8 * ((O / 128 + 7) / 8) * (T / 256);
  uint8_t *a, *w;
  unsigned long long* st;
  float* sink;
  CHECK(hipMalloc(&a, (size_t)T * 1440 + 4096));
  CHECK(hipMalloc(&w, (size_t)O * 1440 + 4096));
  CHECK(hipMemset(a, 1, (size_t)T * 1440 + 4096));
  CHECK(hipMemset(w, 1, (size_t)O * 1440 + 4096));
  CHECK(hipMalloc(&st, (size_t)n_wg * 4 * 2 * 8));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipFuncSetAttribute((const void*)feed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  printf("# feed loop of a 256 x 128 tile GEMM [%d x %d] -> %d without / with matrix work: %d workgroups of 4 wavefronts, 2 per CU, %d steps\n", T, K, O, n_wg, steps);
  printf("# pattern            MFMAs/step  cycles to issue one piece (median)  wait+barrier per step  launch ms  feed TB/s\n");
  for (int mf : {0, 32}) {
    for (auto& c : cases) {
      Args g{a, w, st, sink, T, O, c.row_bytes, steps, c.seg, c.pieces, mf, c.swizzle};
      const size_t lds = 75 * 1024;   // two workgroups per CU, as the GEMM
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipMemset(st, 0, (size_t)n_wg * 4 * 2 * 8));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(feed_kernel, dim3(n_wg), dim3(256), lds, 0, g);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
      }
      std::vector<unsigned long long> h((size_t)n_wg * 4 * 2);
      CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> iss, wt;
      for (size_t k = 0; k < h.size(); k += 2)
        if (h[k]) { iss.push_back((double)h[k] / ((steps - 1) * c.pieces)); wt.push_back((double)h[k + 1] / steps); }
      std::sort(iss.begin(), iss.end());
      std::sort(wt.begin(), wt.end());
      const double bytes = (double)(T / 256) * (O / 128) * steps * c.pieces * 4096.0;
      printf("  %s  %2d          %7.1f                             %7.1f                %.4f     %.2f\n", c.name, mf, iss[iss.size() / 2], wt[wt.size() / 2], best,
             bytes / best / 1e9);
    }
  }
  return 0;
}
