// fpq_kernels.hip - gfx950 (MI355X) fake-quantization kernels behind include/fpq.h.
//
// What the reference computes (PKU-SEC-Lab/FPQVAR, tr/ = models_fp_quant_transform_rotate/):
//   quant/quant_kernel.cu:25-37        nearest entry of a value table, last index wins ties,
//                                      best distance starts at 102400 -> NaN/Inf/far give 0.0
//   tr/quant_utils.py:265-282 ...      ~11 torch ops around that kernel: absmax per row,
//                                      scale = absmax / max|table|, x / scale, fp32 cast,
//                                      kernel, q * scale, cast back
// Here each of those Python bodies is ONE launch: 16-byte coalesced loads, the row
// kept in registers between the absmax reduction and the rounding, cross-lane
// reduction with shuffles (rows of <= 1 KiB live inside one wavefront) or through
// LDS (long rows, one workgroup per row), and the table lookup replaced by a closed
// form on the minifloat structure (no K-step scan).  HBM traffic is the
// algorithmic minimum: every input byte read once, every output byte written once.
//
// Bit-exactness notes
//   * all scale / normalise arithmetic is IEEE fp32 (`/` is correctly rounded:
//     -fhip-fp32-correctly-rounded-divide-sqrt is hipcc's default; contraction is
//     switched off at build time) with the same intermediate roundings to x's dtype
//     that torch performs (fp16 ops = fp32 compute + one rounding);
//   * absmax is taken on the integer bit patterns of |x| so that NaN (largest
//     pattern) propagates exactly like torch's max;
//   * closed form: for r = |xn|, levels of a sign-magnitude minifloat with
//     subnormals have spacing 2^(max(e,emin)-M); a tie between two levels goes to
//     the larger VALUE (the scan's `<=`), i.e. up in magnitude for xn > 0 and down
//     for xn < 0.  tests/ prove it equal to the scan on every fp16 input and on fp32
//     neighbourhoods of every midpoint; fpq_quant_nearest keeps the literal scan.
#include "fpq_common.h"
#include <string.h>

// Experiment switches (fpq_common.h, FPQ_OPTION_LIST; include/fpq.h, fpq_set_option): the table both translation units
// read.  The initialiser below is the ONLY place in the library that touches the environment.
int fpq_option_table[FPQ_OPT_COUNT];
namespace {
struct FpqOptionDesc { const char* name; int is_flag; };
const FpqOptionDesc kOptionDescs[FPQ_OPT_COUNT] = {
#define FPQ_OPT_DESC(name, is_flag) {#name, is_flag},
    FPQ_OPTION_LIST(FPQ_OPT_DESC)
#undef FPQ_OPT_DESC
};
struct FpqOptionInit {
  FpqOptionInit() {
    for (int i = 0; i < FPQ_OPT_COUNT; ++i) {
      const char* e = getenv(kOptionDescs[i].name);
      fpq_option_table[i] = (!e || !*e) ? FPQ_OPTION_DEFAULT : kOptionDescs[i].is_flag ? (strcmp(e, "0") != 0) : atoi(e);
    }
  }
} fpq_option_init;
int option_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < FPQ_OPT_COUNT; ++i)
    if (strcmp(name, kOptionDescs[i].name) == 0) return i;
  return -1;
}
}  // namespace

namespace {

// ---------------------------------------------------------------------------------
// Kernel 1: rows of <= 1 KiB - LPR lanes of one wavefront own a row, one 16-byte
// load per lane, the row never leaves registers.  Grid-stride over wave tiles,
// UNROLL independent loads in flight per lane.
// ---------------------------------------------------------------------------------
template <typename Tin, typename Tout, int LPR, bool DUAL, int UNROLL>
__global__ __launch_bounds__(kBlock) void rows_subwave_kernel(const u32x4* __restrict__ x,
                                                             void* __restrict__ outv, int64_t n_vec,
                                                             Fmt fs, DualArgs dual) {
  constexpr int V = DT<Tin>::kVec;
  // a workgroup owns contiguous tiles of kBlock*UNROLL vectors, dispatched in address order
  constexpr int64_t stride = kBlock;
  const int64_t tile_vecs = (int64_t)kBlock * UNROLL;
  const int64_t tiles = (n_vec + tile_vecs - 1) / tile_vecs;
  float clip = fs.preclamp;
  bool clip_nan = false;
  const bool has_clip = DUAL ? (dual.clip_absmax != nullptr) : (fs.preclamp > 0.0f);
  if (DUAL && has_clip) clip = clip_value<Tin>(dual, &clip_nan);
  bool saw_nan = false;

  // n_vec is a multiple of LPR (whole rows) and LPR divides 64, so a row never
  // straddles the `live` boundary inside a wavefront.
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t v0 = tile * tile_vecs + threadIdx.x;
    u32x4 raw[UNROLL];
    bool live[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int64_t v = v0 + u * stride;
      live[u] = v < n_vec;
      raw[u] = live[u] ? __builtin_nontemporal_load(x + v) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      float xf[V];
      uint32_t mneg = 0, mpos = 0;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float e = DT<Tin>::get(raw[u], i);
        if (has_clip) e = clamp_like_torch(e, clip, clip_nan);
        xf[i] = e;
        uint32_t ab = DT<Tin>::absbits(e);
        if (DUAL) {
          uint32_t bn = (e <= 0.0f) ? ab : 0u, bp = (e > 0.0f) ? ab : 0u;
          saw_nan |= DT<Tin>::bits_nan(ab);
          mneg = mneg > bn ? mneg : bn;
          mpos = mpos > bp ? mpos : bp;
        } else {
          mneg = mneg > ab ? mneg : ab;
        }
      }
      mneg = lanes_max<LPR>(mneg);
      if (DUAL) mpos = lanes_max<LPR>(mpos);
      if (!live[u]) continue;
      float p[V];
      if (DUAL) {
        float sn = scale_of<Tin>(mneg, dual.fneg.gmax), sp = scale_of<Tin>(mpos, dual.fpos.gmax);
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = quant_dual<Tin>(xf[i], sn, sp, dual.fneg, dual.fpos);
      } else {
        float s = scale_of<Tin>(mneg, fs.gmax);
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = quant_sym<Tin>(xf[i], s, fs);
      }
      int64_t v = v0 + u * stride;
      if constexpr (sizeof(Tout) == sizeof(Tin)) {
        u32x4 o = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < V; ++i) DT<Tout>::put(o, i, p[i]);
        __builtin_nontemporal_store(o, (u32x4*)outv + v);
      } else if constexpr (sizeof(Tout) < sizeof(Tin)) {  // f32 -> f16: 4 halves = 8 bytes
        u32x2 o = {f2h(p[0]) | (f2h(p[1]) << 16), f2h(p[2]) | (f2h(p[3]) << 16)};
        __builtin_nontemporal_store(o, (u32x2*)outv + v);
      } else {  // f16 -> f32: 8 floats = 32 bytes
        u32x4 o0 = {fbits(p[0]), fbits(p[1]), fbits(p[2]), fbits(p[3])};
        u32x4 o1 = {fbits(p[4]), fbits(p[5]), fbits(p[6]), fbits(p[7])};
        __builtin_nontemporal_store(o0, (u32x4*)outv + 2 * v);
        __builtin_nontemporal_store(o1, (u32x4*)outv + 2 * v + 1);
      }
    }
  }
  if (DUAL && saw_nan && dual.nan_flag) atomicOr(dual.nan_flag, 1u);
}

// ---------------------------------------------------------------------------------
// Kernel 2: long rows (per-token 1920 / 7680 / 2304 / 9216 ...): one workgroup per
// row, up to MAXC 16-byte vectors per lane kept in registers between the
// reduction (shuffles + LDS) and the rounding; longer rows are re-read (L2).
// ---------------------------------------------------------------------------------

template <typename Tin, typename Tout, bool DUAL, int MAXC>
__global__ __launch_bounds__(kBlock) void rows_block_kernel(const Tin* __restrict__ x,
                                                           Tout* __restrict__ out, int64_t rows,
                                                           int64_t cols, Fmt fs, DualArgs dual) {
  constexpr int V = DT<Tin>::kVec;
  __shared__ uint32_t sh[kBlock / 64];
  const int64_t vec_per_row = cols / V;  // cols % V == 0 guaranteed by the host
  float clip = fs.preclamp;
  bool clip_nan = false;
  const bool has_clip = DUAL ? (dual.clip_absmax != nullptr) : (fs.preclamp > 0.0f);
  if (DUAL && has_clip) clip = clip_value<Tin>(dual, &clip_nan);
  bool saw_nan = false;

  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const u32x4* xr = (const u32x4*)(x + row * cols);
    u32x4 raw[MAXC];
    uint32_t mneg = 0, mpos = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int64_t v = (int64_t)c * kBlock + threadIdx.x;
      raw[c] = (v < vec_per_row) ? __builtin_nontemporal_load(xr + v) : u32x4{0, 0, 0, 0};
    }
    auto scan = [&](const u32x4& r) {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float e = DT<Tin>::get(r, i);
        if (has_clip) e = clamp_like_torch(e, clip, clip_nan);
        uint32_t ab = DT<Tin>::absbits(e);
        if (DUAL) {
          uint32_t bn = (e <= 0.0f) ? ab : 0u, bp = (e > 0.0f) ? ab : 0u;
          saw_nan |= DT<Tin>::bits_nan(ab);
          mneg = mneg > bn ? mneg : bn;
          mpos = mpos > bp ? mpos : bp;
        } else {
          mneg = mneg > ab ? mneg : ab;
        }
      }
    };
#pragma unroll
    for (int c = 0; c < MAXC; ++c) scan(raw[c]);
    for (int64_t v = (int64_t)MAXC * kBlock + threadIdx.x; v < vec_per_row; v += kBlock) scan(xr[v]);
    mneg = block_max(mneg, sh);
    if (DUAL) mpos = block_max(mpos, sh);
    float sn = scale_of<Tin>(mneg, DUAL ? dual.fneg.gmax : fs.gmax);
    float sp = DUAL ? scale_of<Tin>(mpos, dual.fpos.gmax) : 0.0f;

    auto emit = [&](const u32x4& r, int64_t v) {
      float p[V];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float e = DT<Tin>::get(r, i);
        if (has_clip) e = clamp_like_torch(e, clip, clip_nan);
        p[i] = DUAL ? quant_dual<Tin>(e, sn, sp, dual.fneg, dual.fpos) : quant_sym<Tin>(e, sn, fs);
      }
      Tout* o = out + row * cols + v * V;
      if constexpr (sizeof(Tout) == 2) {
        uint32_t w[V / 2];
#pragma unroll
        for (int i = 0; i < V / 2; ++i) w[i] = f2h(p[2 * i]) | (f2h(p[2 * i + 1]) << 16);
        if constexpr (V == 8)
          __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, (u32x4*)o);
        else
          __builtin_nontemporal_store(u32x2{w[0], w[1]}, (u32x2*)o);
      } else {
#pragma unroll
        for (int i = 0; i < V; i += 4)
          __builtin_nontemporal_store(u32x4{fbits(p[i]), fbits(p[i + 1]), fbits(p[i + 2]), fbits(p[i + 3])},
                                      (u32x4*)(o + i));
      }
    };
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int64_t v = (int64_t)c * kBlock + threadIdx.x;
      if (v < vec_per_row) emit(raw[c], v);
    }
    for (int64_t v = (int64_t)MAXC * kBlock + threadIdx.x; v < vec_per_row; v += kBlock) emit(xr[v], v);
  }
  if (DUAL && saw_nan && dual.nan_flag) atomicOr(dual.nan_flag, 1u);
}

// ---------------------------------------------------------------------------------
// Kernel 3: ragged / unaligned rows - scalar accesses, one workgroup per row.
// ---------------------------------------------------------------------------------
template <typename Tin, typename Tout, bool DUAL>
__global__ __launch_bounds__(kBlock) void rows_scalar_kernel(const Tin* __restrict__ x,
                                                            Tout* __restrict__ out, int64_t rows,
                                                            int64_t cols, Fmt fs, DualArgs dual) {
  __shared__ uint32_t sh[kBlock / 64];
  float clip = fs.preclamp;
  bool clip_nan = false;
  const bool has_clip = DUAL ? (dual.clip_absmax != nullptr) : (fs.preclamp > 0.0f);
  if (DUAL && has_clip) clip = clip_value<Tin>(dual, &clip_nan);
  bool saw_nan = false;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const Tin* xr = x + row * cols;
    uint32_t mneg = 0, mpos = 0;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      float e = load_scalar<Tin>(xr + c);
      if (has_clip) e = clamp_like_torch(e, clip, clip_nan);
      uint32_t ab = DT<Tin>::absbits(e);
      if (DUAL) {
        uint32_t bn = (e <= 0.0f) ? ab : 0u, bp = (e > 0.0f) ? ab : 0u;
        saw_nan |= DT<Tin>::bits_nan(ab);
        mneg = mneg > bn ? mneg : bn;
        mpos = mpos > bp ? mpos : bp;
      } else {
        mneg = mneg > ab ? mneg : ab;
      }
    }
    mneg = block_max(mneg, sh);
    if (DUAL) mpos = block_max(mpos, sh);
    float sn = scale_of<Tin>(mneg, DUAL ? dual.fneg.gmax : fs.gmax);
    float sp = DUAL ? scale_of<Tin>(mpos, dual.fpos.gmax) : 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      float e = load_scalar<Tin>(xr + c);
      if (has_clip) e = clamp_like_torch(e, clip, clip_nan);
      float p = DUAL ? quant_dual<Tin>(e, sn, sp, dual.fneg, dual.fpos) : quant_sym<Tin>(e, sn, fs);
      store_scalar<Tout>(out + row * cols + c, p);
    }
  }
  if (DUAL && saw_nan && dual.nan_flag) atomicOr(dual.nan_flag, 1u);
}

// ---------------------------------------------------------------------------------
// "neg reverse" rows (models_fp_quant/quant_utils.py:454-495): the non-positive half is
// shifted up by m = |row min| before it is quantized and shifted back afterwards.
// Three row reductions (min, max|shifted|, max positive), all on data held in registers.
// ---------------------------------------------------------------------------------
// key whose MAX is the row MIN with torch.min's NaN propagation (NaN -> 0xFFFFFFFF)
__device__ __forceinline__ uint32_t min_key(float e) {
  uint32_t b = fbits(e);
  uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // increasing in e
  return (e != e) ? 0xFFFFFFFFu : ~ord;
}
__device__ __forceinline__ float abs_from_min_key(uint32_t k) {
  uint32_t ord = ~k;
  uint32_t b = (ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord;
  return (k == 0xFFFFFFFFu) ? __builtin_nanf("") : fabsf(u2f(b));
}

template <typename T>
__device__ __forceinline__ float negrev_shifted(float xf, float m) {
  return DT<T>::round(((xf <= 0.0f) ? xf : 0.0f) + m);
}

// (T)(x / s).  fp16: x and s carry 11-bit significands, so one residual step on x*rcp(s) lands on the
// correctly rounded quotient (fpq_fast16.h, "exact fp16 division"); where it differs from IEEE
// (s = 0 or non-finite: 0 / NaN instead of inf / 0) the quantizer maps both to level 0.
template <typename T>
__device__ __forceinline__ float div_round(float x, float s) {
  if constexpr (sizeof(T) == 2) {
    float inv = (s == 0.0f) ? 0.0f : __builtin_amdgcn_rcpf(s);
    float y = x * inv;
    float e = __builtin_fmaf(-y, s, x);
    return DT<T>::round(__builtin_fmaf(e, inv, y));
  } else {
    return x / s;
  }
}

template <typename T>
__device__ __forceinline__ float quant_negrev(float xf, float xnr, float m, float snr, float sp, const Fmt& f) {
  float a = div_round<T>(xnr, snr);
  float b = div_round<T>((xf > 0.0f) ? xf : 0.0f, sp);
  uint32_t na = (a < 0.0f) ? 1u : 0u;
  float qa = quant_mag(fabsf(a), na, f);
  qa = (na && qa != 0.0f) ? -qa : qa;
  float qb = quant_mag(b, 0u, f);  // b >= 0 or NaN
  float t = qa * snr;
  t = t - m;
  float u = qb * sp;
  return t + u;
}

template <typename T, int LPR, int UNROLL>
__global__ __launch_bounds__(kBlock) void rows_negrev_subwave_kernel(const u32x4* __restrict__ x,
                                                                    u32x4* __restrict__ out, int64_t n_vec,
                                                                    Fmt fs) {
  constexpr int V = DT<T>::kVec;
  const int64_t tile_vecs = (int64_t)kBlock * UNROLL;
  const int64_t tiles = (n_vec + tile_vecs - 1) / tile_vecs;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t v0 = tile * tile_vecs + threadIdx.x;
    u32x4 raw[UNROLL];
    bool live[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int64_t v = v0 + (int64_t)u * kBlock;
      live[u] = v < n_vec;
      raw[u] = live[u] ? __builtin_nontemporal_load(x + v) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      float xf[V], xnr[V];
      uint32_t kmin = 0, mpos = 0, mnr = 0;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float e = DT<T>::get(raw[u], i);
        xf[i] = e;
        uint32_t k = min_key(e), bp = (e > 0.0f) ? DT<T>::absbits(e) : 0u;
        kmin = kmin > k ? kmin : k;
        mpos = mpos > bp ? mpos : bp;
      }
      kmin = lanes_max<LPR>(kmin);
      mpos = lanes_max<LPR>(mpos);
      const float m = abs_from_min_key(kmin);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        xnr[i] = negrev_shifted<T>(xf[i], m);
        uint32_t ab = DT<T>::absbits(xnr[i]);
        mnr = mnr > ab ? mnr : ab;
      }
      mnr = lanes_max<LPR>(mnr);
      if (!live[u]) continue;
      const float snr = scale_of<T>(mnr, fs.gmax), sp = scale_of<T>(mpos, fs.gmax);
      u32x4 o = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < V; ++i) DT<T>::put(o, i, quant_negrev<T>(xf[i], xnr[i], m, snr, sp, fs));
      __builtin_nontemporal_store(o, out + v0 + (int64_t)u * kBlock);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void rows_negrev_scalar_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                                   int64_t rows, int64_t cols, Fmt fs) {
  __shared__ uint32_t sh[kBlock / 64];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* xr = x + row * cols;
    uint32_t kmin = 0, mpos = 0, mnr = 0;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      float e = load_scalar<T>(xr + c);
      uint32_t k = min_key(e), bp = (e > 0.0f) ? DT<T>::absbits(e) : 0u;
      kmin = kmin > k ? kmin : k;
      mpos = mpos > bp ? mpos : bp;
    }
    kmin = block_max(kmin, sh);
    mpos = block_max(mpos, sh);
    const float m = abs_from_min_key(kmin);
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      uint32_t ab = DT<T>::absbits(negrev_shifted<T>(load_scalar<T>(xr + c), m));
      mnr = mnr > ab ? mnr : ab;
    }
    mnr = block_max(mnr, sh);
    const float snr = scale_of<T>(mnr, fs.gmax), sp = scale_of<T>(mpos, fs.gmax);
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      float e = load_scalar<T>(xr + c);
      store_scalar<T>(out + row * cols + c, quant_negrev<T>(e, negrev_shifted<T>(e, m), m, snr, sp, fs));
    }
  }
}

#include "fpq_fast16.h"
#include "fpq_rotate_mfma.h"
#ifndef FPQ_ROT_BUTTERFLY_BUILD   // 1: the butterfly forms of the rotation everywhere (A/B builds)
#define FPQ_ROT_BUTTERFLY_BUILD 0
#endif
#include "fpq_fast32.h"
#include "fpq_adaln.h"
#include "fpq_codes_mx.h"    // the operand-emitting quantizers of the matrix-core GEMMs (the GEMM kernels themselves,
#include "fpq_codes_fp8.h"   // fpq_gemm_fp4.h / fp8.h / fp6.h, are compiled in fpq_gemm.hip only: an edit there does not
#include "fpq_codes_fp6.h"   // rebuild this translation unit)

// ---------------------------------------------------------------------------------
// L0: literal scan (quant/quant_kernel.cu:25-37), any table of k <= 256 floats.
// The table index is wave-uniform, so table[j] is a scalar load (SGPR broadcast).
// ---------------------------------------------------------------------------------
// One element through the closed form of a built-in table (side: 0 symmetric, 1 values <= 0 only, 2 values >= 0 only)
__device__ __forceinline__ float nearest_closed(float xv, const Fmt& f, int side) {
  if (side == 0) {
    uint32_t neg = xv < 0.0f ? 1u : 0u;
    float qm = quant_mag(fabsf(xv), neg, f);
    return (neg && qm != 0.0f) ? -qm : qm;
  }
  if (side == 1) {
    // table holds only values <= 0: positive inputs fall on 0 (if within reach)
    // (0.0 is also what "nothing within reach" yields, so no reach test is needed here)
    float qm = (xv <= 0.0f) ? quant_mag(fabsf(xv), 1u, f) : 0.0f;
    return (qm != 0.0f) ? -qm : 0.0f;
  }
  return (xv > 0.0f) ? quant_mag(xv, 0u, f) : 0.0f;
}

// The value tables this library knows in closed form, as the reference spells them (sorted, FP6 with two zeros).
// Handed to the scan kernel by value: every workgroup compares the caller's table with them (bit for bit) once,
// and a recognised table takes the closed form (~20 VALU ops per element) instead of the K-step scan
// (3 ops per entry: 45 for K = 15, 190 for K = 64).  Same function of (x, table) either way - the closed form is
// checked against the scan on every fp16 value, the neighbourhoods of every entry / midpoint / reach limit and
// 2^26 random fp32 patterns per table (tests/test_gpu_parity.py).
struct KnownTables {
  float v[272];                 // all tables back to back (262 entries)
  Fmt fmt[FPQ_NUM_TABLES];
  int16_t off[FPQ_NUM_TABLES], k[FPQ_NUM_TABLES];
  int8_t side[FPQ_NUM_TABLES];
};

template <typename T>
__global__ __launch_bounds__(kBlock) void nearest_scan_kernel(const T* __restrict__ x,
                                                             const float* __restrict__ table,
                                                             T* __restrict__ z, int64_t n, int k,
                                                             KnownTables known) {
  __shared__ float tab[256];
  __shared__ int match;
  if ((int)threadIdx.x < k) tab[threadIdx.x] = table[threadIdx.x];
  if (threadIdx.x == 0) match = -1;
  __syncthreads();
  if ((int)threadIdx.x < FPQ_NUM_TABLES && known.k[threadIdx.x] == k) {
    bool same = true;
    const float* kv = known.v + known.off[threadIdx.x];
    for (int j = 0; j < k; ++j) same &= fbits(tab[j]) == fbits(kv[j]);
    if (same) match = (int)threadIdx.x;   // the built-in tables are pairwise different: at most one writer
  }
  __syncthreads();
  const int id = match;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  if (id >= 0) {
    const Fmt f = known.fmt[id];
    const int side = known.side[id];
    if constexpr (sizeof(T) == 4) {
      if ((((uintptr_t)x | (uintptr_t)z) & 15) == 0) {   // 16 bytes per lane, two vectors in flight
        const int64_t n_vec = n >> 2;
        const u32x4* xv = (const u32x4*)x;
        u32x4* zv = (u32x4*)z;
        for (int64_t v = (int64_t)blockIdx.x * (2 * kBlock) + threadIdx.x; v < n_vec; v += 2 * stride) {
          const bool two = v + kBlock < n_vec;
          u32x4 a = __builtin_nontemporal_load(xv + v);
          u32x4 b = two ? __builtin_nontemporal_load(xv + v + kBlock) : u32x4{0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = fbits(nearest_closed(u2f(a[e]), f, side));
            b[e] = fbits(nearest_closed(u2f(b[e]), f, side));
          }
          __builtin_nontemporal_store(a, zv + v);
          if (two) __builtin_nontemporal_store(b, zv + v + kBlock);
        }
        for (int64_t i = (n_vec << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
          z[i] = (T)nearest_closed((float)x[i], f, side);
        return;
      }
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
      z[i] = (T)nearest_closed((float)x[i], f, side);
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float xv = (float)x[i];
    float best = 102400.0f, zv = 0.0f;
    for (int j = 0; j < k; ++j) {
      float y = tab[j];
      float d = fabsf(xv - y);
      if (d <= best) {
        best = d;
        zv = y;
      }
    }
    z[i] = (T)zv;
  }
}

// quantize_to_nearest_grid (tr/quant_utils.py:209-230): grid[argmin_j |x - grid_j|] with torch.argmin's rules -
// the FIRST minimal index wins, a NaN or +-Inf input selects grid[0] - for any table; float32 result.
template <typename T>
__global__ __launch_bounds__(kBlock) void nearest_argmin_kernel(const T* __restrict__ x, const float* __restrict__ table,
                                                               float* __restrict__ z, int64_t n, int k) {
  __shared__ float tab[256];
  if ((int)threadIdx.x < k) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float xv = (float)x[i];
    float best = __builtin_inff(), zv = tab[0];
    for (int j = 0; j < k; ++j) {
      const float y = tab[j];
      const float d = fabsf(xv - y);
      if (d < best) {   // NaN and Inf distances never win: index 0 stays
        best = d;
        zv = y;
      }
    }
    z[i] = zv;
  }
}

__global__ __launch_bounds__(kBlock) void nearest_builtin_kernel(const float* __restrict__ x,
                                                                float* __restrict__ z, int64_t n, Fmt f,
                                                                int side /*0 sym, 1 neg-only, 2 pos-only*/) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float q = nearest_closed(x[i], f, side);
    z[i] = q;
  }
}

// ---------------------------------------------------------------------------------
// "any NaN in the tensor => the whole result is zero" (the reference's global clamp with a
// NaN bound, tr/quant_utils.py:421-422): second launch, a handful of workgroups that read ONE word and exit when the
// flag is clear - no atomic, no barrier on that path (round 3 drew a ticket per workgroup on every call: 64 device-scope
// atomics on one address, ~11 ns each and serialised, tools/probe/ticket_cost.hip - 3 us of the 7 us a 100-row call took).
// Only when the flag is up: scratch[0] = the flag the quantizer raised, scratch[1] = a ticket counter; every workgroup
// zero-fills its share and takes a ticket AFTER it has read the flag, the one that draws the last ticket clears both
// words - the scratch is zero again when the launch ends (no memset per call, and a captured graph can be replayed).
// (One launch instead of two would need every workgroup of the quantizer to release its stores and count itself:
// profiles/r04_ticket_cost.txt - the chip's eight L2s are not coherent with each other, the release is an L2 write-back
// per workgroup, 4 - 40 times the kernel's own time.)
// ---------------------------------------------------------------------------------
constexpr int kFixupBlocks = 64;
__global__ __launch_bounds__(kBlock) void zero_if_flag_kernel(uint8_t* __restrict__ out, int64_t n_bytes, uint32_t* scratch) {
  // the quantizer's atomicOr is visible to a plain (scalar) load here: kernel boundary; uniform branch
  if (__hip_atomic_load(scratch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n16 = ((uintptr_t)out & 15) == 0 ? n_bytes / 16 : 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) ((u32x4*)out)[i] = u32x4{0, 0, 0, 0};
  for (int64_t i = n16 * 16 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_bytes; i += stride) out[i] = 0;
  __syncthreads();   // every wavefront of this workgroup has read the flag
  if (threadIdx.x == 0 && atomicAdd(scratch + 1, 1u) == gridDim.x - 1) {   // every workgroup has read the flag by now
    __hip_atomic_store(scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(scratch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------
// absmax over a whole tensor (bit-pattern max, NaN propagates), atomicMax combine
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void absmax_kernel(const T* __restrict__ x, int64_t n,
                                                       uint32_t* __restrict__ out) {
  __shared__ uint32_t sh[kBlock / 64];
  constexpr int V = DT<T>::kVec;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t nv = n / V;
  uint32_t m = 0;
  const u32x4* xv = (const u32x4*)x;
  const bool aligned = ((uintptr_t)x & 15) == 0;
  if (aligned) {
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += stride) {
      u32x4 r = xv[v];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        uint32_t ab = DT<T>::absbits(DT<T>::get(r, i));
        m = m > ab ? m : ab;
      }
    }
  }
  for (int64_t i = (aligned ? nv * V : 0) + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    uint32_t ab = DT<T>::absbits(load_scalar<T>(x + i));
    m = m > ab ? m : ab;
  }
  m = block_max(m, sh);
  if (threadIdx.x == 0) atomicMax(out, m);
}

// ---------------------------------------------------------------------------------
// Per-tensor quantizer (BASELINE.json config 1; search/baseline/plot_weight_distribution_for_motivation.py:285-294):
//     scale = x.abs().max() / max|table|      both 0-dim -> float32 whatever x's dtype
//     out   = table[argmin |T(x / scale) - table|] * scale        float32
// Two launches, no memset, no atomics: (1) every workgroup writes the maximum of its slice to `partials`,
// (2) every workgroup of the elementwise launch reduces the <= kMaxBlocks partials (L2-resident) to the same
// scale and quantizes its tile with the argmin rules of quant_sym; workgroup 0 also stores the scale.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void absmax_partials_kernel(const T* __restrict__ x, int64_t n,
                                                                uint32_t* __restrict__ partials) {
  __shared__ uint32_t sh[kBlock / 64];
  constexpr int V = DT<T>::kVec;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const bool aligned = ((uintptr_t)x & 15) == 0;
  const int64_t nv = aligned ? n / V : 0;
  const u32x4* xv = (const u32x4*)x;
  uint32_t m = 0;
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += stride) {
    const u32x4 r = xv[v];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const uint32_t ab = DT<T>::absbits(DT<T>::get(r, i));
      m = m > ab ? m : ab;
    }
  }
  for (int64_t i = nv * V + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const uint32_t ab = DT<T>::absbits(load_scalar<T>(x + i));
    m = m > ab ? m : ab;
  }
  m = block_max(m, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = m;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void tensor_argmin_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t n,
                                                              const uint32_t* __restrict__ partials, int n_partials,
                                                              float* __restrict__ scale_out, Fmt f) {
  __shared__ uint32_t sh[kBlock / 64];
  constexpr int V = DT<T>::kVec;
  uint32_t m = 0;
  for (int i = threadIdx.x; i < n_partials; i += kBlock) {
    const uint32_t p = partials[i];
    m = m > p ? m : p;
  }
  m = block_max(m, sh);
  const float s = DT<T>::from_absbits(m) / f.gmax;   // float32 division of two 0-dim tensors
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const bool aligned = (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
  const int64_t nv = aligned ? n / V : 0;
  const u32x4* xv = (const u32x4*)x;
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += stride) {
    const u32x4 r = __builtin_nontemporal_load(xv + v);
    float p[V];
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = quant_sym<T>(DT<T>::get(r, i), s, f);
#pragma unroll
    for (int i = 0; i < V; i += 4)
      __builtin_nontemporal_store(u32x4{fbits(p[i]), fbits(p[i + 1]), fbits(p[i + 2]), fbits(p[i + 3])},
                                  (u32x4*)(out + v * V + i));
  }
  for (int64_t i = nv * V + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    out[i] = quant_sym<T>(load_scalar<T>(x + i), s, f);
}

// ---------------------------------------------------------------------------------
// Codewords: one workgroup per row (any cols); code = index in the sorted
// de-duplicated symmetric table.
// ---------------------------------------------------------------------------------
template <typename Tin>
__global__ __launch_bounds__(kBlock) void rows_codes_kernel(const Tin* __restrict__ x,
                                                           uint8_t* __restrict__ codes,
                                                           Tin* __restrict__ scales, int64_t rows,
                                                           int64_t cols, Fmt fs, int pack) {
  __shared__ uint32_t sh[kBlock / 64];
  const int64_t code_cols = pack ? (cols + 1) / 2 : cols;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const Tin* xr = x + row * cols;
    uint32_t m = 0;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      uint32_t ab = DT<Tin>::absbits(load_scalar<Tin>(xr + c));
      m = m > ab ? m : ab;
    }
    m = block_max(m, sh);
    float s = scale_of<Tin>(m, fs.gmax);
    if (threadIdx.x == 0) store_scalar<Tin>(scales + row, s);
    auto code_of = [&](int64_t c) -> uint32_t {
      if (c >= cols) return (uint32_t)fs.zero_code;
      float xn = DT<Tin>::round(load_scalar<Tin>(xr + c) / s);
      uint32_t neg = (xn < 0.0f) ? 1u : 0u;
      float qm = quant_mag(fabsf(xn), neg, fs);
      int li = level_index(qm, fs);
      return (uint32_t)(neg ? fs.zero_code - li : fs.zero_code + li);
    };
    if (pack) {
      for (int64_t b = threadIdx.x; b < code_cols; b += kBlock)
        codes[row * code_cols + b] = (uint8_t)(code_of(2 * b) | (code_of(2 * b + 1) << 4));
    } else {
      for (int64_t c = threadIdx.x; c < cols; c += kBlock) codes[row * code_cols + c] = (uint8_t)code_of(c);
    }
  }
}

// Vectorised codes for rows of exactly 128 elements (per-group): LPR lanes own a row, 16-byte
// loads, one packed store per lane (FP4: V nibbles, FP6: V bytes), lane 0 of the row writes the scale.
template <typename Tin, bool PACK, bool HW = false>
__device__ __forceinline__ void codes128_body(const u32x4* __restrict__ x, uint8_t* __restrict__ codes,
                                              Tin* __restrict__ scales, int64_t n_vec, const Fmt& fs) {
  constexpr int V = DT<Tin>::kVec;
  constexpr int LPR = 128 / V;
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * kBlock) {
    u32x4 raw = __builtin_nontemporal_load(x + v);
    float xf[V];
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      xf[i] = DT<Tin>::get(raw, i);
      uint32_t ab = DT<Tin>::absbits(xf[i]);
      m = m > ab ? m : ab;
    }
    m = lanes_max<LPR>(m);
    float s = scale_of<Tin>(m, fs.gmax);
    if ((threadIdx.x & (LPR - 1)) == 0) store_scalar<Tin>(scales + v / LPR, s);
    uint32_t c[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float xn = DT<Tin>::round(xf[i] / s);
      uint32_t neg = (xn < 0.0f) ? 1u : 0u;
      float qm = quant_mag(fabsf(xn), neg, fs);
      int li = level_index(qm, fs);
      if constexpr (HW) c[i] = (uint32_t)li | ((neg && li != 0) ? 8u : 0u);   // OCP sign-magnitude nibble
      else c[i] = (uint32_t)(neg ? fs.zero_code - li : fs.zero_code + li);
    }
    if constexpr (PACK && V == 8) {
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) w |= c[i] << (4 * i);
      ((uint32_t*)codes)[v] = w;
    } else if constexpr (PACK && V == 4) {
      ((uint16_t*)codes)[v] = (uint16_t)(c[0] | (c[1] << 4) | (c[2] << 8) | (c[3] << 12));
    } else if constexpr (V == 8) {
      u32x2 w = {c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24), c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24)};
      ((u32x2*)codes)[v] = w;
    } else {
      ((uint32_t*)codes)[v] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
    }
  }
}
template <typename Tin, bool PACK, bool HW = false>
__global__ __launch_bounds__(kBlock) void codes128_kernel(const u32x4* __restrict__ x, uint8_t* __restrict__ codes,
                                                         Tin* __restrict__ scales, int64_t n_vec, Fmt fs) {
  codes128_body<Tin, PACK, HW>(x, codes, scales, n_vec, fs);
}
// Many tensors, one launch (fpq_quant_rows_codes_segments): blockIdx.y = segment of a device-resident table, the
// workgroups of the x dimension stride over that segment's vectors (a short segment's surplus workgroups fall through)
struct CodesSeg {
  const void* x;
  uint8_t* codes;
  void* scales;
  int64_t rows;
};
template <typename Tin, bool PACK>
__global__ __launch_bounds__(kBlock) void codes128_segments_kernel(const CodesSeg* __restrict__ segs, Fmt fs) {
  const CodesSeg sg = segs[blockIdx.y];
  codes128_body<Tin, PACK, false>((const u32x4*)sg.x, sg.codes, (Tin*)sg.scales, sg.rows * (128 / DT<Tin>::kVec), fs);
}

// inverse for rows of 128: every lane decodes 8 consecutive elements
template <typename Ts, typename Tout, bool PACK>
__device__ __forceinline__ void decode128_body(const uint8_t* __restrict__ codes, const Ts* __restrict__ scales,
                                               Tout* __restrict__ out, int64_t n_oct, const Fmt& fs) {
  const int nsub = (int)(fs.kmin * fs.inv_step0);
  if constexpr (PACK) {
    // nibble codes (round 4): the two signed levels of every code BYTE from a 256-entry table in LDS, filled once per
    // workgroup from the closed form below - shift, mask, one 8-byte LDS read and two multiplies per pair instead of ~24
    // vector instructions (the decode side of the calibration's packed exchange ran at 0.45 of 8 TB/s on them)
    __shared__ float pair_lut[256][2];
    {
      const int t = threadIdx.x;   // kBlock == 256: one entry per thread
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int li = (int)((t >> (4 * h)) & 15) - fs.zero_code;
        const uint32_t neg = li < 0;
        li = neg ? -li : li;
        const float q = (li < nsub) ? (float)li * fs.step0 : u2f(((uint32_t)li + fs.kmin_code_base) << fs.mshift);
        pair_lut[t][h] = neg ? -q : q;
      }
    }
    __syncthreads();
    // tiles of U x 256 octets: all of a tile's code words are requested before the first is decoded (one octet per
    // thread and trip, as before, cycled workgroups that write 4 KiB each - the rate of the headline quantizer's
    // workgroups, which move twice that)
    constexpr int U = 4;
    const int64_t n_tiles = (n_oct + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      const int64_t v0 = tile * ((int64_t)kBlock * U) + threadIdx.x;
      uint32_t w[U];
      float s[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t v = v0 + u * kBlock;
        w[u] = v < n_oct ? __builtin_nontemporal_load((const uint32_t*)codes + v) : 0u;
        s[u] = v < n_oct ? load_scalar<Ts>(scales + (v >> 4)) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t v = v0 + u * kBlock;
        float p[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t off = j == 0 ? (w[u] << 3) & 0x7F8u : (w[u] >> (8 * j - 3)) & 0x7F8u;   // byte j of the word, times 8
          const v2f_t q = *(const v2f_t*)((const char*)pair_lut + off);
          p[2 * j] = q[0] * s[u];
          p[2 * j + 1] = q[1] * s[u];
        }
        if (v < n_oct) {
          if constexpr (sizeof(Tout) == 2) {
            __builtin_nontemporal_store(u32x4{f2h2(p[0], p[1]), f2h2(p[2], p[3]), f2h2(p[4], p[5]), f2h2(p[6], p[7])}, (u32x4*)out + v);
          } else {
            __builtin_nontemporal_store(u32x4{fbits(p[0]), fbits(p[1]), fbits(p[2]), fbits(p[3])}, (u32x4*)out + 2 * v);
            __builtin_nontemporal_store(u32x4{fbits(p[4]), fbits(p[5]), fbits(p[6]), fbits(p[7])}, (u32x4*)out + 2 * v + 1);
          }
        }
      }
    }
    return;
  }
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n_oct; v += (int64_t)gridDim.x * kBlock) {
    uint32_t c[8];
    if constexpr (PACK) {
      uint32_t w = ((const uint32_t*)codes)[v];
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = (w >> (4 * i)) & 0xFu;
    } else {
      u32x2 w = ((const u32x2*)codes)[v];
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
    }
    float s = load_scalar<Ts>(scales + (v >> 4));
    float p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int li = (int)c[i] - fs.zero_code;
      uint32_t neg = li < 0;
      li = neg ? -li : li;
      float q = (li < nsub) ? (float)li * fs.step0 : u2f(((uint32_t)li + fs.kmin_code_base) << fs.mshift);
      p[i] = (neg ? -q : q) * s;
    }
    if constexpr (sizeof(Tout) == 2) {
      u32x4 o = {f2h(p[0]) | (f2h(p[1]) << 16), f2h(p[2]) | (f2h(p[3]) << 16), f2h(p[4]) | (f2h(p[5]) << 16),
                 f2h(p[6]) | (f2h(p[7]) << 16)};
      __builtin_nontemporal_store(o, (u32x4*)out + v);
    } else {
      __builtin_nontemporal_store(u32x4{fbits(p[0]), fbits(p[1]), fbits(p[2]), fbits(p[3])}, (u32x4*)out + 2 * v);
      __builtin_nontemporal_store(u32x4{fbits(p[4]), fbits(p[5]), fbits(p[6]), fbits(p[7])}, (u32x4*)out + 2 * v + 1);
    }
  }
}
template <typename Ts, typename Tout, bool PACK>
__global__ __launch_bounds__(kBlock) void decode128_kernel(const uint8_t* __restrict__ codes, const Ts* __restrict__ scales,
                                                          Tout* __restrict__ out, int64_t n_oct, Fmt fs) {
  decode128_body<Ts, Tout, PACK>(codes, scales, out, n_oct, fs);
}
struct DecodeSeg {
  const uint8_t* codes;
  const void* scales;
  void* out;
  int64_t rows;
};
template <typename Ts, typename Tout, bool PACK>
__global__ __launch_bounds__(kBlock) void decode128_segments_kernel(const DecodeSeg* __restrict__ segs, Fmt fs) {
  const DecodeSeg sg = segs[blockIdx.y];
  decode128_body<Ts, Tout, PACK>(sg.codes, (const Ts*)sg.scales, (Tout*)sg.out, sg.rows * 16, fs);
}

template <typename Ts, typename Tout>
__global__ __launch_bounds__(kBlock) void rows_decode_kernel(const uint8_t* __restrict__ codes,
                                                            const Ts* __restrict__ scales,
                                                            Tout* __restrict__ out, int64_t rows, int64_t cols,
                                                            Fmt fs, int pack) {
  const int64_t code_cols = pack ? (cols + 1) / 2 : cols;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    float s = load_scalar<Ts>(scales + row);
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      uint32_t code = pack ? ((codes[row * code_cols + (c >> 1)] >> ((c & 1) * 4)) & 0xF)
                           : codes[row * code_cols + c];
      int li = (int)code - fs.zero_code;
      uint32_t neg = li < 0;
      li = neg ? -li : li;
      // level li: below 2^M levels are li*step0; above, mantissa/exponent from the index
      int nsub = (int)(fs.kmin * fs.inv_step0);  // 2^M
      float q = (li < nsub) ? (float)li * fs.step0 : u2f(((uint32_t)li + fs.kmin_code_base) << fs.mshift);
      q = neg ? -q : q;
      store_scalar<Tout>(out + row * cols + c, q * s);
    }
  }
}


template <typename Tin, typename Tout, bool DUAL>
int launch_rows(const void* x, void* out, int64_t rows, int64_t cols, const Fmt& fs, const DualArgs& dual,
                hipStream_t st) {
  constexpr int V = DT<Tin>::kVec;
  const bool aligned = (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
  const int64_t row_bytes = cols * (int64_t)sizeof(Tin);
  if (aligned && cols % V == 0) {
    const int64_t n_vec = rows * (cols / V);
    const int lpr = (int)(cols / V);
    constexpr int U = 2;
    auto go = [&](auto kern) {
      int64_t blocks = (n_vec + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
      hipLaunchKernelGGL(kern, dim3(grid_for(blocks, 1 << 20)), dim3(kBlock), 0, st, (const u32x4*)x, out, n_vec,
                         fs, dual);
      return check_launch();
    };
    if (row_bytes <= 1024 && (lpr & (lpr - 1)) == 0) {
      switch (lpr) {
        case 1: return go(rows_subwave_kernel<Tin, Tout, 1, DUAL, U>);
        case 2: return go(rows_subwave_kernel<Tin, Tout, 2, DUAL, U>);
        case 4: return go(rows_subwave_kernel<Tin, Tout, 4, DUAL, U>);
        case 8: return go(rows_subwave_kernel<Tin, Tout, 8, DUAL, U>);
        case 16: return go(rows_subwave_kernel<Tin, Tout, 16, DUAL, U>);
        case 32: return go(rows_subwave_kernel<Tin, Tout, 32, DUAL, U>);
        case 64: return go(rows_subwave_kernel<Tin, Tout, 64, DUAL, U>);
      }
    }
    // the output row must stay 16-byte (f16 out of f32 in: 8-byte) aligned per vector
    const int64_t vec_per_row = cols / V;
    const int g = grid_for(rows, 65535);
    if (vec_per_row <= (int64_t)kBlock * 2)
      hipLaunchKernelGGL((rows_block_kernel<Tin, Tout, DUAL, 2>), dim3(g), dim3(kBlock), 0, st, (const Tin*)x,
                         (Tout*)out, rows, cols, fs, dual);
    else
      hipLaunchKernelGGL((rows_block_kernel<Tin, Tout, DUAL, 8>), dim3(g), dim3(kBlock), 0, st, (const Tin*)x,
                         (Tout*)out, rows, cols, fs, dual);
    return check_launch();
  }
  hipLaunchKernelGGL((rows_scalar_kernel<Tin, Tout, DUAL>), dim3(grid_for(rows, 65535)), dim3(kBlock), 0, st,
                     (const Tin*)x, (Tout*)out, rows, cols, fs, dual);
  return check_launch();
}

template <bool DUAL>
int dispatch_rows(const void* x, void* out, int64_t rows, int64_t cols, int in_dtype, int out_dtype, const Fmt& fs,
                  const DualArgs& dual, hipStream_t st) {
  if (in_dtype == FPQ_F16 && out_dtype == FPQ_F16)
    return launch_rows<_Float16, _Float16, DUAL>(x, out, rows, cols, fs, dual, st);
  if (in_dtype == FPQ_F32 && out_dtype == FPQ_F32)
    return launch_rows<float, float, DUAL>(x, out, rows, cols, fs, dual, st);
  if (in_dtype == FPQ_F32 && out_dtype == FPQ_F16)
    return launch_rows<float, _Float16, DUAL>(x, out, rows, cols, fs, dual, st);
  if (in_dtype == FPQ_F16 && out_dtype == FPQ_F32)
    return launch_rows<_Float16, float, DUAL>(x, out, rows, cols, fs, dual, st);
  return FPQ_ERR_DTYPE;
}

// ---- fast fp16 -> fp16 path (fpq_fast16.h) ------------------------------------------
inline int table_shift16(int id) {   // rounding thresholds are multiples of 2^shift in fp16 patterns
  return id == FPQ_INT_NEG ? 5 : 9 - kTables[id].mbits;
}

inline bool fast16_aligned(const void* x, const void* out, int64_t cols, int in_dtype, int out_dtype) {
  return in_dtype == FPQ_F16 && out_dtype == FPQ_F16 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0 &&
         cols % 8 == 0;
}

// rows that live inside one wavefront
inline bool fast16_eligible(const void* x, const void* out, int64_t cols, int in_dtype, int out_dtype) {
  if (!fast16_aligned(x, out, cols, in_dtype, out_dtype)) return false;
  const int64_t lpr = cols / 8;
  return lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0;
}

// long rows: one workgroup per row, at most 8 vectors (64 halves) per lane in registers
inline bool fast16_block_eligible(const void* x, const void* out, int64_t cols, int in_dtype, int out_dtype) {
  return fast16_aligned(x, out, cols, in_dtype, out_dtype) && cols / 8 <= (int64_t)kBlock * 8;
}

// host cache of prebuilt tables: built once per (neg, pos) pair, immutable afterwards
struct Lut16Host {
  Lut16Args args;
  Lut16Tab tab;                    // the compressed image that travels in the kernel arguments
  bool tab_valid;                  // false: kernels evaluate the closed form themselves (lut16_fill)
  uint16_t full[kLutLdsEntries];   // the full image (host side: derived code tables are built from it)
};

inline const Lut16Host& lut16_host(int neg_id, int pos_id) {
  static const auto* cache = [] {
    auto* c = new Lut16Host[FPQ_NUM_TABLES * FPQ_NUM_TABLES];
    for (int n = 0; n < FPQ_NUM_TABLES; ++n)
      for (int p = 0; p < FPQ_NUM_TABLES; ++p) {
        Lut16Host& h = c[n * FPQ_NUM_TABLES + p];
        h.args.fneg = make_fmt(n);
        h.args.fpos = make_fmt(p);
        h.args.inv_gneg = 1.0f / h.args.fneg.gmax;
        h.args.inv_gpos = 1.0f / h.args.fpos.gmax;
        h.args.shift = table_shift16(n) < table_shift16(p) ? table_shift16(n) : table_shift16(p);
        h.args.nan_flag = nullptr;
        h.args.clip_absmax = nullptr;
        h.args.clip_strength = 1.0f;
        h.args.gelu_out = nullptr;
        for (int i = 0; i < kLutLdsEntries; ++i) h.full[i] = 0;
        lut16_build_host(h.full, h.args);
        h.tab_valid = lut16_compress(h.full, h.args.shift, &h.tab);
      }
    return c;
  }();
  return cache[neg_id * FPQ_NUM_TABLES + pos_id];
}

// Defaults measured on MI355X with tools/kbench (cold HBM, 4 rotating 252 MB buffer pairs):
// U = 2 with non-temporal loads and stores matches the best plain copy (78 us for
// fp16 [65536x1920] = 6.45 TB/s); grid-stride with a capped grid or U = 8 lose 3-8 %.
#ifndef FPQ_FAST16_U   // vectors per lane of the sub-wavefront row kernels (A/B builds)
#define FPQ_FAST16_U 2
#endif
#ifndef FPQ_FAST16_HW4_U   // ... of the table-free E2M1 form
#define FPQ_FAST16_HW4_U 1
#endif
template <bool DUAL, int U = FPQ_FAST16_U, bool NTL = true, bool NTS = true>
int launch_fast16(const void* x, void* out, int64_t rows, int64_t cols, int neg_id, int pos_id, hipStream_t st,
                  int grid_cap = 1 << 20, uint32_t* nan_flag = nullptr, const void* clip_absmax = nullptr,
                  float clip_strength = 1.0f, bool gelu = false, void* gelu_out = nullptr) {
  const Lut16Host& h = lut16_host(neg_id, pos_id);
  Lut16Args args = h.args;
  args.nan_flag = nan_flag;
  args.clip_absmax = clip_absmax;
  args.clip_strength = clip_strength;
  args.gelu_out = gelu_out;
  const int64_t n_vec = rows * (cols / 8);
  const int lpr = (int)(cols / 8);
  const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
  const int64_t tiles = (n_vec + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
  auto go = [&](auto kern_tab, auto kern_fill) {
    if (h.tab_valid)
      hipLaunchKernelGGL(kern_tab, dim3(grid_for(tiles, grid_cap)), dim3(kBlock), lds, st, (const u32x4*)x,
                         (u32x4*)out, n_vec, args, h.tab);
    else
      hipLaunchKernelGGL(kern_fill, dim3(grid_for(tiles, grid_cap)), dim3(kBlock), lds, st, (const u32x4*)x,
                         (u32x4*)out, n_vec, args, h.tab);
    return check_launch();
  };
  // the headline shape - E2M1, groups of 128 - takes its levels from the FP4 conversion hardware (fpq_fast16.h); FPQ_NO_HW4
  // (read at every call: the exhaustive test sweeps both forms in one process) keeps the bucket table
  if constexpr (DUAL) {
    if (gelu) {          // groups of 128, small tables (fpq_gelu_quant_rows_dual checks): GELU in front of the quantizer, one pass
      if (lpr != 16 || clip_absmax) return FPQ_ERR_SHAPE;
      return go(rows16_lut_subwave_kernel<16, true, U, true, NTL, NTS, false, false, 0, true>,
                rows16_lut_subwave_kernel<16, true, U, false, NTL, NTS, false, false, 0, true>);
    }
    if (clip_absmax) {   // groups of 128 only (fpq_quant_rows_dual checks): the clamping form of the same kernel
      if (lpr != 16) return FPQ_ERR_SHAPE;
      return go(rows16_lut_subwave_kernel<16, true, U, true, NTL, NTS, false, true>, rows16_lut_subwave_kernel<16, true, U, false, NTL, NTS, false, true>);
    }
  }
  if constexpr (!DUAL) {
    if (lpr == 16 && neg_id == FPQ_E2M1 && pos_id == FPQ_E2M1 && !fpq_flag(OPT_FPQ_NO_HW4)) {
      // no table to stage, so nothing to amortise over a tile: ONE vector per lane on the full grid, the best plain-copy
      // shape of this chip (profiles/r02_copy_persistent_probe.txt: 0.81 against 0.777 for 8 KiB tiles); same-process
      // A/B against U = 2: 77.0 - 77.5 vs 79.0 - 79.4 us in steady state (profiles/r03_headline_u1.txt)
      constexpr int U1 = FPQ_FAST16_HW4_U;
      const int64_t tiles1 = (n_vec + (int64_t)kBlock * U1 - 1) / ((int64_t)kBlock * U1);
      hipLaunchKernelGGL((rows16_lut_subwave_kernel<16, false, U1, true, NTL, NTS, true>), dim3(grid_for(tiles1, grid_cap)),
                         dim3(kBlock), lds, st, (const u32x4*)x, (u32x4*)out, n_vec, args, h.tab);
      return check_launch();
    }
  }
  if constexpr (!DUAL) {
    // E2M3 / E3M2 per group of 128 and per row of 64 (the KV cache's head rows): levels from the FP6 conversion hardware, four
    // vectors per lane = the 32 values of one conversion, full grid (no table to amortise); FPQ_NO_HW6 keeps the table
    if ((lpr == 16 || lpr == 8) && neg_id == pos_id && (neg_id == FPQ_E2M3 || neg_id == FPQ_E3M2) && !fpq_flag(OPT_FPQ_NO_HW6)) {
      const int64_t tiles4 = (n_vec + (int64_t)kBlock * 4 - 1) / ((int64_t)kBlock * 4);
      const dim3 g4(grid_for(tiles4, 1 << 20));
#define FPQ_HW6_GO(L, H)                                                                                                         \
  hipLaunchKernelGGL((rows16_lut_subwave_kernel<L, false, 4, true, NTL, NTS, false, false, H>), g4, dim3(kBlock), lds, st,      \
                     (const u32x4*)x, (u32x4*)out, n_vec, args, h.tab)
      if (lpr == 16) { if (neg_id == FPQ_E2M3) FPQ_HW6_GO(16, 1); else FPQ_HW6_GO(16, 2); }
      else { if (neg_id == FPQ_E2M3) FPQ_HW6_GO(8, 1); else FPQ_HW6_GO(8, 2); }
#undef FPQ_HW6_GO
      return check_launch();
    }
  }
#define FPQ_FAST16_CASE(L) \
  case L: return go(rows16_lut_subwave_kernel<L, DUAL, U, true, NTL, NTS>, rows16_lut_subwave_kernel<L, DUAL, U, false, NTL, NTS>);
  switch (lpr) {
    FPQ_FAST16_CASE(1) FPQ_FAST16_CASE(2) FPQ_FAST16_CASE(4) FPQ_FAST16_CASE(8)
    FPQ_FAST16_CASE(16) FPQ_FAST16_CASE(32) FPQ_FAST16_CASE(64)
  }
#undef FPQ_FAST16_CASE
  return FPQ_ERR_SHAPE;
}

inline const Lut16Tab& lut16_mx_codes_e2m1();

// k-major images are addressed with 32-bit byte offsets (and int buffer ranges) by their producers: the whole image must stay below 2 GiB
static bool km_image_fits(int64_t rows, int64_t row_bytes) { return rows < (1ll << 31) && rows * row_bytes < (1ll << 31); }

template <typename Tin>
int launch_rotate_quant(const void* x, void* out, void* rot_out, int64_t rows, int64_t cols, const float* smooth,
                        const uint32_t sign[4], int table_id, hipStream_t st, uint16_t* code_scales = nullptr,
                        bool km = false /* FP4 codes into a k-major image (include/fpq.h): the matrix-core form only */) {
  const Lut16Host& h = lut16_host(table_id, table_id);
  if (!h.tab_valid) return FPQ_ERR_TABLE;
  const Lut16Tab& tab = code_scales ? lut16_mx_codes_e2m1() : h.tab;
  RotArgs r;
  r.code_scales = code_scales;
  r.code_bits = 8;
  r.km_rows = km ? (uint32_t)rows : 0u;
  r.km_gpr = fast_div((uint32_t)(cols / 128));
  if (km && (!code_scales || !km_image_fits(rows, cols / 2))) return FPQ_ERR_SHAPE;
  r.smooth = smooth;
  for (int i = 0; i < 4; ++i) r.sign[i] = sign[i];
  r.c_h = h2f(f2h(1.0f / __builtin_sqrtf(128.0f)));   // torch.tensor(128).sqrt() is float32; autocast makes Q fp16
  r.vec_per_row = cols / 8;
  const int64_t n_vec = rows * (cols / 8);
  #ifndef FPQ_ROT_U
#define FPQ_ROT_U 2
#endif
  constexpr int U = FPQ_ROT_U;
  const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
  const int64_t tiles = (n_vec + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
  const dim3 grid(grid_for(tiles, 1 << 20));
  // values out: the transform on the matrix cores (fpq_rotate_mfma.h), one 32-group tile per wavefront
  const bool butterfly = FPQ_ROT_BUTTERFLY_BUILD || fpq_flag(OPT_FPQ_ROT_BUTTERFLY);
  if (!butterfly) {
    // Every workgroup the same number of passes over its tiles.  With a bucket table to stage per workgroup the grid is
    // two generations of the FPQ_ROT_WAVES workgroups a CU holds (3072: 84.2 us against 85.9 for one generation, round 2).
    // The table-free E2M1 forms have next to no prologue and want SHORT workgroups - the grid drains faster at its end:
    // values out, 3072 / 7680 / 12288 / 16384 workgroups: 83.2 / 82.2 / 82.1 / 80.3 us (one pass each at [65536 x 1920]);
    // codes out: 57.3 / 52.8 / 53.0 / 53.6 us (profiles/r03_rotate_grid.txt).  FPQ_ROT_WGS overrides.
    const bool hw4 = table_id == FPQ_E2M1 && !fpq_flag(OPT_FPQ_NO_HW4);   // E2M1 values or FP4 operands: levels / codes from the conversion hardware
    const int64_t per_wg = (int64_t)(kBlock / 64) * kRqTileVec;
    const int64_t wg_tiles = (n_vec + per_wg - 1) / per_wg;
    const int64_t resident_env = fpq_opt(OPT_FPQ_ROT_WGS, 0);
    // (with a smoothing vector every workgroup stages it - 7.5 KiB at C = 1920 - so a few passes each: 7680 / 2560)
    const int64_t resident = resident_env > 0 ? resident_env : !hw4 ? 2 * 256ll * FPQ_ROT_WAVES
                             : smooth ? (code_scales ? 2560 : 7680) : code_scales ? 8192 : 16384;
    const int64_t passes = (wg_tiles + resident - 1) / resident;
    const dim3 mgrid((unsigned)((wg_tiles + passes - 1) / passes));
#define FPQ_ROT_MFMA(EMIT, SMOOTH, ...)                                                                             \
  hipLaunchKernelGGL((rotate_quant_mfma_kernel<Tin, EMIT, SMOOTH, ##__VA_ARGS__>), mgrid, dim3(kBlock), lds, st, x, \
                     (u32x4*)out, (u32x4*)rot_out, n_vec, r, h.args, tab)
    if (code_scales && hw4) { if (smooth) FPQ_ROT_MFMA(false, true, true, true); else FPQ_ROT_MFMA(false, false, true, true); }
    else if (code_scales) { if (smooth) FPQ_ROT_MFMA(false, true, true); else FPQ_ROT_MFMA(false, false, true); }
    else if (rot_out && hw4) { if (smooth) FPQ_ROT_MFMA(true, true, false, true); else FPQ_ROT_MFMA(true, false, false, true); }
    else if (rot_out) { if (smooth) FPQ_ROT_MFMA(true, true); else FPQ_ROT_MFMA(true, false); }
    else if (hw4) { if (smooth) FPQ_ROT_MFMA(false, true, false, true); else FPQ_ROT_MFMA(false, false, false, true); }
    else { if (smooth) FPQ_ROT_MFMA(false, true); else FPQ_ROT_MFMA(false, false); }
#undef FPQ_ROT_MFMA
    return check_launch();
  }
  if (km) return FPQ_ERR_SHAPE;   // the butterfly forms (a build / experiment switch) write row-major codes only
  if (code_scales)
    hipLaunchKernelGGL((rotate_quant16_kernel<Tin, false, U, true>), grid, dim3(kBlock), lds, st, x, (u32x4*)out,
                       (u32x4*)nullptr, n_vec, r, h.args, tab);
  else if (rot_out)
    hipLaunchKernelGGL((rotate_quant16_kernel<Tin, true, U>), grid, dim3(kBlock), lds, st, x, (u32x4*)out,
                       (u32x4*)rot_out, n_vec, r, h.args, tab);
  else
    hipLaunchKernelGGL((rotate_quant16_kernel<Tin, false, U>), grid, dim3(kBlock), lds, st, x, (u32x4*)out,
                       (u32x4*)nullptr, n_vec, r, h.args, tab);
  return check_launch();
}

template <typename Tin, typename Tmod>
int launch_adaln_rotate_quant(const void* x, void* out, void* h_out, void* y_out, int64_t rows, int64_t cols,
                              const AdaLnArgs& ad, const float* smooth, const uint32_t sign[4], int table_id,
                              hipStream_t st, int lanes_per_row, uint16_t* code_scales = nullptr,
                              int token_mode = 0 /*1: per-token values, 2: per-token E4M3 codes, 3: per-token packed 6-bit codes*/,
                              const Lut16Tab* token_code_tab = nullptr,
                              bool km = false /* FP4 / 6-bit codes into a k-major image (include/fpq.h): adaln_mfma_kernel only */) {
  const Lut16Host& h = lut16_host(table_id, table_id);
  if (!h.tab_valid) return FPQ_ERR_TABLE;
  const Lut16Tab& tab = token_mode >= 2 ? *token_code_tab : (code_scales && !token_mode ? lut16_mx_codes_e2m1() : h.tab);
  RotArgs r;
  r.code_scales = code_scales;
  r.code_bits = token_mode == 3 ? 6 : 8;
  r.km_rows = km ? (uint32_t)rows : 0u;
  r.km_gpr = fast_div((uint32_t)(cols / 128));
  if (km) {
    const bool fp4_codes = code_scales && !token_mode;
    if (!(fp4_codes || token_mode == 3) || !km_image_fits(rows, token_mode == 3 ? cols / 4 * 3 : cols / 2)) return FPQ_ERR_SHAPE;
    // the forms that write row-major codes only: rows beyond one wavefront, the first-generation kernel, the butterfly build
    if (lanes_per_row != 64 || cols / 8 > 64 * 5 || fpq_flag(OPT_FPQ_ADALN_V1) || FPQ_ROT_BUTTERFLY_BUILD || fpq_flag(OPT_FPQ_ROT_BUTTERFLY))
      return FPQ_ERR_SHAPE;
  }
  r.smooth = smooth;
  for (int i = 0; i < 4; ++i) r.sign[i] = sign[i];
  r.c_h = h2f(f2h(1.0f / __builtin_sqrtf(128.0f)));
  r.vec_per_row = cols / 8;
  const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
  {
    // second generation (fpq_adaln.h): fp16 or fp32 rows of up to 2560 channels, one batch entry per workgroup
    constexpr bool X32 = sizeof(Tin) == 4;
    if (lanes_per_row == 64 && r.vec_per_row <= 64 * 5 && !fpq_flag(OPT_FPQ_ADALN_V1)) {
      if (h.args.shift < 6) return FPQ_ERR_TABLE;   // symmetric tables only (<= 2 x 512 buckets)
      const int64_t L = ad.rows_per_batch;
      const int64_t n_batches = (rows + L - 1) / L;
      const bool adaln_butterfly = FPQ_ROT_BUTTERFLY_BUILD || fpq_flag(OPT_FPQ_ROT_BUTTERFLY);
      const bool rows_env = fpq_opt_set(OPT_FPQ_ADALN_ROWS);
      // Large launches: chunks of 16 rows (4 per wavefront) amortise the staging of the modulation; small launches (the
      // early scale steps of a generation: 100 .. 3600 rows) are latency-bound and want every CU busy: one row per wavefront
      // (profiles/r02_small_steps.json; round 4, cold inputs, 4 / 8 / 12 / 16 rows per workgroup over the ten steps of d30 and
      // d36-512: 4 is the best or within 2 % of it up to 10 000 rows, 8 from 16 900 on - profiles/r04_adaln_rows_sweep.txt).
      // FPQ_ADALN_ROWS=n: n rows per workgroup everywhere; FPQ_ADALN_TAIL=rows: how many rows at the end of the grid go to
      // each of two finer tiers (8 and 4 rows per workgroup).  The tiers are OFF by default (0): they never beat a plain grid
      // of 8 - 12 rows (profiles/r03_adaln_partition.txt); the tier decode stays reachable through the variable and is
      // covered by tests/test_gpu_parity.py::test_adaln_tail_tiers_switch in a child process.
      // (third generation, large launches: 8 rows = two per wavefront for the stream-bound forms - E2M1 values out, fp32
      // rows; 12 for the forms bound by vector issue - operands out or a bucket table, from fp16 rows - where the
      // prologue's instructions per row count: 73.3 -> 70.5 us for codes, 89.7 -> 87.1 for E4M3 bytes, 96.2 -> 93.4 for
      // per-token E2M3 values; profiles/r03_adaln_partition.txt)
      const bool issue_bound = !X32 && (code_scales != nullptr || token_mode != 0 || table_id != FPQ_E2M1);
      int rows_per_wg = rows_env ? fpq_opt(OPT_FPQ_ADALN_ROWS, 0) : (rows >= 8192 ? (adaln_butterfly && rows >= 32768 ? 16 : issue_bound && rows >= 32768 ? 12 : 8) : 4);
      if (rows_per_wg < 1) rows_per_wg = 1;
      // rows of exactly 8 groups (C = 1024): two rows per tile (fpq_adaln.h, PAIR2) - workgroups of an even number of rows
      const bool pair2 = !adaln_butterfly && !X32 && r.vec_per_row == 128 && token_mode == 0 && !h_out && !y_out &&
                         !fpq_flag(OPT_FPQ_ADALN_NO_PAIR2);
      if (pair2) rows_per_wg = rows_env ? ((rows_per_wg + 1) & ~1) : (rows >= 8192 ? 16 : 8);
      const int64_t per_batch = (L + rows_per_wg - 1) / rows_per_wg;
      if (n_batches * per_batch > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
      AdalnTiers tiers = {};
      const int tail_rows = fpq_opt(OPT_FPQ_ADALN_TAIL, 0);
      int64_t nb2 = 0, nb1 = 0;
      if (!adaln_butterfly && tail_rows > 0 && rows_per_wg > 4) {
        nb2 = (tail_rows + L - 1) / L;                                   // batch entries cut into chunks of 4 rows
        if (rows_per_wg > 8) nb1 = (tail_rows + L - 1) / L;              // ... of 8 rows
        if (nb2 > n_batches) nb2 = n_batches;
        if (nb1 > n_batches - nb2) nb1 = n_batches - nb2;
      }
      tiers.rows[0] = rows_per_wg;
      tiers.rows[1] = 8;
      tiers.rows[2] = 4;
      for (int t = 0; t < 3; ++t) tiers.per_batch[t] = (int)((L + tiers.rows[t] - 1) / tiers.rows[t]);
      tiers.batches[0] = (int)(n_batches - nb1 - nb2);
      tiers.batches[1] = (int)nb1;
      {
        const int64_t nb_of[3] = {n_batches - nb1 - nb2, nb1, nb2};
        for (int t = 0; t < 3; ++t) {   // id / d == (id * ceil(2^32 / d)) >> 32 whenever id * d < 2^32
          const uint64_t d = (uint64_t)tiers.per_batch[t], ids = (uint64_t)nb_of[t] * d;
          tiers.magic[t] = (d >= 2 && ids * d < (1ull << 32)) ? (uint32_t)(((1ull << 32) + d - 1) / d) : 0u;
        }
      }
      const int64_t n_wg3 = (int64_t)tiers.batches[0] * tiers.per_batch[0] + nb1 * tiers.per_batch[1] + nb2 * tiers.per_batch[2];
      if (n_wg3 > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
      const dim3 g2((unsigned)(n_batches * per_batch));         // second generation (butterfly form): flat grid of equal chunks
      const dim3 g3((unsigned)n_wg3);
      const size_t lds2 = 0;   // table, modulation planes and images live in static LDS
      // E2M1 values per group: levels from the FP4 conversion hardware, no table (fpq_adaln.h)
      const bool hw4 = table_id == FPQ_E2M1 && !token_mode && !fpq_flag(OPT_FPQ_NO_HW4);
      const bool tight_ok = FPQ_ADALN_TIGHT && !fpq_flag(OPT_FPQ_ADALN_NO_TIGHT);
      // E2M3 / E3M2 values (per group, or per token: token_mode 1): levels from the FP6 conversion hardware, no table
      const int hw6 = (token_mode <= 1 && !code_scales && !fpq_flag(OPT_FPQ_NO_HW6)) ? (table_id == FPQ_E2M3 ? 1 : table_id == FPQ_E3M2 ? 2 : 0) : 0;
#define FPQ_ADALN3(M, CODES, EMIT, TOKEN, HW4, TIGHT)                                                                  \
  hipLaunchKernelGGL((adaln_mfma_kernel<Tmod, M, CODES, EMIT, TOKEN, X32, HW4, TIGHT>), g3, dim3(kBlock), lds2, st,    \
                     (const u32x4*)x, (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out, rows, ad, r, h.args, tab, tiers)
#define FPQ_ADALN2K(M, CODES, EMIT, TOKEN)                                                                             \
  do {                                                                                                                 \
    if (!adaln_butterfly) {                                                                                            \
      if constexpr ((M == 4 || M == 5) && !(CODES) && !(EMIT)) {   /* E2M3 / E3M2 values, rows of 13 .. 20 groups: hardware levels */ \
        if (hw6 == 1) {                                                                                                \
          hipLaunchKernelGGL((adaln_mfma_kernel<Tmod, M, false, false, TOKEN, X32, false, false, 4, false, 1>), g3,    \
                             dim3(kBlock), lds2, st, (const u32x4*)x, (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out,       \
                             rows, ad, r, h.args, tab, tiers);                                                         \
          break;                                                                                                       \
        }                                                                                                              \
        if (hw6 == 2) {                                                                                                \
          hipLaunchKernelGGL((adaln_mfma_kernel<Tmod, M, false, false, TOKEN, X32, false, false, 4, false, 2>), g3,    \
                             dim3(kBlock), lds2, st, (const u32x4*)x, (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out,       \
                             rows, ad, r, h.args, tab, tiers);                                                         \
          break;                                                                                                       \
        }                                                                                                              \
      }                                                                                                                \
      if constexpr (M == 2 && !X32 && !(EMIT) && !(TOKEN)) {                                                           \
        if (pair2) {                                                                                                   \
          if (hw4)                                                                                                     \
            hipLaunchKernelGGL((adaln_mfma_kernel<Tmod, 4, CODES, false, false, false, true, false, 4, true>), g3,     \
                               dim3(kBlock), lds2, st, (const u32x4*)x, (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out,     \
                               rows, ad, r, h.args, tab, tiers);                                                       \
          else                                                                                                         \
            hipLaunchKernelGGL((adaln_mfma_kernel<Tmod, 4, CODES, false, false, false, false, false, 4, true>), g3,    \
                               dim3(kBlock), lds2, st, (const u32x4*)x, (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out,     \
                               rows, ad, r, h.args, tab, tiers);                                                       \
          break;                                                                                                       \
        }                                                                                                              \
      }                                                                                                                \
      if constexpr (!(TOKEN)) {                                                                                        \
        if constexpr (M == 4 && !X32 && !(EMIT) && !(CODES)) {                                                                     \
          if (hw4 && tight_ok && r.vec_per_row == 240) {   /* VAR-d30: 31 KiB of LDS, five workgroups per CU */        \
            FPQ_ADALN3(M, CODES, EMIT, TOKEN, true, true);                                                             \
            break;                                                                                                     \
          }                                                                                                            \
        }                                                                                                              \
        if (hw4) {                                                                                                     \
          FPQ_ADALN3(M, CODES, EMIT, TOKEN, true, false);                                                              \
          break;                                                                                                       \
        }                                                                                                              \
      }                                                                                                                \
      FPQ_ADALN3(M, CODES, EMIT, TOKEN, false, false);                                                                 \
      break;                                                                                                           \
    }                                                                                                                  \
    hipLaunchKernelGGL((adaln_rq16_kernel<Tmod, M, CODES, EMIT, TOKEN, X32>), g2, dim3(kBlock), lds2, st,              \
                       (const u32x4*)x, (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out, rows, ad, r, h.args, tab,           \
                       rows_per_wg, (int)per_batch);                                                                   \
  } while (0)
#ifdef FPQ_ADALN_STAMPS
#define FPQ_ADALN_EMIT (h_out != nullptr)   /* diagnostic build: y_out alone is the stamp buffer */
#else
#define FPQ_ADALN_EMIT (h_out || y_out)
#endif
#define FPQ_ADALN2(M)                                                                                                  \
  do {                                                                                                                 \
    const bool emit = FPQ_ADALN_EMIT;                                                                                  \
    if (token_mode >= 2) FPQ_ADALN2K(M, true, false, true);                                                            \
    else if (token_mode == 1 && emit) FPQ_ADALN2K(M, false, true, true);                                               \
    else if (token_mode == 1) FPQ_ADALN2K(M, false, false, true);                                                      \
    else if (code_scales) FPQ_ADALN2K(M, true, false, false);                                                          \
    else if (emit) FPQ_ADALN2K(M, false, true, false);                                                                 \
    else FPQ_ADALN2K(M, false, false, false);                                                                          \
  } while (0)
      switch ((int)((r.vec_per_row + 63) / 64)) {   // MAXC = ceil(vectors per row / 64), exactly
        case 1: FPQ_ADALN2(1); break;
        case 2: FPQ_ADALN2(2); break;
        case 3: FPQ_ADALN2(3); break;
        case 4: FPQ_ADALN2(4); break;
        default: FPQ_ADALN2(5); break;
      }
#undef FPQ_ADALN3
#undef FPQ_ADALN2
#undef FPQ_ADALN2K
      return check_launch();
    }
  }
  const int64_t rows_per_wg = kBlock / lanes_per_row;
  int64_t g64 = (rows + rows_per_wg - 1) / rows_per_wg;
  const int64_t cap = fpq_opt(OPT_FPQ_ADALN_GRID, 8192);   // every workgroup stages the table once, then walks rows
  if (g64 > cap) g64 = cap;
  const dim3 g((unsigned)g64);
  const int maxc = (int)((r.vec_per_row + lanes_per_row - 1) / lanes_per_row);
#define FPQ_ADALN(L, M)                                                                                              \
  do {                                                                                                               \
    if (code_scales)                                                                                                 \
      hipLaunchKernelGGL((adaln_rotate_quant16_kernel<Tin, Tmod, L, M, true>), g, dim3(kBlock), lds, st, x,         \
                         (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out, rows, ad, r, h.args, tab);                       \
    else                                                                                                             \
      hipLaunchKernelGGL((adaln_rotate_quant16_kernel<Tin, Tmod, L, M>), g, dim3(kBlock), lds, st, x, (u32x4*)out,  \
                         (u32x4*)h_out, (u32x4*)y_out, rows, ad, r, h.args, tab);                                    \
  } while (0)
  if (token_mode) {   // one wavefront per row only (C <= 2560)
#define FPQ_ADALN_TOK(M)                                                                                             \
  do {                                                                                                               \
    if (token_mode >= 2)                                                                                             \
      hipLaunchKernelGGL((adaln_rotate_quant16_kernel<Tin, Tmod, 64, M, true, true>), g, dim3(kBlock), lds, st, x,  \
                         (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out, rows, ad, r, h.args, tab);                       \
    else                                                                                                             \
      hipLaunchKernelGGL((adaln_rotate_quant16_kernel<Tin, Tmod, 64, M, false, true>), g, dim3(kBlock), lds, st, x, \
                         (u32x4*)out, (u32x4*)h_out, (u32x4*)y_out, rows, ad, r, h.args, tab);                       \
  } while (0)
    if (lanes_per_row != 64 || maxc > 5) return FPQ_ERR_SHAPE;
    if (maxc <= 4) FPQ_ADALN_TOK(4);
    else FPQ_ADALN_TOK(5);
#undef FPQ_ADALN_TOK
    return check_launch();
  }
  if (lanes_per_row == 256) {
    if (maxc <= 1) FPQ_ADALN(256, 1);
    else FPQ_ADALN(256, 2);
  } else {
    if (maxc <= 4) FPQ_ADALN(64, 4);
    else if (maxc <= 5) FPQ_ADALN(64, 5);
    else FPQ_ADALN(64, 8);
  }
#undef FPQ_ADALN
  return check_launch();
}

template <bool DUAL>
int launch_fast16_pair8(const void* x, void* out, int64_t rows, int neg_id, int pos_id, hipStream_t st,
                        uint32_t* nan_flag = nullptr) {
  const Lut16Host& h = lut16_host(neg_id, pos_id);
  Lut16Args args = h.args;
  args.nan_flag = nan_flag;
  const int64_t n_vec = rows * 16;   // rows of 128 halves
  const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
  const int64_t tiles = (n_vec + (int64_t)kBlock * 2 - 1) / ((int64_t)kBlock * 2);
  if (h.tab_valid)
    hipLaunchKernelGGL((rows16_lut_pair_kernel<8, DUAL, true>), dim3(grid_for(tiles, 1 << 20)), dim3(kBlock), lds, st,
                       (const u32x4*)x, (u32x4*)out, n_vec, args, h.tab);
  else
    hipLaunchKernelGGL((rows16_lut_pair_kernel<8, DUAL, false>), dim3(grid_for(tiles, 1 << 20)), dim3(kBlock), lds, st,
                       (const u32x4*)x, (u32x4*)out, n_vec, args, h.tab);
  return check_launch();
}

template <bool DUAL>
int launch_fast16_block(const void* x, void* out, int64_t rows, int64_t cols, int neg_id, int pos_id,
                        hipStream_t st, uint32_t* nan_flag = nullptr, bool gelu = false, void* gelu_out = nullptr) {
  const Lut16Host& h = lut16_host(neg_id, pos_id);
  Lut16Args args = h.args;
  args.nan_flag = nan_flag;
  args.gelu_out = gelu_out;
  const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
  const int64_t vec_per_row = cols / 8;
  if (vec_per_row <= 64 * 5 && !fpq_flag(OPT_FPQ_NO_WAVE_ROWS) && !gelu) {
    // one wavefront per row: 4 rows per workgroup pass, enough workgroups to keep every CU busy while the
    // table staging stays amortised
    const int mc = (int)((vec_per_row + 63) / 64);
    int64_t g = (rows + 3) / 4;
    const int64_t capw = h.tab_valid ? 16384 : 2048;
    if (g > capw) g = capw;
#define FPQ_WAVE(M) do { if (h.tab_valid) hipLaunchKernelGGL((rows16_lut_wave_kernel<DUAL, M, true>), dim3((unsigned)g), dim3(kBlock), lds, st, (const uint16_t*)x, (uint16_t*)out, rows, cols, args, h.tab); \
                         else hipLaunchKernelGGL((rows16_lut_wave_kernel<DUAL, M, false>), dim3((unsigned)g), dim3(kBlock), lds, st, (const uint16_t*)x, (uint16_t*)out, rows, cols, args, h.tab); } while (0)
    // rows of four or five vectors per lane on E2M3 / E3M2 (per-token FP6 at C = 1920, 2048, 2304): levels from the FP6
    // conversion hardware, no table (fpq_fast16.h, fp6_levels_hw32); FPQ_NO_HW6 (read at every call) keeps the table
    if constexpr (!DUAL) {
      if ((mc == 4 || mc == 5) && neg_id == pos_id && (neg_id == FPQ_E2M3 || neg_id == FPQ_E3M2) && !fpq_flag(OPT_FPQ_NO_HW6)) {
        int64_t g6 = (rows + 3) / 4;
        if (g6 > (1 << 20)) g6 = 1 << 20;   // nothing to amortise: one pass of four rows per workgroup
#define FPQ_WAVE6(M, H) hipLaunchKernelGGL((rows16_lut_wave_kernel<false, M, true, H>), dim3((unsigned)g6), dim3(kBlock), lds, st, \
                                           (const uint16_t*)x, (uint16_t*)out, rows, cols, args, h.tab)
        if (mc == 4) { if (neg_id == FPQ_E2M3) FPQ_WAVE6(4, 1); else FPQ_WAVE6(4, 2); }
        else { if (neg_id == FPQ_E2M3) FPQ_WAVE6(5, 1); else FPQ_WAVE6(5, 2); }
#undef FPQ_WAVE6
        return check_launch();
      }
    }
    if (mc <= 1) FPQ_WAVE(1);
    else if (mc <= 2) FPQ_WAVE(2);
    else if (mc <= 4) FPQ_WAVE(4);
    else FPQ_WAVE(5);
#undef FPQ_WAVE
    return check_launch();
  }
  const int maxc = (int)((vec_per_row + kBlock - 1) / kBlock);
  // enough workgroups to fill the chip several times over, each walking consecutive rows
  // one row per workgroup, dispatched in address order, when the table arrives as a kernel argument (measured on
  // [65536 x 7680]: 0.785 of 8 TB/s vs 0.676 with four consecutive rows per workgroup); a table that every workgroup
  // has to evaluate itself is amortised over more rows
  const int64_t target_wgs = h.tab_valid ? (1 << 20) : 2048;
  int64_t rpb = (rows + target_wgs - 1) / target_wgs;
  if (rpb < 1) rpb = 1;
  if (h.tab_valid && (1 << (16 - h.args.shift)) >= 1024) {   // 2 x 512 (E2M3: [16384 x 7680] 86.3 -> 83.4 us) or 2 x 1024 buckets to stage: two rows per workgroup
    rpb = fpq_opt(OPT_FPQ_BIGTAB_RPB, 2);
    if (rpb < 1) rpb = 1;
  }
  const int64_t grid = (rows + rpb - 1) / rpb;
  auto go = [&](auto kern_tab, auto kern_fill) {
    if (h.tab_valid)
      hipLaunchKernelGGL(kern_tab, dim3((unsigned)grid), dim3(kBlock), lds, st, (const uint16_t*)x, (uint16_t*)out,
                         rows, cols, rpb, args, h.tab);
    else
      hipLaunchKernelGGL(kern_fill, dim3((unsigned)grid), dim3(kBlock), lds, st, (const uint16_t*)x,
                         (uint16_t*)out, rows, cols, rpb, args, h.tab);
    return check_launch();
  };
  if constexpr (DUAL) {
    if (gelu) {   // GELU in front of the quantizer (fpq_gelu_quant_rows_dual): one workgroup per row for every row length
      if (maxc <= 1) return go(rows16_lut_block_kernel<true, 1, true, true>, rows16_lut_block_kernel<true, 1, false, true>);
      if (maxc <= 2) return go(rows16_lut_block_kernel<true, 2, true, true>, rows16_lut_block_kernel<true, 2, false, true>);
      if (maxc <= 4) return go(rows16_lut_block_kernel<true, 4, true, true>, rows16_lut_block_kernel<true, 4, false, true>);
      if (maxc <= 5) return go(rows16_lut_block_kernel<true, 5, true, true>, rows16_lut_block_kernel<true, 5, false, true>);
      return go(rows16_lut_block_kernel<true, 8, true, true>, rows16_lut_block_kernel<true, 8, false, true>);
    }
  }
  if (maxc <= 1) return go(rows16_lut_block_kernel<DUAL, 1, true>, rows16_lut_block_kernel<DUAL, 1, false>);
  if (maxc <= 2) return go(rows16_lut_block_kernel<DUAL, 2, true>, rows16_lut_block_kernel<DUAL, 2, false>);
  if (maxc <= 4) return go(rows16_lut_block_kernel<DUAL, 4, true>, rows16_lut_block_kernel<DUAL, 4, false>);
  if (maxc <= 5) return go(rows16_lut_block_kernel<DUAL, 5, true>, rows16_lut_block_kernel<DUAL, 5, false>);
  return go(rows16_lut_block_kernel<DUAL, 8, true>, rows16_lut_block_kernel<DUAL, 8, false>);
}

// ---- fp32 rows of 128 (weights): fpq_fast32.h -------------------------------------------------
inline bool fast32_eligible(const void* x, const void* out, int64_t cols, int in_dtype, int table_id) {
  return in_dtype == FPQ_F32 && cols == 128 && kTables[table_id].symmetric && (((uintptr_t)x | (uintptr_t)out) & 15) == 0 &&
         !fpq_flag(OPT_FPQ_NO_FAST32);
}

// one tensor (segs == nullptr, `one` by value) or a device-resident segment table (grid.y = segment)
inline int launch_fast32(const Seg32* segs, int n_segs, const Seg32& one, int64_t max_rows, int table_id, int out_dtype,
                         hipStream_t st) {
  constexpr int U = 4;
  const Lut32Args a = lut32_args(table_id);
  const int64_t tiles = (max_rows * 32 + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
  if (tiles > 0x7FFFFFFF || n_segs > 65535) return FPQ_ERR_SHAPE;
  const dim3 grid((unsigned)tiles, (unsigned)n_segs);
  if (out_dtype == FPQ_F16)
    hipLaunchKernelGGL((groups32_lut_kernel<_Float16, U>), grid, dim3(kBlock), 0, st, segs, one, a);
  else
    hipLaunchKernelGGL((groups32_lut_kernel<float, U>), grid, dim3(kBlock), 0, st, segs, one, a);
  return check_launch();
}

// fp32 groups of 128 -> codes + fp32 scales: one tensor (segs == nullptr) or a device-resident segment table
inline int launch_codes32(const CodesSeg32* segs, int n_segs, const CodesSeg32& one, int64_t max_rows, int table_id, bool pack,
                          hipStream_t st) {
  constexpr int U = 4;
  const Lut32Args a = lut32_args(table_id);
  const int64_t tiles = (max_rows * 32 + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
  if (tiles > 0x7FFFFFFF || n_segs > 65535) return FPQ_ERR_SHAPE;
  const dim3 grid((unsigned)tiles, (unsigned)n_segs);
  if (pack) hipLaunchKernelGGL((groups32_codes_kernel<true, U>), grid, dim3(kBlock), 0, st, segs, one, a);
  else hipLaunchKernelGGL((groups32_codes_kernel<false, U>), grid, dim3(kBlock), 0, st, segs, one, a);
  return check_launch();
}

// long fp32 rows (per-channel weights): one wavefront or one workgroup per row (fpq_fast32.h)
inline bool rows32_eligible(const void* x, const void* out, int64_t cols, int in_dtype, int table_id) {
  return in_dtype == FPQ_F32 && cols % 8 == 0 && cols >= 512 && cols / 4 <= 256 * 10 && kTables[table_id].symmetric &&
         (((uintptr_t)x | (uintptr_t)out) & 15) == 0 && !fpq_flag(OPT_FPQ_NO_FAST32);
}

template <typename Tout>
int launch_rows32(const void* x, void* out, int64_t rows, int64_t cols, int table_id, hipStream_t st) {
  const Lut32Args a = lut32_args(table_id);
  const int64_t vpr = cols / 4;
#define FPQ_R32(L, M)                                                                                              \
  do {                                                                                                             \
    const int64_t wgs = (rows + (kBlock / L) - 1) / (kBlock / L);                                                  \
    hipLaunchKernelGGL((rows32_lut_kernel<Tout, L, M>), dim3(grid_for(wgs, 1 << 16)), dim3(kBlock), 0, st,        \
                       (const float*)x, (Tout*)out, rows, cols, a);                                               \
    return check_launch();                                                                                         \
  } while (0)
  if (vpr <= 64 * 2) FPQ_R32(64, 2);
  if (vpr <= 64 * 4) FPQ_R32(64, 4);
  if (vpr <= 64 * 8) FPQ_R32(64, 8);
  if (vpr <= 256 * 3) FPQ_R32(256, 3);
  if (vpr <= 256 * 4) FPQ_R32(256, 4);
  if (vpr <= 256 * 6) FPQ_R32(256, 6);
  if (vpr <= 256 * 8) FPQ_R32(256, 8);
  FPQ_R32(256, 10);
#undef FPQ_R32
}

// ---- F2: hardware-nibble codes + FP4 MFMA GEMM ---------------------------------------
inline const Lut16Tab& lut16_mx_codes_e2m1() {
  static const Lut16Tab* tab = [] {
    auto* t = new Lut16Tab;
    const Lut16Host& h = lut16_host(FPQ_E2M1, FPQ_E2M1);
    const int n = 1 << (16 - h.args.shift);
    uint16_t full[kLutLdsEntries] = {0};
    for (int i = 0; i < n; ++i) {
      uint32_t u = (uint32_t)i << h.args.shift;
      bool neg = (u >> 15) != 0;
      float qm = quant_mag(h2f(u & 0x7FFFu), 0u, h.args.fpos);
      // level -> magnitude index 0..7 (host twin of level_index)
      int li = (qm >= h.args.fpos.kmin) ? (int)((fbits(qm) >> h.args.fpos.mshift) - h.args.fpos.kmin_code_base)
                                        : (int)(qm * h.args.fpos.inv_step0);
      full[i] = (uint16_t)(li | ((neg && li != 0) ? 8 : 0));
    }
    if (!lut16_compress(full, h.args.shift, t)) abort();   // E2M1: 2 x 128 buckets, always fits
    return t;
  }();
  return *tab;
}

}  // namespace

// =================================================================================
// C ABI
// =================================================================================
template <typename T>
int launch_negrev(const void* x, void* out, int64_t rows, int64_t cols, const Fmt& fs, hipStream_t st) {
  constexpr int V = DT<T>::kVec;
  constexpr int U = 2;
  const bool aligned = (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
  const int64_t lpr = cols / V;
  if (aligned && cols % V == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0) {
    const int64_t n_vec = rows * lpr;
    auto go = [&](auto kern) {
      int64_t blocks = (n_vec + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
      hipLaunchKernelGGL(kern, dim3(grid_for(blocks, 1 << 20)), dim3(kBlock), 0, st, (const u32x4*)x, (u32x4*)out,
                         n_vec, fs);
      return check_launch();
    };
    switch ((int)lpr) {
      case 1: return go(rows_negrev_subwave_kernel<T, 1, U>);
      case 2: return go(rows_negrev_subwave_kernel<T, 2, U>);
      case 4: return go(rows_negrev_subwave_kernel<T, 4, U>);
      case 8: return go(rows_negrev_subwave_kernel<T, 8, U>);
      case 16: return go(rows_negrev_subwave_kernel<T, 16, U>);
      case 32: return go(rows_negrev_subwave_kernel<T, 32, U>);
      case 64: return go(rows_negrev_subwave_kernel<T, 64, U>);
    }
  }
  hipLaunchKernelGGL((rows_negrev_scalar_kernel<T>), dim3(grid_for(rows, 65535)), dim3(kBlock), 0, st, (const T*)x,
                     (T*)out, rows, cols, fs);
  return check_launch();
}

extern "C" {

int fpq_version(void) { return FPQ_VERSION; }

#ifndef FPQ_BUILD_TAG
#define FPQ_BUILD_TAG "stock"
#endif
const char* fpq_build_tag(void) { return FPQ_BUILD_TAG; }

int fpq_internal_dual_lut(int neg_table, int pos_table, void* args_out, size_t args_bytes, void* tab_out, size_t tab_bytes) {
  if (neg_table < 0 || neg_table >= FPQ_NUM_TABLES || pos_table < 0 || pos_table >= FPQ_NUM_TABLES) return FPQ_ERR_TABLE;
  const Lut16Host& h = lut16_host(neg_table, pos_table);
  if (!h.tab_valid || args_bytes != sizeof(Lut16Args) || tab_bytes != sizeof(Lut16Tab)) return FPQ_ERR_TABLE;
  memcpy(args_out, &h.args, sizeof(Lut16Args));
  memcpy(tab_out, &h.tab, sizeof(Lut16Tab));
  return FPQ_OK;
}

int fpq_internal_zero_if_flag(void* out, int64_t n_bytes, void* scratch, void* stream) {
  hipLaunchKernelGGL(zero_if_flag_kernel, dim3(kFixupBlocks), dim3(kBlock), 0, (hipStream_t)stream, (uint8_t*)out, n_bytes, (uint32_t*)scratch);
  return check_launch();
}

int fpq_set_option(const char* name, int value) {
  const int i = option_index(name);
  if (i < 0) return FPQ_ERR_ARG;
  __atomic_store_n(&fpq_option_table[i], value, __ATOMIC_RELAXED);
  return FPQ_OK;
}
int fpq_get_option(const char* name, int* value_out) {
  const int i = option_index(name);
  if (i < 0 || !value_out) return FPQ_ERR_ARG;
  *value_out = fpq_opt_raw(i);
  return FPQ_OK;
}
const char* fpq_option_name(int index) { return (index >= 0 && index < FPQ_OPT_COUNT) ? kOptionDescs[index].name : nullptr; }

const char* fpq_strerror(int status) {
  switch (status) {
    case FPQ_OK: return "ok";
    case FPQ_ERR_ARG: return "invalid argument (null pointer or negative size)";
    case FPQ_ERR_DTYPE: return "unsupported dtype for this entry point";
    case FPQ_ERR_SHAPE: return "unsupported shape";
    case FPQ_ERR_TABLE: return "unknown table id, or half table passed where a symmetric table is required";
    case FPQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case FPQ_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown fpq status";
  }
}

int fpq_table_values(int table_id, float* host_out) {
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES) return FPQ_ERR_TABLE;
  float pos[64];
  int np = pos_levels(table_id, pos);
  int n = 0;
  const bool neg_half = (table_id == FPQ_E1M2_NEG || table_id == FPQ_INT_NEG || table_id == FPQ_E2M1_NEG);
  const bool pos_half = (table_id == FPQ_E2M1_POS || table_id == FPQ_E2M3_POS);
  const bool dup_zero = (table_id == FPQ_E2M3 || table_id == FPQ_E3M2);
  if (!pos_half) {
    for (int i = np - 1; i >= 1; --i, ++n)
      if (host_out) host_out[n] = -pos[i];
    if (neg_half || dup_zero) {
      if (host_out) host_out[n] = 0.0f;
      ++n;
    }
  }
  if (!neg_half)
    for (int i = 0; i < np; ++i, ++n)
      if (host_out) host_out[n] = pos[i];
  return n;
}

static const KnownTables& known_tables() {
  static const KnownTables* kt = [] {
    auto* t = new KnownTables();
    int off = 0;
    for (int id = 0; id < FPQ_NUM_TABLES; ++id) {
      t->off[id] = (int16_t)off;
      t->k[id] = (int16_t)fpq_table_values(id, t->v + off);
      off += t->k[id];
      t->fmt[id] = make_fmt(id);
      t->side[id] = kTables[id].symmetric ? 0 : ((id == FPQ_E2M1_POS || id == FPQ_E2M3_POS) ? 2 : 1);
    }
    return t;
  }();
  return *kt;
}

int fpq_quant_nearest(const void* x, const float* table, void* z, int64_t n, int k, int dtype,
                      fpq_stream_t stream) {
  if (n < 0) return FPQ_ERR_ARG;
  if (k < 1 || k > 256) return FPQ_ERR_SHAPE;
  if (dtype != FPQ_F32 && dtype != FPQ_F64) return FPQ_ERR_DTYPE;
  if (n == 0) return FPQ_OK;
  if (!x || !table || !z) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int g = grid_for((n + kBlock - 1) / kBlock);
  if (dtype == FPQ_F32)
    hipLaunchKernelGGL(nearest_scan_kernel<float>, dim3(grid_for((n / 4 + 2 * kBlock - 1) / (2 * kBlock) + 1, 4096)),
                       dim3(kBlock), 0, st, (const float*)x, table, (float*)z, n, k, known_tables());
  else
    hipLaunchKernelGGL(nearest_scan_kernel<double>, dim3(g), dim3(kBlock), 0, st, (const double*)x, table,
                       (double*)z, n, k, known_tables());
  return check_launch();
}

int fpq_quant_nearest_argmin(const void* x, const float* table, float* z, int64_t n, int k, int dtype,
                             fpq_stream_t stream) {
  if (n < 0) return FPQ_ERR_ARG;
  if (k < 1 || k > 256) return FPQ_ERR_SHAPE;
  if (dtype != FPQ_F16 && dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (n == 0) return FPQ_OK;
  if (!x || !table || !z) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for((n + kBlock - 1) / kBlock, 8192);
  if (dtype == FPQ_F32)
    hipLaunchKernelGGL(nearest_argmin_kernel<float>, dim3(g), dim3(kBlock), 0, st, (const float*)x, table, z, n, k);
  else
    hipLaunchKernelGGL(nearest_argmin_kernel<_Float16>, dim3(g), dim3(kBlock), 0, st, (const _Float16*)x, table, z, n, k);
  return check_launch();
}

int fpq_quant_nearest_builtin(const float* x, float* z, int64_t n, int table_id, fpq_stream_t stream) {
  if (n < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES) return FPQ_ERR_TABLE;
  if (n == 0) return FPQ_OK;
  if (!x || !z) return FPQ_ERR_ARG;
  int side = kTables[table_id].symmetric ? 0 : ((table_id == FPQ_E1M2_NEG || table_id == FPQ_INT_NEG || table_id == FPQ_E2M1_NEG) ? 1 : 2);
  hipLaunchKernelGGL(nearest_builtin_kernel, dim3(grid_for((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, x, z, n, make_fmt(table_id), side);
  return check_launch();
}

int fpq_quant_rows(const void* x, void* out, int64_t rows, int64_t cols, int table_id, int in_dtype, int out_dtype,
                   fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if ((in_dtype != FPQ_F16 && in_dtype != FPQ_F32) || (out_dtype != FPQ_F16 && out_dtype != FPQ_F32))
    return FPQ_ERR_DTYPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !out) return FPQ_ERR_ARG;
  if (fast16_eligible(x, out, cols, in_dtype, out_dtype)) {
    // tables of >= 1024 buckets (E2M3, int: 2 KiB to stage per workgroup, a quarter of an 8 KiB tile's own bytes): at most
    // 16384 workgroups, so that at the large shapes every workgroup stages once for two tiles or more - E2M3 g = 128 and
    // KV rows of 64 at [65536 x 1920]: 84.2 -> 80.8 us (caps 10240 .. 24576: 82.3 .. 80.6; U = 4 and U = 1 are slower,
    // profiles/r03_e2m3_grid.txt)
    const int cap = (1 << (16 - lut16_host(table_id, table_id).args.shift)) >= 1024 ? 16384 : 1 << 20;
    return launch_fast16<false>(x, out, rows, cols, table_id, table_id, (hipStream_t)stream, cap);
  }
  if (fast16_block_eligible(x, out, cols, in_dtype, out_dtype))
    return launch_fast16_block<false>(x, out, rows, cols, table_id, table_id, (hipStream_t)stream);
  if (fast32_eligible(x, out, cols, in_dtype, table_id)) {
    const Seg32 one = {x, out, rows};
    return launch_fast32(nullptr, 1, one, rows, table_id, out_dtype, (hipStream_t)stream);
  }
  if (rows32_eligible(x, out, cols, in_dtype, table_id))
    return out_dtype == FPQ_F16 ? launch_rows32<_Float16>(x, out, rows, cols, table_id, (hipStream_t)stream)
                                : launch_rows32<float>(x, out, rows, cols, table_id, (hipStream_t)stream);
  DualArgs dual = {};
  dual.nan_flag = nullptr;
  return dispatch_rows<false>(x, out, rows, cols, in_dtype, out_dtype, make_fmt(table_id), dual,
                              (hipStream_t)stream);
}

int fpq_quant_rows_multi(const fpq_segment_t* segments_host, int n_segments, int64_t cols, int table_id, int in_dtype,
                         int out_dtype, fpq_stream_t stream) {
  if (n_segments < 0 || cols < 0 || (n_segments > 0 && !segments_host)) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if ((in_dtype != FPQ_F16 && in_dtype != FPQ_F32) || (out_dtype != FPQ_F16 && out_dtype != FPQ_F32)) return FPQ_ERR_DTYPE;
  for (int i = 0; i < n_segments; ++i) {
    if (segments_host[i].rows < 0) return FPQ_ERR_ARG;
    if (segments_host[i].rows > 0 && cols > 0 && (!segments_host[i].x || !segments_host[i].out)) return FPQ_ERR_ARG;
  }
  if (n_segments == 0 || cols == 0) return FPQ_OK;
  hipStream_t st = (hipStream_t)stream;
  bool one_launch = in_dtype == FPQ_F16 && out_dtype == FPQ_F16 && n_segments <= kMaxMulti && cols % 8 == 0;
  const int64_t lpr = cols / 8;
  one_launch = one_launch && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && lut16_host(table_id, table_id).tab_valid;
  int64_t max_vec = 0;
  for (int i = 0; one_launch && i < n_segments; ++i) {
    if ((((uintptr_t)segments_host[i].x | (uintptr_t)segments_host[i].out) & 15) != 0) one_launch = false;
    max_vec = segments_host[i].rows * lpr > max_vec ? segments_host[i].rows * lpr : max_vec;
  }
  if (!one_launch) {   // anything the fused multi-tensor kernel does not cover: still one C call, one launch per tensor
    for (int i = 0; i < n_segments; ++i)
      if (int rc = fpq_quant_rows(segments_host[i].x, segments_host[i].out, segments_host[i].rows, cols, table_id, in_dtype,
                                  out_dtype, stream))
        return rc;
    return FPQ_OK;
  }
  if (max_vec == 0) return FPQ_OK;
  constexpr int U = 2;
  Multi16 m = {};
  for (int i = 0; i < n_segments; ++i) {
    m.x[i] = (const u32x4*)segments_host[i].x;
    m.out[i] = (u32x4*)segments_host[i].out;
    m.n_vec[i] = segments_host[i].rows * lpr;
  }
  const Lut16Host& h = lut16_host(table_id, table_id);
  const int64_t tiles = (max_vec + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
  if (tiles > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  const dim3 grid((unsigned)tiles, (unsigned)n_segments);
#define FPQ_MULTI_CASE(L) \
  case L: hipLaunchKernelGGL((rows16_lut_multi_kernel<L, U>), grid, dim3(kBlock), 0, st, m, h.args, h.tab); break;
  switch ((int)lpr) {
    FPQ_MULTI_CASE(1) FPQ_MULTI_CASE(2) FPQ_MULTI_CASE(4) FPQ_MULTI_CASE(8) FPQ_MULTI_CASE(16) FPQ_MULTI_CASE(32) FPQ_MULTI_CASE(64)
  }
#undef FPQ_MULTI_CASE
  return check_launch();
}

int fpq_quant_rows_segments(const fpq_segment_t* segments_device, int n_segments, int64_t max_rows, int64_t cols,
                            int table_id, int in_dtype, int out_dtype, fpq_stream_t stream) {
  static_assert(sizeof(fpq_segment_t) == sizeof(Seg32), "fpq_segment_t and the kernels' Seg32 share one layout");
  if (n_segments < 0 || max_rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F32 || (out_dtype != FPQ_F16 && out_dtype != FPQ_F32)) return FPQ_ERR_DTYPE;
  if (cols != 128) return FPQ_ERR_SHAPE;
  if (n_segments == 0 || max_rows == 0) return FPQ_OK;
  if (!segments_device || (((uintptr_t)segments_device) & 7) != 0) return FPQ_ERR_ARG;
  const Seg32 none = {nullptr, nullptr, 0};
  return launch_fast32((const Seg32*)segments_device, n_segments, none, max_rows, table_id, out_dtype, (hipStream_t)stream);
}

int fpq_kv_cache_step(void* cache, int64_t batch, int64_t max_len, int64_t row_elems, int64_t quant_start,
                      int64_t quant_stop, const void* new_k, const void* new_v, int64_t new_batch_pitch,
                      int64_t new_token_pitch, int64_t new_start, int64_t n_new, int64_t group, int table_id,
                      fpq_stream_t stream) {
  if (batch < 0 || max_len < 0 || row_elems <= 0 || n_new < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (quant_start < 0 || quant_stop < quant_start || new_start < quant_stop || new_start + n_new > max_len)
    return FPQ_ERR_ARG;
  if (group != 8 && group != 16 && group != 32 && group != 64 && group != 128 && group != 256 && group != 512)
    return FPQ_ERR_SHAPE;                      // rows of 8 * {1..64} halves, owned by 1..64 lanes
  if (row_elems % group != 0 || batch > 65535) return FPQ_ERR_SHAPE;
  if (new_batch_pitch % 8 != 0 || new_token_pitch % 8 != 0 || new_token_pitch < 0 || new_batch_pitch < 0) return FPQ_ERR_SHAPE;
  const int64_t n_quant = quant_stop - quant_start;
  if (batch == 0 || (n_quant == 0 && n_new == 0)) return FPQ_OK;
  if (!cache || (n_new > 0 && (!new_k || !new_v))) return FPQ_ERR_ARG;
  if ((((uintptr_t)cache | (uintptr_t)new_k | (uintptr_t)new_v) & 15) != 0) return FPQ_ERR_ARG;
  const Lut16Host& h = lut16_host(table_id, table_id);
  if (!h.tab_valid) return FPQ_ERR_TABLE;
  constexpr int U = 2;
  KvStepArgs k;
  k.cache = (u32x4*)cache;
  k.row_vec = (int)(row_elems / 8);
  k.slab_vec = max_len * k.row_vec;
  k.batch = (int)batch;
  k.q_first_vec = quant_start * k.row_vec;
  k.q_vecs = n_quant * k.row_vec;
  const int64_t q_tiles = (k.q_vecs + kBlock * U - 1) / (kBlock * U);
  k.src[0] = (const uint16_t*)new_k;
  k.src[1] = (const uint16_t*)new_v;
  k.src_batch_pitch = new_batch_pitch;
  k.src_token_pitch = new_token_pitch;
  k.new_first_vec = new_start * k.row_vec;
  k.new_vecs = n_new * k.row_vec;
  const int64_t c_tiles = (k.new_vecs + kBlock * U - 1) / (kBlock * U);
  if (q_tiles + c_tiles > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  k.q_tiles = (int)q_tiles;
  const dim3 grid((unsigned)(q_tiles + c_tiles), (unsigned)batch, 2);
  const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
  hipStream_t st = (hipStream_t)stream;
#define FPQ_KV_CASE(L) \
  case L: hipLaunchKernelGGL((kv16_step_kernel<L, U>), grid, dim3(kBlock), lds, st, k, h.args, h.tab); break;
  switch ((int)(group / 8)) {
    FPQ_KV_CASE(1) FPQ_KV_CASE(2) FPQ_KV_CASE(4) FPQ_KV_CASE(8) FPQ_KV_CASE(16) FPQ_KV_CASE(32) FPQ_KV_CASE(64)
  }
#undef FPQ_KV_CASE
  return check_launch();
}

int fpq_quant_rows_argmin(const void* x, float* out, int64_t rows, int64_t cols, int table_id, int in_dtype,
                          int clamp3, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !out) return FPQ_ERR_ARG;
  Fmt f = make_fmt(table_id);
  f.argmin = 1;
  f.preclamp = clamp3 ? 3.0f : 0.0f;
  DualArgs dual = {};
  dual.nan_flag = nullptr;
  return dispatch_rows<false>(x, out, rows, cols, in_dtype, FPQ_F32, f, dual, (hipStream_t)stream);
}

int fpq_quant_rows_dual(const void* x, void* out, int64_t rows, int64_t cols, int neg_table, int pos_table,
                        int in_dtype, int out_dtype, const void* clip_absmax, float clip_strength, void* nan_flag,
                        fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (neg_table != FPQ_E1M2_NEG && neg_table != FPQ_INT_NEG && neg_table != FPQ_E2M1_NEG) return FPQ_ERR_TABLE;
  if (pos_table != FPQ_E2M1_POS && pos_table != FPQ_E2M3_POS) return FPQ_ERR_TABLE;
  if ((in_dtype != FPQ_F16 && in_dtype != FPQ_F32) || (out_dtype != FPQ_F16 && out_dtype != FPQ_F32))
    return FPQ_ERR_DTYPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !out) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  uint32_t* flag = (uint32_t*)nan_flag;
  if (flag && (((uintptr_t)flag) & 7) != 0) return FPQ_ERR_ARG;
  int rc;
  const bool bigtab = (1 << (16 - lut16_host(neg_table, pos_table).args.shift)) > 1024;
  if (clip_absmax && clip_strength >= 0.0f && cols == 128 && !bigtab && fast16_eligible(x, out, cols, in_dtype, out_dtype)) {
    // the global clamp on the fast path (fp16 groups of 128, strength >= 0; anything else below, through the generic kernel)
    rc = launch_fast16<true>(x, out, rows, cols, neg_table, pos_table, st, 1 << 20, flag, clip_absmax, clip_strength);
  } else if (!clip_absmax && fast16_eligible(x, out, cols, in_dtype, out_dtype)) {
    // int_neg/e2m3_pos needs a 2048-entry table (too big for the kernel arguments): every workgroup
    // evaluates it once, so give each workgroup many tiles (measured: 88 us vs 127 us with a full grid)
    // tables of 2 x 1024 buckets (int_neg / e2m3_pos) cost a workgroup 8 stores per lane to stage: give each
    // workgroup many tiles (capped grid, U = 4); the smaller ones run one tile per workgroup on a full grid
    if ((1 << (16 - lut16_host(neg_table, pos_table).args.shift)) <= 1024)
      rc = launch_fast16<true>(x, out, rows, cols, neg_table, pos_table, st, 1 << 20, flag);
    else {
      const int cap = fpq_opt(OPT_FPQ_BIGTAB_CAP, 16384);   // measured on [65536 x 7680]: 4096 -> 366 us, 16384 -> 348 us, full grid -> 367 us
      if (fpq_opt(OPT_FPQ_BIGTAB_U, 4) == 8) rc = launch_fast16<true, 8>(x, out, rows, cols, neg_table, pos_table, st, cap, flag);
      else rc = launch_fast16<true, 4>(x, out, rows, cols, neg_table, pos_table, st, cap, flag);
    }
  } else if (!clip_absmax && fast16_block_eligible(x, out, cols, in_dtype, out_dtype)) {
    rc = launch_fast16_block<true>(x, out, rows, cols, neg_table, pos_table, st, flag);
  } else {
    DualArgs dual;
    dual.fneg = make_fmt(neg_table);
    dual.fpos = make_fmt(pos_table);
    dual.clip_absmax = clip_absmax;
    dual.clip_strength = clip_strength;
    dual.nan_flag = flag;
    rc = dispatch_rows<true>(x, out, rows, cols, in_dtype, out_dtype, dual.fneg, dual, st);
  }
  if (rc != FPQ_OK || !flag) return rc;
  const int64_t n_bytes = rows * cols * (out_dtype == FPQ_F16 ? 2 : 4);
  hipLaunchKernelGGL(zero_if_flag_kernel, dim3(kFixupBlocks), dim3(kBlock), 0, st, (uint8_t*)out, n_bytes, flag);
  return check_launch();
}

int fpq_gelu_quant_rows_dual(const void* x, void* out, void* gelu_out, int64_t rows, int64_t cols, int neg_table, int pos_table,
                             void* nan_flag, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (neg_table != FPQ_E1M2_NEG && neg_table != FPQ_INT_NEG && neg_table != FPQ_E2M1_NEG) return FPQ_ERR_TABLE;
  if (pos_table != FPQ_E2M1_POS && pos_table != FPQ_E2M3_POS) return FPQ_ERR_TABLE;
  // groups of 128 (rows inside a wavefront) or rows of at most 16384 elements (one workgroup per row: the per-token forms)
  if (cols <= 0 || cols % 8 != 0 || cols > 8 * 8 * kBlock) return FPQ_ERR_SHAPE;
  if (rows == 0) return FPQ_OK;
  if (!x || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gelu_out) & 15) != 0 || ((uintptr_t)nan_flag & 7) != 0) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  uint32_t* flag = (uint32_t*)nan_flag;
  int rc;
  if (cols != 128)
    rc = launch_fast16_block<true>(x, out, rows, cols, neg_table, pos_table, st, flag, true, gelu_out);
  else if ((1 << (16 - lut16_host(neg_table, pos_table).args.shift)) <= 1024)
    rc = launch_fast16<true>(x, out, rows, cols, neg_table, pos_table, st, 1 << 20, flag, nullptr, 1.0f, true, gelu_out);
  else   // int_neg / e2m3_pos: 2 x 1024 buckets to stage per workgroup - many tiles per workgroup, as fpq_quant_rows_dual
    rc = launch_fast16<true, 4>(x, out, rows, cols, neg_table, pos_table, st, fpq_opt(OPT_FPQ_BIGTAB_CAP, 16384), flag, nullptr, 1.0f, true, gelu_out);
  if (rc != FPQ_OK) return rc;
  if (!nan_flag) return FPQ_OK;
  hipLaunchKernelGGL(zero_if_flag_kernel, dim3(kFixupBlocks), dim3(kBlock), 0, st, (uint8_t*)out, rows * cols * 2, (uint32_t*)nan_flag);
  return check_launch();
}

int fpq_quant_rows_neg_reverse(const void* x, void* out, int64_t rows, int64_t cols, int table_id, int dtype,
                               fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (dtype != FPQ_F16 && dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !out) return FPQ_ERR_ARG;
  if (dtype == FPQ_F16) return launch_negrev<_Float16>(x, out, rows, cols, make_fmt(table_id), (hipStream_t)stream);
  return launch_negrev<float>(x, out, rows, cols, make_fmt(table_id), (hipStream_t)stream);
}

static int rotate_quant_impl(const void* x, void* out, void* rotated_out, void* code_scales, int64_t rows, int64_t cols,
                             int in_dtype, const float* smooth, const uint32_t* sign_mask_host, int table_id,
                             fpq_stream_t stream, bool km = false) {
  if (rows < 0 || cols < 0 || !sign_mask_host) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (cols % 128 != 0) return FPQ_ERR_SHAPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)rotated_out | (uintptr_t)smooth) & 15) != 0) return FPQ_ERR_ARG;
  if (in_dtype == FPQ_F16)
    return launch_rotate_quant<_Float16>(x, out, rotated_out, rows, cols, smooth, sign_mask_host, table_id,
                                         (hipStream_t)stream, (uint16_t*)code_scales, km);
  return launch_rotate_quant<float>(x, out, rotated_out, rows, cols, smooth, sign_mask_host, table_id,
                                    (hipStream_t)stream, (uint16_t*)code_scales, km);
}

int fpq_quant_rows_dual_argmin(const void* x, float* out, int64_t rows, int64_t cols, int neg_table, int pos_table,
                               int in_dtype, const void* clip_absmax, float clip_strength, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (neg_table != FPQ_E1M2_NEG && neg_table != FPQ_INT_NEG && neg_table != FPQ_E2M1_NEG) return FPQ_ERR_TABLE;
  if (pos_table != FPQ_E2M1_POS && pos_table != FPQ_E2M3_POS) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !out) return FPQ_ERR_ARG;
  DualArgs dual;
  dual.fneg = make_fmt(neg_table);
  dual.fpos = make_fmt(pos_table);
  dual.fneg.argmin = dual.fpos.argmin = 1;
  dual.clip_absmax = clip_absmax;
  dual.clip_strength = clip_strength;
  dual.nan_flag = nullptr;
  return dispatch_rows<true>(x, out, rows, cols, in_dtype, FPQ_F32, dual.fneg, dual, (hipStream_t)stream);
}

int fpq_rotate_quant_rows(const void* x, void* out, void* rotated_out, int64_t rows, int64_t cols, int in_dtype,
                          const float* smooth, const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream) {
  return rotate_quant_impl(x, out, rotated_out, nullptr, rows, cols, in_dtype, smooth, sign_mask_host, table_id, stream);
}

int fpq_rotate_quant_rows_codes_mx(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int in_dtype,
                                   const float* smooth, const uint32_t* sign_mask_host, fpq_stream_t stream) {
  if (rows > 0 && cols > 0 && !scales) return FPQ_ERR_ARG;
  return rotate_quant_impl(x, codes, nullptr, scales, rows, cols, in_dtype, smooth, sign_mask_host, FPQ_E2M1, stream);
}
int fpq_rotate_quant_rows_codes_mx_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols, int in_dtype,
                                      const float* smooth, const uint32_t* sign_mask_host, fpq_stream_t stream) {
  if (rows > 0 && cols > 0 && !scales) return FPQ_ERR_ARG;
  return rotate_quant_impl(x, image, nullptr, scales, rows, cols, in_dtype, smooth, sign_mask_host, FPQ_E2M1, stream, true);
}

static int adaln_rotate_quant_impl(const void* x, void* out, void* h_out, void* rotated_out, void* code_scales,
                                   int64_t rows, int64_t cols, int in_dtype, const void* scale, const void* shift,
                                   int mod_dtype, int64_t rows_per_batch, float eps, const float* smooth,
                                   const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream,
                                   int token_mode = 0, const Lut16Tab* token_code_tab = nullptr, bool km = false) {
  if (rows < 0 || cols < 0 || rows_per_batch <= 0 || !sign_mask_host) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if ((in_dtype != FPQ_F16 && in_dtype != FPQ_F32) || (mod_dtype != FPQ_F16 && mod_dtype != FPQ_F32))
    return FPQ_ERR_DTYPE;
  if (cols % 128 != 0 || cols > 4096) return FPQ_ERR_SHAPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !out || !scale || !shift) return FPQ_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)h_out | (uintptr_t)rotated_out | (uintptr_t)scale |
        (uintptr_t)shift | (uintptr_t)smooth) & 15) != 0)
    return FPQ_ERR_ARG;
  AdaLnArgs ad;
  ad.scale = scale;
  ad.shift = shift;
  ad.mod_is_f16 = mod_dtype == FPQ_F16;
  ad.rows_per_batch = rows_per_batch;
  ad.eps = eps;
  ad.cols = cols;
  // One wavefront per row while the row fits 5 vectors per lane (C <= 2560: no barrier in the row
  // loop; measured 0.180 ms vs 0.199 ms per [65500 x 1920] on MI355X), one workgroup per row beyond.
  // FPQ_ADALN_LANES / FPQ_ADALN_GRID override the choice for experiments.
  const int lanes = (fpq_opt_set(OPT_FPQ_ADALN_LANES) && !token_mode) ? fpq_opt(OPT_FPQ_ADALN_LANES, 64) : (cols / 8 <= 64 * 5 ? 64 : 256);
  const int lpr = (lanes == 64) ? 64 : 256;
  if (token_mode && lpr != 64) return FPQ_ERR_SHAPE;   // the per-token form keeps a row inside one wavefront: C <= 2560
  hipStream_t st = (hipStream_t)stream;
#define FPQ_GO(TI, TM) return launch_adaln_rotate_quant<TI, TM>(x, out, h_out, rotated_out, rows, cols, ad, smooth, \
                                                              sign_mask_host, table_id, st, lpr, (uint16_t*)code_scales, \
                                                              token_mode, token_code_tab, km)
  if (in_dtype == FPQ_F16 && mod_dtype == FPQ_F16) FPQ_GO(_Float16, _Float16);
  if (in_dtype == FPQ_F16) FPQ_GO(_Float16, float);
  if (mod_dtype == FPQ_F16) FPQ_GO(float, _Float16);
  FPQ_GO(float, float);
#undef FPQ_GO
}

int fpq_adaln_rotate_quant_rows(const void* x, void* out, void* h_out, void* rotated_out, int64_t rows, int64_t cols,
                                int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                int64_t rows_per_batch, float eps, const float* smooth,
                                const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream) {
  return adaln_rotate_quant_impl(x, out, h_out, rotated_out, nullptr, rows, cols, in_dtype, scale, shift, mod_dtype,
                                 rows_per_batch, eps, smooth, sign_mask_host, table_id, stream);
}

int fpq_adaln_rotate_quant_rows_codes_mx(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols,
                                         int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                         int64_t rows_per_batch, float eps, const float* smooth,
                                         const uint32_t* sign_mask_host, fpq_stream_t stream) {
  if (rows > 0 && cols > 0 && !scales) return FPQ_ERR_ARG;
  return adaln_rotate_quant_impl(x, codes, nullptr, nullptr, scales, rows, cols, in_dtype, scale, shift, mod_dtype,
                                 rows_per_batch, eps, smooth, sign_mask_host, FPQ_E2M1, stream);
}
int fpq_adaln_rotate_quant_rows_codes_mx_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols,
                                            int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                            int64_t rows_per_batch, float eps, const float* smooth,
                                            const uint32_t* sign_mask_host, fpq_stream_t stream) {
  if (rows > 0 && cols > 0 && !scales) return FPQ_ERR_ARG;
  return adaln_rotate_quant_impl(x, image, nullptr, nullptr, scales, rows, cols, in_dtype, scale, shift, mod_dtype,
                                 rows_per_batch, eps, smooth, sign_mask_host, FPQ_E2M1, stream, 0, nullptr, true);
}

static int quant_rows_codes_mx_impl(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int in_dtype,
                                    bool km, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (km && in_dtype != FPQ_F16) return FPQ_ERR_DTYPE;   // fp32 rows (weights): fpq_quant_rows_codes_mx + fpq_codes_to_kmajor
  if (cols % 128 != 0 || (km && !km_image_fits(rows, cols / 2))) return FPQ_ERR_SHAPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !codes || !scales) return FPQ_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)codes | (uintptr_t)scales) & 15) != 0) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == FPQ_F16) {
    const Lut16Host& h = lut16_host(FPQ_E2M1, FPQ_E2M1);
    const int64_t n_vec = rows * (cols / 8);
    const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
    hipLaunchKernelGGL(rows16_codes_mx_kernel, dim3(grid_for((n_vec + kBlock - 1) / kBlock, 16384)), dim3(kBlock), lds, st,
                       (const u32x4*)x, (uint32_t*)codes, (uint16_t*)scales, n_vec, h.args, lut16_mx_codes_e2m1(),
                       km ? (uint32_t)rows : 0u, fast_div((uint32_t)(cols / 128)));
  } else {
    const int64_t n_vec = rows * (cols / 4);
    hipLaunchKernelGGL((codes128_kernel<float, true, true>), dim3(grid_for((n_vec + kBlock - 1) / kBlock, 1 << 20)),
                       dim3(kBlock), 0, st, (const u32x4*)x, codes, (float*)scales, n_vec, make_fmt(FPQ_E2M1));
  }
  return check_launch();
}
int fpq_quant_rows_codes_mx(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int in_dtype,
                            fpq_stream_t stream) {
  return quant_rows_codes_mx_impl(x, codes, scales, rows, cols, in_dtype, false, stream);
}
int fpq_quant_rows_codes_mx_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols, int in_dtype,
                               fpq_stream_t stream) {
  return quant_rows_codes_mx_impl(x, image, scales, rows, cols, in_dtype, true, stream);
}

// host: OCP E4M3 byte of a value that is exactly representable (every level of the symmetric tables is)
static uint8_t e4m3_of(float v) {
  if (v == 0.0f) return 0;
  const uint8_t sgn = v < 0.0f ? 0x80 : 0;
  int e;
  const float m = frexpf(fabsf(v), &e);        // |v| = m * 2^e, m in [0.5, 1)
  const int ex = e - 1;                         // |v| = (2m) * 2^(e-1), 2m in [1, 2)
  const int man = (int)((2.0f * m - 1.0f) * 8.0f);
  return (uint8_t)(sgn | ((ex + 7) << 3) | man);
}

// bucket -> E4M3 code tables for the fast fp16 path, one per symmetric table, built once (immutable afterwards)
static const Lut16Tab& lut16_codes8(int table_id) {
  static const Lut16Tab* tabs = [] {
    auto* t = new Lut16Tab[FPQ_NUM_TABLES]();
    for (int id = 0; id < FPQ_NUM_TABLES; ++id) {
      if (!kTables[id].symmetric) continue;
      const Lut16Host& h = lut16_host(id, id);
      if (!h.tab_valid) continue;
      const int n = 1 << (16 - h.args.shift);
      uint16_t full[kLutLdsEntries] = {0};
      for (int i = 0; i < n; ++i) full[i] = e4m3_of(h2f(h.full[i]));
      if (!lut16_compress(full, h.args.shift, &t[id])) abort();   // same structure as the level table it is derived from
    }
    return t;
  }();
  return tabs[table_id];
}

int fpq_quant_rows_codes_fp8(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int table_id,
                             int in_dtype, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !codes || !scales) return FPQ_ERR_ARG;
  if (in_dtype == FPQ_F16 && cols % 8 == 0 && cols <= 4096 && (((uintptr_t)x | (uintptr_t)codes) & 15) == 0 &&
      lut16_host(table_id, table_id).tab_valid) {
    const Lut16Host& h = lut16_host(table_id, table_id);
    const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
    const int64_t wgs = (rows + kBlock / 64 - 1) / (kBlock / 64);
    const dim3 gw(grid_for(wgs, 8192));
    const int maxc = (int)((cols / 8 + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
#define FPQ_C8(M) hipLaunchKernelGGL((rows16_codes8_wave_kernel<M>), gw, dim3(kBlock), lds, st, (const uint16_t*)x, codes, \
                                     (uint16_t*)scales, rows, cols, h.args, lut16_codes8(table_id))
    if (maxc <= 2) FPQ_C8(2);
    else if (maxc <= 4) FPQ_C8(4);
    else FPQ_C8(8);
#undef FPQ_C8
    return check_launch();
  }
  const dim3 g(grid_for(rows, 65535));
  if (in_dtype == FPQ_F16)
    hipLaunchKernelGGL(rows_codes_fp8_kernel<_Float16>, g, dim3(kBlock), 0, (hipStream_t)stream, (const _Float16*)x,
                       codes, (_Float16*)scales, rows, cols, make_fmt(table_id));
  else
    hipLaunchKernelGGL(rows_codes_fp8_kernel<float>, g, dim3(kBlock), 0, (hipStream_t)stream, (const float*)x, codes,
                       (float*)scales, rows, cols, make_fmt(table_id));
  return check_launch();
}

int fpq_adaln_rotate_quant_token_rows(const void* x, void* out, void* h_out, void* rotated_out, void* row_scales,
                                      int64_t rows, int64_t cols, int in_dtype, const void* scale, const void* shift,
                                      int mod_dtype, int64_t rows_per_batch, float eps, const float* smooth,
                                      const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream) {
  if ((((uintptr_t)row_scales) & 1) != 0) return FPQ_ERR_ARG;
  return adaln_rotate_quant_impl(x, out, h_out, rotated_out, row_scales, rows, cols, in_dtype, scale, shift, mod_dtype,
                                 rows_per_batch, eps, smooth, sign_mask_host, table_id, stream, 1, nullptr);
}

int fpq_adaln_rotate_quant_token_rows_codes_fp8(const void* x, uint8_t* codes, void* row_scales, int64_t rows, int64_t cols,
                                                int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                                int64_t rows_per_batch, float eps, const float* smooth,
                                                const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream) {
  if (rows > 0 && cols > 0 && !row_scales) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  return adaln_rotate_quant_impl(x, codes, nullptr, nullptr, row_scales, rows, cols, in_dtype, scale, shift, mod_dtype,
                                 rows_per_batch, eps, smooth, sign_mask_host, table_id, stream, 2, &lut16_codes8(table_id));
}

static const Lut16Tab& lut16_codes6_e2m3() {
  static const Lut16Tab* tab = [] {
    auto* t = new Lut16Tab();
    const Lut16Host& h = lut16_host(FPQ_E2M3, FPQ_E2M3);
    const int n = 1 << (16 - h.args.shift);
    uint16_t full[kLutLdsEntries] = {0};
    for (int i = 0; i < n; ++i) full[i] = (uint16_t)e2m3_of_level(h2f(h.full[i]));
    if (!lut16_compress(full, h.args.shift, t)) abort();
    return t;
  }();
  return *tab;
}

int fpq_adaln_rotate_quant_token_rows_codes_fp6(const void* x, uint8_t* codes, void* row_scales, int64_t rows, int64_t cols,
                                                int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                                int64_t rows_per_batch, float eps, const float* smooth,
                                                const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream) {
  if (rows > 0 && cols > 0 && !row_scales) return FPQ_ERR_ARG;
  if (table_id != FPQ_E2M3) return FPQ_ERR_TABLE;
  if (cols % 32 != 0) return FPQ_ERR_SHAPE;
  if ((((uintptr_t)codes) & 7) != 0) return FPQ_ERR_ARG;
  return adaln_rotate_quant_impl(x, codes, nullptr, nullptr, row_scales, rows, cols, in_dtype, scale, shift, mod_dtype,
                                 rows_per_batch, eps, smooth, sign_mask_host, table_id, stream, 3, &lut16_codes6_e2m3());
}
int fpq_adaln_rotate_quant_token_rows_codes_fp6_km(const void* x, uint8_t* image, void* row_scales, int64_t rows, int64_t cols,
                                                   int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                                   int64_t rows_per_batch, float eps, const float* smooth,
                                                   const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream) {
  if (rows > 0 && cols > 0 && !row_scales) return FPQ_ERR_ARG;
  if (table_id != FPQ_E2M3) return FPQ_ERR_TABLE;
  if (cols % 128 != 0) return FPQ_ERR_SHAPE;
  return adaln_rotate_quant_impl(x, image, nullptr, nullptr, row_scales, rows, cols, in_dtype, scale, shift, mod_dtype,
                                 rows_per_batch, eps, smooth, sign_mask_host, table_id, stream, 3, &lut16_codes6_e2m3(), true);
}

static int quant_rows_codes_fp6_impl(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int table_id,
                                     int in_dtype, bool km, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id != FPQ_E2M3) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (cols % 32 != 0 || (km && (cols % 128 != 0 || !km_image_fits(rows, cols / 4 * 3)))) return FPQ_ERR_SHAPE;
  const uint32_t km_rows = km ? (uint32_t)rows : 0u;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !codes || !scales) return FPQ_ERR_ARG;
  if ((((uintptr_t)codes) & 7) != 0) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == FPQ_F16 && cols <= 8192 && (((uintptr_t)x) & 15) == 0) {
    const Lut16Host& h = lut16_host(table_id, table_id);
    const size_t lds = 0;   // the bucket table lives in static LDS (fpq_fast16.h)
    const int64_t wgs = (rows + kBlock / 64 - 1) / (kBlock / 64);
    const dim3 gw(grid_for(wgs, 8192));
    const int maxc = (int)((cols / 32 + 63) / 64);
#define FPQ_C6(M) hipLaunchKernelGGL((rows16_codes6_wave_kernel<M>), gw, dim3(kBlock), lds, st, (const uint16_t*)x, codes, \
                                     (uint16_t*)scales, rows, cols, h.args, lut16_codes6_e2m3(), km_rows)
    if (maxc <= 1) FPQ_C6(1);
    else if (maxc <= 2) FPQ_C6(2);
    else FPQ_C6(4);
#undef FPQ_C6
    return check_launch();
  }
  const dim3 g(grid_for(rows, 65535));
  if (in_dtype == FPQ_F16)
    hipLaunchKernelGGL(rows_codes_fp6_kernel<_Float16>, g, dim3(kBlock), 0, st, (const _Float16*)x, codes,
                       (_Float16*)scales, rows, cols, make_fmt(table_id), km_rows);
  else
    hipLaunchKernelGGL(rows_codes_fp6_kernel<float>, g, dim3(kBlock), 0, st, (const float*)x, codes, (float*)scales, rows,
                       cols, make_fmt(table_id), km_rows);
  return check_launch();
}
int fpq_quant_rows_codes_fp6(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int table_id,
                             int in_dtype, fpq_stream_t stream) {
  return quant_rows_codes_fp6_impl(x, codes, scales, rows, cols, table_id, in_dtype, false, stream);
}
int fpq_quant_rows_codes_fp6_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols, int table_id,
                                int in_dtype, fpq_stream_t stream) {
  return quant_rows_codes_fp6_impl(x, image, scales, rows, cols, table_id, in_dtype, true, stream);
}

int fpq_absmax(const void* x, int64_t n, int dtype, void* out, fpq_stream_t stream) {
  if (n < 0 || !out) return FPQ_ERR_ARG;
  if (dtype != FPQ_F16 && dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, 4, st) != hipSuccess) return FPQ_ERR_LAUNCH;
  if (n == 0) return FPQ_OK;
  if (!x) return FPQ_ERR_ARG;
  int64_t per_block = (int64_t)kBlock * 16;
  int g = grid_for((n + per_block - 1) / per_block);
  if (dtype == FPQ_F16)
    hipLaunchKernelGGL(absmax_kernel<_Float16>, dim3(g), dim3(kBlock), 0, st, (const _Float16*)x, n, (uint32_t*)out);
  else
    hipLaunchKernelGGL(absmax_kernel<float>, dim3(g), dim3(kBlock), 0, st, (const float*)x, n, (uint32_t*)out);
  return check_launch();
}

int fpq_quant_tensor_argmin(const void* x, float* out, float* scale_out, void* workspace, int64_t n, int table_id,
                            int in_dtype, fpq_stream_t stream) {
  if (n < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (!scale_out || !workspace || (((uintptr_t)workspace | (uintptr_t)scale_out) & 3) != 0) return FPQ_ERR_ARG;
  if (n > 0 && (!x || !out)) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Fmt f = make_fmt(table_id);
  f.argmin = 1;
  constexpr int64_t per_block = (int64_t)kBlock * 16;
  const int g1 = grid_for((n + per_block - 1) / per_block, FPQ_TENSOR_WORKSPACE_BYTES / 4);   // also 1 when n == 0
  const int V = in_dtype == FPQ_F16 ? 8 : 4;
  const int g2 = grid_for((n / V + 2 * kBlock - 1) / (2 * kBlock) + 1, 1 << 16);
  if (in_dtype == FPQ_F16) {
    hipLaunchKernelGGL(absmax_partials_kernel<_Float16>, dim3(g1), dim3(kBlock), 0, st, (const _Float16*)x, n,
                       (uint32_t*)workspace);
    hipLaunchKernelGGL(tensor_argmin_kernel<_Float16>, dim3(g2), dim3(kBlock), 0, st, (const _Float16*)x, out, n,
                       (const uint32_t*)workspace, g1, scale_out, f);
  } else {
    hipLaunchKernelGGL(absmax_partials_kernel<float>, dim3(g1), dim3(kBlock), 0, st, (const float*)x, n,
                       (uint32_t*)workspace);
    hipLaunchKernelGGL(tensor_argmin_kernel<float>, dim3(g2), dim3(kBlock), 0, st, (const float*)x, out, n,
                       (const uint32_t*)workspace, g1, scale_out, f);
  }
  return check_launch();
}

int fpq_quant_rows_codes(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int table_id,
                         int in_dtype, int pack_nibbles, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (pack_nibbles && kTables[table_id].n_pos > 8) return FPQ_ERR_SHAPE;  // FP6 codes do not fit a nibble
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!x || !codes || !scales) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Fmt f = make_fmt(table_id);
  if (in_dtype == FPQ_F32 && cols == 128 && (((uintptr_t)x | (uintptr_t)codes) & 15) == 0 && (((uintptr_t)scales) & 3) == 0 &&
      !fpq_flag(OPT_FPQ_NO_FAST32)) {   // fp32 weights: the approximate-then-verify path (fpq_fast32.h, groups32_codes_kernel)
    const CodesSeg32 one = {x, codes, scales, rows};
    return launch_codes32(nullptr, 1, one, rows, table_id, pack_nibbles != 0, st);
  }
  if (cols == 128 && (((uintptr_t)x | (uintptr_t)codes | (uintptr_t)scales) & 15) == 0) {
    const int64_t n_vec = rows * (in_dtype == FPQ_F16 ? 16 : 32);
    const int gv = grid_for((n_vec + kBlock - 1) / kBlock, 1 << 20);
    if (in_dtype == FPQ_F16 && pack_nibbles)
      hipLaunchKernelGGL((codes128_kernel<_Float16, true>), dim3(gv), dim3(kBlock), 0, st, (const u32x4*)x, codes, (_Float16*)scales, n_vec, f);
    else if (in_dtype == FPQ_F16)
      hipLaunchKernelGGL((codes128_kernel<_Float16, false>), dim3(gv), dim3(kBlock), 0, st, (const u32x4*)x, codes, (_Float16*)scales, n_vec, f);
    else if (pack_nibbles)
      hipLaunchKernelGGL((codes128_kernel<float, true>), dim3(gv), dim3(kBlock), 0, st, (const u32x4*)x, codes, (float*)scales, n_vec, f);
    else
      hipLaunchKernelGGL((codes128_kernel<float, false>), dim3(gv), dim3(kBlock), 0, st, (const u32x4*)x, codes, (float*)scales, n_vec, f);
    return check_launch();
  }
  int g = grid_for(rows, 65535);
  if (in_dtype == FPQ_F16)
    hipLaunchKernelGGL(rows_codes_kernel<_Float16>, dim3(g), dim3(kBlock), 0, st, (const _Float16*)x, codes,
                       (_Float16*)scales, rows, cols, f, pack_nibbles ? 1 : 0);
  else
    hipLaunchKernelGGL(rows_codes_kernel<float>, dim3(g), dim3(kBlock), 0, st, (const float*)x, codes,
                       (float*)scales, rows, cols, f, pack_nibbles ? 1 : 0);
  return check_launch();
}

int fpq_dequant_rows_codes(const uint8_t* codes, const void* scales, void* out, int64_t rows, int64_t cols,
                           int table_id, int scale_dtype, int out_dtype, int pack_nibbles, fpq_stream_t stream) {
  if (rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if ((scale_dtype != FPQ_F16 && scale_dtype != FPQ_F32) || (out_dtype != FPQ_F16 && out_dtype != FPQ_F32))
    return FPQ_ERR_DTYPE;
  if (pack_nibbles && kTables[table_id].n_pos > 8) return FPQ_ERR_SHAPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!codes || !scales || !out) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Fmt f = make_fmt(table_id);
  if (cols == 128 && (((uintptr_t)out | (uintptr_t)codes) & 15) == 0) {
    const int64_t n_oct = rows * 16;
    const int64_t per_wg = pack_nibbles ? 4 * kBlock : kBlock;   // the nibble form decodes tiles of 4 x 256 octets
    const int gv = grid_for((n_oct + per_wg - 1) / per_wg, 1 << 20);
#define FPQ_DEC(TS, TO, PK) hipLaunchKernelGGL((decode128_kernel<TS, TO, PK>), dim3(gv), dim3(kBlock), 0, st, codes, (const TS*)scales, (TO*)out, n_oct, f)
    if (scale_dtype == FPQ_F16 && out_dtype == FPQ_F16) { if (pack_nibbles) FPQ_DEC(_Float16, _Float16, true); else FPQ_DEC(_Float16, _Float16, false); }
    else if (scale_dtype == FPQ_F16) { if (pack_nibbles) FPQ_DEC(_Float16, float, true); else FPQ_DEC(_Float16, float, false); }
    else if (out_dtype == FPQ_F16) { if (pack_nibbles) FPQ_DEC(float, _Float16, true); else FPQ_DEC(float, _Float16, false); }
    else { if (pack_nibbles) FPQ_DEC(float, float, true); else FPQ_DEC(float, float, false); }
#undef FPQ_DEC
    return check_launch();
  }
  int g = grid_for(rows, 65535);
  int pk = pack_nibbles ? 1 : 0;
  if (scale_dtype == FPQ_F16 && out_dtype == FPQ_F16)
    hipLaunchKernelGGL((rows_decode_kernel<_Float16, _Float16>), dim3(g), dim3(kBlock), 0, st, codes,
                       (const _Float16*)scales, (_Float16*)out, rows, cols, f, pk);
  else if (scale_dtype == FPQ_F16 && out_dtype == FPQ_F32)
    hipLaunchKernelGGL((rows_decode_kernel<_Float16, float>), dim3(g), dim3(kBlock), 0, st, codes,
                       (const _Float16*)scales, (float*)out, rows, cols, f, pk);
  else if (scale_dtype == FPQ_F32 && out_dtype == FPQ_F16)
    hipLaunchKernelGGL((rows_decode_kernel<float, _Float16>), dim3(g), dim3(kBlock), 0, st, codes,
                       (const float*)scales, (_Float16*)out, rows, cols, f, pk);
  else
    hipLaunchKernelGGL((rows_decode_kernel<float, float>), dim3(g), dim3(kBlock), 0, st, codes,
                       (const float*)scales, (float*)out, rows, cols, f, pk);
  return check_launch();
}

int fpq_quant_rows_codes_segments(const fpq_codes_segment_t* segments_device, int n_segments, int64_t max_rows,
                                  int64_t cols, int table_id, int in_dtype, int pack_nibbles, fpq_stream_t stream) {
  static_assert(sizeof(fpq_codes_segment_t) == sizeof(CodesSeg), "fpq_codes_segment_t and the kernels' CodesSeg share one layout");
  if (n_segments < 0 || max_rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if (in_dtype != FPQ_F16 && in_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (cols != 128 || n_segments > 65535) return FPQ_ERR_SHAPE;
  if (pack_nibbles && kTables[table_id].n_pos > 8) return FPQ_ERR_SHAPE;
  if (n_segments == 0 || max_rows == 0) return FPQ_OK;
  if (!segments_device || (((uintptr_t)segments_device) & 7) != 0) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == FPQ_F32 && !fpq_flag(OPT_FPQ_NO_FAST32)) {   // (segments are 16-byte aligned by contract, include/fpq.h)
    static_assert(sizeof(CodesSeg32) == sizeof(CodesSeg), "one segment layout");
    return launch_codes32((const CodesSeg32*)segments_device, n_segments, CodesSeg32{nullptr, nullptr, nullptr, 0}, max_rows, table_id,
                          pack_nibbles != 0, st);
  }
  const Fmt f = make_fmt(table_id);
  const int64_t n_vec = max_rows * (in_dtype == FPQ_F16 ? 16 : 32);
  // a few vectors per thread in the largest segment: the grid's y dimension multiplies it by the segment count
  const dim3 grid((unsigned)grid_for((n_vec + 4 * kBlock - 1) / (4 * kBlock), 1 << 16), (unsigned)n_segments);
  const CodesSeg* sg = (const CodesSeg*)segments_device;
  if (in_dtype == FPQ_F16 && pack_nibbles) hipLaunchKernelGGL((codes128_segments_kernel<_Float16, true>), grid, dim3(kBlock), 0, st, sg, f);
  else if (in_dtype == FPQ_F16) hipLaunchKernelGGL((codes128_segments_kernel<_Float16, false>), grid, dim3(kBlock), 0, st, sg, f);
  else if (pack_nibbles) hipLaunchKernelGGL((codes128_segments_kernel<float, true>), grid, dim3(kBlock), 0, st, sg, f);
  else hipLaunchKernelGGL((codes128_segments_kernel<float, false>), grid, dim3(kBlock), 0, st, sg, f);
  return check_launch();
}

int fpq_dequant_rows_codes_segments(const fpq_decode_segment_t* segments_device, int n_segments, int64_t max_rows,
                                    int64_t cols, int table_id, int scale_dtype, int out_dtype, int pack_nibbles,
                                    fpq_stream_t stream) {
  static_assert(sizeof(fpq_decode_segment_t) == sizeof(DecodeSeg), "fpq_decode_segment_t and the kernels' DecodeSeg share one layout");
  if (n_segments < 0 || max_rows < 0 || cols < 0) return FPQ_ERR_ARG;
  if (table_id < 0 || table_id >= FPQ_NUM_TABLES || !kTables[table_id].symmetric) return FPQ_ERR_TABLE;
  if ((scale_dtype != FPQ_F16 && scale_dtype != FPQ_F32) || (out_dtype != FPQ_F16 && out_dtype != FPQ_F32))
    return FPQ_ERR_DTYPE;
  if (cols != 128 || n_segments > 65535) return FPQ_ERR_SHAPE;
  if (pack_nibbles && kTables[table_id].n_pos > 8) return FPQ_ERR_SHAPE;
  if (n_segments == 0 || max_rows == 0) return FPQ_OK;
  if (!segments_device || (((uintptr_t)segments_device) & 7) != 0) return FPQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const Fmt f = make_fmt(table_id);
  const int64_t n_oct = max_rows * 16;
  const dim3 grid((unsigned)grid_for((n_oct + 4 * kBlock - 1) / (4 * kBlock), 1 << 16), (unsigned)n_segments);
  const DecodeSeg* sg = (const DecodeSeg*)segments_device;
#define FPQ_DECS(TS, TO, PK) hipLaunchKernelGGL((decode128_segments_kernel<TS, TO, PK>), grid, dim3(kBlock), 0, st, sg, f)
  if (scale_dtype == FPQ_F16 && out_dtype == FPQ_F16) { if (pack_nibbles) FPQ_DECS(_Float16, _Float16, true); else FPQ_DECS(_Float16, _Float16, false); }
  else if (scale_dtype == FPQ_F16) { if (pack_nibbles) FPQ_DECS(_Float16, float, true); else FPQ_DECS(_Float16, float, false); }
  else if (out_dtype == FPQ_F16) { if (pack_nibbles) FPQ_DECS(float, _Float16, true); else FPQ_DECS(float, _Float16, false); }
  else { if (pack_nibbles) FPQ_DECS(float, float, true); else FPQ_DECS(float, float, false); }
#undef FPQ_DECS
  return check_launch();
}

}  // extern "C"
