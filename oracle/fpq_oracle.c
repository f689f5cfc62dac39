/* fpq_oracle.c - plain-C CPU restatement of FPQVAR's fake-quant path.
 * TEST INFRASTRUCTURE ONLY (see oracle/fpq_oracle.py): used by tests/ to cross-check
 * the numpy/torch restatement with an independent scalar implementation, and by
 * bench.py as a second CPU figure.  Never linked into or called by the product.
 *
 * Follows, in the reference (PKU-SEC-Lab/FPQVAR):
 *   scan            quant/quant_kernel.cu:25-37
 *   rows            models_fp_quant_transform_rotate/quant_utils.py:265-282,313-330,361-378,503-574
 *   rows_dual       models_fp_quant_transform_rotate/quant_utils.py:415-452,577-646
 * fp16 arithmetic is done the way torch does it: operands widened to fp32, one
 * fp32 operation, result rounded to fp16 (software conversions below; gcc 11 on
 * x86-64 has no _Float16).  Build: see oracle/Makefile (no fast-math, no contraction).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float h2f(uint16_t h) {
  uint32_t s = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
  if (e == 0) {
    if (m == 0) return u2f(s);
    int sh = 0;
    while (!(m & 1024u)) { m <<= 1; ++sh; }
    return u2f(s | ((uint32_t)(113 - sh) << 23) | ((m & 1023u) << 13));
  }
  if (e == 31) return u2f(s | 0x7F800000u | (m << 13));
  return u2f(s | ((e + 112u) << 23) | (m << 13));
}

/* round-to-nearest-even fp32 -> fp16 (the rounding mode is the default one) */
static uint16_t f2h(float f) {
  uint32_t x = f2u(f);
  uint16_t s = (uint16_t)((x >> 16) & 0x8000u);
  x &= 0x7FFFFFFFu;
  if (x > 0x7F800000u) return (uint16_t)(s | 0x7E00u);           /* NaN */
  if (x == 0x7F800000u) return (uint16_t)(s | 0x7C00u);
  float a = u2f(x);
  if (a < 6.103515625e-05f) {                                      /* below 2^-14: fixed point, unit 2^-24 */
    float r = nearbyintf(a * 16777216.0f);
    return (uint16_t)(s | (uint16_t)r);                            /* 1024 = smallest normal */
  }
  int ex;
  float mant = frexpf(a, &ex);                                     /* a = mant * 2^ex, mant in [0.5,1) */
  float r = nearbyintf(mant * 2048.0f);                            /* 1024..2048 */
  if (r == 2048.0f) { r = 1024.0f; ++ex; }
  int he = ex + 14;                                                /* biased half exponent */
  if (he >= 31) return (uint16_t)(s | 0x7C00u);
  return (uint16_t)(s | (he << 10) | ((uint16_t)r - 1024));
}

/* quant/quant_kernel.cu:25-37 */
static float scan(float xv, const float* tab, int k) {
  float best = 102400.0f, z = 0.0f;
  for (int j = 0; j < k; ++j) {
    float d = fabsf(xv - tab[j]);
    if (d <= best) { best = d; z = tab[j]; }
  }
  return z;
}

static float tab_absmax(const float* tab, int k) {
  float m = 0.0f;
  for (int j = 0; j < k; ++j) if (fabsf(tab[j]) > m) m = fabsf(tab[j]);
  return m;
}

void fpq_oracle_nearest_f32(const float* x, const float* tab, float* z, int64_t n, int k) {
  for (int64_t i = 0; i < n; ++i) z[i] = scan(x[i], tab, k);
}

/* torch.max over a row of |x|: NaN propagates */
static float row_absmax(const float* v, int64_t n) {
  float m = 0.0f;
  int nan = 0;
  for (int64_t i = 0; i < n; ++i) {
    float a = fabsf(v[i]);
    if (a != a) nan = 1;
    else if (a > m) m = a;
  }
  return nan ? NAN : m;
}

/* One row, symmetric table.  half_in: x holds fp16 values (already widened), every
 * intermediate is rounded to fp16.  half_out: result rounded to fp16 (kept widened). */
static void row_sym(const float* x, float* out, int64_t n, const float* tab, int k, int half_in, int half_out) {
  float g = tab_absmax(tab, k);
  float s = row_absmax(x, n) / g;
  if (half_in) s = h2f(f2h(s));
  for (int64_t i = 0; i < n; ++i) {
    float xn = x[i] / s;
    if (half_in) xn = h2f(f2h(xn));
    volatile float p = scan(xn, tab, k) * s;      /* fp32 product, materialised */
    out[i] = half_out ? h2f(f2h(p)) : p;
  }
}

static void row_dual(const float* x, float* out, int64_t n, const float* tneg, int kn, const float* tpos, int kp,
                     int half_in, int half_out) {
  float mn = 0.0f, mp = 0.0f;
  for (int64_t i = 0; i < n; ++i) {      /* where(x<=0,x,0) / where(x>0,x,0): NaN takes neither side */
    if (x[i] <= 0.0f) { if (fabsf(x[i]) > mn) mn = fabsf(x[i]); }
    else if (x[i] > 0.0f) { if (x[i] > mp) mp = x[i]; }
  }
  float sn = mn / tab_absmax(tneg, kn), sp = mp / tab_absmax(tpos, kp);
  if (half_in) { sn = h2f(f2h(sn)); sp = h2f(f2h(sp)); }
  for (int64_t i = 0; i < n; ++i) {
    float xneg = (x[i] <= 0.0f) ? x[i] : 0.0f, xpos = (x[i] > 0.0f) ? x[i] : 0.0f;
    float a = xneg / sn, b = xpos / sp;
    if (half_in) { a = h2f(f2h(a)); b = h2f(f2h(b)); }
    volatile float pa = scan(a, tneg, kn) * sn;
    volatile float pb = scan(b, tpos, kp) * sp;
    volatile float p = pa + pb;
    out[i] = half_out ? h2f(f2h(p)) : p;
  }
}

/* Public entry points: fp16 tensors travel as uint16 bit patterns. */
#define MAXROW 65536
static float bufx[MAXROW], bufo[MAXROW];

int fpq_oracle_rows_f16(const uint16_t* x, uint16_t* out, int64_t rows, int64_t cols, const float* tab, int k) {
  if (cols > MAXROW) return -1;
  for (int64_t r = 0; r < rows; ++r) {
    for (int64_t c = 0; c < cols; ++c) bufx[c] = h2f(x[r * cols + c]);
    row_sym(bufx, bufo, cols, tab, k, 1, 1);
    for (int64_t c = 0; c < cols; ++c) out[r * cols + c] = f2h(bufo[c]);
  }
  return 0;
}

int fpq_oracle_rows_f32(const float* x, float* out_f32, uint16_t* out_f16, int64_t rows, int64_t cols,
                        const float* tab, int k) {
  if (cols > MAXROW) return -1;
  for (int64_t r = 0; r < rows; ++r) {
    row_sym(x + r * cols, bufo, cols, tab, k, 0, out_f16 != 0);
    for (int64_t c = 0; c < cols; ++c) {
      if (out_f16) out_f16[r * cols + c] = f2h(bufo[c]);
      else out_f32[r * cols + c] = bufo[c];
    }
  }
  return 0;
}

int fpq_oracle_rows_dual_f16(const uint16_t* x, uint16_t* out, int64_t rows, int64_t cols, const float* tneg,
                             int kn, const float* tpos, int kp) {
  if (cols > MAXROW) return -1;
  for (int64_t r = 0; r < rows; ++r) {
    for (int64_t c = 0; c < cols; ++c) bufx[c] = h2f(x[r * cols + c]);
    row_dual(bufx, bufo, cols, tneg, kn, tpos, kp, 1, 1);
    for (int64_t c = 0; c < cols; ++c) out[r * cols + c] = f2h(bufo[c]);
  }
  return 0;
}

int fpq_oracle_rows_dual_f32(const float* x, float* out, int64_t rows, int64_t cols, const float* tneg, int kn,
                             const float* tpos, int kp) {
  if (cols > MAXROW) return -1;
  for (int64_t r = 0; r < rows; ++r) row_dual(x + r * cols, out + r * cols, cols, tneg, kn, tpos, kp, 0, 0);
  return 0;
}

/* conversions exported for their own test */
void fpq_oracle_h2f(const uint16_t* h, float* f, int64_t n) { for (int64_t i = 0; i < n; ++i) f[i] = h2f(h[i]); }
void fpq_oracle_f2h(const float* f, uint16_t* h, int64_t n) { for (int64_t i = 0; i < n; ++i) h[i] = f2h(f[i]); }
