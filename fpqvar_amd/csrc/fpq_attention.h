// fpq_attention.h - the consumer of the KV cache: softmax(q k^T * scale) v for head_dim 64 in the layout the
// reference hands to flash_attn_func (SelfAttention.forward, tr/basic_var.py:173,211: q / k / v are [B, L, H, c]
// views, no mask and no dropout at inference because KV caching is on, :159).  Included by fpq_kernels.hip.
//
// One workgroup = 4 wavefronts = 128 query rows of one (batch, head); every wavefront owns 32 query rows and walks the
// keys in tiles of 64 that the workgroup stages once (global -> registers -> LDS, double buffered, one barrier per tile).
//
// Matrix-core mapping (v_mfma_f32_32x32x16_f16, gfx950), chosen so that nothing is ever moved between lanes:
//   S^T[kv x q] = K[kv x c] . Q^T[c x q]     A = K rows from LDS (ds_read_b128), B = Q rows held in registers.
//       The result has the query row on the lane (l & 31) and 16 of a sub-tile's 32 keys in its registers
//       (key = (r & 3) + 8 (r >> 2) + 4 (l >> 5)), so the online softmax is lane-local except for one exchange with
//       lane l ^ 32 per tile (the row maximum).
//   O^T[c x q] = V^T[c x kv] . P^T[kv x q]   B = the S registers 8s .. 8s+7 converted to fp16 in place (their key order is
//       the instruction's own k order), A = V^T read with ds_read_b64_tr_b16 (hardware transpose of 4 keys x 16
//       channels per 16 lanes) from the row-major V tile.  The output again has the query row on the lane, so the
//       running rescale and the final 1 / l are lane-local too.
// LDS images, 128-byte rows (one key): K chunk c of row r at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 in
// the instruction's four 16-lane groups), V bytes of row r XOR 64 * ((r >> 1) & 1) (conflict-free transposed reads in
// both 32-lane halves).
#pragma once

typedef _Float16 attn_h8_t __attribute__((ext_vector_type(8)));
typedef short attn_s4_t __attribute__((ext_vector_type(4)));
typedef float attn_f16_t __attribute__((ext_vector_type(16)));

struct AttnArgs {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* out;                     // [B, Lq, H, 64] contiguous
  int64_t q_batch, q_token;          // element strides; heads are contiguous (stride 64)
  int64_t kv_batch, kv_token;
  int batch, heads, lq, lkv, q_tiles;
  float scale_log2e;                 // softmax scale * log2(e)
};

constexpr int kAttnKv = 64;          // keys per staged tile
constexpr int kAttnTile = kAttnKv * 128;   // bytes of one K (or V) tile

__global__ __launch_bounds__(256, 3) void attn_fwd64_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[4 * kAttnTile];   // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // all query tiles of one (batch, head) on one XCD, next to each other in dispatch order
  const int id = blockIdx.x;
  const int grp = (id >> 3) / a.q_tiles, qt = (id >> 3) - grp * a.q_tiles;
  const int bh = grp * 8 + (id & 7);
  if (bh >= a.batch * a.heads) return;   // uniform over the workgroup
  const int b = bh / a.heads, h = bh - b * a.heads;
  const int ql = lane & 31, hi = lane >> 5;

  // Q fragments: row q, channels 16 ks + 8 hi + j
  int qrow = qt * 128 + wave * 32 + ql;
  const bool q_live = qrow < a.lq;
  const bool wave_live = __builtin_amdgcn_readfirstlane(qt * 128 + wave * 32) < a.lq;   // a wavefront past the last query row only helps staging
  qrow = q_live ? qrow : a.lq - 1;
  attn_h8_t qf[4];
  {
    const uint16_t* qp = a.q + (int64_t)b * a.q_batch + (int64_t)qrow * a.q_token + h * 64 + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(attn_h8_t, *(const u32x4*)(qp + 16 * ks));
  }

  // staging: thread -> (row, 16-byte chunk) of the K and of the V tile, two of each
  const uint16_t* kbase = a.k + (int64_t)b * a.kv_batch + h * 64;
  const uint16_t* vbase = a.v + (int64_t)b * a.kv_batch + h * 64;
  int st_row[2], st_koff[2], st_voff[2], st_ch[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 3, c = idx & 7;
    st_row[i] = r;
    st_ch[i] = c;
    st_koff[i] = r * 128 + 16 * (c ^ ((r >> 1) & 7));
    st_voff[i] = r * 128 + ((16 * c) ^ (64 * ((r >> 1) & 1)));
  }
  u32x4 gk[2], gv[2], gk2[2], gv2[2];   // tile t + 1 and tile t + 2 in flight: two iterations to cover the load latency
  const int n_tiles = (a.lkv + kAttnKv - 1) / kAttnKv;
#define FPQ_ATTN_FETCH(t, GK, GV)                                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                  \
    int kv_ = (t) * kAttnKv + st_row[i_];                                                             \
    kv_ = kv_ < a.lkv ? kv_ : a.lkv - 1;                                                              \
    GK[i_] = *(const u32x4*)(kbase + (int64_t)kv_ * a.kv_token + 8 * st_ch[i_]);                     \
    GV[i_] = *(const u32x4*)(vbase + (int64_t)kv_ * a.kv_token + 8 * st_ch[i_]);                     \
  }
#define FPQ_ATTN_STAGE(buf, GK, GV)                                                                   \
  _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                  \
    *(u32x4*)(smem + (buf) * 2 * kAttnTile + st_koff[i_]) = GK[i_];                                   \
    *(u32x4*)(smem + (buf) * 2 * kAttnTile + kAttnTile + st_voff[i_]) = GV[i_];                       \
  }
  FPQ_ATTN_FETCH(0, gk, gv);
  FPQ_ATTN_STAGE(0, gk, gv);
  // Everything issued so far (the Q fragments too) has to have landed before the loop: otherwise the compiler, not
  // knowing how many prefetches are in flight on each path, waits for vmcnt(0) in front of the loop's first MFMAs -
  // i.e. for the prefetch it has just issued - in every iteration.
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __builtin_amdgcn_sched_barrier(0);
  FPQ_ATTN_FETCH(1, gk, gv);

  // fragment addresses inside a tile
  //   K (A operand of S^T): row 32 u + ql, chunk 2 ks + hi
  int k_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_off[ks] = ql * 128 + 16 * ((2 * ks + hi) ^ ((ql >> 1) & 7));   // + 4096 u (32 rows: same swizzle)
  //   V^T (A operand of O^T): transposed read of rows r0 + (i >> 2), channels c0 + 4 (i & 3), i = lane & 15,
  //   r0 = 32 u + 16 s + 4 hi (+ 8 for the second read), c0 = 32 dt + 16 ((lane >> 4) & 1)
  const int tr_i = lane & 15, tr_row = 4 * hi + (tr_i >> 2), tr_col = 16 * ((lane >> 4) & 1) + 4 * (tr_i & 3);
  // rows 32 u + 16 s + {0, 8} + tr_row: bit 1 of the row is bit 1 of tr_row (the other terms are multiples of 4)
  const int v_off = tr_row * 128 + ((2 * tr_col) ^ (64 * ((tr_row >> 1) & 1)));   // + 128 * (32 u + 16 s + 8 e) , dt: ^ 64

  attn_f16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const float c = a.scale_log2e;

  // One iteration = one tile.  g1 holds tile t + 1 (fetched an iteration ago, staged at the end of this one), g2
  // receives tile t + 2; the two register sets swap roles from one iteration to the next (no copies: a copy would
  // wait for the load it copies).
  auto iter = [&](int t, u32x4 (&gk1)[2], u32x4 (&gv1)[2], u32x4 (&gk2_)[2], u32x4 (&gv2_)[2]) {
    __syncthreads();   // tile t is in buffer t & 1; the other buffer's readers (iteration t - 1) are done
    const bool more = t + 1 < n_tiles;
    FPQ_ATTN_FETCH(t + 2, gk2_, gv2_);   // unconditional (rows past the end clamp to the last key): a fixed number of loads
                                         // in flight lets the staging below wait for exactly its own
    const uint8_t* kt = smem + (t & 1) * 2 * kAttnTile;
    const uint8_t* vt = kt + kAttnTile;

    if (wave_live) {   // uniform over the wavefront: EXEC stays all ones inside (the transposed reads need that)
    attn_f16_t s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const attn_h8_t kf = __builtin_bit_cast(attn_h8_t, *(const u32x4*)(kt + 4096 * u + k_off[ks]));
        s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[u], 0, 0, 0);
      }
    }
    if (!more && (a.lkv & (kAttnKv - 1)) != 0) {   // keys past the end of the last tile (uniform branch)
      const int base = t * kAttnKv + 4 * hi;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (base + 32 * u + (r & 3) + 8 * (r >> 2) >= a.lkv) s[u][r] = -INFINITY;
    }
    float mx = s[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
    {   // the other 16 keys of the sub-tiles live in lane l ^ 32: one v_permlane32_swap, no LDS round trip
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float m_new = fmaxf(m_run, mx);            // finite: every tile holds at least one real key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    m_run = m_new;
    float psum = 0.0f;
    attn_h8_t pf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], c, -mc));
        psum += p;
        pf[u][r >> 3][r & 7] = (_Float16)p;
      }
    l_run = l_run * alpha + psum;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // uniform branch
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int sst = 0; sst < 2; ++sst)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const uint8_t* p0 = vt + 128 * (32 * u + 16 * sst) + (v_off ^ (64 * dt));
          const attn_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_s4_t*)(p0));
          const attn_s4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_s4_t*)(p0 + 128 * 8));
          typedef short attn_s8_t __attribute__((ext_vector_type(8)));
          const attn_s8_t both = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(attn_h8_t, both), pf[u][sst], o[dt], 0, 0, 0);
        }
    }
    if (more) { FPQ_ATTN_STAGE((t + 1) & 1, gk1, gv1); }
  };
  for (int t = 0; t < n_tiles; t += 2) {
    iter(t, gk, gv, gk2, gv2);
    if (t + 1 < n_tiles) iter(t + 1, gk2, gv2, gk, gv);
  }
#undef FPQ_ATTN_FETCH
#undef FPQ_ATTN_STAGE

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_live) {
    uint16_t* op = a.out + (((int64_t)b * a.lq + qrow) * a.heads + h) * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 attn_h4_t __attribute__((ext_vector_type(4)));
        attn_h4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (_Float16)(o[dt][4 * g + e] * inv);
        *(u32x2*)(op + 32 * dt + 8 * g + 4 * hi) = __builtin_bit_cast(u32x2, w);
      }
  }
}
