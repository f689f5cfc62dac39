/* fpq.h - C ABI of libfpq_hip.so: MI355X (gfx950) fake-quantization kernels that
 * replace FPQVAR's `quant_cuda` extension and the torch-op bodies of its
 * `fp_quant_*` / `fp6_quant_*` functions.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the name says `host`;
 *   - nothing allocates, nothing synchronises the host, everything is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - re-entrant and thread-safe: no mutable global state on any launch path.  The ONE piece of process-wide state is
 *     the table of experiment switches below (fpq_set_option): filled from the environment once, when the library is
 *     loaded, and read as plain ints afterwards - no entry point calls getenv();
 *   - returns FPQ_OK (0) or a negative FPQ_ERR_* code; on error nothing is enqueued;
 *   - rows == 0 / n == 0 is valid and enqueues nothing;
 *   - inputs are never written.
 *
 * "reference" below = PKU-SEC-Lab/FPQVAR; tr/ = models_fp_quant_transform_rotate/.
 */
#ifndef FPQ_H
#define FPQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPQ_VERSION 125 /* 0.1.2: + fpq_quant_tensor_argmin, fpq_quant_rows_segments, fpq_quant_rows_multi (round 2);
                           0.1.3: + fpq_quant_rows_codes_segments, fpq_dequant_rows_codes_segments (round 3);
                           123: + fpq_build_tag (round 4);
                           124: + fpq_set_option, fpq_get_option, fpq_option_name, fpq_gemm_fp4_gelu_dual, fpq_gelu_quant_rows_dual (round 5);
                           125: + k-major operand images: fpq_codes_to_kmajor, fpq_scales_to_kmajor, fpq_gemm_fp4_mx_km, fpq_gemm_fp4_gelu_dual_km,
                                fpq_gemm_fp6_rows_km, the *_km producers, fpq_gemm_fp4_mx_split (round 5) */

typedef void* fpq_stream_t; /* hipStream_t */

enum fpq_status {
  FPQ_OK = 0,
  FPQ_ERR_ARG = -1,    /* NULL pointer, negative size */
  FPQ_ERR_DTYPE = -2,  /* dtype not supported by this entry point */
  FPQ_ERR_SHAPE = -3,  /* cols == 0, k out of range, ... */
  FPQ_ERR_TABLE = -4,  /* unknown table id / half table used as symmetric table */
  FPQ_ERR_LAUNCH = -5, /* hipLaunch reported an error */
  FPQ_ERR_NO_DEVICE = -6
};

enum fpq_dtype { FPQ_F16 = 0, FPQ_F32 = 1, FPQ_F64 = 2 };

/* Built-in value tables, literal in the reference at
 * tr/quant_utils.py:233-235 (E3M0/E2M1/E1M2), :458-486 (E2M3/E3M2, two zeros),
 * :418-419 (E1M2_NEG/E2M1_POS), :488-500 (INT_NEG/E2M3_POS). */
enum fpq_table {
  FPQ_E2M1 = 0,
  FPQ_E1M2 = 1,
  FPQ_E3M0 = 2,
  FPQ_E2M3 = 3,
  FPQ_E3M2 = 4,
  FPQ_E1M2_NEG = 5,
  FPQ_E2M1_POS = 6,
  FPQ_INT_NEG = 7,
  FPQ_E2M3_POS = 8,
  FPQ_E2M1_NEG = 9, /* [-6 .. 0]: negative half of fp4_afpq_per_group_cuda, models_fp_quant/quant_utils.py:501 */
  FPQ_NUM_TABLES = 10
};

int fpq_version(void);
const char* fpq_strerror(int status);
/* Which build this is: "stock" for the shipped library, the name given to tools/build_variant.sh for a diagnostic
 * build (-DFPQ_BUILD_TAG).  The A/B tools assert on it so that a timing is never attributed to the wrong library. */
const char* fpq_build_tag(void);

/* Experiment switches - tests and the A/B tools only; a deployment never touches them.  Every switch is an int named
 * like the environment variable that initialises it ("FPQ_NO_HW4", "FPQ_GEMM_CFG", ...; fpq_option_name enumerates
 * them).  The environment is read ONCE, by the library's initialiser (a variable that is set and not empty: flags take
 * 1 unless the text is "0", numbers take atoi of the text); later changes of the environment are not seen.
 * fpq_set_option changes a switch for every later call in the process (value FPQ_OPTION_DEFAULT = back to the built-in
 * choice); it is a relaxed atomic store - call it between launches, not concurrently with them, if the outcome matters.
 * Both return FPQ_ERR_ARG for an unknown name. */
#define FPQ_OPTION_DEFAULT (-2147483647 - 1)
int fpq_set_option(const char* name, int value);
int fpq_get_option(const char* name, int* value_out);
const char* fpq_option_name(int index); /* NULL past the last one */

/* Host-side copy of a built-in table exactly as the reference spells it
 * (ascending, duplicate zeros kept).  Returns the entry count, or FPQ_ERR_TABLE.
 * `host_out` may be NULL to query the size. */
int fpq_table_values(int table_id, float* host_out);

/* ---- L0: the reference's one native symbol ---------------------------------
 * Replaces quant_forward_cuda / quant_forward_cuda_kernel
 * (quant/quant_kernel.cu:11-62, bound as `quant_cuda.quant` at quant/quant.cpp:27-29).
 *   z[i] = table[j*], j* = LAST j in 0..k-1 minimising fabsf((float)x[i] - table[j])
 *   among distances <= 102400.0f; z[i] = +0.0 if there is none (NaN, +-Inf, far).
 * x, z: `dtype` in {FPQ_F32, FPQ_F64} (f64 compared in f32, as the reference does),
 * n elements, contiguous.  table: k floats on the device, 1 <= k <= 256.
 * The reference's second output (`tensor_idx`, never written, all zeros) is the
 * caller's business: the Python binding returns a zero tensor for it. */
int fpq_quant_nearest(const void* x, const float* table, void* z, int64_t n, int k, int dtype,
                      fpq_stream_t stream);

/* quantize_to_nearest_grid (tr/quant_utils.py:209-230; also the GALT / format-search scripts' copy): the lookup of
 * the reference's pure-torch path for ANY table - z[i] = table[argmin_j |(float)x[i] - table[j]|], FIRST minimal
 * index (the earlier entry on a tie), table[0] for a NaN or +-Inf input; x F16 or F32, z always float32. */
int fpq_quant_nearest_argmin(const void* x, const float* table, float* z, int64_t n, int k, int dtype,
                             fpq_stream_t stream);

/* Same lookup against a BUILT-IN table through the closed form the fused kernels
 * use (midpoint counting on the minifloat structure).  f32 only.  Exposed so
 * tests can compare it with fpq_quant_nearest element by element. */
int fpq_quant_nearest_builtin(const float* x, float* z, int64_t n, int table_id, fpq_stream_t stream);

/* ---- L1: one launch per reference function ---------------------------------
 * x: [rows, cols] contiguous, in_dtype in {F16, F32}.  out: same shape, out_dtype.
 * One scale per row of `cols` elements:
 *     s   = (T)(max|x_row| / max|table|)          (in x's dtype T)
 *     xn  = (T)(x / s)                            (in T)
 *     q   = nearest((float)xn)                    (L0 semantics)
 *     out = (Tout)((float)q * (float)s)           (fp32 product, then cast)
 * - per-group functions call it with cols = group_size (128) and rows = numel/128:
 *     fp_quant_e{1,2,3}_per_group_cuda  tr/quant_utils.py:265-282,313-330,361-378
 *     fp6_quant_{e2m3,e3m2}_per_group_cuda  :537-574   (out_dtype = F16)
 *     KV cache, kv_bit 4                tr/basic_var.py:50-67,197-198
 * - per-token functions call it with cols = last dim:
 *     fp6_quant_{e2m3,e3m2}_per_token_cuda  tr/quant_utils.py:503-534 (out_dtype = F16)
 *     KV cache, kv_bit 6 (cols = 64)    tr/basic_var.py:71-85,194-195
 * table_id must be one of the symmetric tables (E2M1, E1M2, E3M0, E2M3, E3M2). */
int fpq_quant_rows(const void* x, void* out, int64_t rows, int64_t cols, int table_id, int in_dtype,
                   int out_dtype, fpq_stream_t stream);

/* out = softmax(q k^T * scale) v per (batch, head) for head_dim 64, fp16, in the [B, L, H, c] layout the reference
 * passes to flash_attn_func(q, k, v, softmax_scale=self.scale) in SelfAttention.forward (tr/basic_var.py:173,211) -
 * the consumer of the KV cache; at inference there is no mask and no dropout (attn_bias is None with KV caching,
 * :159).  q [B, lq, H, 64] and k / v [B, lkv, H, 64] may be views: rows of H * 64 halves contiguous, batch / token
 * pitch in elements (multiples of 8) free (q out of the fused qkv output, k / v out of fpq_kv_cache_step's cache);
 * out [B, lq, H, 64] contiguous; all 16-byte aligned.  fp32 scores and accumulation on the matrix cores, online
 * softmax; agreement with a reference attention is to fp16 tolerance (P is rounded to fp16 before P V, as in flash
 * attention), not bit-exact. */
int fpq_attention_blhc(const void* q, const void* k, const void* v, void* out, int64_t batch, int64_t lq, int64_t lkv,
                       int64_t heads, int64_t head_dim, int64_t q_batch_pitch, int64_t q_token_pitch,
                       int64_t kv_batch_pitch, int64_t kv_token_pitch, float scale, fpq_stream_t stream);

/* One step of an incrementally maintained KV cache.  The reference
 * (SelfAttention.forward, tr/basic_var.py:186-209) re-quantizes the WHOLE cached K and V at every
 * step before concatenating the new k / v; since the quantizer returns its own output unchanged
 * (tests/test_kv_idempotence.py), only the entries appended by the previous step change.  This
 * call does both halves of a step in one launch:
 *   - fake-quantizes tokens [quant_start, quant_stop) of K and V in place, rows of `group`
 *     consecutive halves sharing one scale (64 = fp6_quant_e2m3_per_token_cuda on head_dim 64,
 *     tr/quant_utils.py:503-517; 128 = fp_quant_e2_per_group_cuda, :313-330), and
 *   - copies the new rows new_k / new_v [batch, n_new, row_elems] to tokens
 *     [new_start, new_start + n_new); their rows must be contiguous, batch and token pitch (in
 *     elements, multiples of 8) are free - views of a fused qkv projection need no copy.
 * cache: fp16 [2 (K, V), batch, max_len, row_elems], 16-byte aligned like new_k / new_v.
 * Requires quant_stop <= new_start and new_start + n_new <= max_len. */
int fpq_kv_cache_step(void* cache, int64_t batch, int64_t max_len, int64_t row_elems, int64_t quant_start,
                      int64_t quant_stop, const void* new_k, const void* new_v, int64_t new_batch_pitch,
                      int64_t new_token_pitch, int64_t new_start, int64_t n_new, int64_t group, int table_id,
                      fpq_stream_t stream);

/* The reference's pure-torch quantizers ("CPU path", also what QuantizedLinear uses on
 * the GPU for per_channel / per_token FP4, tr/quant_utils.py:699-704,796-807):
 *     fp_quant_e{1,2,3}_per_token   tr/quant_utils.py:237-247,285-295,333-343   (clamp3 = 1)
 *     fp_quant_e{1,3}_per_group     :250-262,346-358                            (clamp3 = 1)
 *     fp_quant_e2_per_group         :298-310                                    (clamp3 = 0)
 * Same scale / normalise arithmetic as fpq_quant_rows, but the lookup is
 * quantize_to_nearest_grid (:209-230) = torch.argmin over |x - grid|: a tie goes to the
 * SMALLER value, a NaN or +-Inf normalised value selects grid[0] (so an all-zero row
 * yields -0.0), and the result is float32 whatever the input dtype.  clamp3: clamp x to
 * [-3, 3] first.  The input is never written (the reference's fp_quant_e2_per_group
 * divides its argument in place; that side effect is not reproduced). */
int fpq_quant_rows_argmin(const void* x, float* out, int64_t rows, int64_t cols, int table_id, int in_dtype,
                          int clamp3, fpq_stream_t stream);

/* Asymmetric neg/pos dual format: x <= 0 is scaled and rounded on `neg_table`,
 * x > 0 on `pos_table`, each with its own per-row scale; NaN elements take
 * neither side and come out 0 (torch.where semantics).
 *     fp_quant_e1m2_neg_e2m1_pos_per_group_cuda   tr/quant_utils.py:415-452
 *     fp6_quant_int_neg_e2m3_pos_per_group_cuda   :577-611
 *     fp6_quant_int_neg_e2m3_pos_per_token_cuda   :614-646
 *     fp4_afpq_per_group_cuda (E2M1_NEG / E2M1_POS)   models_fp_quant/quant_utils.py:498-535
 * clip_absmax: NULL, or a device scalar (in_dtype) holding max|x| over the WHOLE
 * tensor as written by fpq_absmax; the kernel then clamps x to
 * +-(T)(clip_strength * absmax) first (tr/quant_utils.py:421-422).  A NaN bound
 * turns every element into NaN, hence the whole output into zeros, exactly as
 * torch.clamp does.
 * nan_flag: NULL, or 8 bytes of device scratch (8-byte aligned) that are ZERO on entry.  With
 * clip_strength == 1.0 the clamp is the identity unless x holds a NaN (then the bound is NaN and
 * the result all zeros); passing scratch here reproduces exactly that WITHOUT the absmax pass: the
 * kernel raises word 0 when it meets a NaN and a second, 64-workgroup launch (which exits at once
 * when the flag is clear) zero-fills `out` and leaves the scratch zero again - allocate and zero it
 * once, reuse it for every call on the same stream (two launches per call, no memset; a captured
 * graph replays correctly).  Calls that may overlap on different streams need their own scratch. */
int fpq_quant_rows_dual(const void* x, void* out, int64_t rows, int64_t cols, int neg_table,
                        int pos_table, int in_dtype, int out_dtype, const void* clip_absmax,
                        float clip_strength, void* nan_flag, fpq_stream_t stream);

/* `fc2.act_quant(act(y))` of the reference's FFN in ONE pass over y (tr/basic_var.py:120-121: act = GELU(approximate="tanh");
 * fc2's input quantizer is bound at tr/quant_utils.py:991 (W4A4: fp_quant_e1m2_neg_e2m1_pos_per_group_cuda) and :930-931 (W6A6:
 * fp6_quant_int_neg_e2m3_pos_per_token_cuda)): out = fpq_quant_rows_dual(half(gelu_tanh(float(y)))) on fp16 rows of `cols`
 * elements - 128 (per group) or the token's row (per token, cols % 8 == 0, <= 16384) - for any dual table pair.  nan_flag: the
 * FP4 pair's global-clamp rule at clipping strength 1.0 (as in fpq_quant_rows_dual), NULL for the FP6 pairs (the reference
 * does not clamp there).  gelu_out: NULL, or fp16 [rows, cols] receiving the GELU values the quantizer saw (the quantization is
 * bit-exact on those; they sit within one fp16 ulp of torch's GELU on every fp16 input).  The same fused tail inside the fc1
 * GEMM: fpq_gemm_fp4_gelu_dual. */
int fpq_gelu_quant_rows_dual(const void* y, void* out, void* gelu_out, int64_t rows, int64_t cols, int neg_table, int pos_table,
                             void* nan_flag, fpq_stream_t stream);

/* The pure-torch twin of the FP4 dual format, fp_quant_e1m2_neg_e2m1_pos_per_group (tr/quant_utils.py:381-412;
 * what models_fp_quant_rotate's QuantizedLinear_fc2 binds, rot/quant_utils.py:779): same split and scales as
 * fpq_quant_rows_dual, but BOTH halves of every element go through the argmin lookup (the other half's input
 * is 0), a tie takes the smaller value, a NaN / +-Inf normalised value takes table[0] (so a group without
 * negatives adds -max|neg table| to every element - the reference's behaviour, reproduced, not fixed), and
 * the result is float32: out = (q_neg + q_pos) * (x <= 0 ? s_neg : s_pos).  clip_absmax as in
 * fpq_quant_rows_dual (the reference always clamps; pass fpq_absmax's result). */
int fpq_quant_rows_dual_argmin(const void* x, float* out, int64_t rows, int64_t cols, int neg_table, int pos_table,
                               int in_dtype, const void* clip_absmax, float clip_strength, fpq_stream_t stream);

/* "Neg reverse" rows: out = (T)( (q(x_nr / s_nr) * s_nr - m) + q(x_pos / s_pos) * s_pos ) with
 * m = |min(row)|, x_nr = (T)(min(x, 0) + m), both scales (T)(absmax / max|table|); products,
 * the subtraction and the sum in fp32.  `dtype` is the type of both x and out.
 *   replaces fp_neg_reverse_quant_per_group_cuda   models_fp_quant/quant_utils.py:454-495
 *   (the reference uses FPQ_E2M1 and cols = 128; any symmetric table and row length work).
 * A NaN in a row makes m NaN, hence the whole row NaN, as torch.min does. */
int fpq_quant_rows_neg_reverse(const void* x, void* out, int64_t rows, int64_t cols, int table_id,
                               int dtype, fpq_stream_t stream);

/* Online rotate fused in front of the per-group(128) quantizer.
 * Replaces, for the block-diagonal randomized-Hadamard rotation
 * (rotate_utils/rotation_utils.py:69-104: every 128x128 block = diag(D).H128/sqrt(128)),
 *     x1 = torch.matmul(producer.mul(s), Q)            tr/basic_var.py:263,266 (fp16 autocast)
 *     q  = fp_quant_e{1,2,3}_per_group_cuda(x1, 4, 128) / fp6_quant_*_per_group_cuda
 * by one launch:  h = half(x * s);  y = half(c_h * FWHT128(h * D));  out = quant(y).
 * x: [rows, cols] F16 or F32 (the producer's output), cols % 128 == 0;
 * smooth: device float[cols] (the GALT factor s of this block) or NULL;
 * sign_mask_host: HOST pointer to 4 x uint32, bit j set <=> D[j] == -1 (seed-42 vector);
 * out: fp16 [rows, cols]; rotated_out: NULL, or fp16 [rows, cols] receiving y.
 * Parity: out == fpq_quant_rows(y) bit for bit; y = half(c_h * sum) with the +-h_j summed in fp32 (on the matrix
 * cores; exact for groups of like magnitude): |y - exact| <= 1/2 fp16 ulp + 2^-22 c_h sum|h_j|, i.e. within 1 fp16
 * ulp of the fp64-accumulated half(x*s) @ half(Q) except on cancelling outputs of groups spanning > 2^13 in magnitude
 * (the reference GEMM's summation order is unspecified, and its +-c_h operands round more).  A non-finite input
 * poisons its own group of 128 only (the reference's dense GEMM with the block-diagonal Q: its whole row, 0 * inf).
 * x may be F32: the kernel then reads 16 bytes per lane all the same.  All pointers 16-byte aligned. */
int fpq_rotate_quant_rows(const void* x, void* out, void* rotated_out, int64_t rows, int64_t cols,
                          int in_dtype, const float* smooth, const uint32_t* sign_mask_host, int table_id,
                          fpq_stream_t stream);

/* The complete producer of tr/basic_var.py:263 / :266 plus the activation quantizer, one launch:
 *     h   = half( ((LayerNorm(x) * half(scale + 1)) + shift) * smooth )    (no affine, eps; fp32 ops in
 *                                                                            the reference's order)
 *     y   = half( c_h * FWHT128(h * D) )          (= h @ half(Q_block))
 *     out = per-group(128) quant(y)
 * x: [rows, cols] F16/F32 (F32 is what the reference's autocast run carries: tr/var.py:209, tr/basic_var.py:264,267),
 * cols % 128 == 0, cols <= 4096; scale, shift: [rows / rows_per_batch, cols]
 * in mod_dtype (the block's AdaLN scale1/shift1 or scale2/shift2, one row per batch element);
 * smooth: device float[cols] or NULL; sign_mask_host as in fpq_rotate_quant_rows.
 * h_out / rotated_out: NULL or fp16 [rows, cols] receiving h / y (for verification).
 * Parity: out == fpq_quant_rows(y) bit for bit and y as in fpq_rotate_quant_rows given h; h itself
 * matches torch's chain up to the fp32 rounding of LayerNorm's mean / rstd (1 fp16 ulp on rare
 * elements) - torch's Welford reduction order is not a contract. */
int fpq_adaln_rotate_quant_rows(const void* x, void* out, void* h_out, void* rotated_out, int64_t rows,
                                int64_t cols, int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                int64_t rows_per_batch, float eps, const float* smooth,
                                const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream);

/* The same producer for the per-token configurations (W6A6, run.sh:7 with --rotate --block_rotate): the quantizer
 * behind the rotation is fp6_quant_{e2m3,e3m2}_per_token_cuda (tr/quant_utils.py:503-534), ONE scale per token row
 * of `cols` channels.  A row lives in one wavefront: cols <= 2560 (d30: 1920, d36: 2304).  row_scales: NULL or fp16
 * [rows] receiving the scales.  Parity as fpq_adaln_rotate_quant_rows: out == fpq_quant_rows(y, cols = row length)
 * bit for bit on the rotated y produced here.  The _codes_fp8 form emits the operand format of fpq_gemm_fp8_rows
 * (E4M3 bytes [rows, cols] + fp16 row scales) instead of values. */
int fpq_adaln_rotate_quant_token_rows(const void* x, void* out, void* h_out, void* rotated_out, void* row_scales,
                                      int64_t rows, int64_t cols, int in_dtype, const void* scale, const void* shift,
                                      int mod_dtype, int64_t rows_per_batch, float eps, const float* smooth,
                                      const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream);
int fpq_adaln_rotate_quant_token_rows_codes_fp8(const void* x, uint8_t* codes, void* row_scales, int64_t rows,
                                                int64_t cols, int in_dtype, const void* scale, const void* shift,
                                                int mod_dtype, int64_t rows_per_batch, float eps, const float* smooth,
                                                const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream);
/* ... and the _codes_fp6 form the operand format of fpq_gemm_fp6_rows (dense 6-bit E2M3 codes [rows, cols * 3 / 4] +
 * fp16 row scales): table_id must be FPQ_E2M3, cols % 32 == 0, codes 8-byte aligned. */
int fpq_adaln_rotate_quant_token_rows_codes_fp6(const void* x, uint8_t* codes, void* row_scales, int64_t rows,
                                                int64_t cols, int in_dtype, const void* scale, const void* shift,
                                                int mod_dtype, int64_t rows_per_batch, float eps, const float* smooth,
                                                const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream);

/* max|x| over n elements (NaN-propagating, like torch's x.abs().max()), written
 * as ONE scalar of `dtype` to `out`.  `out` must hold 4 bytes; it is zeroed on the
 * stream first (hipMemsetAsync) and then combined with device atomics. */
int fpq_absmax(const void* x, int64_t n, int dtype, void* out, fpq_stream_t stream);

/* Many tensors, ONE launch (weight calibration: every Linear a rank owns, QuantizedLinear.from_float's
 * fp_quant_e{1,2,3}_per_group_cuda / fp6_quant_*_per_group_cuda on the fp32 weight, tr/quant_utils.py:828-855,
 * with the driver's later var.half() as the F16 output form).  Same results as fpq_quant_rows per segment.
 * segments_device: DEVICE-resident array of n_segments descriptors (the caller uploads it once and reuses it);
 * every segment is `rows` contiguous rows of `cols` elements, x and out 16-byte aligned; segments may have
 * different row counts, max_rows = the largest (sizes the grid: grid.y = segment, workgroups beyond a shorter
 * segment's end exit at once).  This version: cols == 128, in_dtype F32, out_dtype F16 or F32, n_segments <= 65535.
 * The descriptors' pointers are NOT validated (they live on the device): the caller guarantees them. */
typedef struct {
  const void* x;
  void* out;
  int64_t rows;
} fpq_segment_t;
int fpq_quant_rows_segments(const fpq_segment_t* segments_device, int n_segments, int64_t max_rows, int64_t cols,
                            int table_id, int in_dtype, int out_dtype, fpq_stream_t stream);

/* A few tensors, one call: the reference quantizes independent tensors back to back in places - the cached K and
 * the cached V of a generation step (tr/basic_var.py:192-200: two fp6_quant_e2m3_per_token_cuda / two
 * fp_quant_e2_per_group_cuda calls), the samples of a format search.  segments_host: HOST array (read before the
 * call returns; the descriptors travel in the kernel arguments, nothing is copied to the device), every segment
 * `rows` contiguous rows of `cols` elements; same results as fpq_quant_rows per segment.  ONE launch when
 * in_dtype == out_dtype == F16, n_segments <= 8, cols in {8, 16, ..., 512} and everything 16-byte aligned; otherwise
 * one launch per segment behind the same call. */
int fpq_quant_rows_multi(const fpq_segment_t* segments_host, int n_segments, int64_t cols, int table_id, int in_dtype,
                         int out_dtype, fpq_stream_t stream);

/* Per-tensor quantizer of the reference's pure-torch path (BASELINE.json config 1):
 *   replaces fp_quant_e2_per_tensor   search/baseline/plot_weight_distribution_for_motivation.py:285-294
 *     scale = x.abs().max() / max|table|     (two 0-dim tensors: float32 for F16 and F32 input alike)
 *     out   = table[argmin_j |T(x / scale) - table[j]|] * scale      (float32; T = x's dtype)
 * with torch.argmin's rules (first minimal index = the smaller value on a tie, NaN / +-Inf -> table[0]).
 * x: n elements F16 or F32; out: float32 [n]; scale_out: ONE float32 (the function's second result);
 * workspace: FPQ_TENSOR_WORKSPACE_BYTES of device scratch (per-workgroup maxima; no memset, no atomics).
 * Two launches (maxima, then the elementwise pass).  n == 0 stores scale 0. */
#define FPQ_TENSOR_WORKSPACE_BYTES 8192
int fpq_quant_tensor_argmin(const void* x, float* out, float* scale_out, void* workspace, int64_t n, int table_id,
                            int in_dtype, fpq_stream_t stream);

/* Codeword output (build-defined; the reference's kernel never emits codes,
 * quant_kernel.cu:18,49).  code = index into the sorted, de-duplicated table
 * (0..14 for the FP4 tables, 0..62 for FP6); for the FP4 tables two codes are
 * packed per byte (element 2i in the low nibble).  scales: one per row, in
 * x's dtype.  dequant: table_dedup[code] * scale reproduces fpq_quant_rows. */
int fpq_quant_rows_codes(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols,
                         int table_id, int in_dtype, int pack_nibbles, fpq_stream_t stream);

/* ---- F2: a real FP4 consumer --------------------------------------------
 * Per-group(128) FP4-E2M1 quantization straight to HARDWARE codes: one OCP E2M1 nibble per element
 * (bit 3 sign, bits 2:0 index into {0,.5,1,1.5,2,3,4,6}; element 2i in the low nibble of byte i) and
 * one scale per group in x's dtype.  Same scale / normalise / rounding arithmetic as fpq_quant_rows,
 * i.e. level * scale reproduces fp_quant_e2_per_group_cuda exactly.  x: [rows, cols] F16 or F32,
 * cols % 128 == 0; codes: [rows, cols/2]; scales: [rows, cols/128]. */
int fpq_quant_rows_codes_mx(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols,
                            int in_dtype, fpq_stream_t stream);

/* The fused producers of F1 emitting the FP4 GEMM's operand format directly (hardware E2M1 codes + one fp16
 * scale per 128-group) instead of fake-quantized values: same arguments and the same arithmetic as
 * fpq_rotate_quant_rows / fpq_adaln_rotate_quant_rows with table FPQ_E2M1, i.e. level(code) * scale is bit-equal
 * to their `out`.  codes: [rows, cols/2]; scales: fp16 [rows, cols/128].  With these the activation never exists
 * in HBM as fp16 between the block's LayerNorm and its matrix product (2 B read + 0.53 B written per element). */
int fpq_rotate_quant_rows_codes_mx(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols,
                                   int in_dtype, const float* smooth, const uint32_t* sign_mask_host,
                                   fpq_stream_t stream);
int fpq_adaln_rotate_quant_rows_codes_mx(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols,
                                         int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                         int64_t rows_per_batch, float eps, const float* smooth,
                                         const uint32_t* sign_mask_host, fpq_stream_t stream);

/* out[t, o] = bias[o] + sum_g a_scale[t,g] * w_scale[o,g] * dot_128(levels(a)[t,g,:], levels(w)[o,g,:])
 * on the gfx950 block-scaled FP4 matrix cores (one 16x16x128 MFMA per group and 16x16 output tile,
 * exact products, fp32 accumulation); replaces F.linear(act_quant(x), W_q, b) of
 * tr/quant_utils.py:765-767 for the W4A4 per-group E2M1 configuration.  a_codes [tokens, k/2],
 * a_scales fp16 [tokens, k/128], w_codes [outs, k/2], w_scales [outs, k/128] in w_scale_dtype,
 * bias fp16 [outs] or NULL, out fp16 [tokens, outs]; k % 128 == 0, outs % 8 == 0.
 * Numerics: does NOT round the de-quantized operands to fp16 first as the reference's fp16 GEMM does;
 * agreement with it is to fp16-GEMM tolerance, not bit-exact. */
int fpq_gemm_fp4_mx(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                    int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs, int64_t k,
                    fpq_stream_t stream);

/* ---- F2 for the per-token / per-channel configurations (W6A6, run.sh:7) ----------------------------------------
 * Per-row quantization straight to one OCP FP8 E4M3 byte per element (every level of the symmetric FP4 / FP6 tables
 * is exactly an E4M3 number) + one scale per row in x's dtype: same arithmetic as fpq_quant_rows with cols = row
 * length, i.e. e4m3(code) * scale reproduces fp6_quant_{e2m3,e3m2}_per_token_cuda (tr/quant_utils.py:503-534) and the
 * per-channel weight quantization of QuantizedLinear.from_float (:808-829).  codes: [rows, cols]; scales: [rows]. */
int fpq_quant_rows_codes_fp8(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int table_id,
                             int in_dtype, fpq_stream_t stream);

/* out[t, o] = bias[o] + a_scale[t] * w_scale[o] * sum_k e4m3(a[t,k]) * e4m3(w[o,k]) on the FP8 matrix cores (exact
 * products, fp32 accumulation, nothing but matrix instructions in the K loop: with one scale per row the scales
 * leave the sum); replaces F.linear(act_quant(x), W_q, b) of tr/quant_utils.py:765-767 for per_token activations x
 * per_channel weights.  a_codes [tokens, k], w_codes [outs, k], bias fp16 [outs] or NULL, out fp16 [tokens, outs];
 * k % 128 == 0, outs % 8 == 0, code arrays and out 16-byte aligned.  Tolerance-level agreement with the reference's
 * fp16 GEMM on the de-quantized tensors (as fpq_gemm_fp4_mx). */
int fpq_gemm_fp8_rows(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                      const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs,
                      int64_t k, fpq_stream_t stream);

/* The same pair with the operands in the matrix instruction's 6-bit packed form (FP6 E2M3: sign, 2 exponent bits with
 * bias 1, 3 mantissa bits; element j of a row in bits [6j, 6j+6) of the row's little-endian bit string, i.e. dense
 * packing, 3/4 byte per element): 25 % less operand traffic in a kernel bound by exactly that.  table_id must be
 * FPQ_E2M3, cols % 32 == 0 (GEMM: k % 128 == 0); codes: [rows, cols * 3 / 4]; everything else as the FP8 pair. */
int fpq_quant_rows_codes_fp6(const void* x, uint8_t* codes, void* scales, int64_t rows, int64_t cols, int table_id,
                             int in_dtype, fpq_stream_t stream);
int fpq_gemm_fp6_rows(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                      const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs,
                      int64_t k, fpq_stream_t stream);

/* The three GEMMs with an optional fused tail, for the AdaLN blocks' `x = x + attn(...).mul_(gamma1)` /
 * `x + ffn(...).mul(gamma2)` (tr/basic_var.py:264,267): out = residual + y * gate[t / rows_per_gate, :], y being the
 * fp16 result of the plain call, each operation in fp16 with one rounding (bit-identical to the two torch ops on y).
 * Either pointer may be NULL; `residual` may be `out` itself; both 16-byte aligned.  epilogue == NULL: the plain call. */
typedef struct fpq_gemm_epilogue {
  const void* gate;      /* fp16 [ceil(tokens / rows_per_gate), outs] or NULL (gamma viewed as [B, C]) */
  const void* residual;  /* fp16 [tokens, outs] or NULL */
  int64_t rows_per_gate; /* consecutive token rows sharing a gate row (tokens per batch entry), >= 1 */
} fpq_gemm_epilogue_t;
/* The same tail as a call of its own, for a Linear that runs elsewhere (fc2's fp16 GEMM): out = residual +
 * y * gate[row / rows_per_gate, :], all fp16 [rows, cols] (gate [ceil(rows / rows_per_gate), cols]), cols % 8 == 0,
 * 16-byte aligned; out may be y or residual. */
int fpq_gate_residual(const void* y, const void* gate, const void* residual, void* out, int64_t rows, int64_t cols,
                      int64_t rows_per_gate, fpq_stream_t stream);
int fpq_gemm_fp4_mx_ex(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                       int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs, int64_t k,
                       const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream);
/* fc1 of the AdaLN block's FFN with everything up to fc2's GEMM in its epilogue (tr/basic_var.py:120-121; fc2's input
 * quantizer is bound at tr/quant_utils.py:991 and defined at :415-452):
 *     y   = half(dequant(a) @ dequant(w).T + bias)                         the Linear output of fpq_gemm_fp4_mx, bit for bit
 *     h   = half(gelu_tanh(float(y)))                                      F.gelu(y, approximate="tanh") on the fp16 tensor
 *     out = fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(h, 4, 128)           groups of 128 consecutive outputs, strength 1.0
 * out: fp16 [tokens, outs], outs % 128 == 0.  gelu_out: NULL, or fp16 [tokens, outs] receiving h.  nan_flag: as in
 * fpq_quant_rows_dual (8 bytes of zeroed persistent scratch: "a NaN anywhere in h => the whole result is zero", one tiny
 * second launch; NULL skips the rule).  Parity: out == fpq_quant_rows_dual(h) bit for bit on the h produced here; h within
 * one fp16 ulp of torch's GELU of y on every fp16 input (torch's formula in torch's operation order, the device library's
 * tanh restated).  Replaces three launches and 10 B of HBM traffic per element behind the GEMM. */
int fpq_gemm_fp4_gelu_dual(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                           int w_scale_dtype, const void* bias, void* out, void* gelu_out, int64_t tokens, int64_t outs,
                           int64_t k, void* nan_flag, fpq_stream_t stream);
int fpq_gemm_fp8_rows_ex(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                         const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens,
                         int64_t outs, int64_t k, const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream);
int fpq_gemm_fp6_rows_ex(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                         const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens,
                         int64_t outs, int64_t k, const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream);

/* ---- K-MAJOR OPERAND IMAGES ------------------------------------------------------------------------------------------
 * The FP4 and FP6 matrix-core GEMMs above read ROW-MAJOR code tensors [rows, k/2] / [rows, k*3/4]; per K step of 128
 * elements their LDS-DMA engine then gathers 64 / 96 bytes out of each of 384 rows that lie a whole row apart, and issuing
 * those gathers is what bounds them (profiles/r05_gemm6_stamps.txt, r05_lds_dma_issue.txt: 100 - 150 cycles per 1 KiB
 * piece against 65 - 70 for a contiguous one).  The *_km entry points take the same codes as a K-MAJOR IMAGE instead:
 *
 *     image[(s * image_rows + j) * seg + p * 16 + b]      s = K step (k / 128 of them), j = image row, seg = 64 (FP4) or
 *                                                          96 (FP6) bytes, p = 16-byte chunk inside the segment, b = byte
 *   = codes[row(j), s * seg + c(j, p) * 16 + b]
 *
 *   chunk order (the GEMM's bank-conflict-free LDS image, so that a DMA piece is one linear 1 KiB copy):
 *     FP4: c = p ^ pi(j & 15),  pi(q) = (0, 2, 3, 1)[q >> 2]          FP6: c = (p - ((j >> 3) & 1)) mod 6
 *   activation side: image_rows = rows, row(j) = j (no padding: the image has exactly the row-major tensor's size);
 *   weight side ("dealt", the order the kernels hand a wavefront's 64 outputs to its four 16-row tiles):
 *     image_rows = rows rounded up to 64, row(j) = (j & ~63) + 4 * (j & 15) + ((j >> 4) & 3), zero where row(j) >= rows.
 *
 * The per-group scales of the FP4 GEMM travel the same way - row-major [rows, k/128] costs every tile 90 strided loads and as
 * many LDS writes, 8 % of the kernel - as a K-MAJOR SCALE IMAGE, always fp32:
 *     scale_image[g * image_rows + r] = (float)scales[r, g]        image_rows = rows rounded up to 4 (activation side) or to 64
 *                                                                   (weight side; natural row order, NOT dealt), padding = 0
 * (fp16 -> fp32 is exact: the GEMM multiplies the same numbers).  The FP6 GEMM's one scale per row stays a plain vector.
 *
 * Same arithmetic, same results bit for bit (tests/test_gpu_kmajor.py); bias, out and the epilogue are unchanged.
 * fpq_codes_to_kmajor / fpq_scales_to_kmajor convert row-major codes / scales (weights once at load time; any producer's
 * output as a fallback); the *_km producers below write the activation images directly.  code_bits: 4 or 6;
 * scale_dtype: FPQ_F16 or FPQ_F32; all image pointers 16-byte aligned. */
int fpq_codes_to_kmajor(const uint8_t* codes, uint8_t* image, int64_t rows, int64_t k, int code_bits, int dealt,
                        fpq_stream_t stream);
int fpq_scales_to_kmajor(const void* scales, int scale_dtype, float* image, int64_t rows, int64_t groups, int weight_side,
                         fpq_stream_t stream);
/* fpq_gemm_fp4_mx_ex / fpq_gemm_fp4_gelu_dual / fpq_gemm_fp6_rows_ex on k-major images (a_image: activation side,
 * w_image: dealt weight side).  FP4: a_scales / w_scales are k-major SCALE images (fp32: w_scale_dtype must be FPQ_F32;
 * tokens, outs < 2^28); FP6: the row scale vectors as before.  bias must be 8-byte aligned; k limited by the LDS-DMA
 * kernels' scale tiles (FP4: the row-major entry point falls back to a register-staged kernel for very long k, this one
 * returns FPQ_ERR_SHAPE). */
int fpq_gemm_fp4_mx_km(const uint8_t* a_image, const void* a_scales, const uint8_t* w_image, const void* w_scales,
                       int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs, int64_t k,
                       const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream);
int fpq_gemm_fp4_gelu_dual_km(const uint8_t* a_image, const void* a_scales, const uint8_t* w_image, const void* w_scales,
                              int w_scale_dtype, const void* bias, void* out, void* gelu_out, int64_t tokens, int64_t outs,
                              int64_t k, void* nan_flag, fpq_stream_t stream);
/* The activation producers writing the k-major images directly (same arguments as the forms without _km; image:
 * rows * cols / 2 bytes (FP4) or rows * cols * 3 / 4 (FP6), below 2 GiB, 16-byte aligned; cols % 128 == 0).  The FP4
 * producers' `scales` is the fp32 k-major scale image [cols / 128][rows rounded up to 4] (padding rows are not written);
 * the FP6 producers' `scales` / `row_scales` stay one value per row.
 * fpq_quant_rows_codes_mx_km: fp16 rows only.  The fused adaLN / rotation producers: the matrix-core forms only (rows of up
 * to 2560 channels for adaLN), FPQ_ERR_SHAPE otherwise - use the row-major producer + fpq_codes_to_kmajor there. */
int fpq_quant_rows_codes_mx_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols, int in_dtype,
                               fpq_stream_t stream);
int fpq_rotate_quant_rows_codes_mx_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols, int in_dtype,
                                      const float* smooth, const uint32_t* sign_mask_host, fpq_stream_t stream);
int fpq_adaln_rotate_quant_rows_codes_mx_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols,
                                            int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                            int64_t rows_per_batch, float eps, const float* smooth,
                                            const uint32_t* sign_mask_host, fpq_stream_t stream);
int fpq_quant_rows_codes_fp6_km(const void* x, uint8_t* image, void* scales, int64_t rows, int64_t cols, int table_id,
                                int in_dtype, fpq_stream_t stream);
int fpq_adaln_rotate_quant_token_rows_codes_fp6_km(const void* x, uint8_t* image, void* row_scales, int64_t rows, int64_t cols,
                                                   int in_dtype, const void* scale, const void* shift, int mod_dtype,
                                                   int64_t rows_per_batch, float eps, const float* smooth,
                                                   const uint32_t* sign_mask_host, int table_id, fpq_stream_t stream);
/* The FP4 GEMM with a SPLIT OUTPUT: the outs columns are n_parts (<= 3) parts of part_cols (% 128 == 0) each, and every part has
 * its own destination: token t = b * rows_per_batch + l of part p is written to row  b * batch_stride[p] + row0[p] + l  of out[p]
 * (fp16, row_stride[p] elements per row, >= part_cols, % 4 == 0, 8-byte aligned).  What it is for: mat_qkv of an attention block
 * (tr/basic_var.py:173-209) - q to its own [B, L, H * c] tensor, k and v straight into the KV cache's slots [B, max_len, H * c] at
 * token position `len` (batch_stride = max_len, row0 = len), so that the cache's copy-in pass (60 % of the bytes
 * fpq_kv_cache_step moves) never runs: fpq_kv_cache_step then only quantizes the previous step's entries (n = 0).
 * Same arithmetic as fpq_gemm_fp4_mx_ex / _km (kmajor: 0 row-major operands, 1 k-major images); no gate / residual tail. */
typedef struct fpq_gemm_split {
  int64_t part_cols;
  int32_t n_parts;
  void* out[3];
  int64_t row_stride[3];
  int64_t rows_per_batch;
  int64_t batch_stride[3];
  int64_t row0[3];
} fpq_gemm_split_t;
int fpq_gemm_fp4_mx_split(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales, int w_scale_dtype,
                          const void* bias, int64_t tokens, int64_t outs, int64_t k, const fpq_gemm_split_t* split, int kmajor,
                          fpq_stream_t stream);
int fpq_gemm_fp6_rows_km(const uint8_t* a_image, const void* a_scales, int a_scale_dtype, const uint8_t* w_image,
                         const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens,
                         int64_t outs, int64_t k, const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream);

/* Inverse of fpq_quant_rows_codes: out = (Tout)((float)table_dedup[code] * (float)scale). */
int fpq_dequant_rows_codes(const uint8_t* codes, const void* scales, void* out, int64_t rows,
                           int64_t cols, int table_id, int scale_dtype, int out_dtype,
                           int pack_nibbles, fpq_stream_t stream);

/* The two above over many tensors in ONE launch each: the packed exchange format of the sharded weight calibration
 * (every Linear a rank owns -> nibble codes + one scale per group straight into its slot of the all-gather slab; after
 * the gather every rank decodes all layers of all ranks; reference: quantize_VAR, tr/quant_utils.py:1095-1167 over
 * QuantizedLinear.from_float :828-837, whose result this reproduces bit for bit at 0.53 B per element on the wire).
 * segments_device: DEVICE-resident array of n_segments descriptors, as for fpq_quant_rows_segments; every segment is
 * `rows` groups of `cols` elements; x / codes / out 16-byte aligned, scales aligned to their dtype; max_rows = the
 * largest segment (sizes grid.x; grid.y = segment).  This version: cols == 128, n_segments <= 65535; scales have
 * in_dtype (quantize) / scale_dtype (decode).  Same results as the single-tensor calls per segment.  The descriptors'
 * pointers are NOT validated (they live on the device): the caller guarantees them. */
typedef struct {
  const void* x;
  uint8_t* codes;
  void* scales;
  int64_t rows;
} fpq_codes_segment_t;
int fpq_quant_rows_codes_segments(const fpq_codes_segment_t* segments_device, int n_segments, int64_t max_rows,
                                  int64_t cols, int table_id, int in_dtype, int pack_nibbles, fpq_stream_t stream);
typedef struct {
  const uint8_t* codes;
  const void* scales;
  void* out;
  int64_t rows;
} fpq_decode_segment_t;
int fpq_dequant_rows_codes_segments(const fpq_decode_segment_t* segments_device, int n_segments, int64_t max_rows,
                                    int64_t cols, int table_id, int scale_dtype, int out_dtype, int pack_nibbles,
                                    fpq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FPQ_H */
