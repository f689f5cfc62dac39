// quant_cuda_ext.cpp - the compiled host binding of the boundary: fpqvar_amd/_native.*.so.
//
// The reference's `quant_cuda` is a compiled pybind extension (quant/quant.cpp:17-29) and its quant_utils.py functions are
// Python around it.  Round 2's binding of this library was Python + ctypes: 8 us of host time per eager call against
// ~4 us for the HIP launch itself (profiles/r02_small_steps.json) - and the early scale steps of a generation are small
// launches.  This module is what a torch extension of the reference's kind looks like on top of the C ABI
// (include/fpq.h, libfpq_hip.so): argument checks, output allocation, torch's current stream, ONE C call.  It exports
//   quant(x, y) -> (z, idx)                                   quant/quant.cpp:27-29
//   the reference-named per-group / per-token / dual functions  tr/quant_utils.py:265-282,313-330,361-378,415-452,503-646
//   quant_rows / quant_rows_dual                               the generic forms fpqvar_amd.ops wraps
// PyTorch is plumbing here (tensors in, tensors out, device memory, streams); no kernel and no arithmetic lives in this
// file.  fpqvar_amd.quant_utils / quant_cuda bind these when the module is built and fall back to the ctypes path
// otherwise (same C entry points either way; tests cover both).
#include <torch/extension.h>
// torch-ROCm presents its HIP devices as "cuda": the guard and stream accessors of that masquerade
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <array>
#include <map>
#include <mutex>
#include <utility>

#include "fpq.h"

namespace {

int dtype_id(at::ScalarType t, const char* what) {
  switch (t) {
    case at::kHalf: return FPQ_F16;
    case at::kFloat: return FPQ_F32;
    case at::kDouble: return FPQ_F64;
    default: TORCH_CHECK(false, what, ": unsupported dtype ", t);
  }
}

void check(int status, const char* what) {
  TORCH_CHECK(status == 0, what, ": fpq error ", status, ": ", fpq_strerror(status));
}

void require_gpu(const at::Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), what, ": expected a tensor on the GPU, got device ", t.device(),
              " (fpqvar_amd has no CPU path; the CPU restatement lives in oracle/ for tests only)");
}

fpq_stream_t current_stream(const at::Tensor& x) {
  return (fpq_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream();
}

// ---- quant_cuda.quant (quant/quant.cpp:17-29, quant/quant_kernel.cu:42-62) ----
std::tuple<at::Tensor, at::Tensor> quant(const at::Tensor& x, const at::Tensor& y) {
  require_gpu(x, "quant_nearest(x)");
  require_gpu(y, "quant_nearest(table)");
  TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kDouble, "quant_nearest: x must be float32 or float64, got ",
              x.scalar_type());
  TORCH_CHECK(x.is_contiguous(), "quant_nearest: x must be contiguous");
  TORCH_CHECK(y.device() == x.device(), "quant_nearest: table must live on x's device");
  const int64_t k = y.numel();
  TORCH_CHECK(k >= 1 && k <= 256, "quant_nearest: table must hold 1..256 entries, got ", k);
  const at::Tensor tab = (y.scalar_type() == at::kFloat && y.is_contiguous() && y.dim() == 1) ? y : y.detach().reshape({-1}).to(at::kFloat).contiguous();
  at::Tensor z = at::empty_like(x);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  check(fpq_quant_nearest(x.data_ptr(), (const float*)tab.data_ptr(), z.data_ptr(), x.numel(), (int)k,
                          dtype_id(x.scalar_type(), "quant_nearest"), current_stream(x)), "fpq_quant_nearest");
  // the reference's second output is allocated and never written (quant_kernel.cu:18,49): a zero-stride view of one zero
  at::Tensor idx = at::zeros({}, x.options()).expand(x.sizes());
  return {z, idx};
}

// ---- one scale per row of `cols` elements, symmetric table ----
at::Tensor quant_rows(const at::Tensor& x, int64_t table_id, int64_t cols, c10::optional<at::ScalarType> out_dtype) {
  require_gpu(x, "quant_rows");
  TORCH_CHECK(x.scalar_type() == at::kHalf || x.scalar_type() == at::kFloat, "quant_rows: x must be float16 or float32, got ",
              x.scalar_type());
  const at::ScalarType od = out_dtype.value_or(x.scalar_type());
  const int64_t n = x.numel();
  TORCH_CHECK(cols > 0 && n % cols == 0, "quant_rows: numel ", n, " is not a multiple of the row length ", cols);
  const at::Tensor xc = x.is_contiguous() ? x : x.contiguous();   // the reference reshapes (copying when needed) before its kernel
  at::Tensor out = at::empty(x.sizes(), x.options().dtype(od));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  check(fpq_quant_rows(xc.data_ptr(), out.data_ptr(), n / cols, cols, (int)table_id, dtype_id(x.scalar_type(), "quant_rows"),
                       dtype_id(od, "quant_rows"), current_stream(x)), "fpq_quant_rows");
  return out;
}

// 8 zeroed bytes per (device, stream) for fpq_quant_rows_dual's NaN flag (include/fpq.h): the fix-up launch leaves
// them zero, so one allocation serves every call on that stream.  A stream that is being captured gets scratch of its
// own per capture-time call: a graph bakes the pointer in and may be replayed beside eager calls on the same stream.
at::Tensor nan_scratch(const at::Tensor& x, fpq_stream_t st) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
    static std::mutex m;
    static std::vector<at::Tensor> keep;   // lives as long as the graphs that reference it may
    at::Tensor t;
    {
      c10::hip::HIPStreamCaptureModeGuard relaxed(hipStreamCaptureModeRelaxed);
      t = at::zeros({2}, x.options().dtype(at::kInt));
    }
    std::lock_guard<std::mutex> lock(m);
    keep.push_back(t);
    return t;
  }
  static std::mutex m;
  static std::map<std::pair<int, void*>, at::Tensor> cache;
  std::lock_guard<std::mutex> lock(m);
  auto key = std::make_pair((int)x.device().index(), (void*)st);
  auto it = cache.find(key);
  if (it == cache.end()) it = cache.emplace(key, at::zeros({2}, x.options().dtype(at::kInt))).first;
  return it->second;
}

at::Tensor quant_rows_dual(const at::Tensor& x, int64_t neg_id, int64_t pos_id, int64_t cols, c10::optional<double> clipping_strength,
                           c10::optional<at::ScalarType> out_dtype) {
  require_gpu(x, "quant_rows_dual");
  TORCH_CHECK(x.scalar_type() == at::kHalf || x.scalar_type() == at::kFloat, "quant_rows_dual: x must be float16 or float32, got ",
              x.scalar_type());
  const at::ScalarType od = out_dtype.value_or(x.scalar_type());
  const int64_t n = x.numel();
  TORCH_CHECK(cols > 0 && n % cols == 0, "quant_rows_dual: numel ", n, " is not a multiple of the row length ", cols);
  const at::Tensor xc = x.is_contiguous() ? x : x.contiguous();
  at::Tensor out = at::empty(x.sizes(), x.options().dtype(od));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  const fpq_stream_t st = current_stream(x);
  const int in_id = dtype_id(x.scalar_type(), "quant_rows_dual");
  const void* clip_ptr = nullptr;
  float strength = 1.0f;
  void* flag_ptr = nullptr;
  at::Tensor scratch, amax;
  if (clipping_strength.has_value() && (float)*clipping_strength == 1.0f) {
    scratch = nan_scratch(x, st);
    flag_ptr = scratch.data_ptr();
  } else if (clipping_strength.has_value()) {
    amax = at::empty({x.scalar_type() == at::kHalf ? 2 : 1}, x.options());   // fpq_absmax writes through 4 bytes
    check(fpq_absmax(xc.data_ptr(), n, in_id, amax.data_ptr(), st), "fpq_absmax");
    clip_ptr = amax.data_ptr();
    strength = (float)*clipping_strength;
  }
  const int status = fpq_quant_rows_dual(xc.data_ptr(), out.data_ptr(), n / cols, cols, (int)neg_id, (int)pos_id, in_id,
                                         dtype_id(od, "quant_rows_dual"), clip_ptr, strength, flag_ptr, st);
  if (status != 0 && flag_ptr) (void)hipMemsetAsync(flag_ptr, 0, 8, (hipStream_t)st);   // a failed launch must not leave the words raised
  check(status, "fpq_quant_rows_dual");
  return out;
}

// the reference's `assert n_bits == 4` / `== 6` (tr/quant_utils.py:266,314,362,416,504,...): an AssertionError, as there
void assert_bits(int64_t n_bits, int64_t want) {
  if (n_bits != want) {
    PyErr_SetString(PyExc_AssertionError, want == 4 ? "n_bits == 4" : "n_bits == 6");
    throw py::error_already_set();
  }
}

// ---- the reference's names (tr/quant_utils.py); table ids: include/fpq.h ----
at::Tensor fp_quant_e3_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size) {
  assert_bits(n_bits, 4);
  return quant_rows(x, FPQ_E3M0, group_size, c10::nullopt);
}
at::Tensor fp_quant_e2_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size) {
  assert_bits(n_bits, 4);
  return quant_rows(x, FPQ_E2M1, group_size, c10::nullopt);
}
at::Tensor fp_quant_e1_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size) {
  assert_bits(n_bits, 4);
  return quant_rows(x, FPQ_E1M2, group_size, c10::nullopt);
}
at::Tensor fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size, double clipping_strength) {
  assert_bits(n_bits, 4);
  return quant_rows_dual(x, FPQ_E1M2_NEG, FPQ_E2M1_POS, group_size, clipping_strength, c10::nullopt);
}
at::Tensor fp4_afpq_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size, double clipping_strength) {
  assert_bits(n_bits, 4);
  return quant_rows_dual(x, FPQ_E2M1_NEG, FPQ_E2M1_POS, group_size, clipping_strength, c10::nullopt);
}
at::Tensor fp6_quant_e2m3_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size) {
  assert_bits(n_bits, 6);
  return quant_rows(x, FPQ_E2M3, group_size, at::kHalf);
}
at::Tensor fp6_quant_e3m2_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size) {
  assert_bits(n_bits, 6);
  return quant_rows(x, FPQ_E3M2, group_size, at::kHalf);
}
at::Tensor fp6_quant_int_neg_e2m3_pos_per_group_cuda(const at::Tensor& x, int64_t n_bits, int64_t group_size) {
  assert_bits(n_bits, 6);
  return quant_rows_dual(x, FPQ_INT_NEG, FPQ_E2M3_POS, group_size, c10::nullopt, c10::nullopt);
}
// per token: the reference flattens with .view(-1), which torch refuses for some non-contiguous layouts
// (fpqvar_amd.quant_utils._require_viewable decides with torch's own stride rule): contiguous input only here
at::Tensor fp6_quant_per_token_contig(const at::Tensor& x, int64_t n_bits, int64_t table_id) {
  assert_bits(n_bits, 6);
  TORCH_CHECK(x.is_contiguous() && x.dim() >= 1, "fp6 per token (native): contiguous input only");
  return quant_rows(x, table_id, x.size(-1), at::kHalf);
}
at::Tensor fp6_quant_int_neg_e2m3_pos_per_token_contig(const at::Tensor& x, int64_t n_bits) {
  assert_bits(n_bits, 6);
  TORCH_CHECK(x.is_contiguous() && x.dim() >= 1, "fp6 dual per token (native): contiguous input only");
  return quant_rows_dual(x, FPQ_INT_NEG, FPQ_E2M3_POS, x.size(-1), c10::nullopt, c10::nullopt);
}

// ---- the fused producers (tr/basic_var.py:263,266 + tr/quant_utils.py:765), value output, contiguous arguments: the hot
// calls of a generation step; everything else (intermediates, operand outputs, odd layouts) stays with rotation.py ----
at::Tensor rotate_quant(const at::Tensor& x, int64_t table_id, const std::array<uint32_t, 4>& sign_mask, const c10::optional<at::Tensor>& smooth) {
  require_gpu(x, "rotate_quant");
  TORCH_CHECK(x.scalar_type() == at::kHalf || x.scalar_type() == at::kFloat, "rotate_quant: x must be float16 or float32, got ", x.scalar_type());
  TORCH_CHECK(x.dim() >= 1 && x.size(-1) % 128 == 0, "rotate_quant: the last dimension must be a multiple of 128");
  const int64_t c = x.size(-1);
  const at::Tensor xc = x.is_contiguous() ? x : x.contiguous();
  const float* sm = nullptr;
  if (smooth.has_value()) {
    TORCH_CHECK(smooth->scalar_type() == at::kFloat && smooth->is_contiguous() && smooth->numel() == c && smooth->device() == x.device(),
                "rotate_quant (native): smooth must be a contiguous float32 [C] tensor on x's device");
    sm = (const float*)smooth->data_ptr();
  }
  at::Tensor out = at::empty(x.sizes(), x.options().dtype(at::kHalf));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  check(fpq_rotate_quant_rows(xc.data_ptr(), out.data_ptr(), nullptr, x.numel() / c, c, dtype_id(x.scalar_type(), "rotate_quant"), sm,
                              sign_mask.data(), (int)table_id, current_stream(x)), "fpq_rotate_quant_rows");
  return out;
}

at::Tensor adaln_rotate_quant(const at::Tensor& x, const at::Tensor& scale, const at::Tensor& shift, int64_t table_id,
                              const std::array<uint32_t, 4>& sign_mask, const c10::optional<at::Tensor>& smooth, double eps) {
  require_gpu(x, "adaln_rotate_quant");
  TORCH_CHECK(x.dim() == 3, "adaln_rotate_quant: x must be [B, L, C]");
  TORCH_CHECK(x.scalar_type() == at::kHalf || x.scalar_type() == at::kFloat, "adaln_rotate_quant: x must be float16 or float32, got ", x.scalar_type());
  const int64_t b = x.size(0), l = x.size(1), c = x.size(2);
  TORCH_CHECK(c % 128 == 0 && c <= 4096, "adaln_rotate_quant: C must be a multiple of 128 and at most 4096");
  TORCH_CHECK(scale.scalar_type() == shift.scalar_type() && (scale.scalar_type() == at::kHalf || scale.scalar_type() == at::kFloat),
              "adaln_rotate_quant: scale and shift must both be float16 or both float32");
  TORCH_CHECK(x.is_contiguous() && scale.is_contiguous() && shift.is_contiguous() && scale.numel() == b * c && shift.numel() == b * c &&
              scale.device() == x.device() && shift.device() == x.device(),
              "adaln_rotate_quant (native): contiguous x and [B, C] modulation rows on x's device");
  const float* sm = nullptr;
  if (smooth.has_value()) {
    TORCH_CHECK(smooth->scalar_type() == at::kFloat && smooth->is_contiguous() && smooth->numel() == c && smooth->device() == x.device(),
                "adaln_rotate_quant (native): smooth must be a contiguous float32 [C] tensor on x's device");
    sm = (const float*)smooth->data_ptr();
  }
  at::Tensor out = at::empty(x.sizes(), x.options().dtype(at::kHalf));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  check(fpq_adaln_rotate_quant_rows(x.data_ptr(), out.data_ptr(), nullptr, nullptr, b * l, c, dtype_id(x.scalar_type(), "adaln_rotate_quant"),
                                    scale.data_ptr(), shift.data_ptr(), dtype_id(scale.scalar_type(), "adaln_rotate_quant"), l, (float)eps, sm,
                                    sign_mask.data(), (int)table_id, current_stream(x)), "fpq_adaln_rotate_quant_rows");
  return out;
}

// ---- the "Q" path: the same producers emitting what the matrix-core GEMMs consume, the KV-cache step and the FP4 GEMM
// (rotation.py / gemm.py / ops.py hold the general forms and dispatch here for the usual arguments) ----
struct ProducerArgs {
  int64_t b, l, c;
  const float* smooth;
};

ProducerArgs producer_checks(const char* what, const at::Tensor& x, const at::Tensor* scale, const at::Tensor* shift,
                             const c10::optional<at::Tensor>& smooth, int64_t max_c) {
  require_gpu(x, what);
  TORCH_CHECK(x.scalar_type() == at::kHalf || x.scalar_type() == at::kFloat, what, ": x must be float16 or float32, got ", x.scalar_type());
  TORCH_CHECK(x.is_contiguous(), what, " (native): contiguous x");
  ProducerArgs a{1, 1, 0, nullptr};
  if (scale) {
    TORCH_CHECK(x.dim() == 3, what, ": x must be [B, L, C]");
    a.b = x.size(0), a.l = x.size(1), a.c = x.size(2);
    TORCH_CHECK(scale->scalar_type() == shift->scalar_type() && (scale->scalar_type() == at::kHalf || scale->scalar_type() == at::kFloat),
                what, ": scale and shift must both be float16 or both float32");
    TORCH_CHECK(scale->is_contiguous() && shift->is_contiguous() && scale->numel() == a.b * a.c && shift->numel() == a.b * a.c &&
                scale->device() == x.device() && shift->device() == x.device(), what, " (native): [B, C] modulation rows on x's device");
  } else {
    TORCH_CHECK(x.dim() >= 1, what, ": x must have a last dimension");
    a.c = x.size(-1);
    a.l = a.c ? x.numel() / a.c : 0;
  }
  TORCH_CHECK(a.c % 128 == 0 && a.c <= max_c, what, ": C must be a multiple of 128 and at most ", max_c);
  if (smooth.has_value()) {
    TORCH_CHECK(smooth->scalar_type() == at::kFloat && smooth->is_contiguous() && smooth->numel() == a.c && smooth->device() == x.device(),
                what, " (native): smooth must be a contiguous float32 [C] tensor on x's device");
    a.smooth = (const float*)smooth->data_ptr();
  }
  return a;
}

// FP4 operand codes: row-major [rows, C / 2], or (kmajor) the activation side's k-major image [C / 128, rows, 64] (include/fpq.h)
static at::Tensor mx_codes_tensor(const at::Tensor& x, int64_t rows, int64_t c, bool kmajor) {
  return kmajor ? at::empty({c / 128, rows, 64}, x.options().dtype(at::kByte)) : at::empty({rows, c / 2}, x.options().dtype(at::kByte));
}
// ... and their scales: fp16 [rows, C / 128], or (kmajor) the fp32 k-major scale image [C / 128, rows rounded up to 4], padding zeroed
static at::Tensor mx_scales_tensor(const at::Tensor& x, int64_t rows, int64_t c, bool kmajor) {
  if (!kmajor) return at::empty({rows, c / 128}, x.options().dtype(at::kHalf));
  const int64_t pad = (rows + 3) / 4 * 4;
  at::Tensor t = at::empty({c / 128, pad}, x.options().dtype(at::kFloat));
  if (pad != rows) t.narrow(1, rows, pad - rows).zero_();
  return t;
}

std::tuple<at::Tensor, at::Tensor> rotate_quant_mx(const at::Tensor& x, const std::array<uint32_t, 4>& sign_mask,
                                                   const c10::optional<at::Tensor>& smooth, bool kmajor) {
  const ProducerArgs a = producer_checks("rotate_quant_mx", x, nullptr, nullptr, smooth, 1 << 30);
  const int64_t rows = a.l;
  at::Tensor codes = mx_codes_tensor(x, rows, a.c, kmajor);
  at::Tensor scales = mx_scales_tensor(x, rows, a.c, kmajor);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  check((kmajor ? fpq_rotate_quant_rows_codes_mx_km : fpq_rotate_quant_rows_codes_mx)(
            x.data_ptr(), (uint8_t*)codes.data_ptr(), scales.data_ptr(), rows, a.c, dtype_id(x.scalar_type(), "rotate_quant_mx"), a.smooth,
            sign_mask.data(), current_stream(x)),
        kmajor ? "fpq_rotate_quant_rows_codes_mx_km" : "fpq_rotate_quant_rows_codes_mx");
  return {codes, scales};
}

std::tuple<at::Tensor, at::Tensor> adaln_rotate_quant_mx(const at::Tensor& x, const at::Tensor& scale, const at::Tensor& shift,
                                                         const std::array<uint32_t, 4>& sign_mask,
                                                         const c10::optional<at::Tensor>& smooth, double eps, bool kmajor) {
  const ProducerArgs a = producer_checks("adaln_rotate_quant_mx", x, &scale, &shift, smooth, 4096);
  const int64_t rows = a.b * a.l;
  at::Tensor codes = mx_codes_tensor(x, rows, a.c, kmajor);
  at::Tensor scales = mx_scales_tensor(x, rows, a.c, kmajor);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  check((kmajor ? fpq_adaln_rotate_quant_rows_codes_mx_km : fpq_adaln_rotate_quant_rows_codes_mx)(
            x.data_ptr(), (uint8_t*)codes.data_ptr(), scales.data_ptr(), rows, a.c, dtype_id(x.scalar_type(), "adaln_rotate_quant_mx"),
            scale.data_ptr(), shift.data_ptr(), dtype_id(scale.scalar_type(), "adaln_rotate_quant_mx"), a.l, (float)eps, a.smooth,
            sign_mask.data(), current_stream(x)),
        kmajor ? "fpq_adaln_rotate_quant_rows_codes_mx_km" : "fpq_adaln_rotate_quant_rows_codes_mx");
  return {codes, scales};
}

// per-token configurations (W6A6): values, or the operands of the row-scaled GEMMs - code_bits 8: E4M3 bytes, 6: dense E2M3
at::Tensor adaln_rotate_quant_token(const at::Tensor& x, const at::Tensor& scale, const at::Tensor& shift, int64_t table_id,
                                    const std::array<uint32_t, 4>& sign_mask, const c10::optional<at::Tensor>& smooth, double eps) {
  const ProducerArgs a = producer_checks("adaln_rotate_quant_token", x, &scale, &shift, smooth, 2560);
  at::Tensor out = at::empty(x.sizes(), x.options().dtype(at::kHalf));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  check(fpq_adaln_rotate_quant_token_rows(x.data_ptr(), out.data_ptr(), nullptr, nullptr, nullptr, a.b * a.l, a.c,
                                          dtype_id(x.scalar_type(), "adaln_rotate_quant_token"), scale.data_ptr(), shift.data_ptr(),
                                          dtype_id(scale.scalar_type(), "adaln_rotate_quant_token"), a.l, (float)eps, a.smooth,
                                          sign_mask.data(), (int)table_id, current_stream(x)), "fpq_adaln_rotate_quant_token_rows");
  return out;
}

std::tuple<at::Tensor, at::Tensor> adaln_rotate_quant_token_codes(const at::Tensor& x, const at::Tensor& scale, const at::Tensor& shift,
                                                                  int64_t table_id, int64_t code_bits,
                                                                  const std::array<uint32_t, 4>& sign_mask,
                                                                  const c10::optional<at::Tensor>& smooth, double eps, bool kmajor) {
  const ProducerArgs a = producer_checks("adaln_rotate_quant_token_codes", x, &scale, &shift, smooth, 2560);
  TORCH_CHECK(code_bits == 8 || (code_bits == 6 && table_id == FPQ_E2M3), "adaln_rotate_quant_token_codes: E4M3 bytes (8) or dense E2M3 codes (6)");
  TORCH_CHECK(!kmajor || code_bits == 6, "adaln_rotate_quant_token_codes: kmajor is a layout of the dense 6-bit codes");
  const int64_t rows = a.b * a.l;
  at::Tensor codes = kmajor ? at::empty({a.c / 128, rows, 96}, x.options().dtype(at::kByte))   // the activation side's k-major image (include/fpq.h)
                            : at::empty({rows, code_bits == 8 ? a.c : a.c * 3 / 4}, x.options().dtype(at::kByte));
  at::Tensor scales = at::empty({rows}, x.options().dtype(at::kHalf));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
  const auto fn = code_bits == 8 ? fpq_adaln_rotate_quant_token_rows_codes_fp8
                  : kmajor       ? fpq_adaln_rotate_quant_token_rows_codes_fp6_km
                                 : fpq_adaln_rotate_quant_token_rows_codes_fp6;
  check(fn(x.data_ptr(), (uint8_t*)codes.data_ptr(), scales.data_ptr(), rows, a.c, dtype_id(x.scalar_type(), "adaln_rotate_quant_token_codes"),
           scale.data_ptr(), shift.data_ptr(), dtype_id(scale.scalar_type(), "adaln_rotate_quant_token_codes"), a.l, (float)eps, a.smooth,
           sign_mask.data(), (int)table_id, current_stream(x)), "fpq_adaln_rotate_quant_token_rows_codes");
  return {codes, scales};
}

// one step of the incrementally kept KV cache (ops.kv_cache_step holds the argument checks' prose)
void kv_cache_step(const at::Tensor& cache, int64_t quant_start, int64_t quant_stop, const at::Tensor& k, const at::Tensor& v,
                   int64_t new_start, int64_t group, int64_t table_id) {
  require_gpu(cache, "kv_cache_step");
  TORCH_CHECK(cache.scalar_type() == at::kHalf && k.scalar_type() == at::kHalf && v.scalar_type() == at::kHalf,
              "kv_cache_step: cache, k and v must be float16");
  TORCH_CHECK(cache.dim() == 5 && cache.size(0) == 2 && cache.is_contiguous(), "kv_cache_step: cache must be a contiguous [2, B, max_len, H, c] tensor");
  const int64_t B = cache.size(1), max_len = cache.size(2), H = cache.size(3), c = cache.size(4);
  TORCH_CHECK(k.sizes() == v.sizes() && k.dim() == 4 && k.size(0) == B && k.size(2) == H && k.size(3) == c && k.device() == cache.device() &&
              v.device() == cache.device(), "kv_cache_step: k / v must be [B, n, H, c] on the cache's device");
  const int64_t n = k.size(1);
  if (n) {
    TORCH_CHECK(k.stride(3) == 1 && k.stride(2) == c && v.stride(3) == 1 && v.stride(2) == c, "kv_cache_step: the (H, c) rows of k / v must be contiguous");
    TORCH_CHECK(k.strides() == v.strides(), "kv_cache_step: k and v must share their strides");
  }
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(cache.device());
  check(fpq_kv_cache_step(cache.data_ptr(), B, max_len, H * c, quant_start, quant_stop, k.data_ptr(), v.data_ptr(), n ? k.stride(0) : 0,
                          n ? k.stride(1) : 0, new_start, n, (int)group, (int)table_id, current_stream(cache)), "fpq_kv_cache_step");
}

void check_operand(const char* what, const at::Tensor& codes, const at::Tensor& scales, int64_t rows, int64_t row_bytes, int64_t n_scales,
                   const at::Device& dev) {
  TORCH_CHECK(codes.scalar_type() == at::kByte && codes.is_contiguous() && codes.device() == dev, what, ": codes must be a contiguous uint8 tensor on ", dev);
  TORCH_CHECK(codes.numel() == rows * row_bytes, what, ": codes hold ", codes.numel(), " bytes, expected ", rows, " x ", row_bytes);
  TORCH_CHECK((scales.scalar_type() == at::kHalf || scales.scalar_type() == at::kFloat) && scales.is_contiguous() && scales.device() == dev,
              what, ": scales must be a contiguous float16 / float32 tensor on ", dev);
  TORCH_CHECK(scales.numel() == n_scales, what, ": ", scales.numel(), " scales, expected ", n_scales);
}

// Operand shapes of the FP4 GEMMs: both row-major codes [rows, K / 2] (2-D) or both k-major images [K / 128, image rows, 64] (3-D,
// include/fpq.h; the weight image has outs rounded up to 64 rows, outs itself comes from the scales [outs, K / 128]).
struct Fp4Shapes {
  int64_t tokens, outs, k;
  bool kmajor;
};
static Fp4Shapes fp4_shapes(const char* what, const at::Tensor& a_codes, const at::Tensor& a_scales, const at::Tensor& w_codes,
                            const at::Tensor& w_scales, const c10::optional<at::Tensor>& bias, const c10::optional<int64_t>& outs_arg) {
  Fp4Shapes sh;
  TORCH_CHECK((a_codes.dim() == 2 && w_codes.dim() == 2) || (a_codes.dim() == 3 && w_codes.dim() == 3), what,
              ": both operands must be row-major codes [rows, K / 2] or both k-major images [K / 128, rows, 64]");
  sh.kmajor = a_codes.dim() == 3;
  const at::Device dev = a_codes.device();
  if (sh.kmajor) {
    TORCH_CHECK(a_codes.size(2) == 64 && w_codes.size(2) == 64 && a_codes.size(0) == w_codes.size(0) && w_codes.size(1) % 64 == 0, what,
                ": k-major images must be [K / 128, rows, 64] with the same K and a weight image of a multiple of 64 rows");
    const int64_t groups = a_codes.size(0), w_rows = w_codes.size(1);
    sh.tokens = a_codes.size(1);
    sh.k = groups * 128;
    // the Linear's width: named by the caller, or the bias' length, or the weight image's row count
    sh.outs = outs_arg.has_value() ? *outs_arg : bias.has_value() ? bias->numel() : w_rows;
    TORCH_CHECK(sh.outs > w_rows - 64 && sh.outs <= w_rows, what, ": outs = ", sh.outs, " does not belong to a weight image of ", w_rows, " rows");
    const auto image_ok = [&](const at::Tensor& t, int64_t rows) {
      return t.scalar_type() == at::kFloat && t.dim() == 2 && t.size(0) == groups && t.size(1) == rows && t.is_contiguous() && t.device() == dev;
    };
    TORCH_CHECK(image_ok(a_scales, (sh.tokens + 3) / 4 * 4) && image_ok(w_scales, w_rows), what,
                ": the k-major scale images must be contiguous float32 [K / 128, rows rounded up to 4 (activation) | weight image rows]");
    TORCH_CHECK(a_codes.scalar_type() == at::kByte && w_codes.scalar_type() == at::kByte && a_codes.is_contiguous() && w_codes.is_contiguous() &&
                    w_codes.device() == dev, what, ": the k-major images must be contiguous uint8 tensors on ", dev);
    return sh;
  }
  sh.tokens = a_codes.size(0);
  sh.outs = w_codes.size(0);
  sh.k = a_codes.size(1) * 2;
  TORCH_CHECK(w_codes.size(1) * 2 == sh.k, what, ": operand shapes mismatch");
  TORCH_CHECK(a_scales.scalar_type() == at::kHalf && sh.k % 128 == 0, what, ": operand shapes / activation scale dtype mismatch");
  check_operand((std::string(what) + "(activation)").c_str(), a_codes, a_scales, sh.tokens, sh.k / 2, sh.tokens * (sh.k / 128), dev);
  check_operand((std::string(what) + "(weight)").c_str(), w_codes, w_scales, sh.outs, sh.k / 2, sh.outs * (sh.k / 128), dev);
  return sh;
}

// fp16 [tokens, outs] = dequant(a) @ dequant(w).T + bias on the FP4 matrix cores, with the AdaLN block's gated residual in
// the epilogue when given (tr/quant_utils.py:767, tr/basic_var.py:264): gemm.linear_fp4
at::Tensor linear_fp4(const at::Tensor& a_codes, const at::Tensor& a_scales, const at::Tensor& w_codes, const at::Tensor& w_scales,
                      const c10::optional<at::Tensor>& bias, const c10::optional<at::Tensor>& gate, const c10::optional<at::Tensor>& residual,
                      const c10::optional<int64_t>& outs_arg) {
  require_gpu(a_codes, "linear_fp4");
  const Fp4Shapes sh = fp4_shapes("linear_fp4", a_codes, a_scales, w_codes, w_scales, bias, outs_arg);
  const int64_t tokens = sh.tokens, outs = sh.outs, k = sh.k;
  const at::Device dev = a_codes.device();
  fpq_gemm_epilogue_t ep{nullptr, nullptr, 1};
  at::Tensor g, r, b;
  at::Tensor out = at::empty({tokens, outs}, a_codes.options().dtype(at::kHalf));
  if (tokens == 0 || outs == 0) return out;   // as the C ABI: valid, nothing to enqueue (and no reshape({-1, 0}) below)
  // the epilogue reads gate / residual / bias in 8- and 16-byte pieces: a contiguous view at an odd storage offset is cloned
  // here (the C ABI rejects the pointer; a misaligned bias would otherwise route to the slower register-staged kernel)
  auto aligned = [](const at::Tensor& t) { return (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0 ? t : t.clone(); };
  if (gate.has_value()) {
    g = gate->reshape({-1, outs});
    TORCH_CHECK(g.scalar_type() == at::kHalf && g.size(0) > 0 && tokens % g.size(0) == 0 && g.device() == dev,
                "linear_fp4: gate must be float16 [B, outs] with tokens % B == 0");
    g = aligned(g.contiguous());
    ep.gate = g.data_ptr();
    ep.rows_per_gate = std::max<int64_t>(tokens / g.size(0), 1);
  }
  if (residual.has_value()) {
    r = residual->reshape({-1, outs});
    TORCH_CHECK(r.scalar_type() == at::kHalf && r.size(0) == tokens && r.device() == dev, "linear_fp4: residual must be float16 with ", tokens, " rows of ", outs);
    r = aligned(r.contiguous());
    ep.residual = r.data_ptr();
  }
  if (bias.has_value()) {
    TORCH_CHECK(bias->numel() == outs && bias->device() == dev, "linear_fp4: bias must hold one value per output on the operands' device");
    b = aligned(bias->detach().to(at::kHalf).reshape({-1}).contiguous());
  }
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
  check((sh.kmajor ? fpq_gemm_fp4_mx_km : fpq_gemm_fp4_mx_ex)(
            (const uint8_t*)a_codes.data_ptr(), a_scales.data_ptr(), (const uint8_t*)w_codes.data_ptr(), w_scales.data_ptr(),
            dtype_id(w_scales.scalar_type(), "linear_fp4"), b.defined() ? b.data_ptr() : nullptr, out.data_ptr(), tokens, outs, k,
            (gate.has_value() || residual.has_value()) ? &ep : nullptr, current_stream(a_codes)),
        sh.kmajor ? "fpq_gemm_fp4_mx_km" : "fpq_gemm_fp4_mx_ex");
  return out;
}

// fc1 with the FFN's GELU and fc2's dual-format input quantizer in the GEMM's epilogue (tr/basic_var.py:120-121,
// tr/quant_utils.py:415-452,991): gemm.linear_fp4_gelu_dual.  Returns (out, gelu values or an undefined tensor).
std::tuple<at::Tensor, c10::optional<at::Tensor>> linear_fp4_gelu_dual(const at::Tensor& a_codes, const at::Tensor& a_scales, const at::Tensor& w_codes,
                                                                        const at::Tensor& w_scales, const c10::optional<at::Tensor>& bias,
                                                                        bool return_gelu, const c10::optional<int64_t>& outs_arg) {
  require_gpu(a_codes, "linear_fp4_gelu_dual");
  const Fp4Shapes sh = fp4_shapes("linear_fp4_gelu_dual", a_codes, a_scales, w_codes, w_scales, bias, outs_arg);
  const int64_t tokens = sh.tokens, outs = sh.outs, k = sh.k;
  TORCH_CHECK(outs % 128 == 0, "linear_fp4_gelu_dual: outs must be a multiple of 128");
  const at::Device dev = a_codes.device();
  at::Tensor out = at::empty({tokens, outs}, a_codes.options().dtype(at::kHalf));
  c10::optional<at::Tensor> h;
  if (return_gelu) h = at::empty({tokens, outs}, a_codes.options().dtype(at::kHalf));
  if (tokens == 0 || outs == 0) return {out, h};
  at::Tensor b;
  if (bias.has_value()) {
    TORCH_CHECK(bias->numel() == outs && bias->device() == dev, "linear_fp4_gelu_dual: bias must hold one value per output on the operands' device");
    b = bias->detach().to(at::kHalf).reshape({-1}).contiguous();
    if ((reinterpret_cast<uintptr_t>(b.data_ptr()) & 15) != 0) b = b.clone();
  }
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
  const fpq_stream_t st = current_stream(a_codes);
  at::Tensor scratch = nan_scratch(a_codes, st);
  const int status = (sh.kmajor ? fpq_gemm_fp4_gelu_dual_km : fpq_gemm_fp4_gelu_dual)(
      (const uint8_t*)a_codes.data_ptr(), a_scales.data_ptr(), (const uint8_t*)w_codes.data_ptr(), w_scales.data_ptr(),
      dtype_id(w_scales.scalar_type(), "linear_fp4_gelu_dual"), b.defined() ? b.data_ptr() : nullptr, out.data_ptr(),
      h.has_value() ? h->data_ptr() : nullptr, tokens, outs, k, scratch.data_ptr(), st);
  if (status != 0) (void)hipMemsetAsync(scratch.data_ptr(), 0, 8, (hipStream_t)st);
  check(status, sh.kmajor ? "fpq_gemm_fp4_gelu_dual_km" : "fpq_gemm_fp4_gelu_dual");
  return {out, h};
}

// `fc2.act_quant(act(y))` in one pass over the fp16 fc1 output y: ops.gelu_quant_rows_dual
std::tuple<at::Tensor, c10::optional<at::Tensor>> gelu_quant_rows_dual(const at::Tensor& y, int64_t neg_id, int64_t pos_id, int64_t cols, bool nan_rule,
                                                                        bool return_gelu) {
  require_gpu(y, "gelu_quant_rows_dual");
  TORCH_CHECK(y.scalar_type() == at::kHalf && cols > 0 && cols % 8 == 0 && y.numel() % cols == 0,
              "gelu_quant_rows_dual: y must be float16 and hold whole rows of ", cols, " (a multiple of 8) elements");
  const at::Tensor yc = y.is_contiguous() ? y : y.contiguous();
  at::Tensor out = at::empty(y.sizes(), y.options());
  c10::optional<at::Tensor> h;
  if (return_gelu) h = at::empty(y.sizes(), y.options());
  if (yc.numel() == 0) return {out, h};
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(y.device());
  const fpq_stream_t st = current_stream(y);
  at::Tensor scratch;
  if (nan_rule) scratch = nan_scratch(y, st);
  const int status = fpq_gelu_quant_rows_dual(yc.data_ptr(), out.data_ptr(), h.has_value() ? h->data_ptr() : nullptr, yc.numel() / cols, cols, (int)neg_id,
                                              (int)pos_id, nan_rule ? scratch.data_ptr() : nullptr, st);
  if (status != 0 && nan_rule) (void)hipMemsetAsync(scratch.data_ptr(), 0, 8, (hipStream_t)st);
  check(status, "fpq_gelu_quant_rows_dual");
  return {out, h};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled binding of libfpq_hip.so for the reference's quant_cuda / quant_utils boundary";
  m.def("fpq_version", [] { return fpq_version(); });
  m.def("fpq_build_tag", [] { return std::string(fpq_build_tag()); });
  m.def("quant", &quant, py::arg("x"), py::arg("y"));
  m.def("quant_rows", &quant_rows, py::arg("x"), py::arg("table_id"), py::arg("cols"), py::arg("out_dtype") = py::none());
  m.def("quant_rows_dual", &quant_rows_dual, py::arg("x"), py::arg("neg_table_id"), py::arg("pos_table_id"), py::arg("cols"),
        py::arg("clipping_strength") = py::none(), py::arg("out_dtype") = py::none());
  m.def("fp_quant_e3_per_group_cuda", &fp_quant_e3_per_group_cuda, py::arg("x"), py::arg("n_bits"), py::arg("group_size") = 128);
  m.def("fp_quant_e2_per_group_cuda", &fp_quant_e2_per_group_cuda, py::arg("x"), py::arg("n_bits"), py::arg("group_size") = 128);
  m.def("fp_quant_e1_per_group_cuda", &fp_quant_e1_per_group_cuda, py::arg("x"), py::arg("n_bits"), py::arg("group_size") = 128);
  m.def("fp_quant_e1m2_neg_e2m1_pos_per_group_cuda", &fp_quant_e1m2_neg_e2m1_pos_per_group_cuda, py::arg("x"), py::arg("n_bits"),
        py::arg("group_size") = 128, py::arg("clipping_strength") = 1.0);
  m.def("fp4_afpq_per_group_cuda", &fp4_afpq_per_group_cuda, py::arg("x"), py::arg("n_bits"), py::arg("group_size") = 128,
        py::arg("clipping_strength") = 1.0);
  m.def("fp6_quant_e2m3_per_group_cuda", &fp6_quant_e2m3_per_group_cuda, py::arg("x"), py::arg("n_bits"), py::arg("group_size") = 128);
  m.def("fp6_quant_e3m2_per_group_cuda", &fp6_quant_e3m2_per_group_cuda, py::arg("x"), py::arg("n_bits"), py::arg("group_size") = 128);
  m.def("fp6_quant_int_neg_e2m3_pos_per_group_cuda", &fp6_quant_int_neg_e2m3_pos_per_group_cuda, py::arg("x"), py::arg("n_bits"),
        py::arg("group_size") = 128);
  m.def("rotate_quant", &rotate_quant, py::arg("x"), py::arg("table_id"), py::arg("sign_mask"), py::arg("smooth") = py::none());
  m.def("adaln_rotate_quant", &adaln_rotate_quant, py::arg("x"), py::arg("scale"), py::arg("shift"), py::arg("table_id"),
        py::arg("sign_mask"), py::arg("smooth") = py::none(), py::arg("eps") = 1e-6);
  m.def("rotate_quant_mx", &rotate_quant_mx, py::arg("x"), py::arg("sign_mask"), py::arg("smooth") = py::none(), py::arg("kmajor") = false);
  m.def("adaln_rotate_quant_mx", &adaln_rotate_quant_mx, py::arg("x"), py::arg("scale"), py::arg("shift"), py::arg("sign_mask"),
        py::arg("smooth") = py::none(), py::arg("eps") = 1e-6, py::arg("kmajor") = false);
  m.def("adaln_rotate_quant_token", &adaln_rotate_quant_token, py::arg("x"), py::arg("scale"), py::arg("shift"), py::arg("table_id"),
        py::arg("sign_mask"), py::arg("smooth") = py::none(), py::arg("eps") = 1e-6);
  m.def("adaln_rotate_quant_token_codes", &adaln_rotate_quant_token_codes, py::arg("x"), py::arg("scale"), py::arg("shift"),
        py::arg("table_id"), py::arg("code_bits"), py::arg("sign_mask"), py::arg("smooth") = py::none(), py::arg("eps") = 1e-6,
        py::arg("kmajor") = false);
  m.def("kv_cache_step", &kv_cache_step, py::arg("cache"), py::arg("quant_start"), py::arg("quant_stop"), py::arg("k"), py::arg("v"),
        py::arg("new_start"), py::arg("group"), py::arg("table_id"));
  m.def("linear_fp4", &linear_fp4, py::arg("a_codes"), py::arg("a_scales"), py::arg("w_codes"), py::arg("w_scales"),
        py::arg("bias") = py::none(), py::arg("gate") = py::none(), py::arg("residual") = py::none(), py::arg("outs") = py::none());
  m.def("linear_fp4_gelu_dual", &linear_fp4_gelu_dual, py::arg("a_codes"), py::arg("a_scales"), py::arg("w_codes"), py::arg("w_scales"),
        py::arg("bias") = py::none(), py::arg("return_gelu") = false, py::arg("outs") = py::none());
  m.def("gelu_quant_rows_dual", &gelu_quant_rows_dual, py::arg("y"), py::arg("neg_table_id"), py::arg("pos_table_id"), py::arg("cols") = 128,
        py::arg("nan_rule") = true, py::arg("return_gelu") = false);
  m.def("fp6_quant_per_token_contig", &fp6_quant_per_token_contig, py::arg("x"), py::arg("n_bits"), py::arg("table_id"));
  m.def("fp6_quant_int_neg_e2m3_pos_per_token_contig", &fp6_quant_int_neg_e2m3_pos_per_token_contig, py::arg("x"), py::arg("n_bits"));
}
