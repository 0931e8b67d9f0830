// api.hip -- error plumbing, version, and the instruction-layout self-test.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "common.h"

namespace vtx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return VTX_ELAUNCH;
  }
  return VTX_OK;
}

// ---- options -------------------------------------------------------------------------
static const char* const kNtNames[] = {"auto", "pp256", "dma2", "ring128x3", "ring128x4k32", "ring256x3", "ring256x3k32", "ring256x4k32"};
static const char* const kTnNames[] = {"auto", "pp256", "ring", "dma2", "w4"};

static int parse_enum(const char* v, const char* const* names, int n) {
  for (int i = 0; i < n; ++i)
    if (strcmp(v, names[i]) == 0) return i;
  return -1;
}

static int set_option(Options& o, const char* name, const char* value) {
  if (!name || !value) return VTX_EINVAL;
  if (strcmp(name, "gemm_nt") == 0) { const int e = parse_enum(value, kNtNames, 8); if (e < 0) return VTX_EINVAL; o.gemm_nt = e; return VTX_OK; }
  if (strcmp(name, "gemm_tn") == 0) { const int e = parse_enum(value, kTnNames, 5); if (e < 0) return VTX_EINVAL; o.gemm_tn = e; return VTX_OK; }
  if (strcmp(name, "gemm_nodma") == 0) { o.gemm_nodma = atoi(value) != 0; return VTX_OK; }
  if (strcmp(name, "tn_safe") == 0) { o.tn_safe = atoi(value) != 0; return VTX_OK; }
  if (strcmp(name, "tn_cus") == 0) { const int g = atoi(value); if (g < 32 || g > 1024) return VTX_EINVAL; o.tn_cus = g; return VTX_OK; }
  if (strcmp(name, "attn_valu") == 0) { o.attn_valu = atoi(value) != 0; return VTX_OK; }
  if (strcmp(name, "attn_hw_fwd") == 0) { const int g = atoi(value); if (g < 0) return VTX_EINVAL; o.attn_hw_fwd = g; return VTX_OK; }
  if (strcmp(name, "attn_hw_bwd") == 0) { const int g = atoi(value); if (g < 0) return VTX_EINVAL; o.attn_hw_bwd = g; return VTX_OK; }
  if (strcmp(name, "attn_fused") == 0) { const int v = atoi(value); o.attn_fused = v < 0 ? 0 : v > 2 ? 2 : v; return VTX_OK; }
  if (strcmp(name, "attn_fwd_stream") == 0) { o.attn_fwd_stream = atoi(value) != 0; return VTX_OK; }
  if (strcmp(name, "attn_dkv") == 0) { const int g = atoi(value); if (g < 0 || g > 4) return VTX_EINVAL; o.attn_dkv = g; return VTX_OK; }
  if (strcmp(name, "pp_grid") == 0) { const int g = atoi(value); if (g < 8 || g > 4096 || g % 8) return VTX_EINVAL; o.pp_grid = g; return VTX_OK; }
  if (strcmp(name, "pp_cg") == 0) { const int g = atoi(value); if (g < 0) return VTX_EINVAL; o.pp_cg = g; return VTX_OK; }
  if (strcmp(name, "pp_epi") == 0) { o.pp_epi = atoi(value); return VTX_OK; }
  if (strcmp(name, "pp_cont") == 0) { o.pp_cont = atoi(value) != 0; return VTX_OK; }
  if (strcmp(name, "ln_rows") == 0) { const int g = atoi(value); if (g < 1 || g > 4) return VTX_EINVAL; o.ln_rows = g; return VTX_OK; }
  if (strcmp(name, "pp_trace") == 0) { o.pp_trace = strtoull(value, nullptr, 0); return VTX_OK; }
  return VTX_EINVAL;
}

Options& options() {
  static Options o = [] {
    Options d;
    static const char* const env[][2] = {{"VTX_GEMM_NT", "gemm_nt"}, {"VTX_GEMM_TN", "gemm_tn"}, {"VTX_GEMM_NODMA", "gemm_nodma"},
                                         {"VTX_TN_SAFE", "tn_safe"}, {"VTX_TN_CUS", "tn_cus"}, {"VTX_ATTN_VALU", "attn_valu"}, {"VTX_GEMM_PP_GRID", "pp_grid"},
                                         {"VTX_GEMM_PP_CG", "pp_cg"}, {"VTX_GEMM_PP_EPI", "pp_epi"},
                                         {"VTX_GEMM_PP_CONT", "pp_cont"}, {"VTX_LN_ROWS", "ln_rows"}, {"VTX_ATTN_HW_FWD", "attn_hw_fwd"}, {"VTX_ATTN_HW_BWD", "attn_hw_bwd"}, {"VTX_ATTN_DKV", "attn_dkv"}, {"VTX_ATTN_FWD_STREAM", "attn_fwd_stream"},
                                         {"VTX_ATTN_FUSED", "attn_fused"}};
    for (const auto& e : env) {
      const char* v = getenv(e[0]);
      if (v && *v) set_option(d, e[1], v);       // an unparsable value keeps the default
    }
    return d;
  }();
  return o;
}

// ---- probes (one wave each) ---------------------------------------------------------
// D = A(32x16) * B(16x32) with the operand layout the GEMM kernels assume:
// lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31], j = 0..7; D in the documented C layout.
__global__ void probe_mfma_bf16(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  union { bf16x8 v; bf16raw s[8]; } a, b;
  for (int j = 0; j < 8; ++j) {
    a.s[j] = f2bf(A[(l & 31) * 16 + 8 * (l >> 5) + j]);
    b.s[j] = f2bf(B[(8 * (l >> 5) + j) * 32 + (l & 31)]);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// D = A(32x2) * B(2x32): lane l holds A[l&31][l>>5], B[l>>5][l&31].
__global__ void probe_mfma_f32(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// ds_read_b64_tr_b16 with the address pattern of gemm_tn: LDS holds lds[e] = e.
__global__ void probe_tr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16 * 160];
  const int l = threadIdx.x;
  for (int e = l; e < 16 * 160; e += 64) lds[e] = (unsigned short)e;
  __syncthreads();
  const int row = 8 * (l >> 5) + ((l & 15) >> 2), col = 16 * ((l >> 4) & 1) + 4 * (l & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row * 160 + col));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int hog_selftest(int* bad_mag, int* bad_bin);     // hog.hip

}  // namespace vtx

using namespace vtx;

extern "C" int vtx_version(void) { return 220; }  // 0.2.2: vtx_layernorm_acc_fwd / VTX_BF16_X32 (exact residual stream); 0.2.1: vtx_hog_fwd takes the size of the table blob

extern "C" int vtx_set_option(const char* name, const char* value) {
  const int rc = set_option(options(), name, value);
  if (rc) set_error("vtx_set_option: unknown option or bad value: %s=%s", name ? name : "(null)", value ? value : "(null)");
  return rc;
}
extern "C" const char* vtx_last_error_string(void) { return g_err; }

extern "C" int vtx_selftest(char* report, size_t report_bytes) {
  std::string rep;
  int fails = 0;
  auto say = [&](const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    rep += buf;
  };
  float *dA = nullptr, *dB = nullptr, *dD = nullptr;
  unsigned short* dT = nullptr;
  if (hipMalloc(&dA, 32 * 16 * 4) != hipSuccess || hipMalloc(&dB, 16 * 32 * 4) != hipSuccess ||
      hipMalloc(&dD, 32 * 32 * 4) != hipSuccess || hipMalloc(&dT, 256 * 2) != hipSuccess) {
    set_error("selftest: hipMalloc failed");
    return VTX_ELAUNCH;
  }
  std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)((s >> 24) % 15) - 7); };
  // ---- bf16 32x32x16
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_mfma_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float r = 0.f;
      for (int k = 0; k < 16; ++k) r += A[i * 16 + k] * B[k * 32 + j];
      if (r != D[i * 32 + j]) ++bad;
    }
  say("mfma_f32_32x32x16_bf16 layout: %s (%d/1024 mismatches)\n", bad ? "FAIL" : "ok", bad);
  fails += bad != 0;
  // ---- f32 32x32x2
  std::vector<float> A2(32 * 2), B2(2 * 32);
  for (auto& v : A2) v = rnd();
  for (auto& v : B2) v = rnd();
  hipMemcpy(dA, A2.data(), A2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B2.data(), B2.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_mfma_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      const float r = A2[i * 2] * B2[j] + A2[i * 2 + 1] * B2[32 + j];
      if (r != D[i * 32 + j]) ++bad;
    }
  say("mfma_f32_32x32x2_f32 layout: %s (%d/1024 mismatches)\n", bad ? "FAIL" : "ok", bad);
  fails += bad != 0;
  // ---- ds_read_b64_tr_b16
  std::vector<unsigned short> T(256);
  hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, dT);
  hipMemcpy(T.data(), dT, 512, hipMemcpyDeviceToHost);
  bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = (8 * (l >> 5) + j) * 160 + (l & 31);
      if (T[l * 4 + j] != want) ++bad;
    }
  say("ds_read_b64_tr_b16 gather: %s (%d/256 mismatches)\n", bad ? "FAIL" : "ok", bad);
  if (bad) {
    say("  observed (lane: row,col x4):\n");
    for (int l = 0; l < 64; ++l) {
      say("  %2d:", l);
      for (int j = 0; j < 4; ++j) say(" %d,%d", T[l * 4 + j] / 160, T[l * 4 + j] % 160);
      say("\n");
    }
  }
  fails += bad != 0;
  // ---- HOG: device magnitudes (correctly rounded sqrt + table correction) vs the host's hypot, integer bin rule vs float64
  {
    int bad_mag = -1, bad_bin = -1;
    const int rc = hog_selftest(&bad_mag, &bad_bin);
    say("hog magnitudes, all 65536 gradient pairs vs host hypot: %s (%d mismatches)\n", (rc || bad_mag) ? "FAIL" : "ok", bad_mag);
    say("hog integer bin rule, all 261121 gradient pairs vs the float64 sign tests: %s (%d mismatches)\n", (rc || bad_bin) ? "FAIL" : "ok", bad_bin);
    fails += (rc != 0 || bad_mag != 0) + (rc != 0 || bad_bin != 0);
  }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { say("device error: %s\n", hipGetErrorString(e)); ++fails; }
  hipFree(dA); hipFree(dB); hipFree(dD); hipFree(dT);
  if (report && report_bytes) {
    strncpy(report, rep.c_str(), report_bytes - 1);
    report[report_bytes - 1] = 0;
  }
  return fails;
}
