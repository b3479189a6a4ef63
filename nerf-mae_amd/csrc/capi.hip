// extern "C" layer of libnerfmae_hip.so (see include/nerfmae_hip.h).  No torch types; plain pointers + sizes.
#include "../../include/nerfmae_hip.h"
#include "kernels.hpp"
#include <cstdio>

static inline WinMap to_wm(const int* w) {
  WinMap m{};
  if (w) { m.B = w[0]; m.H = w[1]; m.W = w[2]; m.D = w[3]; m.PH = w[4]; m.PW = w[5]; m.PD = w[6]; m.s0 = w[7]; m.s1 = w[8]; m.s2 = w[9]; }
  return m;
}
#define ST ((hipStream_t)stream)
#define NMH_DT_BF16_C 1
// required pointers: a NULL yields the argument-error code -4 instead of a device fault
#define REQ(...) do { const void* req_[] = {__VA_ARGS__}; for (const void* q_ : req_) if (!q_) return -4; } while (0)
// hipGetLastError() is sticky per thread: clear whatever an unrelated earlier runtime call left behind, so that the
// post-launch check in the k_* launchers reports only this call's launch status
#define CLR() (void)hipGetLastError()

extern "C" {
int nmh_version(void) { return 100; }
const char* nmh_error_string(int code) {
  if (code == 0) return "ok";
  if (code == -1) return "nmh: unsupported operation";
  if (code == -2) return "nmh: shape constraint violated (channel counts must be multiples of 8; head_dim must be 32)";
  if (code == -3) return "nmh: row too wide for the layernorm kernels";
  if (code < 0) return "nmh: invalid argument";
  return hipGetErrorString((hipError_t)code);
}
int nmh_gemm_nt(int dt, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K, void* C, int64_t ldc, const float* bias, int act, void* C2,
                const void* resid, const float* rowscale, int rows_per_scale, int accumulate, void* stream) {
  CLR();
  REQ(A, W, C);
  if (M <= 0) return 0;
  EpiParams ep{C, ldc, bias, act, C2, resid, rowscale, rows_per_scale > 0 ? rows_per_scale : 1, accumulate};
  return k_gemm_nt(dt, A, lda, W, ldw, M, N, K, ep, ST);
}
int nmh_gemm_nt_window_scatter(int dt, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K, void* out, const void* resid, const float* bias,
                               const float* rowscale, int tokens_per_sample, const int* wm, void* stream) {
  CLR();
  REQ(A, W, out, resid, wm);
  if (M <= 0) return 0;
  EpiParams ep{out, N, bias, 0, nullptr, resid, rowscale, tokens_per_sample > 0 ? tokens_per_sample : 1, 0};
  ep.win_on = 1; ep.wm = to_wm(wm);
  return k_gemm_nt(dt, A, lda, W, ldw, M, N, K, ep, ST);
}
int nmh_gemm_tn(int dt, const void* A, int64_t lda, const void* B, int64_t ldb, float* dW, int64_t M, int N, int K, const float* rowscale, int rows_per_scale,
                int omode, int64_t ldo, int p0, int p1, float* dbias, float* ws, int64_t ws_floats, void* stream) {
  CLR();
  REQ(A, B, dW);
  if (M <= 0) return 0;
  TnGeom gm{};
  gm.omode = omode; gm.ldo = ldo; gm.dbias = dbias; gm.ws = ws; gm.ws_floats = ws ? (long)ws_floats : 0;
  if (omode == 2) { gm.Cin = p0; gm.V = (unsigned)p1; gm.dC = make_fdiv(p0); }
  return k_gemm_tn(dt, A, lda, B, ldb, dW, M, N, K, rowscale, rows_per_scale > 0 ? rows_per_scale : 1, gm, ST);
}
int nmh_gemm_tn_grouped(int dt, const nmh_tn_problem* probs, int nprob, float* ws, int64_t ws_floats, void* stream) {
  CLR();
  if (nprob <= 0) return 0;
  if (dt != NMH_DT_BF16_C || probs == nullptr) return -4;
  static_assert(sizeof(nmh_tn_problem) == sizeof(TnProblemHost), "descriptor layouts must agree");
  return k_gemm_tn_grouped(reinterpret_cast<const TnProblemHost*>(probs), nprob, ws, ws ? (long)ws_floats : 0, ST);
}
int nmh_gemm_tn_grouped_fg(int dt, const nmh_tn_problem* probs, int nprob, float* ws, int64_t ws_floats, void* stream) {
  CLR();
  if (nprob <= 0) return 0;
  if (dt != NMH_DT_BF16_C || probs == nullptr) return -4;
  return k_gemm_tn_grouped(reinterpret_cast<const TnProblemHost*>(probs), nprob, ws, ws ? (long)ws_floats : 0, ST, true);
}
int nmh_upconv_fwd(int dt, const void* x, const void* Wt, const float* bias, void* cat, int64_t ldc, int B, int v, int k, int Cin, int Cout, void* stream) {
  CLR();
  REQ(x, Wt, cat);
  return k_upconv_fwd(dt, x, Wt, bias, cat, (long)ldc, B, v, k, Cin, Cout, ST);
}
int nmh_upconv_dgrad(int dt, const void* dcat, int64_t ldc, const void* Wd, void* dx, int B, int v, int k, int Cin, int Cout, void* stream) {
  CLR();
  REQ(dcat, Wd, dx);
  return k_upconv_dgrad(dt, dcat, (long)ldc, Wd, dx, B, v, k, Cin, Cout, ST);
}
int nmh_upconv_wgrad(int dt, const void* dcat, int64_t ldc, const void* x, float* dW, float* dbias, int B, int v, int k, int Cin, int Cout, void* stream) {
  CLR();
  REQ(dcat, x, dW);
  return k_upconv_wgrad(dt, dcat, (long)ldc, x, dW, dbias, B, v, k, Cin, Cout, ST);
}
int nmh_conv3d_k3(int dt, const void* X, const void* Wp, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, float* ws, int64_t ws_floats, void* stream) {
  CLR();
  REQ(X, Wp, Y);
  EpiParams ep{Y, Cout, nullptr, 0, nullptr, nullptr, nullptr, 1, accumulate};
  return k_conv3_nt(dt, X, Wp, B, D, H, W, Cin, Cout, ep, ST, ws, ws ? (long)ws_floats : 0);
}
int nmh_conv3d_k3_bias(int dt, const void* X, const void* Wp, const float* bias, void* Y, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
  CLR();
  REQ(X, Wp, bias, Y);
  EpiParams ep{Y, Cout, bias, 0, nullptr, nullptr, nullptr, 1, 0};
  return k_conv3_nt(dt, X, Wp, B, D, H, W, Cin, Cout, ep, ST);
}
int nmh_conv3d_k3_c64(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, double* stats_acc, const float* bias, void* stream) {
  CLR();
  if (!X || !Wk || !Y) return -4;
  return k_conv64(X, Wk, Y, B, D, H, W, Cin, Cout, accumulate, stats_acc, bias, ST);
}
int nmh_conv3d_k3_c64_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
  CLR();
  if (!dY || !X || !dW || !ws) return -4;
  return k_conv64_wgrad(dY, X, dW, ws, B, D, H, W, Cin, Cout, ST);
}
int64_t nmh_conv3d_k3_c64_wgrad_ws_floats(void) { return (int64_t)k_conv64_wgrad_ws_floats(); }
int64_t nmh_conv3d_k3_c64_pack_numel(int Cin, int Cout) { return (Cin % 64 || Cout % 64) ? -1 : (int64_t)k_conv64_pack_numel(Cin, Cout); }
int nmh_nearest_upsample_add(int dt, const void* coarse, void* fine, int B, int Dc, int Hc, int Wc, int Df, int Hf, int Wf, int C, void* stream) {
  CLR();
  REQ(coarse, fine);
  return k_nearest_up_add(dt, coarse, fine, B, Dc, Hc, Wc, Df, Hf, Wf, C, 0, ST);
}
int nmh_nearest_upsample_add_bwd(int dt, const void* dfine, void* dcoarse, int B, int Dc, int Hc, int Wc, int Df, int Hf, int Wf, int C, void* stream) {
  CLR();
  REQ(dfine, dcoarse);
  return k_nearest_up_add(dt, dcoarse, const_cast<void*>(dfine), B, Dc, Hc, Wc, Df, Hf, Wf, C, 1, ST);
}
int nmh_copy_cols(int dt, const void* src, int64_t lds, void* dst, int64_t ldd, int64_t M, int C, void* stream) {
  CLR();
  REQ(src, dst);
  return k_copy_cols(dt, src, (long)lds, dst, (long)ldd, (long)M, C, ST);
}
int nmh_ndhwc_to_ncdhw(int dt, const void* src, float* dst, int B, int64_t V, int C, void* stream) {
  CLR();
  return k_vc_transpose(dt, src, dst, B, (long)V, C, 0, ST);
}
int nmh_ncdhw_to_ndhwc(int dt, const float* src, void* dst, int B, int64_t V, int C, void* stream) {
  CLR();
  return k_vc_transpose(dt, src, dst, B, (long)V, C, 1, ST);
}
int nmh_conv3d_k3_c48(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int accumulate, double* stats_acc, void* stream) {
  CLR();
  return k_conv48(X, Wk, Y, B, D, H, W, accumulate, stats_acc, ST);
}
int nmh_conv3d_k3_c48_bwd_reduce(const void* dY, const void* Wkd, void* dX, int B, int D, int H, int W, const void* Y1, const float* stats1, float slope,
                                 double* sums, void* stream) {
  CLR();
  REQ(dY, Wkd, dX, Y1, stats1, sums);
  return k_conv48(dY, Wkd, dX, B, D, H, W, 0, sums, ST, Y1, stats1, slope);
}
int nmh_conv48_pack_scaled(const float* W, const float* stats, void* Wk_per_sample, int B, void* stream) {
  CLR();
  REQ(W, stats, Wk_per_sample);
  if (B <= 0) return 0;
  return k_conv48_pack_scaled(W, stats, Wk_per_sample, B, ST);
}
int nmh_conv3d_k3_c48_per_sample(const void* X, const void* Wk_per_sample, void* Y, int B, int D, int H, int W, double* stats_acc, void* stream) {
  CLR();
  REQ(X, Wk_per_sample, Y);
  return k_conv48(X, Wk_per_sample, Y, B, D, H, W, 0, stats_acc, ST, nullptr, nullptr, 0.f, 41L * 3 * 512, 0);
}
int nmh_conv3d_k3_c48_bwd_reduce_centered(const void* dY, const void* Wkd, void* dX, int B, int D, int H, int W, const void* Z, const float* stats1, float slope,
                                          double* sums, void* stream) {
  CLR();
  REQ(dY, Wkd, dX, Z, stats1, sums);
  if (!(slope > 0.f && slope < 1.f)) return -2;
  return k_conv48(dY, Wkd, dX, B, D, H, W, 0, sums, ST, Z, stats1, slope, 0, 1);
}
int nmh_conv3d_k3_c48_wgrad_scaled(const void* dY, const void* Z, const float* stats, float* dW, float* ws, int B, int D, int H, int W, void* stream) {
  CLR();
  REQ(dY, Z, stats, dW, ws);
  return k_conv48_wgrad(dY, Z, dW, ws, B, D, H, W, ST, stats);
}
int nmh_instnorm_bwd_apply_bg_centered(int dt, const void* dout, const void* z, const float* stats, const double* sums, void* dx, int B, int64_t V, int C, float slope, void* stream) {
  CLR();
  REQ(dout, z, stats, sums, dx);
  return k_in_bwd_apply_bg(dt, dout, z, stats, sums, dx, B, (long)V, C, slope, ST, 1);
}
int nmh_conv3d_k3_c48mb(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, void* stream) {
  CLR();
  if (!X || !Wk || !Y) return -4;
  return k_conv48_mb(X, Wk, Y, B, D, H, W, Cin, Cout, accumulate, ST);
}
int nmh_instnorm_finalize(const double* acc, float* stats, int B, int64_t V, int C, float eps, void* stream) {
  CLR();
  return k_in_finalize(0, acc, stats, B, (long)V, C, eps, ST);
}
int nmh_conv3d_k3_c48_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, void* stream) {
  CLR();
  return k_conv48_wgrad(dY, X, dW, ws, B, D, H, W, ST);
}
int nmh_conv3d_k3_wgrad_halo(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
  CLR();
  return k_conv3_wgrad_halo(dY, X, dW, ws, B, D, H, W, Cin, Cout, ST);
}
int64_t nmh_conv3d_k3_c48_wgrad_ws_floats(void) { return (int64_t)k_conv48_wgrad_ws_floats(); }
int nmh_conv3d_k3_wgrad(int dt, const void* dY, const void* X, float* dW, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
  CLR();
  return k_conv3_tn(dt, dY, X, dW, B, D, H, W, Cin, Cout, ST);
}
int nmh_layernorm_fwd(int dt, int src_mode, const void* x, void* out, const float* gamma, const float* beta, float eps, float* mean, float* rstd, int64_t rows, int C,
                      const int* wm, const float* pos, const unsigned char* mask, const float* mask_token, int64_t tokens_per_sample, void* stream) {
  CLR();
  if (rows <= 0) return 0;
  LnArgs a{dt, src_mode, x, out, gamma, beta, eps, mean, rstd, (long)rows, C, to_wm(wm), pos, mask, mask_token, (long)(tokens_per_sample > 0 ? tokens_per_sample : 1), nullptr};
  return k_ln_fwd(a, ST);
}
int nmh_layernorm_fwd_window_tokens(int dt, const void* x, void* out_window, void* out_tokens, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                                    int64_t rows, int C, const int* wm, void* stream) {
  CLR();
  REQ(x, out_window, out_tokens, gamma, beta, mean, rstd, wm);
  if (rows <= 0) return 0;
  LnArgs a{dt, 1, x, out_window, gamma, beta, eps, mean, rstd, (long)rows, C, to_wm(wm), nullptr, nullptr, nullptr, 1, out_tokens};
  return k_ln_fwd(a, ST);
}
int nmh_layernorm_bwd(int dt, int src_mode, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx,
                      float* dgamma, float* dbeta, int64_t rows, int C, const int* wm, const unsigned char* mask, float* dmask_token, int64_t tokens_per_sample,
                      void* dyw, const float* dyw_scale, void* stream) {
  CLR();
  if (rows <= 0) return 0;
  LnBwdArgs a{dt, src_mode, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, C, to_wm(wm), mask, dmask_token, (long)(tokens_per_sample > 0 ? tokens_per_sample : 1),
              dyw, dyw_scale, 0, nullptr};
  return k_ln_bwd(a, ST);
}
int64_t nmh_layernorm_bwd_partial_rows(int64_t rows, int C) { return rows > 0 && C > 0 && C % 8 == 0 ? (int64_t)k_ln_bwd_blocks((long)rows, C) : 0; }
int nmh_layernorm_bwd_deferred(int dt, int src_mode, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx,
                               float* partials, int64_t rows, int C, const int* wm, void* dyw, const float* dyw_scale, int64_t tokens_per_sample, void* stream) {
  CLR();
  REQ(dy, x, gamma, mean, rstd, dx, partials);
  if (rows <= 0) return 0;
  LnBwdArgs a{dt, src_mode, dy, x, gamma, mean, rstd, dres, dx, nullptr, nullptr, (long)rows, C, to_wm(wm), nullptr, nullptr, (long)(tokens_per_sample > 0 ? tokens_per_sample : 1),
              dyw, dyw_scale, 0, partials};
  return k_ln_bwd(a, ST);
}
int nmh_layernorm_param_grad_reduce(const nmh_ln_reduce_item* items, int n, void* stream) {
  CLR();
  static_assert(sizeof(nmh_ln_reduce_item) == sizeof(LnReduceItem), "layout");
  if (n <= 0) return 0;
  REQ(items);
  return k_ln_param_reduce(reinterpret_cast<const LnReduceItem*>(items), n, ST);
}
int64_t nmh_cconv_pack_numel(void) { return (int64_t)k_cconv_pack_numel(); }
int64_t nmh_cconv_pack_ws_floats(void) { return (int64_t)k_cconv_pack_ws_floats(); }
int nmh_cconv_pack(const float* Wt, const float* W1, const float* bt, void* Wcp, float* delta, float* ws, void* stream) {
  CLR();
  REQ(Wt, W1, bt, Wcp, delta, ws);
  return k_cconv_pack(Wt, W1, bt, Wcp, delta, ws, ST);
}
int nmh_cconv_fwd(const void* x, const void* Wcp, const float* delta, void* y1, int B, int v, double* stats_acc, void* stream) {
  CLR();
  REQ(x, Wcp, delta, y1);
  if (B <= 0 || v <= 0) return 0;
  return k_cconv_fwd(x, Wcp, delta, y1, B, v, stats_acc, ST);
}
int nmh_cconv_pack_centered(const float* Wt, const float* W1, const float* bt, void* Wcp, float* delta, float* ws, float* mean_table, void* stream) {
  CLR();
  REQ(Wt, W1, bt, Wcp, delta, ws, mean_table);
  return k_cconv_pack(Wt, W1, bt, Wcp, delta, ws, ST, mean_table);
}
int nmh_cconv_output_mean(const void* x, const float* mean_table, const float* delta, double* class_sums, float* mean, int B, int v, void* stream) {
  CLR();
  REQ(x, mean_table, delta, class_sums, mean);
  if (B <= 0 || v <= 0) return 0;
  return k_cconv_mean(x, mean_table, delta, class_sums, mean, B, v, ST);
}
int nmh_cconv_fwd_centered(const void* x, const void* Wcp, const float* delta, const float* mean, float slope, void* z, int B, int v, double* stats_acc, void* stream) {
  CLR();
  REQ(x, Wcp, delta, mean, z);
  if (B <= 0 || v <= 0) return 0;
  if (!(slope > 0.f && slope < 1.f)) return -2;
  return k_cconv_fwd(x, Wcp, delta, z, B, v, stats_acc, ST, mean, slope);
}
int64_t nmh_cconv_wgrad_ws_floats(void) { return (int64_t)k_cconv_wgrad_ws_floats(); }
int nmh_cconv_wgrad(const void* x, const void* dy1, const float* pack_ws, const float* bt, float* dW1, float* dWt, float* dbt, float* ws, int B, int v, int phase, void* stream) {
  CLR();
  REQ(x, dy1, pack_ws, bt, dW1, ws);
  if (B <= 0 || v <= 0) return 0;
  if (phase < 0 || phase > 2) return -1;
  return k_cconv_wgrad(x, dy1, pack_ws, bt, dW1, dWt, dbt, ws, B, v, phase, ST);
}
int64_t nmh_cconv_dgrad_pack_numel(void) { return (int64_t)k_cconv_dpack_numel(); }
int nmh_cconv_dgrad_pack(const void* Wcp, void* Wdp, void* stream) {
  CLR();
  REQ(Wcp, Wdp);
  return k_cconv_dpack(Wcp, Wdp, ST);
}
int nmh_cconv_dgrad(const void* dy1, const void* Wdp, const void* add, void* dx, int B, int v, void* stream) {
  CLR();
  REQ(dy1, Wdp, dx);
  if (B <= 0 || v <= 0) return 0;
  return k_cconv_dgrad(dy1, Wdp, add, dx, B, v, ST);
}
int64_t nmh_upconv4_pack_numel(void) { return (int64_t)k_upconv4_pack_numel(); }
int nmh_upconv4_pack(const float* pack_ws, void* Wup, void* stream) {
  CLR();
  REQ(pack_ws, Wup);
  return k_upconv4_pack(pack_ws, Wup, ST);
}
int nmh_upconv4_fwd(const void* x, const void* Wup, const float* bt, void* u, int B, int v, void* stream) {
  CLR();
  REQ(x, Wup, bt, u);
  if (B <= 0 || v <= 0) return 0;
  return k_upconv4_fwd(x, Wup, bt, u, B, v, ST);
}
int nmh_mlp_fused_supported(int C) { return k_mlp_fused_supported(C); }
int nmh_mlp_fused_fwd(const void* x1, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* b2, const float* rowscale,
                      int rows_per_scale, void* x2, float* mean, float* rstd, int64_t M, int C, float eps, void* stream) {
  CLR();
  REQ(x1, gamma, beta, W1, b1, W2T, b2, x2);
  if (M <= 0) return 0;
  return k_mlp_fused_fwd(x1, gamma, beta, W1, b1, W2T, b2, rowscale, rows_per_scale, x2, mean, rstd, (long)M, C, eps, ST);
}
int nmh_mlp_fused_bwd(const void* x1, const void* dx2, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* rowscale,
                      int rows_per_scale, void* dx1, void* x1n, void* hact, void* dh, float* dgamma, float* dbeta, void* dyw, const float* dyw_scale, const int* wm,
                      int64_t M, int C, float eps, void* stream) {
  CLR();
  REQ(x1, dx2, gamma, beta, W1, b1, W2T, dx1, x1n, hact, dh, dgamma, dbeta);
  if (M <= 0) return 0;
  if (dyw && !wm) return -4;
  const WinMap w = to_wm(wm);
  return k_mlp_fused_bwd(x1, dx2, gamma, beta, W1, b1, W2T, rowscale, rows_per_scale, dx1, x1n, hact, dh, dgamma, dbeta, dyw, dyw_scale, wm ? &w : nullptr, (long)M, C, eps, ST);
}
int nmh_swin_supported(int C) { return k_swin_supported(C); }
int64_t nmh_swin_stream_numel(int type, int C) { return (int64_t)k_swin_stream_numel(type, C); }
int nmh_swin_pack(const nmh_swin_pack_item* items, int n, void* stream) {
  CLR();
  if (n <= 0) return 0;
  REQ(items);
  static_assert(sizeof(nmh_swin_pack_item) == sizeof(SwinPackItem), "descriptor layouts must agree");
  return k_swin_pack(reinterpret_cast<const SwinPackItem*>(items), n, ST);
}
int nmh_swin_attn_fwd(const void* x, const float* gamma, const float* beta, const void* wstream, const float* bqkv, const float* bias_table, const float* bproj,
                      const float* rowscale, int rows_per_scale, void* xnw, float* mean, float* rstd, void* qkv, void* o, float* lse, void* x1, const int* wm, int C,
                      float eps, int token_saves, void* stream) {
  CLR();
  REQ(x, gamma, beta, wstream, bqkv, bias_table, bproj, xnw, mean, rstd, qkv, o, lse, x1, wm);
  return k_swin_attn_fwd(x, gamma, beta, wstream, bqkv, bias_table, bproj, rowscale, rows_per_scale, xnw, mean, rstd, qkv, o, lse, x1, to_wm(wm), C, eps, token_saves, ST);
}
int nmh_swin_mlp_fwd(const void* x1, const float* gamma, const float* beta, const void* wstream, const float* b1, const float* b2, const float* rowscale, int rows_per_scale,
                     void* x2, void* x1n, void* hp, void* hact, float* mean, float* rstd, int64_t M, int C, float eps, void* split_ws, int64_t split_ws_bytes,
                     void* stream) {
  CLR();
  REQ(x1, gamma, beta, wstream, b1, b2, x2, x1n, hp, mean, rstd);
  if (M <= 0) return 0;
  if ((long)M * 4 * C >= (1L << 31)) return -2;
  return k_swin_mlp_fwd(x1, gamma, beta, wstream, b1, b2, rowscale, rows_per_scale, x2, x1n, hp, hact, mean, rstd, (long)M, C, eps, split_ws, (long)split_ws_bytes, ST);
}
int64_t nmh_swin_mlp_split_ws_bytes(int64_t M, int C) { return k_swin_mlp_split_ws_bytes((long)M, C); }
int nmh_window_scatter_residual(int dt, const void* yw, const void* x, void* out, const float* rowscale, int C, const int* wm, void* stream) {
  CLR();
  return k_window_scatter_residual(dt, yw, x, out, rowscale, C, to_wm(wm), ST);
}
int nmh_window_gather_scale(int dt, const void* dx, void* dyw, const float* rowscale, int C, const int* wm, void* stream) {
  CLR();
  return k_window_gather_scale(dt, dx, dyw, rowscale, C, to_wm(wm), ST);
}
int nmh_window_attn_fwd(int dt, const void* qkv, const float* bias_table, void* out, float* lse, int heads, int C, const int* wm, void* stream) {
  CLR();
  return k_attn_fwd(dt, qkv, bias_table, out, lse, heads, C, to_wm(wm), ST);
}
int nmh_window_attn_fwd_tokens(int dt, const void* qkv, const float* bias_table, void* out_tok, float* lse, int heads, int C, const int* wm, void* stream) {
  CLR();
  REQ(qkv, bias_table, out_tok, lse, wm);
  return k_attn_fwd(dt, qkv, bias_table, out_tok, lse, heads, C, to_wm(wm), ST, 1);
}
int nmh_window_attn_bwd(int dt, const void* qkv, const float* bias_table, const void* dout, const float* lse, void* dqkv, float* dbias_table, int heads, int C,
                        const int* wm, void* stream) {
  CLR();
  return k_attn_bwd(dt, qkv, bias_table, dout, lse, dqkv, dbias_table, heads, C, to_wm(wm), ST);
}
int nmh_window_attn_bwd_tokens(int dt, const void* qkv, const float* bias_table, const void* dout_tok, const float* lse, void* dqkv_tok, void* dqkv_pad, float* dbias_table,
                               int heads, int C, const int* wm, void* stream) {
  CLR();
  REQ(qkv, bias_table, dout_tok, lse, dqkv_tok, dqkv_pad, dbias_table, wm);
  return k_attn_bwd(dt, qkv, bias_table, dout_tok, lse, dqkv_pad, dbias_table, heads, C, to_wm(wm), ST, dqkv_tok);
}
int nmh_window_pad_rows_colsum(int dt, const void* x, int N, const int* wm, float* out, void* stream) {
  CLR();
  REQ(x, wm, out);
  return k_attn_pad_rows_colsum(dt, x, N, to_wm(wm), out, ST);
}
int nmh_window_pad_rows_colsum_grouped(int dt, const void* const* xs, float* const* outs, int n, int N, const int* wm, void* stream) {
  CLR();
  REQ(xs, outs, wm);
  return k_attn_pad_rows_colsum_grouped(dt, xs, outs, n, N, to_wm(wm), ST);
}
int nmh_instnorm_stats(int dt, const void* x, float* stats, double* scratch, int B, int64_t V, int C, float eps, void* stream) {
  CLR();
  return k_in_stats(dt, x, stats, scratch, B, (long)V, C, eps, ST);
}
int nmh_instnorm_apply(int dt, const void* x, const float* stats, const void* r, const float* stats_r, int rmode, void* out, int B, int64_t V, int C, float slope, void* stream) {
  CLR();
  return k_in_apply(dt, x, stats, r, stats_r, rmode, out, B, (long)V, C, slope, ST);
}
int nmh_instnorm_bwd_reduce(int dt, const void* dout, const void* out, const void* x, const float* stats, const void* r, const float* stats_r, int rmode, double* sums,
                            double* sums_r, int B, int64_t V, int C, float slope, void* stream) {
  CLR();
  return k_in_bwd_reduce(dt, dout, out, x, stats, r, stats_r, rmode, sums, sums_r, B, (long)V, C, slope, ST);
}
int nmh_instnorm_bwd_apply(int dt, const void* dout, const void* out, const void* x, const float* stats, const double* sums, const void* r, const float* stats_r,
                           const double* sums_r, int rmode, void* dx, void* dr, int dr_accumulate, int B, int64_t V, int C, float slope, void* stream) {
  CLR();
  return k_in_bwd_apply(dt, dout, out, x, stats, sums, r, stats_r, sums_r, rmode, dx, dr, dr_accumulate, B, (long)V, C, slope, ST);
}
int nmh_instnorm_bwd_apply_bg(int dt, const void* dout, const void* x, const float* stats, const double* sums, void* dx, int B, int64_t V, int C, float slope, void* stream) {
  CLR();
  REQ(dout, x, stats, sums, dx);
  return k_in_bwd_apply_bg(dt, dout, x, stats, sums, dx, B, (long)V, C, slope, ST);
}
int nmh_patch_embed_gather(int dt, const float* x, void* A, int B, int R, void* stream) {
  CLR(); return k_embed_gather(dt, x, A, B, R, ST); }
int nmh_patch_embed_kept_rows(const unsigned char* mask, int n, int cap, int* rowmap, void* stream) {
  CLR();
  REQ(mask, rowmap);
  return k_mask_rowmap(mask, n, cap, rowmap, ST);
}
int nmh_patch_embed_gather_kept(int dt, const float* x, void* A, int B, int R, const int* rowmap, int64_t cap_rows, void* stream) {
  CLR();
  REQ(x, A, rowmap);
  return k_embed_gather(dt, x, A, B, R, ST, rowmap, (long)cap_rows);
}
int nmh_patch_embed_norm_fwd_kept(int dt, const void* y0, void* tok, const float* gamma, const float* beta, float eps, float* mean, float* rstd, int64_t rows, int C, const float* pos,
                                  const unsigned char* mask, const float* mask_token, int64_t tokens_per_sample, const int* rowmap, int64_t cap_rows, void* stream) {
  CLR();
  REQ(y0, tok, gamma, beta, mean, rstd, mask, mask_token, rowmap);
  if (rows <= 0) return 0;
  if (tokens_per_sample <= 0 || rows % tokens_per_sample || cap_rows <= 0) return -2;
  LnArgs a{dt, 0, y0, tok, gamma, beta, eps, mean, rstd, (long)rows, C, WinMap{}, pos, mask, mask_token, (long)tokens_per_sample, nullptr, rowmap, (long)cap_rows};
  return k_ln_fwd(a, ST);
}
int nmh_patch_embed_norm_bwd_kept(int dt, const void* dtok, const void* y0, const float* gamma, const float* mean, const float* rstd, void* dy0, float* dgamma, float* dbeta,
                                  int64_t rows, int C, const unsigned char* mask, float* dmask_token, int64_t tokens_per_sample, const int* rowmap, int64_t cap_rows, void* stream) {
  CLR();
  REQ(dtok, y0, gamma, mean, rstd, dy0, dgamma, dbeta, mask, dmask_token, rowmap);
  if (rows <= 0) return 0;
  if (tokens_per_sample <= 0 || rows % tokens_per_sample || cap_rows <= 0) return -2;
  LnBwdArgs a{dt, 0, dtok, y0, gamma, mean, rstd, nullptr, dy0, dgamma, dbeta, (long)rows, C, WinMap{}, mask, dmask_token, (long)tokens_per_sample, nullptr, nullptr, 0, nullptr,
              rowmap, (long)cap_rows};
  return k_ln_bwd(a, ST);
}
int nmh_upconv_shuffle_fwd(int dt, const void* upre, const float* bias, const void* skip, void* out, int B, int v, int k, int Cout, void* stream) {
  CLR();
  return k_up_cat_fwd(dt, upre, bias, skip, out, B, v, k, Cout, ST);
}
int nmh_upconv_shuffle_bwd(int dt, const void* dcat, void* dupre, void* dskip, float* dbias, int B, int v, int k, int Cout, int has_skip, void* stream) {
  CLR();
  return k_up_cat_bwd(dt, dcat, dupre, dskip, dbias, B, v, k, Cout, has_skip, ST);
}
int nmh_mae_loss_fwd(int dt, const void* d0, const float* Wout, const float* bout, const float* target, const int* extents, const unsigned char* tokmask, int B, int R,
                     int Cd, double* sums, float* losses, float* pred, float* dpred, void* stream) {
  CLR();
  LossArgs a{dt, d0, Wout, bout, target, extents, tokmask, B, R, Cd, sums, pred, dpred};
  int rc = k_loss_fwd(a, ST);
  if (rc) return rc;
  return k_loss_finalize(sums, losses, ST);
}
int nmh_mae_loss_bwd(int dt, const void* d0, const float* Wout, const float* bout, const float* target, const int* extents, const unsigned char* tokmask, int B, int R,
                     int Cd, const double* sums, void* dd0, void* dpred8, float* dWout, float* dbout, void* stream) {
  CLR();
  LossArgs a{dt, d0, Wout, bout, target, extents, tokmask, B, R, Cd, const_cast<double*>(sums), nullptr, nullptr};
  (void)dpred8;  // kept in the ABI for layout stability; the head weight gradient is now fused into the kernel
  return k_loss_bwd(a, dd0, dWout, dbout, ST);
}
int nmh_mae_tail_fwd(int dt, const void* y, const float* stats, const void* r, void* d0, const float* Wout, const float* bout, const float* target, const int* extents,
                     const unsigned char* tokmask, int B, int R, int C, double* sums, float* losses, float* pred, float* dpred, float slope, double* bwd_sums,
                     unsigned char* sign_mask, void* stream) {
  CLR();
  LossArgs a{dt, nullptr, Wout, bout, target, extents, tokmask, B, R, C, sums, pred, dpred, bwd_sums, sign_mask};
  int rc = k_tail_fwd(a, y, stats, r, d0, slope, ST);
  if (rc) return rc;
  return k_loss_finalize(sums, losses, ST);
}
int64_t nmh_tail_residual_pack_numel(void) { return (int64_t)k_tail_r_pack_numel(); }
int nmh_tail_residual_pack(const float* pack_ws, void* Wr, void* stream) {
  CLR();
  REQ(pack_ws, Wr);
  return k_tail_r_pack(pack_ws, Wr, ST);
}
int nmh_mae_tail_fwd_from_coarse(int dt, const void* y, const float* stats, const void* xcoarse, const void* Wr, const float* bt, const float* Wout, const float* bout,
                                 const float* target, const int* extents, const unsigned char* tokmask, int B, int R, int C, double* sums, float* losses, float* pred,
                                 float* dpred, float slope, double* bwd_sums, unsigned char* sign_mask, void* stream) {
  CLR();
  REQ(y, stats, xcoarse, Wr, bt, Wout, bout, target, extents, tokmask, sums, losses, dpred, bwd_sums, sign_mask);
  LossArgs a{dt, nullptr, Wout, bout, target, extents, tokmask, B, R, C, sums, pred, dpred, bwd_sums, sign_mask};
  int rc = k_tail_fwd_coarse(a, y, stats, xcoarse, Wr, bt, slope, ST);
  if (rc) return rc;
  return k_loss_finalize(sums, losses, ST);
}
int nmh_mae_tail_bwd(int dt, const void* d0, const void* r, const void* y, const float* stats, const float* dpred, const double* loss_sums, const float* Wout, double* in_sums,
                     void* dy, void* dr, float slope, float* dWout, float* dbout, int B, int64_t V, int C, const double* bwd_sums, const unsigned char* sign_mask,
                     void* stream) {
  CLR();
  return k_tail_bwd(dt, d0, r, y, stats, dpred, loss_sums, Wout, in_sums, dy, dr, slope, dWout, dbout, B, (long)V, C, bwd_sums, ST, sign_mask);
}
int nmh_grid_prepare(int src_u8, const void* src, int W, int L, int H, float* dst, int R, int flags, void* stream) {
  CLR();
  return k_grid_prepare(src_u8, src, W, L, H, dst, R, flags, ST);
}
int nmh_bias_grad(int dt, const void* dY, float* db, int64_t M, int N, const float* rowscale, int rows_per_scale, void* stream) {
  CLR();
  if (M <= 0) return 0;
  return k_bias_grad(dt, dY, db, (long)M, N, rowscale, rows_per_scale > 0 ? rows_per_scale : 1, ST);
}
int nmh_add(int dt, const void* a, const void* b, void* out, int64_t n, void* stream) {
  CLR();
  if (n <= 0) return 0;
  if (!a || !b || !out) return -4;
  return k_add(dt, out, a, b, (long)n, ST);
}
int nmh_add_inplace(int dt, void* a, const void* b, int64_t n, void* stream) {
  CLR(); return k_add_inplace(dt, a, b, (long)n, ST); }
int nmh_grid_to_cl8(int dt, const float* grid, void* out, int B, int64_t V, void* stream) {
  CLR();
  if (!grid || !out) return -4;
  return k_grid_to_cl8(dt, grid, out, B, (long)V, ST);
}
int nmh_head_upsample_fwd(int dt, const void* y, float* pred, int B, int Co, int Cp, int R, int Ro, float inv_scale, void* stream) {
  CLR();
  if (!y || !pred || Co > Cp) return -4;
  return k_cl_to_ncdhw_up(dt, y, pred, B, Co, Cp, R, Ro, inv_scale, ST);
}
int nmh_head_upsample_bwd(int dt, const float* dpred, void* g, int B, int Co, int Cp, int R, int Ro, float inv_scale, void* stream) {
  CLR();
  if (!dpred || !g || Co > Cp) return -4;
  return k_ncdhw_up_adjoint(dt, dpred, g, B, Co, Cp, R, Ro, inv_scale, ST);
}
int nmh_add_cols_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream) {
  CLR();
  if (!src || !dst) return -4;
  return k_add_cols_f32(src, (long)lds, dst, (long)ldd, (long)M, C, ST);
}
int nmh_voxel_sr_loss_fwd(const float* pred, const float* target, int B, int64_t V, double* sums, float* loss, void* stream) {
  CLR();
  if (!pred || !target || !sums || !loss) return -4;
  return k_sr_loss_fwd(pred, target, B, (long)V, sums, loss, ST);
}
int nmh_voxel_sr_loss_bwd(const float* pred, const float* target, int B, int64_t V, const double* sums, float gscale, float* dpred, void* stream) {
  CLR();
  if (!pred || !target || !sums || !dpred) return -4;
  return k_sr_loss_bwd(pred, target, B, (long)V, sums, gscale, dpred, ST);
}
int nmh_masked_ce_fwd(const float* logits, const float* labels, const float* class_weights, int B, int K, int64_t V, double* sums, double* iou_sums, float* out, void* stream) {
  CLR();
  if (!logits || !labels || !sums || !iou_sums || !out) return -4;
  return k_masked_ce_fwd(logits, labels, class_weights, B, K, (long)V, sums, iou_sums, out, ST);
}
int nmh_masked_ce_bwd(const float* logits, const float* labels, const float* class_weights, int B, int K, int64_t V, const double* sums, float gscale, float* dlogits, void* stream) {
  CLR();
  if (!logits || !labels || !sums || !dlogits) return -4;
  return k_masked_ce_bwd(logits, labels, class_weights, B, K, (long)V, sums, gscale, dlogits, ST);
}
int nmh_step_params(const uint32_t* block_bits, int nb, int g, unsigned char* tokmask, const float* hyper, float* hyper_dev, const int* extents, int n_ext, int* extents_dev,
                    void* stream) {
  CLR();
  if ((tokmask && !block_bits) || (hyper && !hyper_dev) || (extents && !extents_dev)) return -4;
  return k_step_params(block_bits, nb, g, tokmask, hyper, hyper_dev, extents, n_ext, extents_dev, ST);
}
int nmh_grad_to_bf16(const float* g, void* bucket, int64_t n, void* stream) {
  CLR();
  if (n <= 0) return 0;
  if (!g || !bucket) return -4;
  return k_grad_cast(1, g, bucket, (long)n, 1.0f, ST);
}
int nmh_grad_from_bf16(const void* bucket, float* g, int64_t n, float scale, void* stream) {
  CLR();
  if (n <= 0) return 0;
  if (!g || !bucket) return -4;
  return k_grad_cast(0, bucket, g, (long)n, scale, ST);
}
char* g_nmh_arena_base = nullptr;
size_t g_nmh_arena_bytes = 0;
int nmh_set_prezeroed_arena(void* base, int64_t bytes) {
  CLR();
  if (bytes < 0 || (bytes > 0 && !base)) return -4;
  g_nmh_arena_base = (char*)base;
  g_nmh_arena_bytes = (size_t)bytes;
  return 0;
}
int nmh_fill_f32(float* p, float v, int64_t n, void* stream) {
  CLR(); return k_fill_f32(p, v, (long)n, ST); }
int nmh_pack_weights(int dt, const void* descs_dev, const int* blk2desc_dev, const int64_t* blkstart_dev, int nblocks, void* stream) {
  CLR();
  static_assert(sizeof(PackDesc) == 40 || sizeof(PackDesc) == 48, "PackDesc layout");
  if (nblocks <= 0) return 0;
  return k_pack_weights(dt, (const PackDesc*)descs_dev, blk2desc_dev, (const long*)blkstart_dev, nblocks, ST);
}
int nmh_grad_sqnorm(const float* g, int64_t n, double* acc, void* stream) {
  CLR(); return k_sqnorm(g, (long)n, acc, ST); }
int nmh_clip_coef(const double* acc, float max_norm, float* coef, float* norm_out, void* stream) {
  CLR(); return k_clip_coef(acc, max_norm, coef, norm_out, ST); }
int nmh_adamw_step(float* p, float* g, float* m, float* v, int64_t n, const float* hyper, const float* coef, void* stream) {
  CLR(); return k_adamw(p, g, m, v, (long)n, hyper, coef, ST); }
}
