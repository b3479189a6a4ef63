// Internal launcher declarations shared by the .hip translation units and the C-ABI layer (capi.hip).
#pragma once
#include <hip/hip_runtime.h>

struct FDiv { unsigned M; int sh; unsigned d; };
FDiv make_fdiv(unsigned d);

// window partition geometry (3-D shifted windows, window edge 4): real dims, padded dims, effective shifts
struct WinMap { int B, H, W, D, PH, PW, PD, s0, s1, s2; };

// GEMM epilogue description (all optional, evaluated in this order):
//   v = acc (+ bias[col]);  act==1: C2 = v, v = gelu(v);  act==2: v *= gelu'(C2);  v *= rowscale[row/rows_per_scale];
//   v += resid;  accumulate: v += C;  C = v.         C, C2, resid share leading dimension ldc.
struct EpiParams {
  void* C; long ldc;
  const float* bias;
  int act; void* C2;
  const void* resid;
  const float* rowscale; int rows_per_scale;
  int accumulate;
  // ConvTranspose3d(kernel = stride = up_k) pixel shuffle folded into the store (up_k > 0): GEMM row m = coarse voxel (b,z,y,x) of a
  // up_v^3 grid, column = tap*up_cout + co  ->  C[fine voxel (b, z*k+tz, y*k+ty, x*k+tx)][co], row stride ldc; bias indexed by co
  int up_k, up_v, up_cout;
  FDiv up_dv, up_dk, up_dc;
  // window reverse folded into the store (win_on): GEMM row = window-ordered row -> token row (pad rows dropped); bias, rowscale
  // (indexed by token / rows_per_scale) and resid (token order) as usual: x1[tok] = x[tok] + s_b * (o . Wproj^T + b)
  int win_on; WinMap wm;
  // split contraction (implicit-GEMM convs on small volumes: too few output tiles to fill the chip, K = 27*Cin long): grid.z = batch * ksplit,
  // every split stores its fp32 accumulators to kpart[split][batch*M][N]; a second launch sums them and applies `accumulate` (no other epilogue)
  int ksplit; int nbatch; float* kpart;
};

// geometry for gemm_tn gather / output remap
struct TnGeom {
  int omode;            // 0: Out[n*ldo+k]; 1: conv3 weight [Cout][Cin][27]; 2: convT weight [Cin][Cout][k3]
  long ldo;
  int Cin, D, H, W; unsigned V;
  FDiv dC, dW, dH, dV;
  float* dbias;         // optional: dbias[n] += sum_m A[m][n]*rs(m) (column sums of the A operand, e.g. a Linear's bias gradient)
  // up_k > 0: the A operand is the pixel-shuffled view of a fine-grid tensor (ConvTranspose3d k = stride backward):
  // A[m][n = tap*Cout + co] = X[fine voxel (m, tap)][co], X row stride up_ldc, Cout = Cin field; dbias is then indexed by co
  int up_k, up_v; long up_ldc; FDiv up_dv, up_dk;
  // split-contraction partials: with a workspace the M splits store plain fp32 partial tiles [split][N][K] (+ [split][N] bias sums)
  // and a second kernel sums them into Out -- instead of gz fp32 global atomics per output element (measured: the atomics are
  // most of the time of the small-M weight gradients)
  float* ws; long ws_floats; float* part;
};


#ifdef __HIPCC__
// window order <-> token order of the 4x4x4 shifted-window partition (pad -> roll(-shift) -> partition; swin_mae3d.py:62-101)
// (32-bit unsigned arithmetic: every row / token count of this path is < 2^31, and a 64-bit div/mod costs ~150 VALU instructions on
//  gfx950 against ~30 for the 32-bit form -- six of them per row made the window gathers / scatters VALU-bound at stage 0)
__device__ __forceinline__ long win_to_tok(const WinMap& w, long m) {
  const unsigned mu = (unsigned)m;
  const int t = (int)(mu & 63u);
  unsigned win = mu >> 6;
  const unsigned nwy = (unsigned)w.PW >> 2, nwx = (unsigned)w.PD >> 2, nwz = (unsigned)w.PH >> 2;
  unsigned q = win / nwx;
  const int wx = (int)(win - q * nwx);
  win = q; q = win / nwy;
  const int wy = (int)(win - q * nwy);
  win = q; q = win / nwz;
  const int wz = (int)(win - q * nwz);
  const unsigned b = q;
  int sz = wz * 4 + (t >> 4) + w.s0, sy = wy * 4 + ((t >> 2) & 3) + w.s1, sx = wx * 4 + (t & 3) + w.s2;
  if (sz >= w.PH) sz -= w.PH;
  if (sy >= w.PW) sy -= w.PW;
  if (sx >= w.PD) sx -= w.PD;
  if (sz >= w.H || sy >= w.W || sx >= w.D) return -1;
  return (long)(((b * (unsigned)w.H + (unsigned)sz) * (unsigned)w.W + (unsigned)sy) * (unsigned)w.D + (unsigned)sx);
}
__device__ __forceinline__ long tok_to_win(const WinMap& w, long tok) {
  const unsigned tu = (unsigned)tok;
  unsigned q = tu / (unsigned)w.D;
  const int x = (int)(tu - q * (unsigned)w.D);
  unsigned q2 = q / (unsigned)w.W;
  const int y = (int)(q - q2 * (unsigned)w.W);
  const unsigned b = q2 / (unsigned)w.H;
  const int z = (int)(q2 - b * (unsigned)w.H);
  int pz = z - w.s0, py = y - w.s1, px = x - w.s2;
  if (pz < 0) pz += w.PH;
  if (py < 0) py += w.PW;
  if (px < 0) px += w.PD;
  const unsigned win = ((b * ((unsigned)w.PH >> 2) + ((unsigned)pz >> 2)) * ((unsigned)w.PW >> 2) + ((unsigned)py >> 2)) * ((unsigned)w.PD >> 2) + ((unsigned)px >> 2);
  return (long)win * 64 + ((pz & 3) << 4) + ((py & 3) << 2) + (px & 3);
}

#endif
int k_gemm_nt(int dt, const void* A, long lda, const void* Bw, long ldb, int M, int N, int K, const EpiParams& ep, hipStream_t st);
// ConvTranspose3d (kernel = stride = k) as GEMMs with the pixel shuffle folded into addressing (unetr_block.py:151-158):
//   fwd:   cat[fine][0:Cout] = x[coarse] . Wt^T + bias   (Wt packed [(tap,co)][ci])
//   dgrad: dx[coarse][ci] = sum_(tap,co) dcat[fine][co] Wd[ci][(tap,co)]
//   wgrad: dW[ci][co][tap] += sum_coarse x[coarse][ci] dcat[fine][co];  dbias[co] += sum dcat
int k_upconv_fwd(int dt, const void* x, const void* Wt, const float* bias, void* cat, long ldc, int B, int v, int k, int Cin, int Cout, hipStream_t st);
int k_upconv_dgrad(int dt, const void* dcat, long ldc, const void* Wd, void* dx, int B, int v, int k, int Cin, int Cout, hipStream_t st);
int k_upconv_wgrad(int dt, const void* dcat, long ldc, const void* x, float* dW, float* dbias, int B, int v, int k, int Cin, int Cout, hipStream_t st);
int k_conv3_nt(int dt, const void* X, const void* Wp, int B, int D, int H, int W, int Cin, int Cout, const EpiParams& ep, hipStream_t st, float* ws = nullptr, long ws_floats = 0);
int k_gemm_tn(int dt, const void* A, long lda, const void* Bm, long ldb, float* Out, long M, int N, int K, const float* rowscale, int rows_per_scale, const TnGeom& gm, hipStream_t st);
// one problem of a grouped weight-gradient launch (host-side descriptor; mirrors nmh_tn_problem in include/nerfmae_hip.h)
struct TnProblemHost {
  const void* A; long lda;        // [M][N] rows = contraction index (dY)
  const void* B; long ldb;        // [M][K] (layer input)
  float* dW; long ldo;            // [N][K] fp32, accumulated
  float* dbias;                   // optional [N]: += column sums of A (times rowscale)
  const float* rowscale;          // optional [M / rows_per_sample]: factor per sample on the rows of A
  long M; int N, K; int rows_per_sample;
  long stride_k;                  // 0: dW[n*ldo + k]; > 0: dW[n*ldo + k*stride_k] (e.g. the [Cin][Cout][k^3] layout of a ConvTranspose3d weight)
  int up_k, up_v;                 // up_k > 0: A is the pixel-shuffled view of a fine-grid tensor (ConvTranspose3d kernel = stride = up_k backward):
                                  // row m = coarse voxel (b,z,y,x) of an up_v^3 grid -> fine row ((b*V+z*k)*V+y*k)*V+x*k, V = up_v*up_k (A already offset by the tap)
  int bias_atomic;                // dbias is shared with other problems of the call: atomic adds
  int n_inner; long stride_n2;    // n_inner > 0: column n = (n / n_inner, n % n_inner) -> dW[(n % n_inner)*ldo + (n / n_inner)*stride_n2 + k*stride_k], dbias[n % n_inner]
};
int k_gemm_tn_grouped(const TnProblemHost* probs, int nprob, float* ws, long ws_floats, hipStream_t st, bool foreground = false);
int k_conv48(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int accumulate, double* stats_acc, hipStream_t st,
             const void* Y1 = nullptr, const float* stats1 = nullptr, float slope = 0.f, long wk_sample_stride = 0, int centered = 0);
int k_conv48_pack_scaled(const float* W, const float* stats, void* out, int B, hipStream_t st);
int k_conv48_mb(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, hipStream_t st);
int k_conv64(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, double* stats_acc, const float* bias, hipStream_t st);
long k_conv64_pack_numel(int Cin, int Cout);
int k_conv64_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, hipStream_t st);
long k_conv64_wgrad_ws_floats();
int k_in_finalize(int dt, const double* acc, float* stats, int B, long V, int C, float eps, hipStream_t st);
int k_conv48_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, hipStream_t st, const float* scale_stats = nullptr);
long k_conv48_wgrad_ws_floats();
int k_conv3_wgrad_halo(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, hipStream_t st);
int k_conv3_tn(int dt, const void* dY, const void* X, float* dW, int B, int D, int H, int W, int Cin, int Cout, hipStream_t st);
int k_nearest_up_add(int dt, const void* coarse, void* fine, int B, int Dc, int Hc, int Wc, int Df, int Hf, int Wf, int C, int bwd, hipStream_t st);
int k_vc_transpose(int dt, const void* src, void* dst, int B, long V, int C, int dir, hipStream_t st);
int k_copy_cols(int dt, const void* src, long lds, void* dst, long ldd, long M, int C, hipStream_t st);

// ---- norm.hip ----
// src_mode: 0 direct rows, 1 window-ordered output rows (gather tokens, pads -> 0), 2 patch-merge gather (8 tokens -> 8C row)
struct LnArgs {
  int dt; int src_mode;
  const void* x; void* out;             // x: token-major [T,C]; out: [rows, Cout] (Cout = C or 8C)
  const float* gamma; const float* beta; float eps;
  float* mean; float* rstd;             // per token (modes 0,1) or per output row (mode 2)
  long rows; int C;                     // rows = output rows; C = normalized width (8*Cin for mode 2)
  WinMap wm;                            // modes 1, 2 (mode 2 uses B,H,W,D as the *input* grid)
  const float* pos; const unsigned char* mask; const float* mask_token; long tokens_per_sample;  // patch-embed post-ops (mode 0)
  void* out_tok;                        // mode 1, optional: a second, TOKEN-ordered copy of the normalised rows [T,C] (operand of a token-ordered weight gradient)
  // mode 0 with a mask, optional (patch embed of the KEPT tokens only, k_mask_rowmap): x holds the kept tokens of every sample in compact rows
  // [B][cap_rows][C] -- token tl of sample b at row b * cap_rows + rowmap[tl]; masked tokens have no row and are not read
  const int* rowmap; long cap_rows;
};
int k_ln_fwd(const LnArgs& a, hipStream_t st);
// mask [n] (1 = removed) -> rowmap [n + 2]: rowmap[t] = number of kept tokens in front of token t (its compact row), -1 for a removed token (and for kept tokens
// beyond cap rows: rowmap[n + 1] = 1 then); rowmap[n] = min(kept count, cap)
int k_mask_rowmap(const unsigned char* mask, int n, int cap, int* rowmap, hipStream_t st);
struct LnBwdArgs {
  int dt; int src_mode;
  const void* dy;                       // mode 0: [T,C]; mode 1: window-ordered [Tw,C]; mode 2: [rows, 8C]
  const void* x; const float* gamma; const float* mean; const float* rstd;
  const void* dres;                     // optional residual gradient added to dx (modes 0,1)
  void* dx;                             // token-major [T,C]
  float* dgamma; float* dbeta;          // fp32 accumulators (atomicAdd)
  long rows; int C; WinMap wm;
  const unsigned char* mask; float* dmask_token; long tokens_per_sample;
  void* dyw; const float* dyw_scale;    // MODE 0 only: second output in window order (wm), scaled per sample (fused window gather)
  int dyw_pads;                         // set by k_ln_bwd: the window-ordered tensor has pad rows (no token) -- the kernel writes their zeros
  float* part;                          // optional [k_ln_bwd_blocks(rows, C)][2C]: every workgroup leaves its dgamma / dbeta sums here (plain stores) instead of adding
                                        // them to dgamma / dbeta with 2C same-address atomics; k_ln_param_reduce adds the column sums later, off the dependent chain
  // mode 0 with a mask, optional (see LnArgs): x and dx are compact [B][cap_rows][C]; masked tokens write no row, the rows behind the kept count are zeroed
  const int* rowmap; long cap_rows;
};
int k_ln_bwd(const LnBwdArgs& a, hipStream_t st);
long k_ln_bwd_blocks(long rows, int C);
struct LnReduceItem { const float* part; float* dgamma; float* dbeta; long nb; int C; int pad_; };   // == nmh_ln_reduce_item
constexpr int LN_REDUCE_MAX = 96;
int k_ln_param_reduce(const LnReduceItem* items, int n, hipStream_t st);

int k_window_scatter_residual(int dt, const void* yw, const void* x, void* out, const float* rowscale, int C, const WinMap& wm, hipStream_t st);
int k_window_gather_scale(int dt, const void* dx, void* dyw, const float* rowscale, int C, const WinMap& wm, hipStream_t st);

// ---- cconv.hip: ConvTranspose3d(96 -> 48, k = s = 4) composed with the 3x3x3 conv that follows it (decoder1, forward) ----
int k_cconv_pack(const float* Wt, const float* W1, const float* bt, void* Wcp, float* delta, float* ws, hipStream_t st, float* Mtab = nullptr);
int k_cconv_mean(const void* x, const float* Mtab, const float* delta, double* C27, float* mhat, int B, int v, hipStream_t st);
long k_cconv_pack_numel();
long k_cconv_pack_ws_floats();
int k_cconv_fwd(const void* X, const void* Wcp, const float* delta, void* Y, int B, int v, double* stats_acc, hipStream_t st, const float* mhat = nullptr, float slope = 0.f);
int k_cconv_wgrad(const void* X, const void* dY, const float* WtT, const float* bt, float* dW1, float* dWt, float* dbt, float* ws, int B, int v, int phase, hipStream_t st);
long k_cconv_dpack_numel();
int k_cconv_dpack(const void* Wcp, void* Wdp, hipStream_t st);
int k_cconv_dgrad(const void* dY, const void* Wdp, const void* add, void* DX, int B, int v, hipStream_t st);
long k_cconv_wgrad_ws_floats();
long k_upconv4_pack_numel();
int k_upconv4_pack(const float* ws, void* Wup, hipStream_t st);
int k_upconv4_fwd(const void* X, const void* Wup, const float* bt, void* Y, int B, int v, hipStream_t st);

// ---- mlp_fused.hip: LN -> fc1 -> GELU -> fc2 -> row-scale -> + residual in one launch (bf16), and its backward ----
int k_mlp_fused_supported(int C);
int k_mlp_fused_fwd(const void* x1, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* b2, const float* rowscale,
                    int rows_per_scale, void* x2, float* mean, float* rstd, long M, int C, float eps, hipStream_t st);
int k_mlp_fused_bwd(const void* x1, const void* dx2, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* rowscale,
                    int rows_per_scale, void* dx1, void* x1n, void* hact, void* dh, float* dgamma, float* dbeta, void* dyw, const float* dyw_scale, const WinMap* wm,
                    long M, int C, float eps, hipStream_t st);

// ---- swin_block.hip: fused Swin-block kernels (bf16, C = 96 NW) ----
struct SwinPackItem { const float* w0; const float* w1; void* dst; int type; int C; };   // mirrors nmh_swin_pack_item
int k_swin_supported(int C);
long k_swin_stream_numel(int type, int C);
int k_swin_pack(const SwinPackItem* items, int n, hipStream_t st);
long k_swin_mlp_split_ws_bytes(long M, int C);
int k_swin_mlp_fwd(const void* x1, const float* gamma, const float* beta, const void* wstream, const float* b1, const float* b2, const float* rowscale, int rows_per_scale,
                   void* x2, void* x1n, void* hp, void* hact, float* mean, float* rstd, long M, int C, float eps, void* split_ws, long split_ws_bytes, hipStream_t st);
int k_swin_attn_fwd(const void* x, const float* gamma, const float* beta, const void* wstream, const float* bqkv, const float* table, const float* bproj,
                    const float* rowscale, int rows_per_scale, void* xnw, float* mean, float* rstd, void* qkv, void* o, float* lse, void* x1,
                    const WinMap& wm, int C, float eps, int token_saves, hipStream_t st);

// instance norm over channels-last [B, V, C]; stats[b][c] = {mean, rstd}
int k_in_stats(int dt, const void* x, float* stats, double* scratch, int B, long V, int C, float eps, hipStream_t st);
// out = lrelu( IN(x) [+ r | + IN(r)] ); rmode 0 none, 1 plain residual, 2 normalized residual (stats_r)
int k_in_apply(int dt, const void* x, const float* stats, const void* r, const float* stats_r, int rmode, void* out, int B, long V, int C, float slope, hipStream_t st);
// sums[b][c] = {sum g, sum g*xhat} (and sums_r for rmode 2), g = dout * lrelu'(out)
int k_in_bwd_reduce(int dt, const void* dout, const void* out, const void* x, const float* stats, const void* r, const float* stats_r, int rmode,
                    double* sums, double* sums_r, int B, long V, int C, float slope, hipStream_t st);
// dx = rstd*(g - S1/V - xhat*S2/V); rmode 1: dr (+)= g ; rmode 2: dr = IN-bwd wrt r
int k_in_bwd_apply(int dt, const void* dout, const void* out, const void* x, const float* stats, const double* sums, const void* r, const float* stats_r,
                   const double* sums_r, int rmode, void* dx, void* dr, int dr_accumulate, int B, long V, int C, float slope, hipStream_t st);
int k_in_bwd_apply_bg(int dt, const void* dout, const void* x, const float* stats, const double* sums, void* dx, int B, long V, int C, float slope, hipStream_t st, int centered = 0);

// ---- attn.hip ----
int k_attn_fwd(int dt, const void* qkv, const float* bias_table, void* out, float* lse, int heads, int C, const WinMap& wm, hipStream_t st, int tok_out = 0);
int k_attn_bwd(int dt, const void* qkv, const float* bias_table, const void* dout, const float* lse, void* dqkv, float* dbias_table, int heads, int C, const WinMap& wm, hipStream_t st, void* dqkv_tok = nullptr);
int k_attn_pad_rows_colsum(int dt, const void* x, int N, const WinMap& wm, float* out, hipStream_t st);
int k_attn_pad_rows_colsum_grouped(int dt, const void* const* xs, float* const* outs, int n, int N, const WinMap& wm, hipStream_t st);

// ---- misc.hip ----
int k_embed_gather(int dt, const float* x, void* A, int B, int R, hipStream_t st, const int* rowmap = nullptr, long cap_rows = 0);
int k_up_cat_fwd(int dt, const void* upre, const float* bias, const void* skip, void* out, int B, int v, int k, int Cout, hipStream_t st);
int k_up_cat_bwd(int dt, const void* dcat, void* dupre, void* dskip, float* dbias, int B, int v, int k, int Cout, int has_skip, hipStream_t st);
struct LossArgs {
  int dt; const void* d0; const float* Wout; const float* bout; const float* target;  // d0 [B,R^3,Cd]; target fp32 NCDHW (B,4,R,R,R)
  const int* extents;                    // [B][3] valid extent per axis (A0,A1,A2)
  const unsigned char* tokmask;          // [g^3] 1 = removed (shared by the batch)
  int B, R, Cd;
  double* sums;                          // [4]: sum_rgb, n_occ, sum_alpha, n_rm  ([8] with dp: + sum over voxels of dp[o])
  float* pred;                           // optional fp32 NCDHW (B,4,R,R,R)
  float* dp;                             // optional fp32 [B*R^3][4]: un-normalised d(loss)/d(pred) (forward only)
  double* bwd_sums;                      // optional (k_tail_fwd): [B][C][4] {sum g_rgb, sum g_rgb*xhat, sum g_a, sum g_a*xhat} + [4][C] head weight-gradient sums,
                                         // all on the un-normalised d(pred): the InstanceNorm-backward / head reductions of the tail backward, taken in the forward pass
  unsigned char* sign_mask;              // optional (k_tail_fwd, bf16 / 48 channels): [B*R^3][8] bytes (Cd = 48: 6 used), bit j of byte c = [d0[voxel][8c + j] > 0]: all the tail
                                         // backward needs of d0 once its sums come from bwd_sums -- it then reads 6 bytes per voxel instead of the 96-byte residual row
};
int k_loss_fwd(const LossArgs& a, hipStream_t st);
int k_loss_finalize(const double* sums, float* losses, hipStream_t st);
int k_loss_bwd(const LossArgs& a, void* dd0, float* dWout, float* dbout, hipStream_t st);
// d0 = lrelu(IN(x)+r) -> out, fused with the 1x1 head and the loss terms (a.d0 unused; sums/pred/dp as in k_loss_fwd)
int k_tail_fwd(const LossArgs& a, const void* x, const float* stats, const void* r, void* out, float slope, hipStream_t st);
// the training form of the pass with the residual r = ConvT_{k=s=4}(xcoarse) + bt formed inside from the coarse tensor (Wr: k_tail_r_pack); d(pred), bwd_sums, sign_mask required
int k_tail_fwd_coarse(const LossArgs& a, const void* x, const float* stats, const void* xcoarse, const void* Wr, const float* bt, float slope, hipStream_t st);
long k_tail_r_pack_numel();
int k_tail_r_pack(const float* ws, void* Wr, hipStream_t st);
// loss backward + backward of d0 = lrelu(IN(x)+r) in two elementwise passes over (d0, x, dp): dx, dr = g; also head weight/bias gradients
int k_tail_bwd(int dt, const void* d0, const void* r, const void* xin, const float* in_stats, const float* dp, const double* loss_sums, const float* Wout, double* in_sums,
               void* dx, void* dr, float slope, float* dWout, float* dbout, int B, long V, int C, const double* bwd_sums, hipStream_t st, const unsigned char* sign_mask = nullptr);
int k_bias_grad(int dt, const void* dY, float* db, long M, int N, const float* rowscale, int rows_per_scale, hipStream_t st);
int k_add_inplace(int dt, void* a, const void* b, long n, hipStream_t st);
int k_add(int dt, void* out, const void* a, const void* b, long n, hipStream_t st);
int k_fill_f32(float* p, float v, long n, hipStream_t st);
// ---- heads.hip (voxel super-resolution / semantics heads) ----
int k_grid_to_cl8(int dt, const float* src, void* dst, int B, long V, hipStream_t st);
int k_cl_to_ncdhw_up(int dt, const void* src, float* dst, int B, int Co, int Cp, int R, int Ro, float inv_scale, hipStream_t st);
int k_ncdhw_up_adjoint(int dt, const float* dpred, void* g, int B, int Co, int Cp, int R, int Ro, float inv_scale, hipStream_t st);
int k_add_cols_f32(const float* src, long lds, float* dst, long ldd, long M, int C, hipStream_t st);
int k_sr_loss_fwd(const float* pred, const float* tgt, int B, long V, double* sums, float* loss, hipStream_t st);
int k_sr_loss_bwd(const float* pred, const float* tgt, int B, long V, const double* sums, float gscale, float* dpred, hipStream_t st);
int k_masked_ce_fwd(const float* logits, const float* labels, const float* cw, int B, int K, long V, double* sums, double* iou, float* out, hipStream_t st);
int k_masked_ce_bwd(const float* logits, const float* labels, const float* cw, int B, int K, long V, const double* sums, float gscale, float* dlogits, hipStream_t st);
int k_step_params(const unsigned* bits, int nb, int g, unsigned char* tokmask, const float* hyper_host, float* hyper_dev, const int* ext_host, int n_ext, int* ext_dev, hipStream_t st);
int k_grad_cast(int to_bf16, const void* src, void* dst, long n, float scale, hipStream_t st);

struct PackDesc { const float* src; void* dst; int mode; int d0, d1, d2; long n; };
int k_pack_weights(int dt, const PackDesc* descs_dev, const int* blk2desc_dev, const long* blkstart_dev, int nblocks, hipStream_t st);
int k_sqnorm(const float* g, long n, double* acc, hipStream_t st);
int k_clip_coef(const double* acc, float max_norm, float* coef, float* norm_out, hipStream_t st);
int k_adamw(float* p, float* g, float* m, float* v, long n, const float* hyper /*lr,b1,b2,eps,wd,bc1,bc2,zero_g*/, const float* coef, hipStream_t st);

// raw scene (W,L,H,4) fp32|uint8 -> padded (4,R,R,R) fp32 network input; flags: 1 rotate (z-up 90 deg), 2 flip axis 0, 4 flip axis 1, 8 density->alpha
int k_grid_prepare(int src_u8, const void* src, int W, int L, int H, float* dst, int R, int flags, hipStream_t st);
