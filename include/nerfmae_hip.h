/* nerfmae_hip.h -- C ABI of libnerfmae_hip.so: the MI355X (gfx950) kernels of the NeRF-MAE 3-D Swin MAE
 * pre-training hot path (reference class SwinTransformer_MAE3D_New, nerf_mae/model/mae/swin_mae3d.py:1067-1599).
 *
 * The reference has no FFI on this path -- every op is a stock PyTorch call.  This header is therefore
 * the native-op surface a maintainer would bind in place of those calls (INTEGRATION.md shows the ctypes / pybind
 * stubs); each entry cites the reference op it replaces.  Conventions:
 *   - plain device pointers + sizes, no torch types; all launches are asynchronous on `stream` (a hipStream_t);
 *   - return 0 on success, a hipError_t (>0) or a negative argument-error code otherwise (nmh_error_string); a NULL in a required
 *     pointer argument returns -4 (invalid argument), never a device fault;
 *   - `dt`: storage/compute type of activations and packed weights: 0 = fp32 (exact fp32 MFMA; the 1e-3-parity
 *     mode), 1 = bf16 (bf16 MFMA, fp32 accumulate).  Parameters, gradients and statistics are always fp32.
 *   - activations are channels-last token/voxel-major matrices [rows, C]; rows of a (B,A0,A1,A2,C) volume are
 *     ((b*A0+a0)*A1+a1)*A2+a2 (A2 fastest) -- the reference's (B,H,W,D,C) order;
 *   - `wm` points to 10 host ints {B, H, W, D, PH, PW, PD, s0, s1, s2}: token grid, grid padded to multiples of 4,
 *     effective cyclic shifts (0 where window >= padded size; swin_mae3d.py:62-81).
 * Every line below that starts with NMH_API is parsed by the Python binding and by tests/test_host_cpu.py::test_capi_library_loads_and_exports_every_declared_symbol.
 */
#ifndef NERFMAE_HIP_H
#define NERFMAE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define NMH_API __attribute__((visibility("default")))

NMH_API int nmh_version(void);
NMH_API const char* nmh_error_string(int code);

/* C[M,N] = epi(A[M,K] . W[N,K]^T): replaces F.linear (swin_mae3d.py:108,173; torchvision MLP :352-358; PatchMerging.reduction
 * :413), the patch conv as GEMM (:1120-1126), ConvTranspose3d k=s as GEMM (unetr_block.py:151-158) and 1x1x1 convs (:52-55).
 * epilogue order: +bias[N]; act 1: C2=pre-activation, exact-erf GELU; act 2: *= gelu'(C2); *= rowscale[row/rows_per_scale]
 * (stochastic depth, :367-368); += resid; accumulate: += C.  C, C2, resid share ldc. */
NMH_API int nmh_gemm_nt(int dt, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K, void* C, int64_t ldc, const float* bias, int act, void* C2, const void* resid, const float* rowscale, int rows_per_scale, int accumulate, void* stream);
/* Attention output projection with the window reverse folded into the store (swin_mae3d.py:176-197 + the residual of :367):
 * out[tok] = resid[tok] + rowscale[tok / tokens_per_sample] * (A[m] . W^T + bias) for every window-ordered row m that maps to a token
 * (pad rows are dropped); wm as in nmh_window_scatter_residual, which this replaces. */
NMH_API int nmh_gemm_nt_window_scatter(int dt, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K, void* out, const void* resid, const float* bias, const float* rowscale, int tokens_per_sample, const int* wm, void* stream);
/* dW[N,K] += sum_m A[m,N]*rowscale . B[m,K]  (fp32 atomics): weight gradients of the ops above.
 * omode 0: dW[n*ldo+k]; omode 2: ConvTranspose3d weight [Cin=K][Cout=p0][k3=p1] with n = tap*Cout+co.
 * dbias (optional): dbias[n] += sum_m A[m,n]*rowscale -- the layer's bias gradient from the same pass over A.
 * ws (optional, ws_floats fp32): scratch for the partial tiles of a split contraction; with it the splits are summed by a second
 * launch instead of fp32 global atomics (one scratch per stream: concurrent calls must not share it). */
NMH_API int nmh_gemm_tn(int dt, const void* A, int64_t lda, const void* B, int64_t ldb, float* dW, int64_t M, int N, int K, const float* rowscale, int rows_per_scale, int omode, int64_t ldo, int p0, int p1, float* dbias, float* ws, int64_t ws_floats, void* stream);
/* Grouped form of nmh_gemm_tn for bf16 operands (dt must be 1): `probs` is a HOST array of nprob descriptors; every problem is
 * dW[N,K] += sum_m rowscale[m / rows_per_sample] * A[m,N]^T . B[m,K] (+ dbias[N] += column sums of A), M a multiple of rows_per_sample.
 * The four Linear weight gradients of every Swin block of a stage (swin_mae3d.py:366-369 backward: attn.qkv, attn.proj, mlp.0, mlp.3)
 * and the PatchMerging reduction (:413) are issued through this entry once the stage's input-gradient chain is done: up to 16 problems
 * per kernel launch (descriptors travel as kernel arguments: graph-capturable), one 96x96 output tile per workgroup, contraction
 * splits only when a launch has too few tiles to fill the chip (partials in ws, summed by one reduce launch per group).
 * Optional geometry (0 = off): stride_k > 0 writes dW[n*ldo + k*stride_k]; up_k/up_v make A the pixel-shuffled view of a fine-grid tensor, so
 * the tap problems of a ConvTranspose3d(kernel = stride) weight gradient (unetr_block.py:151-158 backward) run as one grouped call that
 * reads the fine gradient once (A = dcat + tap offset, dW = weight + tap, ldo = k^3, stride_k = Cout*k^3, bias_atomic = 1).
 * n_inner > 0: column n of A is the pair (n / n_inner, n % n_inner) and lands at dW[(n % n_inner)*ldo + (n / n_inner)*stride_n2 + k*stride_k],
 * its bias gradient at dbias[n % n_inner] -- the k taps along x of one (tz, ty) tap row are then ONE problem whose A rows are the contiguous
 * k*Cout-element runs of the fine gradient -- with up_k > 0 column n is read at fine row (row + n / n_inner), channel n % n_inner, which is
 * contiguous when lda == n_inner and k strided pieces otherwise (a skip half in the row) -- k^2 problems with full 96-column tiles instead of k^3 with 48. */
typedef struct nmh_tn_problem { const void* A; int64_t lda; const void* B; int64_t ldb; float* dW; int64_t ldo; float* dbias; const float* rowscale; int64_t M; int N; int K; int rows_per_sample;
  int64_t stride_k; int up_k; int up_v; int bias_atomic; int n_inner; int64_t stride_n2; } nmh_tn_problem;
NMH_API int nmh_gemm_tn_grouped(int dt, const nmh_tn_problem* probs, int nprob, float* ws, int64_t ws_floats, void* stream);
/* The same call for a group that has the chip to itself -- the flush of the FIRST encoder stage, issued when the backward pass has nothing but
 * the patch-embedding backward left (swin_mae3d.py:1455-1463 backward): the contraction splits aim at 640 workgroups instead of the 256 that
 * suit a launch running underneath the input-gradient chain.  Same arguments, same results up to the summation order of the splits. */
NMH_API int nmh_gemm_tn_grouped_fg(int dt, const nmh_tn_problem* probs, int nprob, float* ws, int64_t ws_floats, void* stream);
/* Y[(b,z,y,x)][Cout] (+)= conv3d(k=3,pad=1) of channels-last X with packed weights [Cout][27][Cin] (nn.Conv3d in
 * UnetResBlock, unetr_block.py:35-44).  Input gradients use the same entry with the dgrad pack [Cin][27 flipped][Cout].
 * ws (optional, ws_floats fp32): scratch for a split contraction -- on small volumes (the 10^3 / 20^3 decoder levels: < 256 output tiles,
 * K = 27*Cin up to 20736) the K loop is cut into up to 8 splits whose fp32 partial tiles are summed by a second launch. */
NMH_API int nmh_conv3d_k3(int dt, const void* X, const void* Wp, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, float* ws, int64_t ws_floats, void* stream);
/* The same convolution with a per-output-channel bias in the epilogue (nn.Conv3d(C, C, 3, padding=1) of the FPN neck,
 * nerf_rpn/model/fpn.py:104). */
NMH_API int nmh_conv3d_k3_bias(int dt, const void* X, const void* Wp, const float* bias, void* Y, int B, int D, int H, int W, int Cin, int Cout, void* stream);
/* FPN top-down pathway (nerf_rpn/model/fpn.py:150-159): fine[b,zf,yf,xf,:] += coarse[b,src(zf),src(yf),src(xf),:] with
 * F.interpolate(mode="nearest", size=fine) source indices src(i) = min(floor(i * (float)in/out), in-1); channels-last, C % 8 == 0.
 * _bwd is its adjoint: dcoarse += sum of dfine over the voxels that read it. */
NMH_API int nmh_nearest_upsample_add(int dt, const void* coarse, void* fine, int B, int Dc, int Hc, int Wc, int Df, int Hf, int Wf, int C, void* stream);
NMH_API int nmh_nearest_upsample_add_bwd(int dt, const void* dfine, void* dcoarse, int B, int Dc, int Hc, int Wc, int Df, int Hf, int Wf, int C, void* stream);
/* dst[m][0..C) = src[m][0..C), row strides lds / ldd in elements: the skip connection entering (and its gradient leaving) the
 * channel-concatenated decoder tensor -- torch.cat((out, skip), dim=1), unetr_block.py:196-197, in channels-last. */
NMH_API int nmh_copy_cols(int dt, const void* src, int64_t lds, void* dst, int64_t ldd, int64_t M, int C, void* stream);
/* channels-last compute tensor [B][V][C] (dt) <-> fp32 NCDHW feature map [B][C][V], the layout nerf_rpn's heads consume
 * (torch.permute(x,[0,4,1,2,3]).contiguous(), feature_extractor.py:1183-1185), and back for the incoming gradient. */
NMH_API int nmh_ndhwc_to_ncdhw(int dt, const void* src, float* dst, int B, int64_t V, int C, void* stream);
NMH_API int nmh_ncdhw_to_ndhwc(int dt, const float* src, void* dst, int B, int64_t V, int C, void* stream);
/* The same convolution specialised for Cin = Cout = 48, bf16 (decoder1.conv_block at 160^3 -- 75 % of the model's FLOPs): persistent
 * LDS-halo implicit GEMM; Wk = fragment-ordered pack [41 steps][3][64 lanes][8] (pack modes 6 fwd / 7 dgrad). */
NMH_API int nmh_conv3d_k3_c48(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int accumulate, double* stats_acc, void* stream);
/* The input-gradient launch of that kernel when the conv's INPUT was LeakyReLU(InstanceNorm3d(Y1)) (decoder1's conv2, unetr_block.py:60-63 backward):
 * dX = conv^T(dY) as nmh_conv3d_k3_c48 with the dgrad pack, and in the same epilogue the two per-(sample, channel) sums the InstanceNorm backward needs,
 * sums [B][48][2] fp64 (zeroed here) = (sum g, sum g * yhat) with g = dX * lrelu'(Y1 - mean), yhat = (Y1 - mean) * rstd, stats1 = (mean, rstd) pairs
 * [B][48][2] -- what nmh_instnorm_bwd_reduce(dX, Y1, rmode 0) would produce in a separate pass over both tensors. */
NMH_API int nmh_conv3d_k3_c48_bwd_reduce(const void* dY, const void* Wkd, void* dX, int B, int D, int H, int W, const void* Y1, const float* stats1, float slope, double* sums, void* stream);
/* The LDS-halo kernel on 48-channel blocks: Cin, Cout multiples of 48 (the 40^3 decoder level of swin_t/s: 96 / 192 channels;
 * UnetResBlock convs, unetr_block.py:35-44); Wk = one fragment-ordered image per (output block, input block), output-block-major
 * (pack modes 6 / 7 on a [Cout][Cin][27] weight).  No fused statistics. */
NMH_API int nmh_conv3d_k3_c48mb(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, void* stream);
/* The same convolution on 64-channel blocks, bf16, Cin and Cout multiples of 64 (swin_b's decoder1 64 -> 64 at 160^3, BASELINE
 * configs[3]; the FPN neck's 256 -> 256 convolutions, nerf_rpn/model/fpn.py:104): persistent LDS-halo implicit GEMM that loops over the
 * input-channel blocks with the accumulators in registers.  Wk = fragment-ordered pack [Cout/64][Cin/64][54 steps][4][64 lanes][8]
 * (pack modes 8 fwd / 9 dgrad; nmh_conv3d_k3_c64_pack_numel elements).  stats_acc as for the 48-channel kernel: fp64 [B][Cout][2];
 * bias (optional fp32 [Cout]) is added before the store (the FPN convolutions carry a bias, the decoder's do not need theirs). */
NMH_API int nmh_conv3d_k3_c64(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, double* stats_acc, const float* bias, void* stream);
NMH_API int64_t nmh_conv3d_k3_c64_pack_numel(int Cin, int Cout);
/* weight gradient of the 64-channel-block layer (dW fp32 [Cout][Cin][3][3][3], accumulated): X halo + DMA-double-buffered dY tile in
 * swizzled 128-byte LDS rows, both operands by transpose reads, taps split over two workgroup groups; ws = fp32 scratch of
 * nmh_conv3d_k3_c64_wgrad_ws_floats() elements (per-workgroup partials, summed by a second launch). */
NMH_API int nmh_conv3d_k3_c64_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, void* stream);
NMH_API int64_t nmh_conv3d_k3_c64_wgrad_ws_floats(void);
/* stats_acc (optional, fp64 [B][48][2], zeroed by the call): the conv epilogue also accumulates per-(sample,channel) sum and sum of squares of
 * its outputs, i.e. the InstanceNorm3d statistics of the following norm layer; nmh_instnorm_finalize turns them into {mean, rstd}. */
NMH_API int nmh_instnorm_finalize(const double* acc, float* stats, int B, int64_t V, int C, float eps, void* stream);
/* weight gradient of the specialised layer; ws = fp32 scratch of nmh_conv3d_k3_c48_wgrad_ws_floats() elements (per-workgroup partials) */
NMH_API int nmh_conv3d_k3_c48_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, void* stream);
/* the same LDS-halo weight-gradient kernel for any Cin, Cout that are multiples of 48 (bf16): (Cin/48)*(Cout/48) independent 48x48
 * blocks on strided channel slices, every activation row read once per block instead of once per tap; ws as above. */
NMH_API int nmh_conv3d_k3_wgrad_halo(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, void* stream);
NMH_API int64_t nmh_conv3d_k3_c48_wgrad_ws_floats(void);
/* dW[Cout][Cin][3][3][3] (PyTorch layout, fp32) += conv weight gradient */
NMH_API int nmh_conv3d_k3_wgrad(int dt, const void* dY, const void* X, float* dW, int B, int D, int H, int W, int Cin, int Cout, void* stream);

/* LayerNorm eps over the last dim (swin_mae3d.py:341,351,388,1128) with the surrounding data movement folded in:
 * src_mode 0 rows as-is (+ optional patch-embed post-ops: + pos[tok], masked tokens <- mask_token; :1459-1463,1375-1380),
 * src_mode 1 output in window order (pad -> roll -> partition, :62-101; pad rows are zeros),
 * src_mode 2 patch-merge gather of 8 tokens -> 8C row (:390-402).  mean/rstd are saved per token (modes 0,1) / row (2).
 * backward, mode 0: dyw (optional, with wm) receives the same gradient in window order scaled by dyw_scale[tok / tokens_per_sample]
 * (the adjoint of the attention branch's window reverse; replaces nmh_window_gather_scale). */
NMH_API int nmh_layernorm_fwd(int dt, int src_mode, const void* x, void* out, const float* gamma, const float* beta, float eps, float* mean, float* rstd, int64_t rows, int C, const int* wm, const float* pos, const unsigned char* mask, const float* mask_token, int64_t tokens_per_sample, void* stream);
NMH_API int nmh_layernorm_bwd(int dt, int src_mode, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int C, const int* wm, const unsigned char* mask, float* dmask_token, int64_t tokens_per_sample, void* dyw, const float* dyw_scale, void* stream);
/* nmh_layernorm_bwd with the parameter gradients taken off the dependent chain: every workgroup of the launch leaves its dgamma / dbeta partial sums in
 * partials [nmh_layernorm_bwd_partial_rows(rows, C)][2C] (fp32, plain stores) instead of adding them to dgamma / dbeta with 2C same-address atomics, and
 * ONE nmh_layernorm_param_grad_reduce launch for any number of such LayerNorms -- issued wherever the caller runs its weight gradients -- adds the
 * column sums: dgamma[c] += sum_b partials[b][c], dbeta[c] += sum_b partials[b][C + c] (partial_rows = the value nmh_layernorm_bwd_partial_rows gave).
 * The input gradient dx (and dyw) are as nmh_layernorm_bwd's; no token mask (the embedding norm keeps the atomic form).  (norm1 / norm2 of a Swin
 * block, swin_mae3d.py:366-369 backward: 2 x 24 launches per step.) */
typedef struct nmh_ln_reduce_item { const float* partials; float* dgamma; float* dbeta; int64_t partial_rows; int C; int reserved; } nmh_ln_reduce_item;
NMH_API int64_t nmh_layernorm_bwd_partial_rows(int64_t rows, int C);
NMH_API int nmh_layernorm_bwd_deferred(int dt, int src_mode, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx, float* partials, int64_t rows, int C, const int* wm, void* dyw, const float* dyw_scale, int64_t tokens_per_sample, void* stream);
NMH_API int nmh_layernorm_param_grad_reduce(const nmh_ln_reduce_item* items, int n, void* stream);
/* decoder1, forward, bf16: ConvTranspose3d(96 -> 48, kernel = stride = 4) (unetr_block.py:151-158) COMPOSED with the first 3x3x3 conv of the
 * residual block that follows it (unetr_block.py:35-44; swin_mae3d.py:1246-1257): the conv of the up-sampled map at fine voxel 4j + a only
 * touches 1..8 coarse cells j + n, so y1[4j + a] = sum_n x[j + n] . Wc[a][n] with 216 composed 96 x 48 blocks -- 31 instead of 133 kFLOP per
 * voxel; the up-sampled map itself is not read.  nmh_cconv_pack builds the composed weights (fragment order, nmh_cconv_pack_numel() bf16
 * elements) and the [27][48] fp32 border table from the fp32 master parameters: Wt = transp_conv.weight [96][48][4][4][4], W1 =
 * conv_block.conv1.weight [48][48][3][3][3], bt = transp_conv.bias [48]; ws = nmh_cconv_pack_ws_floats() floats of scratch (the two
 * weights transposed to contiguous rows).  nmh_cconv_fwd: x [B][v^3][96] -> y1 [B][(4v)^3][48] (v a multiple
 * of 8); y1 lacks the per-channel constant that the transpose conv's bias contributes in the interior (the affine-free InstanceNorm that
 * follows removes any such constant, like conv1's own bias) but carries its border variation; stats_acc (optional fp64 [B][48][2]) receives
 * sum / sum of squares of the outputs (fused InstanceNorm statistics, as nmh_conv3d_k3_c48).  The residual path still needs the up-sampled
 * map (nmh_upconv4_fwd below); backward: nmh_cconv_wgrad / nmh_cconv_dgrad. */
NMH_API int64_t nmh_cconv_pack_numel(void);
NMH_API int64_t nmh_cconv_pack_ws_floats(void);
NMH_API int nmh_cconv_pack(const float* Wt, const float* W1, const float* bt, void* Wcp, float* delta, float* ws, void* stream);
NMH_API int nmh_cconv_fwd(const void* x, const void* Wcp, const float* delta, void* y1, int B, int v, double* stats_acc, void* stream);
/* The CENTERED form of decoder1's conv1 -> InstanceNorm -> LeakyReLU -> conv2 (unetr_block.py:57-62), round 6.  y1 is linear in the coarse tensor x, so its mean
 * per (sample, channel) is known before y1 exists: sum over the fine voxels = sum_n S_n . M[n] + border constant, S_n = sum of x over the source cells whose
 * target cell lies in the grid, M[n] = the composed blocks summed over the phases (mean_table, [27][96][48] fp32, built by nmh_cconv_pack_centered next to the pack).
 * nmh_cconv_output_mean: x [B][v^3][96] -> mean [B][48] (class_sums: B * (27 * 96 + 48) doubles of scratch).  nmh_cconv_fwd_centered then stores z = lrelu(y1 - mean) and
 * accumulates the statistics of t = y1 - mean (stats_acc as nmh_cconv_fwd: nmh_instnorm_finalize turns them into (residual mean ~ 0, rstd)).  Since rstd > 0 commutes
 * with the LeakyReLU, lrelu(InstanceNorm(y1)) = rstd * z: the scale goes into conv2's weights per sample (nmh_conv48_pack_scaled, nmh_conv3d_k3_c48_per_sample) and the
 * stand-alone normalisation pass over the 160^3 tensor (nmh_instnorm_apply, 6.3 GB at 8 grids) is gone; the backward reads z (nmh_conv3d_k3_c48_bwd_reduce_centered,
 * nmh_instnorm_bwd_apply_bg_centered, nmh_conv3d_k3_c48_wgrad_scaled). */
NMH_API int nmh_cconv_pack_centered(const float* Wt, const float* W1, const float* bt, void* Wcp, float* delta, float* ws, float* mean_table, void* stream);
NMH_API int nmh_cconv_output_mean(const void* x, const float* mean_table, const float* delta, double* class_sums, float* mean, int B, int v, void* stream);
/* conv2 of the centered decoder1: Wk_per_sample [B][41*3*512] bf16 = the forward fragment pack of W[co][ci][tap] * rstd[b][ci] (stats = (mean, rstd) pairs
 * [B][48][2], fp32 master W [48][48][3][3][3]); nmh_conv3d_k3_c48_per_sample = nmh_conv3d_k3_c48 with the weight image of the tile's sample.
 * Backward: _bwd_reduce_centered = nmh_conv3d_k3_c48_bwd_reduce reading z for y1 (y1 - mean = z > 0 ? z : z / slope); _wgrad_scaled = nmh_conv3d_k3_c48_wgrad on
 * (dY, z) with every workgroup's partial scaled by its sample's rstd[ci] in the reduce (B in {1, 2, 4, 8} and whole tile ranges per sample: -2 otherwise);
 * nmh_instnorm_bwd_apply_bg_centered = nmh_instnorm_bwd_apply_bg reading z. */
NMH_API int nmh_conv48_pack_scaled(const float* W, const float* stats, void* Wk_per_sample, int B, void* stream);
NMH_API int nmh_conv3d_k3_c48_per_sample(const void* X, const void* Wk_per_sample, void* Y, int B, int D, int H, int W, double* stats_acc, void* stream);
NMH_API int nmh_conv3d_k3_c48_bwd_reduce_centered(const void* dY, const void* Wkd, void* dX, int B, int D, int H, int W, const void* Z, const float* stats1, float slope, double* sums, void* stream);
NMH_API int nmh_conv3d_k3_c48_wgrad_scaled(const void* dY, const void* Z, const float* stats, float* dW, float* ws, int B, int D, int H, int W, void* stream);
NMH_API int nmh_instnorm_bwd_apply_bg_centered(int dt, const void* dout, const void* z, const float* stats, const double* sums, void* dx, int B, int64_t V, int C, float slope, void* stream);
NMH_API int nmh_cconv_fwd_centered(const void* x, const void* Wcp, const float* delta, const float* mean, float slope, void* z, int B, int v, double* stats_acc, void* stream);
/* Weight gradient of decoder1's conv1 THROUGH the composition above (backward of unetr_block.py:35-44 with respect to conv1.weight; replaces the
 * 48 -> 48 nmh_conv3d_k3_c48_wgrad launch on the up-sampled map): dW1[c][co][d] += sum_a sum_ci Wt[ci][co][(a+d) mod 4] . G[a][n(a,d)][ci][c] with
 * G[a][n] = sum_j x[j+n]^T dy1[4j+a] -- the same 216 blocks as the forward, a quarter of the FLOPs, contraction over the coarse cells.
 * x [B][v^3][96], dy1 [B][(4v)^3][48] (bf16), pack_ws = the scratch nmh_cconv_pack filled in this step (holds the transposed Wt), bt = transp_conv.bias,
 * ws = nmh_cconv_wgrad_ws_floats() floats.  Exact for a dy1 whose per-sample, per-channel sums vanish -- the input gradient of the affine-free
 * InstanceNorm that conv1 feeds (the term bt[co] * sum_p dy1[p][c] is then carried by the border voxels alone, which the entry sums).
 * dWt [96][48][4][4][4] / dbt [48] (fp32, both or neither; NULL = not wanted): += the gradient of the transpose conv's own weight / bias THROUGH conv1, from
 * the same G blocks and border sums; phase: 0 = everything, 1 = the partial G blocks only (the persistent launch), 2 = the small launches behind it (reduce,
 * border sums, chain rules: they only feed weight gradients and may go to a side stream; same ws) (see nmh_cconv_dgrad: with it conv1's input gradient on the fine grid, which nmh_upconv_wgrad would need, does not exist). */
NMH_API int64_t nmh_cconv_wgrad_ws_floats(void);
NMH_API int nmh_cconv_wgrad(const void* x, const void* dy1, const float* pack_ws, const float* bt, float* dW1, float* dWt, float* dbt, float* ws, int B, int v, int phase, void* stream);
/* Input gradient THROUGH the composition: dx = ConvT^T(conv1^T(dy1)) (backward of unetr_block.py:151-158 after unetr_block.py:35-44) is a stride-4
 * convolution of the fine gradient with a 6x6x6 kernel of 48 -> 96 matrices (the transposes of the same 216 blocks): conv1's input gradient on the fine
 * grid -- a full 48 -> 48 conv pass that only fed the transpose conv's backward -- is never formed.  nmh_cconv_dgrad_pack gathers the forward's
 * fragment-ordered weights into this kernel's order (nmh_cconv_dgrad_pack_numel() bf16 elements).  nmh_cconv_dgrad: dy1 [B][(4v)^3][48] ->
 * dx [B][v^3][96] = (add ? add : 0) + gradient (add may alias dx: the part of dx that comes through the residual branch).  With this entry the
 * transpose conv's own parameter gradients through conv1 come from nmh_cconv_wgrad's G blocks: pass dWt [96][48][4][4][4] / dbt [48] (fp32, accumulated;
 * NULL = not wanted) there; the part through the residual branch stays with nmh_upconv_wgrad on the residual gradient alone. */
NMH_API int64_t nmh_cconv_dgrad_pack_numel(void);
NMH_API int nmh_cconv_dgrad_pack(const void* Wcp, void* Wdp, void* stream);
NMH_API int nmh_cconv_dgrad(const void* dy1, const void* Wdp, const void* add, void* dx, int B, int v, void* stream);
/* The residual branch of the same block, bf16: u = ConvTranspose3d(96 -> 48, kernel = stride = 4)(x) + bias (unetr_block.py:151-158, 193-200), as a
 * persistent kernel with the coarse fragments in registers and the 64 phase weights streamed through LDS (replaces nmh_upconv_fwd at this shape).
 * nmh_upconv4_pack: fragment-ordered bf16 weights (nmh_upconv4_pack_numel() elements) from pack_ws = the scratch nmh_cconv_pack filled in this step
 * (holds the transposed Wt).  nmh_upconv4_fwd: x [B][v^3][96] -> u [B][(4v)^3][48], v a multiple of 8. */
NMH_API int64_t nmh_upconv4_pack_numel(void);
NMH_API int nmh_upconv4_pack(const float* pack_ws, void* Wup, void* stream);
NMH_API int nmh_upconv4_fwd(const void* x, const void* Wup, const float* bt, void* u, int B, int v, void* stream);
/* Fused MLP branch of a Swin block, bf16 (swin_mae3d.py:352-358 torchvision MLP + :368 `x + stochastic_depth(mlp(norm2(x)))`):
 *   x2[row] = x1[row] + rowscale[row / rows_per_scale] * (gelu(LN(x1[row]) . W1^T + b1) . W2^T + b2)
 * in ONE launch: LayerNorm in the MFMA operand registers, the hidden dimension walked in chunks whose GELU output feeds the second
 * GEMM from registers -- neither the pre-activation nor the activation is written.  W1 = fc1 weight [4C][C] (bf16, row-major),
 * W2T = fc2 weight TRANSPOSED [4C][C] (bf16; pack mode 1 of nmh_pack_weights), gamma/beta/b1/b2 fp32.  mean/rstd (optional, [M])
 * receive the LayerNorm statistics.  nmh_mlp_fused_supported(C) != 0 for C in {96,128,192,256,384}; other widths return -1 (callers
 * use nmh_layernorm_fwd + 2 x nmh_gemm_nt).  Replaces nmh_layernorm_fwd + nmh_gemm_nt(act=1) + nmh_gemm_nt(resid, rowscale).
 * backward: from dx2 = dL/dx2 recomputes LN and the hidden activations and writes
 *   dx1 = dx2 + LN_backward(dh . W1)       [M][C]      hact = gelu(pre-activation)          [M][4C]  (B operand of dW2 = (s dx2)^T hact)
 *   x1n = LN(x1)                           [M][C]      dh = s (dx2 . W2) * gelu'(pre-act.)  [M][4C]  (A operand of dW1 = dh^T x1n)
 * accumulates dgamma / dbeta (fp32 atomics) and, when dyw != NULL, also stores dx1 in window order scaled by dyw_scale[row /
 * rows_per_scale] (pad rows zeroed; wm as for nmh_layernorm_bwd) -- the attention branch's incoming gradient.  The two weight /
 * bias gradients stay nmh_gemm_tn(_grouped) calls on (dx2, hact, rowscale) and (dh, x1n). */
NMH_API int nmh_mlp_fused_supported(int C);
NMH_API int nmh_mlp_fused_fwd(const void* x1, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* b2, const float* rowscale, int rows_per_scale, void* x2, float* mean, float* rstd, int64_t M, int C, float eps, void* stream);
NMH_API int nmh_mlp_fused_bwd(const void* x1, const void* dx2, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* rowscale, int rows_per_scale, void* dx1, void* x1n, void* hact, void* dh, float* dgamma, float* dbeta, void* dyw, const float* dyw_scale, const int* wm, int64_t M, int C, float eps, void* stream);
/* Fused Swin-block kernels, bf16, widths C = 96 NW with NW in {1, 2, 4} (stages 0-2 of swin_t / swin_s); SURVEY 2a K1 + K2:
 * `x = x + stochastic_depth(attn(norm1(x)))`, `x = x + stochastic_depth(mlp(norm2(x)))` (swin_mae3d.py:366-369) with
 * shifted_window_attention (swin_mae3d.py:27-197) and the torchvision MLP (swin_mae3d.py:352-358) each in ONE forward launch (the backward is the
 * unfused chain: nmh_gemm_nt, nmh_window_attn_bwd(_tokens), nmh_layernorm_bwd(_deferred), nmh_gemm_tn_grouped).
 * A workgroup owns 64 token rows (a 4x4x4 window / 64 consecutive tokens) and keeps them in registers as MFMA operand fragments; every one
 * of its NW waves owns a slice of the output features of every product and streams ITS weights -- a private, linear, pre-packed sequence of
 * 1-KB fragment images -- through an LDS ring with LDS-DMA (csrc/swin_block.hip).
 *   nmh_swin_pack: builds weight streams from the fp32 master parameters (once per optimizer step).  type 0: attention forward
 *     (w0 = qkv.weight [3C][C], w1 = proj.weight [C][C]); 1: MLP forward (w0 = mlp.0.weight [4C][C], w1 = mlp.3.weight [C][4C]).
 *     dst: nmh_swin_stream_numel(type, C) bf16 elements.  Up to any number of items per call.
 *   nmh_swin_attn_fwd: x [T][C] token order -> x1[tok] = x[tok] + rowscale[tok / rows_per_scale] * (proj(attn(LN1(x))) + bproj); pad -> roll ->
 *     window partition and their inverses are address arithmetic (wm as for nmh_layernorm_fwd).  Also writes, in the layouts of the unfused
 *     kernels (nmh_layernorm_fwd src_mode 1, nmh_gemm_nt, nmh_window_attn_fwd): xnw [rows][C] = LN1(x) in window order (pad rows zero), mean / rstd
 *     [T], qkv [rows][3C], o [rows][C], lse [rows * heads] -- the operands of the weight gradients and of the backward chain.
 *     token_saves != 0: xnw and o are written in TOKEN order instead ([T][C]; pad rows, which are zero / carry no gradient, are dropped): the operands of
 *     weight gradients that run on the real tokens only (with nmh_window_attn_bwd_tokens).
 *   nmh_swin_mlp_fwd: x2[row] = x1[row] + rowscale[..] * (gelu(LN2(x1) W1^T + b1) W2^T + b2); also writes x1n = LN2(x1) [M][C], the fc1
 *     pre-activation hp [M][4C], mean / rstd [M] and, when hact != NULL, gelu(hp) [M][4C] (the unfused backward's fc2 weight-gradient operand).
 *     split_ws (optional): nmh_swin_mlp_split_ws_bytes(M, C) bytes, zero before the first call and left zero by every call (calls that share it
 *     must be ordered on one stream).  When that size is > 0 (C = 384 and <= 128 row tiles: fewer workgroups than half of the CUs) and the workspace
 *     is given, two workgroups share a row tile -- half of the hidden units each, fp32 partial sums added by whichever finishes last; the sum of two
 *     terms does not depend on the arrival order, so results are reproducible. */
typedef struct nmh_swin_pack_item { const float* w0; const float* w1; void* dst; int type; int C; } nmh_swin_pack_item;
NMH_API int nmh_swin_supported(int C);
NMH_API int64_t nmh_swin_stream_numel(int type, int C);
NMH_API int nmh_swin_pack(const nmh_swin_pack_item* items, int n, void* stream);
NMH_API int nmh_swin_attn_fwd(const void* x, const float* gamma, const float* beta, const void* wstream, const float* bqkv, const float* bias_table, const float* bproj, const float* rowscale, int rows_per_scale, void* xnw, float* mean, float* rstd, void* qkv, void* o, float* lse, void* x1, const int* wm, int C, float eps, int token_saves, void* stream);
NMH_API int nmh_swin_mlp_fwd(const void* x1, const float* gamma, const float* beta, const void* wstream, const float* b1, const float* b2, const float* rowscale, int rows_per_scale, void* x2, void* x1n, void* hp, void* hact, float* mean, float* rstd, int64_t M, int C, float eps, void* split_ws, int64_t split_ws_bytes, void* stream);
NMH_API int64_t nmh_swin_mlp_split_ws_bytes(int64_t M, int C);
/* out[tok] = x[tok] + rowscale[b]*yw[window_row(tok)]: window reverse + un-roll + un-pad (:176-196) + residual + stochastic depth */
NMH_API int nmh_window_scatter_residual(int dt, const void* yw, const void* x, void* out, const float* rowscale, int C, const int* wm, void* stream);
/* dyw[window_row] = rowscale[b]*dx[tok] (0 for pad rows): adjoint of the above */
NMH_API int nmh_window_gather_scale(int dt, const void* dx, void* dyw, const float* rowscale, int C, const int* wm, void* stream);
/* softmax(q k^T / sqrt(32) + bias_table[rel_index] + shift_mask(-100)) v per (window, head): swin_mae3d.py:109-172, 200-211.
 * qkv [windows*64, 3C] window-ordered ([q|k|v], head-major), out [windows*64, C], lse [windows*heads*64]. head_dim must be 32. */
NMH_API int nmh_window_attn_fwd(int dt, const void* qkv, const float* bias_table, void* out, float* lse, int heads, int C, const int* wm, void* stream);
NMH_API int nmh_window_attn_bwd(int dt, const void* qkv, const float* bias_table, const void* dout, const float* lse, void* dqkv, float* dbias_table, int heads, int C, const int* wm, void* stream);
/* The same backward with window order confined to the kernel: dout_tok [T][C] is the TOKEN-ordered gradient of the attention output (rows of a window that
 * are pads read as zero -- the reference's x[:, :H, :W, :D] slice, swin_mae3d.py:196), and the rows of d(qkv) that belong to a token are written to
 * dqkv_tok [T][3C] at their token, so that the Linear layers either side (proj, qkv: input gradients and weight gradients) run on the T real tokens
 * instead of the padded window rows (58 % of them at 10^3 tokens in 12^3, 24 % at 5^3 in 8^3).  Pad rows' d(qkv) -- not zero: pad tokens are keys and
 * values with q = k = v = bias -- go to dqkv_pad [rows][3C] at their window row (real rows of that buffer are not touched), and
 * nmh_window_pad_rows_colsum adds their column sums to the qkv bias gradient: out[n] += sum over pad rows of x[row][n]. */
/* Forward counterparts for the UNFUSED chain (small launches, fp32 parity mode): nmh_layernorm_fwd_window_tokens = nmh_layernorm_fwd src_mode 1 with a second,
 * token-ordered copy of the normalised rows (out_tokens [T][C]); nmh_window_attn_fwd_tokens = nmh_window_attn_fwd with its output scattered to token
 * order (out_tok [T][C], pad rows dropped) -- proj is then a plain T-row nmh_gemm_nt with the residual / row-scale epilogue. */
NMH_API int nmh_layernorm_fwd_window_tokens(int dt, const void* x, void* out_window, void* out_tokens, const float* gamma, const float* beta, float eps, float* mean, float* rstd, int64_t rows, int C, const int* wm, void* stream);
NMH_API int nmh_window_attn_fwd_tokens(int dt, const void* qkv, const float* bias_table, void* out_tok, float* lse, int heads, int C, const int* wm, void* stream);
NMH_API int nmh_window_attn_bwd_tokens(int dt, const void* qkv, const float* bias_table, const void* dout_tok, const float* lse, void* dqkv_tok, void* dqkv_pad, float* dbias_table, int heads, int C, const int* wm, void* stream);
NMH_API int nmh_window_pad_rows_colsum(int dt, const void* x, int N, const int* wm, float* out, void* stream);
/* the same for n buffers of one geometry in ONE launch (xs / outs: HOST arrays of n device pointers, copied into the kernel arguments): the blocks of an
 * encoder stage, issued with the stage's grouped weight gradients */
NMH_API int nmh_window_pad_rows_colsum_grouped(int dt, const void* const* xs, float* const* outs, int n, int N, const int* wm, void* stream);

/* InstanceNorm3d (eps, no affine, biased var) + LeakyReLU(slope) + residual over channels-last [B][V][C] (unetr_block.py:57-71).
 * stats[b][c] = {mean, rstd}; scratch/sums: fp64 [B][C][2].  rmode 0: lrelu(IN(x)); 1: lrelu(IN(x)+r); 2: lrelu(IN(x)+IN(r)). */
NMH_API int nmh_instnorm_stats(int dt, const void* x, float* stats, double* scratch, int B, int64_t V, int C, float eps, void* stream);
NMH_API int nmh_instnorm_apply(int dt, const void* x, const float* stats, const void* r, const float* stats_r, int rmode, void* out, int B, int64_t V, int C, float slope, void* stream);
NMH_API int nmh_instnorm_bwd_reduce(int dt, const void* dout, const void* out, const void* x, const float* stats, const void* r, const float* stats_r, int rmode, double* sums, double* sums_r, int B, int64_t V, int C, float slope, void* stream);
NMH_API int nmh_instnorm_bwd_apply(int dt, const void* dout, const void* out, const void* x, const float* stats, const double* sums, const void* r, const float* stats_r, const double* sums_r, int rmode, void* dx, void* dr, int dr_accumulate, int B, int64_t V, int C, float slope, void* stream);
/* the rmode-0 apply pass with the sign taken from x (out = NULL), bf16, C = 48, as a background launch: one persistent workgroup per CU with a footprint
 * (<= 96 VGPRs, no dynamic LDS) that fits on a CU beside the persistent 48 -> 48 weight-gradient kernel -- decoder-1's InstanceNorm backward
 * (unetr_block.py:57-63 backward) issued on a forked stream next to nmh_conv3d_k3_c48_wgrad.  Bit-identical to nmh_instnorm_bwd_apply. */
NMH_API int nmh_instnorm_bwd_apply_bg(int dt, const void* dout, const void* x, const float* stats, const double* sums, void* dx, int B, int64_t V, int C, float slope, void* stream);

/* im2row of the 4x4x4 stride-4 patch conv input: x fp32 (B,4,R,R,R) -> A[(b,z,y,x)][256] (swin_mae3d.py:1120-1126) */
NMH_API int nmh_patch_embed_gather(int dt, const float* x, void* A, int B, int R, void* stream);
/* Patch embed of the KEPT tokens only (the embedding of a removed token is replaced by mask_token right after the LayerNorm, swin_mae3d.py:1375-1380: at
 * mask_ratio 0.75 three quarters of the im2row, of the GEMM rows, of the LayerNorm rows and of the weight-gradient contraction are work on values nobody reads).
 * nmh_patch_embed_kept_rows: token mask [n] (1 = removed) -> rowmap [n + 2]: the compact row of every kept token (raster order), -1 for a removed one;
 *   rowmap[n] = kept count (<= cap), rowmap[n + 1] = 1 if kept tokens did not fit into cap rows (they read -1; the caller sizes cap from the mask it drew).
 * nmh_patch_embed_gather_kept: as nmh_patch_embed_gather into A [B][cap_rows][256], rows behind the kept count zeroed.
 * nmh_patch_embed_norm_fwd_kept / _bwd_kept: nmh_layernorm_fwd / _bwd (mode 0, with pos / mask / mask_token) reading y0 [B][cap_rows][C] through rowmap and
 *   writing token rows [rows][C]; the backward writes d(y0) compact (rows behind the kept count zeroed) and adds the removed tokens' gradients to dmask_token. */
NMH_API int nmh_patch_embed_kept_rows(const unsigned char* mask, int n, int cap, int* rowmap, void* stream);
NMH_API int nmh_patch_embed_gather_kept(int dt, const float* x, void* A, int B, int R, const int* rowmap, int64_t cap_rows, void* stream);
NMH_API int nmh_patch_embed_norm_fwd_kept(int dt, const void* y0, void* tok, const float* gamma, const float* beta, float eps, float* mean, float* rstd, int64_t rows, int C,
                                          const float* pos, const unsigned char* mask, const float* mask_token, int64_t tokens_per_sample, const int* rowmap, int64_t cap_rows,
                                          void* stream);
NMH_API int nmh_patch_embed_norm_bwd_kept(int dt, const void* dtok, const void* y0, const float* gamma, const float* mean, const float* rstd, void* dy0, float* dgamma, float* dbeta,
                                          int64_t rows, int C, const unsigned char* mask, float* dmask_token, int64_t tokens_per_sample, const int* rowmap, int64_t cap_rows,
                                          void* stream);
/* ConvTranspose3d(kernel = stride = k) (unetr_block.py:151-158,193-198) as GEMMs with the pixel shuffle folded into the addressing,
 * channels-last.  x [B*v^3][Cin] coarse grid; cat/dcat [B*(v*k)^3][ldc] fine grid, the transpose conv occupies columns [0,Cout)
 * (the skip connection, if any, the rest).  Wt packed [(tap,co)][ci], Wd packed [ci][(tap,co)], dW fp32 [Cin][Cout][k^3]. */
NMH_API int nmh_upconv_fwd(int dt, const void* x, const void* Wt, const float* bias, void* cat, int64_t ldc, int B, int v, int k, int Cin, int Cout, void* stream);
NMH_API int nmh_upconv_dgrad(int dt, const void* dcat, int64_t ldc, const void* Wd, void* dx, int B, int v, int k, int Cin, int Cout, void* stream);
NMH_API int nmh_upconv_wgrad(int dt, const void* dcat, int64_t ldc, const void* x, float* dW, float* dbias, int B, int v, int k, int Cin, int Cout, void* stream);
/* (unfused pieces) ConvTranspose3d(k=stride) pixel shuffle + bias + channel concat with the skip (unetr_block.py:193-198) and its adjoint */
NMH_API int nmh_upconv_shuffle_fwd(int dt, const void* upre, const float* bias, const void* skip, void* out, int B, int v, int k, int Cout, void* stream);
NMH_API int nmh_upconv_shuffle_bwd(int dt, const void* dcat, void* dupre, void* dskip, float* dbias, int B, int v, int k, int Cout, int has_skip, void* stream);
/* UnetOutBlock 1x1 conv (Cd->4) fused with forward_loss (swin_mae3d.py:1513-1549).  target fp32 (B,4,R,R,R); extents [B][3]
 * valid voxels per axis (replaces the pad_tensor ones-mask, torch_utils.py:56-90); tokmask [g^3] 1 = removed token.
 * sums fp64[4] = {sum_rgb, n_occ, sum_alpha, n_removed}; losses fp32[3] = {loss, loss_rgb, loss_alpha}; pred optional (B,4,R,R,R).
 * dpred optional fp32 [B*R^3][4]: un-normalised d(loss)/d(pred) per voxel for nmh_mae_tail_bwd; sums is then fp64[8] (+ sum_v dpred). */
NMH_API int nmh_mae_loss_fwd(int dt, const void* d0, const float* Wout, const float* bout, const float* target, const int* extents, const unsigned char* tokmask, int B, int R, int Cd, double* sums, float* losses, float* pred, float* dpred, void* stream);
NMH_API int nmh_mae_loss_bwd(int dt, const void* d0, const float* Wout, const float* bout, const float* target, const int* extents, const unsigned char* tokmask, int B, int R, int Cd, const double* sums, void* dd0, void* dpred8, float* dWout, float* dbout, void* stream);
/* Forward of the decoder tail in one pass: d0 = lrelu(IN(y) + r) (unetr_block.py:62-71) -> 1x1 head -> loss terms
 * (swin_mae3d.py:1496-1549); arguments as nmh_instnorm_apply (rmode 1) + nmh_mae_loss_fwd.  sign_mask (optional, with bwd_sums, bf16 / 48 channels):
 * [B*R^3][8] bytes (6 used), bit j of byte c = [d0[voxel][8c + j] > 0] -- with the backward's sums taken here, the sign is all nmh_mae_tail_bwd still needs of d0:
 * given the mask it reads 8 bytes per voxel instead of the residual row (r and d0 may then both be NULL there). */
NMH_API int nmh_mae_tail_fwd(int dt, const void* y, const float* stats, const void* r, void* d0, const float* Wout, const float* bout, const float* target, const int* extents, const unsigned char* tokmask, int B, int R, int C, double* sums, float* losses, float* pred, float* dpred, float slope, double* bwd_sums, unsigned char* sign_mask, void* stream);
/* The training form of nmh_mae_tail_fwd for decoder1, whose residual is r = ConvTranspose3d_{k=s=4}(xcoarse) + bt (unetr_block.py:151-158, 193-200): r is formed
 * inside the pass from the coarse tensor xcoarse [B][(R/4)^3][96] (bf16) on the matrix cores instead of being written by nmh_upconv4_fwd and read back -- one
 * 160^3 x 48 write and read less.  Wr: nmh_tail_residual_pack(pack_ws of nmh_cconv_pack) -> nmh_tail_residual_pack_numel() bf16 elements (the 64 phase
 * weights as MFMA fragments).  bf16, C = 48, R a multiple of 4 with (R/4)^3 a multiple of 16, R <= 256; dpred, bwd_sums and sign_mask are required (-4 otherwise:
 * the caller then stores r and uses nmh_mae_tail_fwd).  r enters the sum in fp32 (nmh_mae_tail_fwd reads it rounded to bf16). */
NMH_API int64_t nmh_tail_residual_pack_numel(void);
NMH_API int nmh_tail_residual_pack(const float* pack_ws, void* Wr, void* stream);
NMH_API int nmh_mae_tail_fwd_from_coarse(int dt, const void* y, const float* stats, const void* xcoarse, const void* Wr, const float* bt, const float* Wout, const float* bout, const float* target, const int* extents, const unsigned char* tokmask, int B, int R, int C, double* sums, float* losses, float* pred, float* dpred, float slope, double* bwd_sums, unsigned char* sign_mask, void* stream);
/* Backward of the decoder tail d0 = lrelu(IN(y) + r) -> 1x1 head -> loss in two elementwise passes that never materialise d(d0)
 * (swin_mae3d.py:1496-1549 + unetr_block.py:62-71 backward): d(d0) = Wout^T dpred is recomputed per element from dpred/loss_sums
 * (both from nmh_mae_loss_fwd).  in_sums[b][c] = {sum g, sum g*yhat}, dy = IN-backward, dr = g; dWout/dbout accumulate the head
 * gradients.  stats = {mean, rstd} of y.  d0 may be NULL when r (the residual input of the forward) is given: d0 is then rebuilt bit-exactly
 * from y, stats and r inside the kernel, and nmh_mae_tail_fwd need not store it either (its d0 may be NULL): one 160^3 x 48 write less.
 * bwd_sums (optional, fp64 [B*C*4 + 4*C], zeroed by nmh_mae_tail_fwd): the forward pass also takes the reductions of this backward
 * ({sum g, sum g*yhat} per (sample, channel) and the head weight gradient, RGB and alpha parts of the loss kept apart until their
 * normalisers are known); passed on to nmh_mae_tail_bwd it makes the backward a single apply pass (one 3-tensor read pass less). */
NMH_API int nmh_mae_tail_bwd(int dt, const void* d0, const void* r, const void* y, const float* stats, const float* dpred, const double* loss_sums, const float* Wout, double* in_sums, void* dy, void* dr, float slope, float* dWout, float* dbout, int B, int64_t V, int C, const double* bwd_sums, const unsigned char* sign_mask, void* stream);
/* Input pipeline in one pass (nerf_rpn/datasets.py:88-101,198-233,247-248 + torch_utils.py:56-90): one stored scene
 * rgbsigma (W,L,H,4), fp32 or uint8, already in device memory -> the (4,R,R,R) fp32 slot of the padded batch: uint8/255, optional
 * density->alpha = clip(1-exp(-exp(sigma)/100),0,1) (fp32 scenes), (W,L,H,C)->(C,W,L,H), the z-up augmentations (flags: 1 = 90-degree
 * rotation [transpose axes 0,1 then flip axis 0], 2 = flip axis 0, 4 = flip axis 1, applied in that order), zero padding at the high
 * end of each axis.  The valid extents of the result are (rot ? L : W, rot ? W : L, H). */
NMH_API int nmh_grid_prepare(int src_u8, const void* src, int W, int L, int H, float* dst, int R, int flags, void* stream);
NMH_API int nmh_bias_grad(int dt, const void* dY, float* db, int64_t M, int N, const float* rowscale, int rows_per_scale, void* stream);
NMH_API int nmh_add_inplace(int dt, void* a, const void* b, int64_t n, void* stream);
/* out = a + b (out may alias a): the sum of the two gradients of an encoder feature map that feeds both the next stage and a decoder
 * skip connection (swin_mae3d.py:1465-1470 + unetr_block.py:196-197: autograd's accumulation, as a HIP kernel). */
NMH_API int nmh_add(int dt, const void* a, const void* b, void* out, int64_t n, void* stream);
NMH_API int nmh_fill_f32(float* p, float v, int64_t n, void* stream);
/* Accumulator arena.  Every entry point that reduces into a caller-provided accumulator (InstanceNorm statistics and backward sums, the
 * fused conv statistics, loss sums, the gradient norm) clears it first with a launch of its own -- 24 launches of ~5 us on the dependent
 * chain of a training step.  After nmh_set_prezeroed_arena(base, bytes) the caller guarantees that any accumulator lying wholly inside
 * [base, base + bytes) is zero on entry (it clears the used part of the arena with ONE launch per step and never hands out a slice twice
 * between two clears), and those clearing launches are skipped; accumulators outside the range are cleared as before.
 * bytes = 0 removes the arena.  One arena per process (one process per GPU).  Reference: the reductions are ATen kernels there
 * (InstanceNorm3d, unetr_block.py:57-71; forward_loss, swin_mae3d.py:1513-1563) and allocate + clear their own workspaces. */
NMH_API int nmh_set_prezeroed_arena(void* base, int64_t bytes);
/* ---- dense-prediction heads on the pretrained encoder + decoder (nerf_rpn/model/feature_extractor.py:1898-2244 VoxelSR, 2521-2848
 * VoxelSemantics).  Their convolutions / norms / GEMMs are the entries above; these are the head-specific pieces. ----
 * nmh_grid_to_cl8: (B,4,V) fp32 NCDHW grid -> [B*V][8] channels-last in dt (channels 4..7 zero): input of `encoder1`, whose first
 *   3x3x3 conv (4 -> E/2, unetr_block.py:35-44) and 1x1x1 residual conv run on weights padded to 8 input channels (pack mode 12).
 * nmh_head_upsample_fwd: y [B*R^3][Cp] (dt, head GEMM output, first Co columns) -> pred (B,Co,Ro,Ro,Ro) fp32 with nn.Upsample(scale_factor,
 *   mode="nearest") source indices min(floorf(dst * inv_scale), R-1) (feature_extractor.py:2036,2224-2229; a 1x1x1 conv commutes with
 *   nearest upsampling, so `voxel_out` runs before it); Ro == R, inv_scale == 1: plain channels-last -> NCDHW (the semantics head).
 * nmh_head_upsample_bwd: its adjoint, g [B*R^3][Cp] (dt) = sum of dpred over the output voxels that read each source voxel (columns >= Co zero).
 * nmh_add_cols_f32: dst[m][0:C] += src[m][0:C] (row strides in elements): un-padding of gradients computed on padded operands. */
NMH_API int nmh_grid_to_cl8(int dt, const float* grid, void* out, int B, int64_t V, void* stream);
NMH_API int nmh_head_upsample_fwd(int dt, const void* y, float* pred, int B, int Co, int Cp, int R, int Ro, float inv_scale, void* stream);
NMH_API int nmh_head_upsample_bwd(int dt, const float* dpred, void* g, int B, int Co, int Cp, int R, int Ro, float inv_scale, void* stream);
NMH_API int nmh_add_cols_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream);
/* VoxelSR loss (feature_extractor.py:2133-2160): loss = sum_v m (pred_rgb - target_rgb)^2 / sum_v m, m = target_alpha > 0.01; pred and
 * target (B,4,V) fp32; sums fp64[2] = {numerator, sum m}.  _bwd: dpred = gscale * d(loss)/d(pred) (alpha plane 0). */
NMH_API int nmh_voxel_sr_loss_fwd(const float* pred, const float* target, int B, int64_t V, double* sums, float* loss, void* stream);
NMH_API int nmh_voxel_sr_loss_bwd(const float* pred, const float* target, int B, int64_t V, const double* sums, float gscale, float* dpred, void* stream);
/* VoxelSemantics loss (nerf_rpn/model/metrics.py:540-553 with nn.CrossEntropyLoss(weight), feature_extractor.py:2700-2722): logits (B,K,V)
 * fp32, labels (B,V) fp32 class ids (0 = unlabelled), mask m = label > 0, cross entropy over ALL voxels of (logits*m, label*m) with
 * optional class weights [K]; sums fp64[2] = {sum w nll, sum w}; iou_sums fp64 [B][K-1][3] = {sum m p_k, #(label = k), sum_{label=k} p_k}
 * (mIoULoss_new, metrics.py:194-245); out fp32[2] = {loss, mean soft IoU}.  _bwd: dlogits = gscale * d(loss)/d(logits).  K <= 32. */
NMH_API int nmh_masked_ce_fwd(const float* logits, const float* labels, const float* class_weights, int B, int K, int64_t V, double* sums, double* iou_sums, float* out, void* stream);
NMH_API int nmh_masked_ce_bwd(const float* logits, const float* labels, const float* class_weights, int B, int K, int64_t V, const double* sums, float gscale, float* dlogits, void* stream);
/* Per-step host parameters of the training step in ONE launch, passed as kernel arguments (no host-to-device copies):
 *  - block_bits (HOST, nb^3 bits, bit (a*nb+b)*nb+c = 1: the 4x4x4-token block (a,b,c) is removed; window_masking_3d's raster order,
 *    swin_mae3d.py:1366-1373) is expanded to the token mask tokmask[g^3] (device, uint8; tokens outside the nb^3 blocks stay 0);
 *  - hyper (HOST, 8 floats {lr, beta1, beta2, eps, wd, 1-beta1^t, 1-beta2^t, zero_g}) -> hyper_dev (nmh_adamw_step);
 *  - extents (HOST, n_ext <= 48 ints: [B][3] valid voxels per axis) -> extents_dev (nmh_mae_loss_fwd / nmh_mae_tail_fwd).
 * Any of the three groups may be NULL.  nb^3 <= 4096. */
NMH_API int nmh_step_params(const uint32_t* block_bits, int nb, int g, unsigned char* tokmask, const float* hyper, float* hyper_dev, const int* extents, int n_ext, int* extents_dev, void* stream);
/* bf16 gradient buckets of the data-parallel exchange (DDP's gradient all-reduce, run_swin_mae3d.py:355-357, with half the bytes on the
 * xGMI links): a contiguous range of the flat fp32 gradient buffer -> bf16 bucket (round to nearest even) before the collective, and
 * bucket * scale -> fp32 gradients after it (scale = 1/world for backends that sum).  n % 8 == 0, 16-byte aligned pointers. */
NMH_API int nmh_grad_to_bf16(const float* g, void* bucket, int64_t n, void* stream);
NMH_API int nmh_grad_from_bf16(const void* bucket, float* g, int64_t n, float scale, void* stream);

/* fp32 master weights -> compute-dtype GEMM operand layouts, one launch for the whole model.  descs: device array of
 * {const float* src; void* dst; int mode; int d0,d1,d2; int64 n} (40 bytes, see kernels.hpp PackDesc); one block per 1024 dst elements
 * (blkstart = first element), except the tiled modes, where blkstart = block id: mode 1 (2-D transpose) one block per 32x32 tile of the
 * d0 x d1 source, row-major over ceil(d0/32) x ceil(d1/32); mode 2 (conv fwd) d0 x ceil(d1/32) blocks (co, 32 ci); mode 3 (conv dgrad)
 * d1 x ceil(d0/32) blocks (ci, 32 co). */
NMH_API int nmh_pack_weights(int dt, const void* descs_dev, const int* blk2desc_dev, const int64_t* blkstart_dev, int nblocks, void* stream);
/* clip_grad_norm_ + AdamW (run_swin_mae3d.py:588-592,665-668) over the flat fp32 parameter buffer; hyper (device fp32[8]) =
 * {lr, beta1, beta2, eps, weight_decay, 1-beta1^t, 1-beta2^t, zero_g}; coef (device) = min(1, max_norm/(norm+1e-6)); zero_g != 0:
 * the step also clears g (the next step's zero_grad for free). */
NMH_API int nmh_grad_sqnorm(const float* g, int64_t n, double* acc, void* stream);
NMH_API int nmh_clip_coef(const double* acc, float max_norm, float* coef, float* norm_out, void* stream);
NMH_API int nmh_adamw_step(float* p, float* g, float* m, float* v, int64_t n, const float* hyper, const float* coef, void* stream);

#ifdef __cplusplus
}
#endif
#endif
