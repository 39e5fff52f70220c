/*
 * itcv_hip.h -- C ABI of libitcv_hip.so, the MI355X (gfx950) hot path of the Soft-Intro
 * beta-TC-VAE training step.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream (as void*); no
 * torch / C++ types cross this boundary.  All tensors are fp32, dense, NCHW.  Entry points
 * return 0 on success and a non-zero code on error; itcv_last_error() returns the message
 * of the last failure on the calling thread.  Nothing here allocates: scratch memory is
 * passed in by the caller ("ws"), with the size given by the matching *_workspace() query.
 * All launches are asynchronous on `stream`; nothing synchronises the device.
 *
 * The reference (meffmadd/intro-tc-vae) is pure Python/PyTorch, so there is no FFI layer to
 * mirror; each entry point below names the reference call site (file:line under
 * /root/reference) whose ATen work it replaces.  The Python mirror of the reference's
 * module surface (models.py / ops.py / solvers) binds these with ctypes
 * (intro-tc-vae_amd/hipvae/abi.py); see INTEGRATION.md.
 */
#ifndef ITCV_HIP_H
#define ITCV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITCV_ABI_VERSION 3

/* ---- library ------------------------------------------------------------------------- */
int itcv_abi_version(void);
const char* itcv_last_error(void);
/* Launch-shape options (process-wide; the library reads NO environment variable).  Every option and value is part of
 * the test matrix: results are bit-identical for any band_persist_blocks, and equal to rounding for band_m16.
 *   "band_m16"             1 (default): v_mfma_f32_16x16x32_bf16 in the band / 128-pixel planes kernels; 0: 32x32x16
 *   "band_persist_blocks"  256 (default): blocks of the persistent band kernel (used when a launch has more tiles);
 *                          0: one tile per block; range 0..1024
 *   "wgrad_m16"            1 (default): v_mfma_f32_16x16x32 in itcv_conv2d_wgrad_bf16p; 0: 32x32x16 (equal to rounding)
 *   "planes_mfma_waves"    8 (default) or 4: MFMA waves of the 128 x 128 tile of the 128-pixel planes kernel (bit-identical)
 * itcv_set_option returns non-zero for an unknown name or a value out of range; itcv_get_option returns -1 for an unknown name. */
int itcv_set_option(const char* name, int value);
int itcv_get_option(const char* name);

/* Optional per-launch timing of the GEMM-class kernels (bench.py's roofline leg): between _begin and
 * _end every conv entry point records a HIP event pair on its launch stream around its MAIN kernel
 * (not the split-K reduce).  _end waits for the events and returns the record count; record i is
 * (code = kind | KS<<4 | BM<<8 | up2<<16 | NS<<20 with kind 0 fwd fp32, 1 fwd split-bf16, 2 wgrad fp32,
 * 3 wgrad split-bf16, 4 small-Cout direct, 5 small-Cin direct, 6 fwd on planes, 7 wgrad on planes, 8 / 9 band-form fwd on planes (one tile / persistent) -- for 7..9 the KS field
 * holds log2(W) -- 10 small-Cout on planes, 11 small-Cin on the matrix cores, 12 5x5 weight gradient on planes (BM field = narrow side's channels);
 * algorithmic FLOP; elapsed ms). */
int itcv_profile_begin(void);
int itcv_profile_end(void);
int itcv_profile_get(int i, int* code, double* flop, float* ms);
int itcv_profile_clear(void);

/* ---- convolution / linear: implicit GEMM on v_mfma_f32_32x32x2_f32 -------------------
 * Stride 1, square odd kernel KS in {1,3,5}, zero padding KS/2 ("same"), groups 1.
 * Replaces nn.Conv2d / nn.Linear forward+backward: models.py:28-47 (3x3 blocks), :213
 * (5x5 stem), :290 (5x5 predict, with bias), :233 / :270 (Linear == KS 1, H=W=1).
 *
 * Weights are consumed in a packed, zero-padded, K-major layout wp[KS*KS*Cp][Mp] with the
 * reduction index ordered TAP-MAJOR, k = tap*Cp + c (Cp = reduction channels rounded up to 16,
 * Mp = M rounded up to the 32/64/128-row tile the kernel picks for M):
 *   for_dgrad = 0:  M = Co, c = ci:  wp[tap*Cp + ci][co] = w[co][ci][tap]
 *   for_dgrad = 1:  M = Ci, c = co:  wp[tap*Cp + co][ci] = w[co][ci][KK-1-tap]
 * so that the data-gradient is the same kernel run on dy with the roles of Ci/Co swapped. */
size_t itcv_conv2d_packed_weight_elems(int Co, int Ci, int KS, int for_dgrad);
int itcv_conv2d_pack_weight(const float* w, float* wp, int Co, int Ci, int KS, int for_dgrad,
                            void* stream);
/* y[B][Co][H][W] = conv(x[B][Ci][H][W], w) (+ bias[Co] if non-NULL).  `up2` != 0 reads x as the
 * nearest-neighbour x2 upsampling of a [B][Ci][H/2][W/2] tensor (models.py:284-286 fused into the
 * consumer). */
size_t itcv_conv2d_fwd_workspace(int B, int Ci, int H, int W, int Co, int KS);
int itcv_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, int B, int Ci,
                    int H, int W, int Co, int KS, int up2, void* ws, size_t ws_bytes, void* stream);
/* Split-bf16 throughput variant of itcv_conv2d_fwd (forward and data-gradient): every fp32 operand
 * is split into ns bf16 planes and the product is accumulated in fp32 on v_mfma_f32_32x32x16_bf16:
 * ns = 2 ("bf16x3", 3 MFMAs per product, ~2^-16 relative per product), ns = 3 ("bf16x6", 6 MFMAs,
 * fp32-class ~2^-23).  Supported for KS in {1,3}, Ci a multiple of 32, Co > 32 (see _supported);
 * weights come pre-split from itcv_conv2d_pack_weight_bf16s. */
int itcv_conv2d_bf16s_supported(int Ci, int Co, int KS);
size_t itcv_conv2d_packed_weight_bytes_bf16s(int Co, int Ci, int KS, int for_dgrad, int ns);
int itcv_conv2d_pack_weight_bf16s(const float* w, void* wp, int Co, int Ci, int KS, int for_dgrad, int ns,
                                  void* stream);
/* The same packing for many layers in ONE launch (the conv weights of a network after its optimiser step,
 * solvers/intro.py:116,160, solvers/vae.py:109-110): the caller fills a host array of n descriptors of itcv_pack_desc_bytes() each with
 * itcv_conv2d_pack_desc_bf16s (returns the number of blocks the layer adds, or a negative error; block0 = the running
 * sum of those), copies it to the device once and launches itcv_conv2d_pack_weights_bf16s with the total. */
size_t itcv_pack_desc_bytes(void);
int itcv_conv2d_pack_desc_bf16s(void* host_desc, const float* w, void* wp, int Co, int Ci, int KS, int for_dgrad,
                                int ns, int block0);
int itcv_conv2d_pack_weights_bf16s(const void* dev_table, int n, int total_blocks, int ns, void* stream);
size_t itcv_conv2d_fwd_bf16s_workspace(int B, int Ci, int H, int W, int Co, int KS);
int itcv_conv2d_fwd_bf16s(const float* x, const void* wp, const float* bias, float* y, int B, int Ci, int H,
                          int W, int Co, int KS, int up2, int ns, void* ws, size_t ws_bytes, void* stream);
/* Pre-split operand ("planes"): planes[p][b][c/8][h][w] = one 16-byte chunk of 8 bf16 = plane p of
 * channels c..c+7 of one pixel (C % 8 == 0; p < ns).  A producing pass (itcv_split_planes, or the
 * BatchNorm apply / backward kernels through their `planes` argument) writes it next to the fp32
 * tensor; itcv_conv2d_fwd_bf16p then moves both operands global -> LDS by LDS-DMA (no gather, no
 * conversion in the conv kernel).  Same contract and shapes (its own workspace query) and -- bit for bit -- results as
 * itcv_conv2d_fwd_bf16s (replaces the same ATen conv forward / data-gradient, models.py:28-47). */
/* Plane formats (the `ns` argument of every planes entry point):
 *   2  two bf16 planes  ("bf16x3": 3 products, ~2^-16 per product)
 *   3  three bf16 planes ("bf16x6": 6 products, fp32 class)
 *   4  ITCV_PLANES_F16X2: two FP16 planes hi = fp16(S x), lo = fp16(S x - hi) of the tensor times a power-of-two scale S
 *      ("f16x3": the same 3 products on v_mfma_f32_*_f16 carry 22 significand bits, ~2^-21 per product -- fp32 class at the
 *      bf16x3 rate).  The record {S, 1/S} (16 bytes) sits behind the second plane; producers write it, consumers multiply
 *      their result by the exact inverse.  Activations use S = 1 (O(1) values; |x| >= 65504 becomes inf and surfaces as a
 *      NaN loss); packed weights S = 2^8; gradient tensors a scale derived from a rigorous bound of their magnitude
 *      (BatchNorm backward: from per-channel maxima it computes anyway; itcv_absmax for loose fp32 tensors). */
#define ITCV_PLANES_F16X2 4
size_t itcv_planes_bytes(int B, int C, int HW, int ns);
int itcv_split_planes(const float* x, void* planes, int B, int C, int HW, int ns, void* stream);
/* parts[256] = block maxima of |x| (x 16-byte aligned); feeds the `amax` arguments below, which derive the power-of-two
 * scale of an fp16 split from them on the device (no host round trip).  amax == NULL means S = 1. */
int itcv_absmax(const float* x, size_t n, float* parts, void* stream);
int itcv_split_planes_scaled(const float* x, void* planes, int B, int C, int HW, int ns, const float* amax, void* stream);
size_t itcv_conv2d_fwd_bf16p_workspace(int B, int Ci, int H, int W, int Co, int KS, int ns);
int itcv_conv2d_fwd_bf16p(const void* xplanes, const void* wp, const float* bias, float* y, int B, int Ci, int H,
                          int W, int Co, int KS, int up2, int ns, void* ws, size_t ws_bytes, void* stream);
/* The same with the BatchNorm statistics of the consumer layer (models.py:37: conv -> BatchNorm) fused into the conv
 * epilogue: tile_stats[(k*Co + c)*T + t], k = 0: sum, k = 1: sum of squares of output channel c over the t-th tile of
 * 256 consecutive (image, pixel) positions, T = itcv_conv2d_fwd_bf16p_stat_tiles(...) (0 = not available for the shape:
 * pass NULL).  itcv_bn_train_fwd folds them (fp64) instead of reading the tensor once more. */
int itcv_conv2d_fwd_bf16p_stat_tiles(int B, int Ci, int H, int W, int Co, int KS, int ns);
int itcv_conv2d_fwd_bf16p_st(const void* xplanes, const void* wp, const float* bias, float* y, int B, int Ci, int H,
                             int W, int Co, int KS, int up2, int ns, float* tile_stats, void* ws, size_t ws_bytes,
                             void* stream);
/* Weight gradient from the same planes (x: [2][B][Ci/8][Hs][Ws], dy: [2][B][Co/8][H][W]); the pixel
 * reduction runs through the gfx950 transposing LDS read, so no pixel-major copy is needed.  bf16x3
 * only; KS = 3, W a power of two in 4..64, H a power of two, B*H*W % 64 == 0 (see _supported).  Replaces
 * the ATen conv weight-gradient (backward of models.py:28-47).  Deterministic split-K (fp32 slabs). */
int itcv_conv2d_wgrad_bf16p_supported(int B, int Ci, int H, int W, int Co, int KS);
size_t itcv_conv2d_wgrad_bf16p_workspace(int B, int Ci, int H, int W, int Co, int KS);
int itcv_conv2d_wgrad_bf16p(const void* xplanes, const void* dyplanes, float* dw, int B, int Ci, int H, int W,
                            int Co, int KS, int up2, int ns /* 2 or 4 */, int accumulate, void* ws, size_t ws_bytes,
                            void* stream);
/* accumulate == 2 DEFERS the slab reduce of itcv_conv2d_wgrad_bf16p: the call leaves its itcv_conv2d_wgrad_bf16p_slabs(...)
 * fp32 slabs in `ws` (which must then stay alive), and ONE itcv_wgrad_reduce_many launch folds the slabs of many layers
 * into their dw (a whole backward pass: solvers/intro.py:109-116).  The caller fills a host array of n descriptors of
 * itcv_wgrad_reduce_desc_bytes() each (returns the blocks the layer adds, < 0 on error; block0 = their running sum; up to
 * four slab sources per dw, folded in the given order: a weight used by several passes of the backward), copies it to the
 * device and launches with the total.  Results are bitwise those of the per-call reduces. */
int itcv_conv2d_wgrad_bf16p_slabs(int B, int Ci, int H, int W, int Co, int KS);
size_t itcv_wgrad_reduce_desc_bytes(void);
int itcv_wgrad_reduce_desc(void* host_desc, const float* const* slabs, const int* splits, int nsrc, float* dw, int Co,
                           int Ci, int accumulate, int block0);
int itcv_wgrad_reduce_many(const void* dev_table, int n, int total_blocks, void* stream);
int itcv_wgrad_reduce_max_descs(void);   /* descriptors one table (one launch) may hold */
/* nn.Linear(K -> N) at batch B (models.py:233,270) as skinny exact-fp32 MFMA GEMMs, in every conv-math mode:
 * y[B][N] = x[B][K] w[N][K]^T + bias;  dx[B][K] = dy[B][N] w;  dw[N][K] (+)= dy^T x.  Deterministic
 * split-K; itcv_linear_workspace serves all three. */
size_t itcv_linear_workspace(int B, int K, int N);
int itcv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, void* ws,
                    size_t ws_bytes, void* stream);
int itcv_linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, void* ws, size_t ws_bytes,
                      void* stream);
int itcv_linear_wgrad(const float* dy, const float* x, float* dw, int B, int K, int N, int accumulate, void* ws,
                      size_t ws_bytes, void* stream);
/* 5x5 weight gradients with a <= 3-channel side (stem 3 -> 64: stem = 1, small = x, big_planes = planes of dy;
 * predict 64 -> 3: stem = 0, small = dy, big_planes = planes of x), bf16x3 on the matrix cores: rows (small channel,
 * filter column), pixel reduction through the transposing LDS read.  W in {32, 64}.  Deterministic slab reduce. */
int itcv_conv2d_wgrad5_bf16p_supported(int Cs, int Cb, int H, int W);
size_t itcv_conv2d_wgrad5_bf16p_workspace(int B, int H);
int itcv_conv2d_wgrad5_bf16p(const float* small, const void* big_planes, float* dw, int B, int Cs, int H, int W,
                             int stem, int ns /* 2 or 4 */, const float* small_amax /* itcv_absmax of `small`, or NULL */,
                             int accumulate, void* ws, size_t ws_bytes, void* stream);
/* Direct (vector-ALU, exact fp32) convolution for layers with at most 4 output channels -- the 5x5
 * predict conv 64->3 (models.py:290) and the data-gradient of the 5x5 stem (models.py:213), where a
 * 32-row MFMA tile would be >90 % padding.  for_dgrad = 0: w is [Co][C][KS][KS]; for_dgrad = 1: w is the
 * forward layer's [C][Co][KS][KS] and its transposed, flipped filter is applied to x = dy. */
int itcv_conv2d_small_cout_supported(int Co, int KS);
/* The same layer on the bf16 matrix cores (bf16x3) from pre-split planes of a 64-channel input: MFMA rows are
 * (output channel, filter column), every operand fragment is one plane chunk loaded straight from global memory. */
int itcv_conv2d_small_cout_bf16p_supported(int C, int Co, int KS);
int itcv_conv2d_small_cout_fwd_bf16p(const void* xplanes, const float* w, const float* bias, float* y, int B, int C,
                                     int H, int W, int Co, int KS, int for_dgrad, int ns /* 2 or 4 */, void* stream);
int itcv_conv2d_small_cout_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C,
                               int H, int W, int Co, int KS, int for_dgrad, void* stream);
/* ... and for layers with at most 4 REDUCTION channels (the 5x5 stem 3->64 forward, models.py:213, and
 * the data-gradient of the predict conv): the pixel's input window lives in registers. */
int itcv_conv2d_small_cin_supported(int C, int KS);
int itcv_conv2d_small_cin_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int H,
                              int W, int Co, int KS, int for_dgrad, void* stream);

/* The same conv for C <= 3, Co == 64, KS == 5, W % 32 == 0 (the stem layer models.py:199-204 and the data-gradient of
 * the prediction layer models.py:271) as split-bf16 (bf16x3) products on the matrix cores. */
int itcv_conv2d_small_cin_bf16x3_supported(int C, int Co, int KS, int W);
int itcv_conv2d_small_cin_fwd_bf16x3(const float* x, const float* w, const float* bias, float* y, int B, int C, int H,
                                     int W, int Co, int KS, int for_dgrad, int ns /* 2 or 4 */,
                                     const float* x_amax /* itcv_absmax of x, or NULL */, void* stream);
/* Split-bf16 weight gradient (same arithmetic, same workspace size as itcv_conv2d_wgrad_workspace):
 * needs KS in {1,3}, Ci % 32 == 0, W % 8 == 0, Co > 32 and a materialised (not virtually upsampled) x. */
int itcv_conv2d_wgrad_bf16s_supported(int Ci, int H, int W, int Co, int KS);
int itcv_conv2d_wgrad_bf16s(const float* x, const float* dy, float* dw, int B, int Ci, int H, int W, int Co,
                            int KS, int ns, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* dw[Co][Ci][KS][KS] (+)= sum_{b,h,w} dy[b][co][h][w] * x[b][ci][h+kh-p][w+kw-p]; `up2` as in _fwd
 * (x is the low-resolution [B][Ci][H/2][W/2] tensor, H/W are the dims of dy). */
size_t itcv_conv2d_wgrad_workspace(int B, int Ci, int H, int W, int Co, int KS);
int itcv_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int Ci, int H, int W,
                      int Co, int KS, int up2, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* Kernel instantiation a call resolves to, for profiling buckets: bits 0-7 block rows BM,
 * 8-15 KS, 16 up2, 20-31 split-K factor. */
int itcv_conv2d_fwd_variant(int B, int Ci, int H, int W, int Co, int KS, int up2);
int itcv_conv2d_wgrad_variant(int B, int Ci, int H, int W, int Co, int KS, int up2);
/* db[C] (+)= sum_{b,hw} dy[b][c][hw]  (bias gradients: models.py:290 predict, :233/:270 Linear) */
size_t itcv_bias_grad_workspace(int B, int C, int HW);
int itcv_bias_grad(const float* dy, float* db, int B, int C, int HW, int accumulate, void* ws,
                   size_t ws_bytes, void* stream);

/* ---- BatchNorm2d (+ LeakyReLU, + AvgPool2d(2)) ---------------------------------------
 * Replaces nn.BatchNorm2d(eps) -> nn.LeakyReLU(0.2) [-> nn.AvgPool2d(2)]: models.py:37-38,48-49,
 * 214-216,225.  Train-mode statistics are computed in fp64 from per-channel (sum, sum of
 * squares); the moments are exposed so that a data-parallel caller can all-reduce them
 * (Sync-BN) between _moments and _finalize. */
size_t itcv_bn_workspace(int B, int C, int HW);
/* sums[0..C) = sum x, sums[C..2C) = sum x^2 over (b,hw); deterministic two-stage reduction */
int itcv_bn_moments(const float* x, double* sums, int B, int C, int HW, void* ws, size_t ws_bytes,
                    void* stream);
/* single-rank fast path: moments + finalize of this rank's own batch in two launches */
int itcv_bn_train_stats(const float* x, int B, int C, int HW, float eps, float momentum, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd, void* ws,
                        size_t ws_bytes, void* stream);
/* mean/rstd from the moments of `count` samples; updates running_mean/var (momentum, unbiased
 * variance) and num_batches_tracked when those pointers are non-NULL. */
int itcv_bn_finalize(const double* sums, double count, float eps, float momentum, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd, int C,
                     void* stream);
/* eval mode: mean = running_mean, rstd = 1/sqrt(running_var + eps) */
int itcv_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean,
                       float* rstd, int C, void* stream);
/* y = pool(lrelu(gamma*(x-mean)*rstd + beta, slope)); slope = 1 disables the activation; pool in
 * {0: none (y [B][C][H][W]), 1: 2x2 average (y [B][C][H/2][W/2])}.  If `skip` is non-NULL it is
 * added before the activation (ResidualBlock, models.py:113-114). */
int itcv_bn_act_fwd(const float* x, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, const float* skip, float* y, int B, int C, int H, int W,
                    float slope, int pool, void* planes, int ns, size_t plane_stride, void* stream);
/* `plane_stride` (16-byte chunks; also in itcv_bn_act_bwd_apply / itcv_bn_train_fwd / itcv_bn_train_bwd): distance
 * between consecutive planes of `planes` / `dx_planes`.  0 = B*(C/8)*Ho*Wo, i.e. the call covers the whole tensor.
 * Non-zero: x / y / planes point at ONE BatchNorm GROUP of a larger batched tensor -- the solvers push several
 * independent network passes (each a BatchNorm batch of its own, models.py:37) through the conv GEMMs as one batch,
 * and normalise every group with its own call: statistics, running-buffer updates and their order stay those of
 * separate passes.
 * `planes` (may be NULL): the same launch also writes the output as pre-split bf16 planes
 * [ns][B][C/8][Ho][Wo] for the consumer conv (itcv_conv2d_fwd_bf16p / _wgrad_bf16p); needs
 * itcv_bn_act_planes_supported(C, H, W, pool).  The fp32 output is bitwise unchanged; with planes given,
 * `y` may be NULL (fp32 output not written: the consumer GEMMs read only the planes).  The same holds
 * for `dx_planes` / `dx` of itcv_bn_act_bwd_apply (planes of dx, pool = 0 in the support query). */
int itcv_bn_act_planes_supported(int C, int H, int W, int pool);
/* backward, stage 1: dsums[0..C) = sum g, dsums[C..2C) = sum g*xhat where
 * g = unpool(dy) * lrelu'(bn_out (+skip)); `up2`!=0 means dy is the gradient of the x2-upsampled
 * output (dy [B][C][2H][2W], summed 2x2 on the fly: adjoint of models.py:284-286).  When non-NULL,
 * dgamma (+)= dsums[C+c] and dbeta (+)= dsums[c] are written by the same launch (the rank's own
 * sums are the parameter gradients, also under Sync-BN). */
int itcv_bn_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, const float* skip, double* dsums,
                           float* dgamma, float* dbeta, int accumulate, int B, int C, int H, int W,
                           float slope, int pool, int up2, void* ws, size_t ws_bytes, void* stream);
/* backward, stage 2: dx = gamma*rstd*(g - dsums[c]/count - xhat*dsums[C+c]/count);
 * dgamma (+)= local_dsums[C+c], dbeta (+)= local_dsums[c] when non-NULL (local_dsums = the rank's
 * own sums; dsums may have been all-reduced for Sync-BN); dskip = g when non-NULL. */
int itcv_bn_act_bwd_apply(const float* x, const float* dy, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, const float* skip, const double* dsums,
                          const double* local_dsums, double count, float* dx, float* dskip,
                          float* dgamma, float* dbeta, int accumulate, int B, int C, int H, int W,
                          float slope, int pool, int up2, void* dx_planes, int ns, size_t plane_stride, void* stream);

/* Single-rank training forms: itcv_bn_train_fwd == itcv_bn_train_stats + itcv_bn_act_fwd and
 * itcv_bn_train_bwd == itcv_bn_act_bwd_reduce + itcv_bn_act_bwd_apply (count = B*H*W), same arguments and results;
 * where the planes kernels apply and the reduction is sliced, the apply launch folds the slices itself (two
 * launches per layer instead of three).  Workspace: itcv_bn_workspace. */
int itcv_bn_train_fwd(const float* x, const float* gamma, const float* beta, const float* skip, float* y, void* planes,
                      int ns, int B, int C, int H, int W, float slope, int pool, float eps, float momentum,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd,
                      void* ws, size_t ws_bytes, size_t plane_stride, const float* tile_stats, int tiles, int tile_pitch,
                      int groups, void* stream);
/* groups > 1 (itcv_bn_train_fwd / _bwd): x / y / planes (dy / dx / dx_planes) hold `groups` BatchNorm groups of B images
 * each, stacked along the batch dimension; mean / rstd are [groups][C], dsums [groups][2C]; plane_stride is that of the
 * whole tensor.  Every group is normalised with its own statistics and advances the running buffers on its own, in
 * order; small layers do it in one statistics launch + one apply launch for all groups. */
/* tile_stats (may be NULL): per-tile sums of x written by the producing conv (itcv_conv2d_fwd_bf16p_st):
 * sum at tile_stats[c*tile_pitch + t], sum of squares at tile_stats[(C + c)*tile_pitch + t], t < tiles -- the tiles
 * that make up THIS call's B images; the statistics are then folded from them and x is read once (apply) only. */
int itcv_bn_train_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const float* skip, double* dsums, float* dx, float* dskip, void* dx_planes,
                      int ns, float* dgamma, float* dbeta, int accumulate, int B, int C, int H, int W, float slope,
                      int pool, int up2, void* ws, size_t ws_bytes, size_t plane_stride, int groups, void* stream);

/* ---- pointwise / resampling ----------------------------------------------------------- */
int itcv_lrelu_fwd(const float* x, float* y, size_t n, float slope, void* stream);     /* models.py:271 */
int itcv_lrelu_bwd(const float* x, const float* dy, float* dx, size_t n, float slope, void* stream);
int itcv_sigmoid_fwd(const float* x, float* y, size_t n, void* stream);                /* models.py:291 */
int itcv_sigmoid_bwd(const float* y, const float* dy, float* dx, size_t n, void* stream);
int itcv_avgpool2_fwd(const float* x, float* y, int BC, int H, int W, void* stream);   /* models.py:216,225 */
int itcv_avgpool2_bwd(const float* dy, float* dx, int BC, int H, int W, void* stream);
int itcv_upsample2_fwd(const float* x, float* y, int BC, int H, int W, void* stream);  /* models.py:284 */
int itcv_upsample2_bwd(const float* dy, float* dx, int BC, int H, int W, void* stream);
int itcv_add(const float* a, const float* b, float* out, size_t n, void* stream);      /* models.py:114,182 */

/* ---- latent math (ops.py) ------------------------------------------------------------- */
/* ops.py:166-185  z = mu + eps*exp(0.5*logvar) */
int itcv_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, size_t n,
                     void* stream);
int itcv_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar,
                     size_t n, void* stream);
/* ops.py:161-163  kl[j] = -0.5 * sum_l (1 + lv - exp(lv) - mu^2) */
int itcv_kl_rows_fwd(const float* logvar, const float* mu, float* kl, int B, int D, void* stream);
int itcv_kl_rows_bwd(const float* g, const float* logvar, const float* mu, float* dlogvar, float* dmu,
                     int B, int D, void* stream);

/* ops.py:15-29,32-49,52-115: pairwise Gaussian log-density + minibatch stratified / weighted
 * sampling, fused; the [B,B,D] tensor is never materialised.
 *   rows j: the caller's local samples z[Bl][D] (global row index = row_offset + j)
 *   cols i: all samples' means mu_all[Bt][D] (all-gathered in data-parallel runs)
 *   logvar: [Bl][D] (rows) with ITCV_TC_VAR_FROM_ROW, else [Bt][D] (columns)
 *   flags: ITCV_TC_*   variance source, density flavour and sampler
 * Outputs: prodm[Bl] = sum_l logsumexp_i(logW[j,i] + lp[j,i,l]),
 *          logqz[Bl] = logsumexp_i(logW[j,i] + sum_l lp[j,i,l]),
 *          lse[Bl][D]      (saved per-dimension logsumexp, needed by the backward),
 *          sjoint[Bl][Bt]  (the joint terms logW[j,i] + sum_l lp[j,i,l] -- without logW for the weighted sampler --
 *                           saved for the backward, which does not recompute the forward).
 * Workspace: itcv_tc_fwd_workspace (per-chunk logsumexp partials). */
#define ITCV_TC_VAR_FROM_ROW 0x1 /* ops.py:81 logvar.unsqueeze(1): variance of sample row j (live path) */
#define ITCV_TC_EPS_DENSITY 0x2  /* ops.py:15-21 density (var clamp 1e-4, straight-through) else ops.py:24-29 */
#define ITCV_TC_WEIGHTED 0x4     /* ops.py:92-101 MWS; default ops.py:104-115 MSS */
#define ITCV_TC_LIVE (ITCV_TC_VAR_FROM_ROW | ITCV_TC_EPS_DENSITY)
size_t itcv_tc_fwd_workspace(int Bl, int Bt, int D);
int itcv_tc_fwd(const float* z, const float* mu_all, const float* logvar, float* prodm, float* logqz,
                float* lse, float* sjoint, int Bl, int Bt, int row_offset, int D, int64_t dataset_size, int flags,
                void* ws, size_t ws_bytes, void* stream);
/* gradient of sum_j g[j] * (logqz[j] - prodm[j]) for the live path (flags == ITCV_TC_LIVE):
 * dz[Bl][D], dlogvar[Bl][D] (rows) and dmu_all[Bt][D] (columns; partial over this rank's rows). */
size_t itcv_tc_bwd_workspace(int Bl, int Bt);
int itcv_tc_bwd(const float* g, const float* z, const float* mu_all, const float* logvar,
                const float* logqz, const float* lse, const float* sjoint, float* dz, float* dmu_all,
                float* dlogvar, int Bl, int Bt, int row_offset, int D, int64_t dataset_size, int flags, void* ws,
                size_t ws_bytes, void* stream);
/* The whole KL hook of the TC solvers, solvers/tc.py:69-89: (beta - 1) * total_correlation + kl_divergence with the
 * hook's reduction, fused into the estimator's launches: out[Bl] (reduction 0 none) or out[1] (1 sum, 2 mean) of
 * coef_tc * (logqz_j - prodm_j) + coef_kl * kl_j, kl_j = ops.py:161-163 on this rank's rows (mu_all + row_offset*D, logvar).
 * rows[Bl]: scratch of the reduced forms.  prodm / logqz / lse / sjoint as itcv_tc_fwd (kept for _bwd).
 * _bwd: g is [Bl] (none) or [1]; dz, dlogvar (rows) and dmu_all (columns; the KL's d/dmu added at this rank's rows). */
int itcv_tc_kl_fwd(const float* z, const float* mu_all, const float* logvar, float* out, float* rows, float* prodm,
                   float* logqz, float* lse, float* sjoint, int Bl, int Bt, int row_offset, int D, int64_t dataset_size,
                   float coef_tc, float coef_kl, int reduction, void* ws, size_t ws_bytes, void* stream);
int itcv_tc_kl_bwd(const float* g, const float* z, const float* mu_all, const float* logvar, const float* logqz,
                   const float* lse, const float* sjoint, float* dz, float* dmu_all, float* dlogvar, int Bl, int Bt,
                   int row_offset, int D, int64_t dataset_size, float coef_tc, float coef_kl, int reduction, void* ws,
                   size_t ws_bytes, void* stream);
/* ops.kl_divergence with its reduction and the hook's `beta *` (ops.py:136-163, solvers/vae.py:63-77) in one launch:
 * reduction 0: out[B] = scale * kl_j; 1 / 2: out[1] = scale * sum_j kl_j [/ B]; _bwd for g of that shape. */
int itcv_kl_loss_fwd(const float* logvar, const float* mu, float* out, int B, int D, int reduction, float scale, void* stream);
int itcv_kl_loss_bwd(const float* g, const float* logvar, const float* mu, float* dlogvar, float* dmu, int B, int D,
                     int reduction, float scale, void* stream);
/* solvers/tc.py:104-121: per-sample log q(z|x) (ops.py:24-29 density, own mu/logvar) and log p(z) */
int itcv_diag_logdensity_rows(const float* z, const float* mu, const float* logvar, float* logq_cx,
                              float* logpz, int B, int D, void* stream);
/* Materialising forms of the reference's named helpers, for callers that use the pieces one by one
 * (`from ops import gaussian_log_density, ...`, solvers/tc.py:5-11); the training step uses the fused kernels above.
 * ops.py:15-21 (eps_density != 0: variance floor 1e-4, straight-through) / ops.py:24-29: out[n0][n1][n2] =
 * clamp(log N(x; mu, exp(logvar)), min=-50) with the three operands broadcast to dims[3] through element strides
 * sx/sm/sl[3] (0 = broadcast dimension).  _bwd: elementwise gradients at the broadcast shape, dx (dmu = -dx) and
 * dlogvar, zero where the clamp is active; the caller sums them over its broadcast dimensions. */
int itcv_gauss_logdensity_fwd(const float* x, const float* mu, const float* logvar, float* out, const int64_t* dims,
                              const int64_t* sx, const int64_t* sm, const int64_t* sl, int eps_density, void* stream);
int itcv_gauss_logdensity_bwd(const float* g, const float* x, const float* mu, const float* logvar, float* dx,
                              float* dlogvar, const int64_t* dims, const int64_t* sx, const int64_t* sm,
                              const int64_t* sl, int eps_density, void* stream);
/* ops.py:104-115 (weighted == 0) / ops.py:92-101 on a materialised lp[B][B][D]: prodm[B], logqz[B]; lse[B][D] and
 * sjoint[B][B] are kept for _bwd, which returns dlp[B][B][D] for gradients g_prodm[B], g_logqz[B]. */
int itcv_sampling_fwd(const float* lp, float* prodm, float* logqz, float* lse, float* sjoint, int B, int D,
                      int64_t dataset_size, int weighted, void* stream);
int itcv_sampling_bwd(const float* g_prodm, const float* g_logqz, const float* lp, const float* lse,
                      const float* sjoint, const float* logqz, float* dlp, int B, int D, int64_t dataset_size,
                      int weighted, void* stream);
/* ops.py:118-122 for x[m][n] with m == n or m == 1: diag[min(m,n)], off[m][n][n] = x - diag_embed(x) */
int itcv_on_off_diag(const float* x, float* diag, float* off, int m, int n, void* stream);

/* ---- reconstruction loss (ops.py:188-236) --------------------------------------------- */
#define ITCV_LOSS_MSE 0
#define ITCV_LOSS_L1 1
#define ITCV_LOSS_BCE 2
/* rows[b] = sum_p err(recon[b][p], x[b][p]) */
int itcv_recon_rows_fwd(const float* x, const float* recon, float* rows, int B, size_t P, int loss_type,
                        void* ws, size_t ws_bytes, void* stream);
size_t itcv_recon_workspace(int B, size_t P);
/* drecon[b][p] = g[b] * d err / d recon */
int itcv_recon_rows_bwd(const float* x, const float* recon, const float* g, float* drecon, int B, size_t P,
                        int loss_type, void* stream);

/* The whole of ops.reconstruction_loss plus the hook's `beta *` (ops.py:219-236, solvers/vae.py:79-87) in two launches:
 * reduction 0 none: out[B] = scale * rows; 1 sum / 2 mean: out[1] = scale * sum_b rows [/ B].  _bwd: drecon for the
 * gradient g of out ([B] for none, [1] else).  Workspace: itcv_recon_workspace. */
int itcv_recon_loss_fwd(const float* x, const float* recon, float* out, int B, size_t P, int loss_type, int reduction,
                        float scale, void* ws, size_t ws_bytes, void* stream);
int itcv_recon_loss_bwd(const float* x, const float* recon, const float* g, float* drecon, int B, size_t P,
                        int loss_type, int reduction, float scale, void* stream);
/* solvers/intro.py:102-103: out[0] = mean_j exp(c * (a[j] + b[j])) (c = -2 * scale); w[B] is kept for _bwd, which
 * writes da[j] = db[j] = g[0] * d out / d a[j] (db may be NULL). */
int itcv_exp_elbo_fwd(const float* a, const float* b, float* out, float* w, int B, float c, void* stream);
int itcv_exp_elbo_bwd(const float* g, const float* w, float* da, float* db, int B, void* stream);
/* out[0] = sum_k weights[k] * terms[k][0], n <= 8 device scalars (solvers/intro.py:105-108,149-151, solvers/vae.py:106:
 * the scalar arithmetic of a loss in one launch); _bwd: grads[k] = g[0] * weights[k].  `terms` / `weights` are HOST arrays. */
int itcv_lincomb_fwd(const float* const* terms, const float* weights, int n, float* out, void* stream);
int itcv_lincomb_bwd(const float* g, const float* weights, int n, float* grads, void* stream);

/* ---- optimiser (train.py:141-144, solvers/intro.py:109-116,153-160) --------------------- */
/* out[0] = sum x^2 (fp64, deterministic) */
size_t itcv_sumsq_workspace(size_t n);
int itcv_sumsq(const float* x, size_t n, double* out, void* ws, size_t ws_bytes, void* stream);
/* torch.nn.utils.clip_grad_norm_: total = sqrt(sum_k sumsq[k]); coef = min(1, clip/(total+1e-6));
 * writes norm_out[0] = total, coef_out[0] = coef (device scalars, no host sync) */
int itcv_clip_coef(const double* sumsq, int nparts, double clip, float* norm_out, float* coef_out,
                   void* stream);
int itcv_scale_by_dev(float* x, size_t n, const float* coef_dev, void* stream);
/* torch.optim.Adam defaults (no amsgrad / weight decay), one flat launch; step is 1-based */
int itcv_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                   float beta2, float eps, int step, void* stream);
/* same update with the 0-based count of completed steps read from (and incremented in) device
 * memory, so that a captured launch (hipGraph replay) advances the bias correction */
int itcv_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                       float beta2, float eps, int* step_dev, void* stream);
int itcv_fill(float* x, size_t n, float value, void* stream);
/* Input pipeline (dataset.py:219-224 transforms.RandomHorizontalFlip, moved behind the host -> device copy):
 * y[b] = x[b] mirrored along W where flip[b] != 0, else x[b]; x, y are [B][rows_per_image][W] (rows = C*H), x != y. */
int itcv_hflip(const float* x, float* y, const unsigned char* flip, int B, int rows_per_image, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ITCV_HIP_H */
