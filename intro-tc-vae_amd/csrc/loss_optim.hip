// Reconstruction losses (ops.py:188-236) and the optimiser-side kernels (global gradient norm,
// clip coefficient, fused flat Adam; train.py:141-144, solvers/intro.py:109-116,153-160).
// All HBM-bound streaming kernels with deterministic fp64 reductions.
#include <math.h>

#include "common.h"

namespace itcv {

template <int LT>
__device__ __forceinline__ float rec_err(float r, float t) {
  if (LT == ITCV_LOSS_MSE) {
    const float d = r - t;
    return d * d;
  }
  if (LT == ITCV_LOSS_L1) return fabsf(r - t);
  // F.binary_cross_entropy clamps both log terms at -100
  return -(t * fmaxf(logf(r), -100.f) + (1.f - t) * fmaxf(logf(1.f - r), -100.f));
}
template <int LT>
__device__ __forceinline__ float rec_derr(float r, float t) {
  if (LT == ITCV_LOSS_MSE) return 2.f * (r - t);
  if (LT == ITCV_LOSS_L1) return r > t ? 1.f : (r < t ? -1.f : 0.f);
  return (r - t) / fmaxf((1.f - r) * r, 1e-12f);  // ATen binary_cross_entropy_backward
}

// grid (B, splits): partial[b][s] = sum over the slice of row b
template <int LT>
__global__ __launch_bounds__(256) void recon_partial_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ recon,
                                                           double* __restrict__ part, size_t P, int splits) {
  __shared__ double scratch[4];
  const int b = blockIdx.x, s = blockIdx.y;
  const size_t chunk = ((P + splits - 1) / splits + 3) & ~(size_t)3;
  const size_t beg = (size_t)s * chunk, end = beg + chunk < P ? beg + chunk : P;
  const float* xr = x + (size_t)b * P;
  const float* rr = recon + (size_t)b * P;
  double acc = 0.0;
  if ((P & 3) == 0) {
    for (size_t i = beg + (size_t)threadIdx.x * 4; i < end; i += 1024) {
      const float4 t = *reinterpret_cast<const float4*>(xr + i);
      const float4 r = *reinterpret_cast<const float4*>(rr + i);
      acc += (double)(rec_err<LT>(r.x, t.x) + rec_err<LT>(r.y, t.y)) +
             (double)(rec_err<LT>(r.z, t.z) + rec_err<LT>(r.w, t.w));
    }
  } else {
    for (size_t i = beg + threadIdx.x; i < end; i += 256) acc += (double)rec_err<LT>(rr[i], xr[i]);
  }
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) part[(size_t)b * splits + s] = acc;
}
__global__ void recon_combine_kernel(const double* __restrict__ part, float* __restrict__ rows, int B, int splits) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  rows[b] = (float)fold_strided(0.0, part + (size_t)b * splits, (size_t)1, splits);
}
template <int LT>
__global__ void recon_bwd_kernel(const float* __restrict__ x, const float* __restrict__ recon,
                                 const float* __restrict__ g, float* __restrict__ drecon, size_t P, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    drecon[i] = g[i / P] * rec_derr<LT>(recon[i], x[i]);
}

// Fused tail of ops.reconstruction_loss (ops.py:230-236) and of the solver hook's `beta *` (solvers/vae.py:79-87): one
// block folds the slice sums into rows[b] (kept for a caller that wants them) and writes
//   reduction 0 (none): out[b] = scale * rows[b];  1 (sum): out[0] = scale * sum_b rows[b];  2 (mean): out[0] = scale * mean_b rows[b]
__global__ __launch_bounds__(256) void recon_finish_kernel(const double* __restrict__ part, float* __restrict__ out, int B,
                                                          int splits, int reduction, float scale) {
  __shared__ double scratch[4];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float row = (float)fold_strided(0.0, part + (size_t)b * splits, (size_t)1, splits);   // the reference's fp32 row sum
    if (reduction == 0) out[b] = scale * row;
    acc += (double)row;
  }
  if (reduction == 0) return;
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[0] = scale * (float)(reduction == 2 ? acc / (double)B : acc);
}
// d recon of the above: coefficient g[b] * scale (none) or g[0] * scale [/ B] (sum / mean)
template <int LT>
__global__ void recon_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ recon,
                                      const float* __restrict__ g, float* __restrict__ drecon, size_t P, size_t n,
                                      int reduction, float coef) {
  const float g0 = reduction ? g[0] * coef : 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    drecon[i] = (reduction ? g0 : g[i / P] * coef) * rec_derr<LT>(recon[i], x[i]);
}

// solvers/intro.py:102-103: out[0] = mean_j exp(c * (a[j] + b[j])), c = -2 * scale; w[j] = exp(...) * c / B kept for the
// backward (d out / d a[j] = d out / d b[j] = w[j])
__global__ __launch_bounds__(256) void exp_elbo_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, float* __restrict__ w, int B, float c) {
  __shared__ float scratch[4];
  float acc = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float e = expf(c * (a[j] + b[j]));
    w[j] = e * c / (float)B;
    acc += e;
  }
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[0] = acc / (float)B;
}
__global__ void exp_elbo_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ da,
                                    float* __restrict__ db, int B) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B) return;
  const float v = g[0] * w[j];
  da[j] = v;
  if (db) db[j] = v;
}

// out[0] = sum_k weight[k] * *term[k] (k < n <= 8): the scalar arithmetic of a solver's loss (solvers/intro.py:105-108,
// 149-151; solvers/vae.py:106) in one launch; the backward writes grad[k] = g * weight[k]
struct LinComb {
  const float* term[8];
  float weight[8];
  int n;
};
__global__ void lincomb_fwd_kernel(LinComb a, float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  float s = 0.f;
  for (int k = 0; k < a.n; ++k) s += a.weight[k] * a.term[k][0];
  out[0] = s;
}
__global__ void lincomb_bwd_kernel(const float* __restrict__ g, LinComb a, float* __restrict__ grads) {
  const int k = threadIdx.x;
  if (k < a.n) grads[k] = g[0] * a.weight[k];
}

static inline int recon_splits(int B, size_t P) {
  int s = cdiv(1024, B);
  const size_t maxs = cdivz(P, 2048);
  if ((size_t)s > maxs) s = (int)maxs;
  return s < 1 ? 1 : s;
}

// ---- optimiser ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, size_t n,
                                                           double* __restrict__ part) {
  __shared__ double scratch[4];
  double acc = 0.0;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += (double)x[i] * x[i];
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ part, int nparts,
                                                         double* __restrict__ out) {
  __shared__ double scratch[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[0] = acc;
}
__global__ void clip_coef_kernel(const double* sumsq, int nparts, double clip, float* norm_out, float* coef_out) {
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += sumsq[i];
  const double total = sqrt(s);
  double coef = clip / (total + 1e-6);  // torch.nn.utils.clip_grad_norm_
  if (coef > 1.0) coef = 1.0;
  norm_out[0] = (float)total;
  coef_out[0] = (float)coef;
}
__global__ void scale_by_dev_kernel(float* __restrict__ x, size_t n, const float* __restrict__ coef) {
  const float c = coef[0];
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x *= c, v.y *= c, v.z *= c, v.w *= c;
    reinterpret_cast<float4*>(x)[i] = v;
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) x[i] *= c;
}
// torch.optim.Adam (single-tensor form): m.lerp_(g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float step_size, float b1, float b2, float eps,
                            float sqrt_bc2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
  }
}
// graph-replay-safe variant: the step count lives in device memory (a captured launch would freeze
// a host-side count); bias corrections are evaluated per thread in fp64 like torch's Python floats
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                const int* __restrict__ step_dev) {
  const int step = step_dev[0] + 1;
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1), sqrt_bc2 = (float)sqrt(bc2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
  }
}
__global__ void bump_step_kernel(int* step_dev) { step_dev[0] += 1; }

// transforms.RandomHorizontalFlip on the device (dataset.py:219-224): image b is mirrored along W where flip[b] != 0
__global__ void hflip_kernel(const float* __restrict__ x, float* __restrict__ y, const unsigned char* __restrict__ flip,
                             size_t rows_per_image, int W, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / W;
    const int w = (int)(i - row * W);
    y[i] = flip[row / rows_per_image] ? x[row * W + (W - 1 - w)] : x[i];
  }
}

__global__ void fill_kernel(float* __restrict__ x, size_t n, float value) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = value;
}

static inline int stream_grid(size_t n, int per_thread) {
  size_t b = cdivz(cdivz(n, per_thread), 256);
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
constexpr int kSumsqBlocks = 1024;

}  // namespace itcv

using namespace itcv;

extern "C" {

size_t itcv_recon_workspace(int B, size_t P) {
  return B > 0 && P > 0 ? (size_t)B * recon_splits(B, P) * sizeof(double) : 0;
}

int itcv_recon_rows_fwd(const float* x, const float* recon, float* rows, int B, size_t P, int loss_type, void* ws,
                        size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(x && recon && rows && B > 0 && P > 0, "itcv_recon_rows_fwd");
  const int splits = recon_splits(B, P);
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)B * splits * sizeof(double), "itcv_recon_rows_fwd(workspace)");
  double* part = static_cast<double*>(ws);
  dim3 grid(B, splits);
  hipStream_t st = S(stream);
  if (loss_type == ITCV_LOSS_MSE)
    hipLaunchKernelGGL(recon_partial_kernel<ITCV_LOSS_MSE>, grid, dim3(256), 0, st, x, recon, part, P, splits);
  else if (loss_type == ITCV_LOSS_L1)
    hipLaunchKernelGGL(recon_partial_kernel<ITCV_LOSS_L1>, grid, dim3(256), 0, st, x, recon, part, P, splits);
  else if (loss_type == ITCV_LOSS_BCE)
    hipLaunchKernelGGL(recon_partial_kernel<ITCV_LOSS_BCE>, grid, dim3(256), 0, st, x, recon, part, P, splits);
  else
    return fail("%s: unknown loss type %lld", "itcv_recon_rows_fwd", loss_type);
  ITCV_CHECK_LAUNCH("itcv_recon_rows_fwd");
  hipLaunchKernelGGL(recon_combine_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, part, rows, B, splits);
  ITCV_CHECK_LAUNCH("itcv_recon_rows_fwd(combine)");
  return 0;
}

int itcv_recon_rows_bwd(const float* x, const float* recon, const float* g, float* drecon, int B, size_t P,
                        int loss_type, void* stream) {
  ITCV_REQUIRE(x && recon && g && drecon && B > 0 && P > 0, "itcv_recon_rows_bwd");
  const size_t n = (size_t)B * P;
  dim3 grid(stream_grid(n, 1));
  hipStream_t st = S(stream);
  if (loss_type == ITCV_LOSS_MSE)
    hipLaunchKernelGGL(recon_bwd_kernel<ITCV_LOSS_MSE>, grid, dim3(256), 0, st, x, recon, g, drecon, P, n);
  else if (loss_type == ITCV_LOSS_L1)
    hipLaunchKernelGGL(recon_bwd_kernel<ITCV_LOSS_L1>, grid, dim3(256), 0, st, x, recon, g, drecon, P, n);
  else if (loss_type == ITCV_LOSS_BCE)
    hipLaunchKernelGGL(recon_bwd_kernel<ITCV_LOSS_BCE>, grid, dim3(256), 0, st, x, recon, g, drecon, P, n);
  else
    return fail("%s: unknown loss type %lld", "itcv_recon_rows_bwd", loss_type);
  ITCV_CHECK_LAUNCH("itcv_recon_rows_bwd");
  return 0;
}

int itcv_recon_loss_fwd(const float* x, const float* recon, float* out, int B, size_t P, int loss_type, int reduction,
                        float scale, void* ws, size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(x && recon && out && B > 0 && P > 0 && reduction >= 0 && reduction <= 2, "itcv_recon_loss_fwd");
  const int splits = recon_splits(B, P);
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)B * splits * sizeof(double), "itcv_recon_loss_fwd(workspace)");
  double* part = static_cast<double*>(ws);
  dim3 grid(B, splits);
  hipStream_t st = S(stream);
  if (loss_type == ITCV_LOSS_MSE)
    hipLaunchKernelGGL(recon_partial_kernel<ITCV_LOSS_MSE>, grid, dim3(256), 0, st, x, recon, part, P, splits);
  else if (loss_type == ITCV_LOSS_L1)
    hipLaunchKernelGGL(recon_partial_kernel<ITCV_LOSS_L1>, grid, dim3(256), 0, st, x, recon, part, P, splits);
  else if (loss_type == ITCV_LOSS_BCE)
    hipLaunchKernelGGL(recon_partial_kernel<ITCV_LOSS_BCE>, grid, dim3(256), 0, st, x, recon, part, P, splits);
  else
    return fail("%s: unknown loss type %lld", "itcv_recon_loss_fwd", loss_type);
  ITCV_CHECK_LAUNCH("itcv_recon_loss_fwd");
  hipLaunchKernelGGL(recon_finish_kernel, dim3(1), dim3(256), 0, st, part, out, B, splits, reduction, scale);
  ITCV_CHECK_LAUNCH("itcv_recon_loss_fwd(finish)");
  return 0;
}

int itcv_recon_loss_bwd(const float* x, const float* recon, const float* g, float* drecon, int B, size_t P, int loss_type,
                        int reduction, float scale, void* stream) {
  ITCV_REQUIRE(x && recon && g && drecon && B > 0 && P > 0 && reduction >= 0 && reduction <= 2, "itcv_recon_loss_bwd");
  const size_t n = (size_t)B * P;
  dim3 grid(stream_grid(n, 1));
  hipStream_t st = S(stream);
  const float coef = reduction == 2 ? scale / (float)B : scale;
  if (loss_type == ITCV_LOSS_MSE)
    hipLaunchKernelGGL(recon_loss_bwd_kernel<ITCV_LOSS_MSE>, grid, dim3(256), 0, st, x, recon, g, drecon, P, n, reduction, coef);
  else if (loss_type == ITCV_LOSS_L1)
    hipLaunchKernelGGL(recon_loss_bwd_kernel<ITCV_LOSS_L1>, grid, dim3(256), 0, st, x, recon, g, drecon, P, n, reduction, coef);
  else if (loss_type == ITCV_LOSS_BCE)
    hipLaunchKernelGGL(recon_loss_bwd_kernel<ITCV_LOSS_BCE>, grid, dim3(256), 0, st, x, recon, g, drecon, P, n, reduction, coef);
  else
    return fail("%s: unknown loss type %lld", "itcv_recon_loss_bwd", loss_type);
  ITCV_CHECK_LAUNCH("itcv_recon_loss_bwd");
  return 0;
}

int itcv_exp_elbo_fwd(const float* a, const float* b, float* out, float* w, int B, float c, void* stream) {
  ITCV_REQUIRE(a && b && out && w && B > 0, "itcv_exp_elbo_fwd");
  hipLaunchKernelGGL(exp_elbo_fwd_kernel, dim3(1), dim3(256), 0, S(stream), a, b, out, w, B, c);
  ITCV_CHECK_LAUNCH("itcv_exp_elbo_fwd");
  return 0;
}
int itcv_exp_elbo_bwd(const float* g, const float* w, float* da, float* db, int B, void* stream) {
  ITCV_REQUIRE(g && w && da && B > 0, "itcv_exp_elbo_bwd");
  hipLaunchKernelGGL(exp_elbo_bwd_kernel, dim3(cdiv(B, 256)), dim3(256), 0, S(stream), g, w, da, db, B);
  ITCV_CHECK_LAUNCH("itcv_exp_elbo_bwd");
  return 0;
}

int itcv_lincomb_fwd(const float* const* terms, const float* weights, int n, float* out, void* stream) {
  ITCV_REQUIRE(terms && weights && out && n >= 1 && n <= 8, "itcv_lincomb_fwd");
  LinComb a;
  memset(&a, 0, sizeof(a));
  a.n = n;
  for (int k = 0; k < n; ++k) {
    ITCV_REQUIRE(terms[k], "itcv_lincomb_fwd(term)");
    a.term[k] = terms[k], a.weight[k] = weights[k];
  }
  hipLaunchKernelGGL(lincomb_fwd_kernel, dim3(1), dim3(64), 0, S(stream), a, out);
  ITCV_CHECK_LAUNCH("itcv_lincomb_fwd");
  return 0;
}
int itcv_lincomb_bwd(const float* g, const float* weights, int n, float* grads, void* stream) {
  ITCV_REQUIRE(g && weights && grads && n >= 1 && n <= 8, "itcv_lincomb_bwd");
  LinComb a;
  memset(&a, 0, sizeof(a));
  a.n = n;
  for (int k = 0; k < n; ++k) a.weight[k] = weights[k];
  hipLaunchKernelGGL(lincomb_bwd_kernel, dim3(1), dim3(64), 0, S(stream), g, a, grads);
  ITCV_CHECK_LAUNCH("itcv_lincomb_bwd");
  return 0;
}

size_t itcv_sumsq_workspace(size_t n) {
  (void)n;
  return kSumsqBlocks * sizeof(double);
}

int itcv_sumsq(const float* x, size_t n, double* out, void* ws, size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(x && out && ws && ws_bytes >= kSumsqBlocks * sizeof(double), "itcv_sumsq");
  ITCV_REQUIRE(((uintptr_t)x & 15) == 0, "itcv_sumsq(16-byte aligned input)");
  int blocks = stream_grid(n, 4);
  if (blocks > kSumsqBlocks) blocks = kSumsqBlocks;
  double* part = static_cast<double*>(ws);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, S(stream), x, n, part);
  ITCV_CHECK_LAUNCH("itcv_sumsq");
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, S(stream), part, blocks, out);
  ITCV_CHECK_LAUNCH("itcv_sumsq(final)");
  return 0;
}

int itcv_clip_coef(const double* sumsq, int nparts, double clip, float* norm_out, float* coef_out, void* stream) {
  ITCV_REQUIRE(sumsq && nparts > 0 && norm_out && coef_out, "itcv_clip_coef");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, S(stream), sumsq, nparts, clip, norm_out, coef_out);
  ITCV_CHECK_LAUNCH("itcv_clip_coef");
  return 0;
}

int itcv_scale_by_dev(float* x, size_t n, const float* coef_dev, void* stream) {
  ITCV_REQUIRE(x && coef_dev, "itcv_scale_by_dev");
  ITCV_REQUIRE(((uintptr_t)x & 15) == 0, "itcv_scale_by_dev(16-byte aligned input)");
  if (!n) return 0;
  hipLaunchKernelGGL(scale_by_dev_kernel, dim3(stream_grid(n, 4)), dim3(256), 0, S(stream), x, n, coef_dev);
  ITCV_CHECK_LAUNCH("itcv_scale_by_dev");
  return 0;
}

int itcv_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                   float eps, int step, void* stream) {
  ITCV_REQUIRE(p && g && m && v && step >= 1, "itcv_adam_step");
  if (!n) return 0;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1), sqrt_bc2 = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n, 1)), dim3(256), 0, S(stream), p, g, m, v, n, step_size, beta1,
                     beta2, eps, sqrt_bc2);
  ITCV_CHECK_LAUNCH("itcv_adam_step");
  return 0;
}

int itcv_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                       float eps, int* step_dev, void* stream) {
  ITCV_REQUIRE(p && g && m && v && step_dev, "itcv_adam_step_dev");
  if (!n) return 0;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(stream_grid(n, 1)), dim3(256), 0, S(stream), p, g, m, v, n, lr, beta1, beta2,
                     eps, step_dev);
  ITCV_CHECK_LAUNCH("itcv_adam_step_dev");
  hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, S(stream), step_dev);
  ITCV_CHECK_LAUNCH("itcv_adam_step_dev(bump)");
  return 0;
}

int itcv_hflip(const float* x, float* y, const unsigned char* flip, int B, int rows_per_image, int W, void* stream) {
  ITCV_REQUIRE(x && y && flip && x != y && B > 0 && rows_per_image > 0 && W > 0, "itcv_hflip");
  const size_t n = (size_t)B * rows_per_image * W;
  hipLaunchKernelGGL(hflip_kernel, dim3(stream_grid(n, 1)), dim3(256), 0, S(stream), x, y, flip, (size_t)rows_per_image, W, n);
  ITCV_CHECK_LAUNCH("itcv_hflip");
  return 0;
}

int itcv_fill(float* x, size_t n, float value, void* stream) {
  ITCV_REQUIRE(x, "itcv_fill");
  if (!n) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(n, 1)), dim3(256), 0, S(stream), x, n, value);
  ITCV_CHECK_LAUNCH("itcv_fill");
  return 0;
}

}  // extern "C"
