// Latent-space math of the beta-TC-VAE term: reparameterisation, analytic KL and the fused
// O(B^2 D) pairwise Gaussian log-density + minibatch stratified / weighted sampling estimator.
// Replaces /root/reference/ops.py:15-29,32-49,52-115,136-185 and solvers/tc.py:104-121.
//
// The [B,B,D] log-density tensor of the reference is never materialised: one workgroup owns
// one sample row j, lanes run over the latent dimension l (coalesced reads of mu[i][:]), the
// per-(j,l) logsumexp over i lives in registers and the per-(j,i) sum over l is a wavefront
// reduction accumulated in LDS.  VALU/transcendental bound (exp/log per element), not a GEMM:
// the clamp(min=-50) sits inside the sum over l and the per-dimension logsumexp needs every
// element, so the contraction cannot be moved to the matrix cores without changing results.
#include <math.h>

#include "common.h"

namespace itcv {

constexpr float kHalfLog2Pi = 0.9189385332046727f;  // 0.5*log(2*pi)
constexpr float kLog2Pi = 1.8378770664093453f;
constexpr float kVarEps = 1e-4f;   // ops.py:18
constexpr float kFloor = -50.f;    // ops.py:21,29
constexpr int kTcThreads = 256;

struct TcConst {
  float lw_n, lw_s, lw_m;  // log(1/N), log((N-M)/(N M)), log(1/M)        (ops.py:42-49)
  float log_bn;            // log(B*N)                                      (ops.py:96,99)
  int M;                   // B_total - 1
};

// log importance weight of element (global row jg, column i): ops.py:45-48 -- the flat stride
// M+1 == B addresses COLUMNS 0 and 1 of every row, then [M-1, 0] is overwritten.
__device__ __forceinline__ float log_iw(const TcConst& c, int jg, int i) {
  if (i == 0) return jg == c.M - 1 ? c.lw_s : c.lw_n;
  return i == 1 ? c.lw_s : c.lw_m;
}

// unclamped log density; EPS: ops.py:15-21 (variance floor 1e-4), else ops.py:24-29
template <bool EPS>
__device__ __forceinline__ float logdens(float d, float lv) {
  if (EPS) {
    const float vh = fmaxf(expf(lv), kVarEps);
    return -(0.5f * (logf(vh) + d * d / vh) + kHalfLog2Pi);
  }
  return -0.5f * (d * d * expf(-lv) + lv + kLog2Pi);
}

// One block per local row j.  dynamic LDS: spart[nwaves][Bt]
// The estimator kernels walk the B_total rows of mu serially per latent column; straight from global memory every
// step of that walk is an exposed L2 round trip (~0.7 us: 47 us for a 64 x 128 problem).  The rows are therefore
// staged through LDS tc_chunk(D) (<= 32) at a time (coalesced, once per pass); the arithmetic and its order are unchanged.
constexpr int kTcChunkMax = 32;
__host__ __device__ inline int tc_chunk(int D) {   // rows staged per pass: <= 32 and <= 32 KB per array
  const int r = 8192 / (D > 0 ? D : 1);
  return r < 1 ? 1 : (r > kTcChunkMax ? kTcChunkMax : r);
}
__device__ __forceinline__ void tc_stage_rows(float* dst, const float* __restrict__ src, int row0, int rows, int D) {
  const int n = rows * D;   // rows are contiguous in memory: one flat copy
  const float* s0 = src + (size_t)row0 * D;
  for (int i = threadIdx.x; i < n; i += kTcThreads) dst[i] = s0[i];
}

template <bool VROW, bool EPS, bool MWS>
__global__ __launch_bounds__(kTcThreads) void tc_fwd_kernel(const float* __restrict__ z,
                                                            const float* __restrict__ mu_all,
                                                            const float* __restrict__ logvar,
                                                            float* __restrict__ prodm, float* __restrict__ logqz,
                                                            float* __restrict__ lse, int Bt, int row_offset, int D,
                                                            TcConst c) {
  const int kTcChunk = tc_chunk(D);
  extern __shared__ __attribute__((aligned(16))) float spart[];
  __shared__ float red[kTcThreads / 64];
  const int j = blockIdx.x, jg = row_offset + j;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = kTcThreads / 64;
  float* mu_s = spart + NW * Bt;                 // [kTcChunk][D]
  float* lv_s = mu_s + kTcChunk * D;             // [kTcChunk][D], only when the variance comes from row i
  for (int i = tid; i < NW * Bt; i += kTcThreads) spart[i] = 0.f;
  __syncthreads();

  float prod_acc = 0.f;  // this thread's share of sum_l logsumexp_i
  for (int l0 = 0; l0 < D; l0 += kTcThreads) {
    const int l = l0 + tid;
    const bool act = l < D;
    const float zj = act ? z[(size_t)j * D + l] : 0.f;
    const float lvj = (act && VROW) ? logvar[(size_t)j * D + l] : 0.f;
    // pass 1: running max over i of (logW + lp); row sums S[j,i] via wave reductions
    float mx = -INFINITY;
    for (int i0 = 0; i0 < Bt; i0 += kTcChunk) {
      const int rows = min(kTcChunk, Bt - i0);
      __syncthreads();
      tc_stage_rows(mu_s, mu_all, i0, rows, D);
      if (!VROW) tc_stage_rows(lv_s, logvar, i0, rows, D);
      __syncthreads();
      for (int ii = 0; ii < rows; ++ii) {
        const int i = i0 + ii;
        float lp = 0.f;
        if (act) {
          const float lv = VROW ? lvj : lv_s[ii * D + l];
          lp = fmaxf(logdens<EPS>(zj - mu_s[ii * D + l], lv), kFloor);
          const float v = MWS ? lp : lp + log_iw(c, jg, i);
          mx = fmaxf(mx, v);
        }
        const float s = wave_sum(lp);
        if (lane == 0) spart[wid * Bt + i] += s;
      }
    }
    // pass 2: sum of exp(v - max)
    float se = 0.f;
    for (int i0 = 0; i0 < Bt; i0 += kTcChunk) {
      const int rows = min(kTcChunk, Bt - i0);
      __syncthreads();
      tc_stage_rows(mu_s, mu_all, i0, rows, D);
      if (!VROW) tc_stage_rows(lv_s, logvar, i0, rows, D);
      __syncthreads();
      if (act) {
        for (int ii = 0; ii < rows; ++ii) {
          const int i = i0 + ii;
          const float lv = VROW ? lvj : lv_s[ii * D + l];
          const float lp = fmaxf(logdens<EPS>(zj - mu_s[ii * D + l], lv), kFloor);
          const float v = MWS ? lp : lp + log_iw(c, jg, i);
          se += expf(v - mx);
        }
      }
    }
    if (act) {
      float r = mx + logf(se);
      lse[(size_t)j * D + l] = r;
      if (MWS) r -= c.log_bn;
      prod_acc += r;
    }
  }
  __syncthreads();
  // log q(z_j) = logsumexp_i(logW + sum_l lp)
  float mx = -INFINITY;
  for (int i = tid; i < Bt; i += kTcThreads) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += spart[w * Bt + i];
    if (!MWS) s += log_iw(c, jg, i);
    spart[i] = s;  // wave 0's slot doubles as the combined row (each i touched by one thread)
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int i = tid; i < Bt; i += kTcThreads) se += expf(spart[i] - mx);
  se = block_sum(se, red);
  const float pm = block_sum(prod_acc, red);
  if (tid == 0) {
    logqz[j] = mx + logf(se) - (MWS ? c.log_bn : 0.f);
    prodm[j] = pm;
  }
}

// ---- backward of sum_j g[j]*(logqz[j]-prodm[j]), live path (VROW, EPS, MSS) ------------------
// Row kernel: block per row j.  Writes wq[j][i] = g_j * softmax_i(logW + S[j,:])[i] to scratch and
// the row gradients dz[j][:], dlogvar[j][:].
__global__ __launch_bounds__(kTcThreads) void tc_bwd_rows_kernel(
    const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ mu_all,
    const float* __restrict__ logvar, const float* __restrict__ logqz, const float* __restrict__ lse,
    float* __restrict__ wq, float* __restrict__ dz, float* __restrict__ dlogvar, int Bt, int row_offset, int D,
    TcConst c) {
  const int kTcChunk = tc_chunk(D);
  extern __shared__ __attribute__((aligned(16))) float spart[];
  const int j = blockIdx.x, jg = row_offset + j;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = kTcThreads / 64;
  float* mu_s = spart + NW * Bt;                 // [kTcChunk][D]
  for (int i = tid; i < NW * Bt; i += kTcThreads) spart[i] = 0.f;
  __syncthreads();
  // S[j,i] exactly as in the forward
  for (int l0 = 0; l0 < D; l0 += kTcThreads) {
    const int l = l0 + tid;
    const bool act = l < D;
    const float zj = act ? z[(size_t)j * D + l] : 0.f;
    const float lvj = act ? logvar[(size_t)j * D + l] : 0.f;
    for (int i0 = 0; i0 < Bt; i0 += kTcChunk) {
      const int rows = min(kTcChunk, Bt - i0);
      __syncthreads();
      tc_stage_rows(mu_s, mu_all, i0, rows, D);
      __syncthreads();
      for (int ii = 0; ii < rows; ++ii) {
        float lp = 0.f;
        if (act) lp = fmaxf(logdens<true>(zj - mu_s[ii * D + l], lvj), kFloor);
        const float s = wave_sum(lp);
        if (lane == 0) spart[wid * Bt + i0 + ii] += s;
      }
    }
  }
  __syncthreads();
  const float gj = g[j], lq = logqz[j];
  for (int i = tid; i < Bt; i += kTcThreads) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += spart[w * Bt + i];
    const float q = gj * expf(s + log_iw(c, jg, i) - lq);
    spart[i] = q;
    wq[(size_t)j * Bt + i] = q;
  }
  __syncthreads();
  for (int l0 = 0; l0 < D; l0 += kTcThreads) {
    const int l = l0 + tid;
    const bool act = l < D;
    const float zj = act ? z[(size_t)j * D + l] : 0.f, lvj = act ? logvar[(size_t)j * D + l] : 0.f;
    const float var = expf(lvj), vh = fmaxf(var, kVarEps), lvh = logf(vh);
    const float ls = act ? lse[(size_t)j * D + l] : 0.f;
    float az = 0.f, av = 0.f;
    for (int i0 = 0; i0 < Bt; i0 += kTcChunk) {
      const int rows = min(kTcChunk, Bt - i0);
      __syncthreads();
      tc_stage_rows(mu_s, mu_all, i0, rows, D);
      __syncthreads();
      if (act) {
        for (int ii = 0; ii < rows; ++ii) {
          const int i = i0 + ii;
          const float d = zj - mu_s[ii * D + l];
          const float lp = -(0.5f * (lvh + d * d / vh) + kHalfLog2Pi);
          if (lp >= kFloor) {  // clamp(min=-50) passes the gradient where lp >= -50
            const float G = spart[i] - gj * expf(lp + log_iw(c, jg, i) - ls);
            const float dv = d / vh;
            az -= G * dv;
            // d lp / d var at the clamped value, times d var / d logvar of the UNCLAMPED variance
            av -= G * 0.5f * (1.f / vh - dv * dv);
          }
        }
      }
    }
    if (act) {
      dz[(size_t)j * D + l] = az;
      dlogvar[(size_t)j * D + l] = av * var;
    }
  }
}

// Column kernel: block per column i; dmu_all[i][l] = sum_j G[j,i,l] * (z_j - mu_i)/vhat_j
__global__ __launch_bounds__(kTcThreads) void tc_bwd_cols_kernel(
    const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ mu_all,
    const float* __restrict__ logvar, const float* __restrict__ lse, const float* __restrict__ wq,
    float* __restrict__ dmu_all, int Bl, int Bt, int row_offset, int D, TcConst c) {
  const int kTcChunk = tc_chunk(D);
  extern __shared__ __attribute__((aligned(16))) float cs[];   // z, logvar, lse rows [3][kTcChunk][D], then wq, g [2][kTcChunk]
  float* z_s = cs;
  float* lv_s = z_s + kTcChunk * D;
  float* ls_s = lv_s + kTcChunk * D;
  float* wq_s = ls_s + kTcChunk * D;
  float* g_s = wq_s + kTcChunk;
  const int i = blockIdx.x;
  for (int l0 = 0; l0 < D; l0 += kTcThreads) {
    const int l = l0 + threadIdx.x;
    const bool act = l < D;
    const float mi = act ? mu_all[(size_t)i * D + l] : 0.f;
    float acc = 0.f;
    for (int j0 = 0; j0 < Bl; j0 += kTcChunk) {
      const int rows = min(kTcChunk, Bl - j0);
      __syncthreads();
      tc_stage_rows(z_s, z, j0, rows, D);
      tc_stage_rows(lv_s, logvar, j0, rows, D);
      tc_stage_rows(ls_s, lse, j0, rows, D);
      if ((int)threadIdx.x < rows) {
        wq_s[threadIdx.x] = wq[(size_t)(j0 + threadIdx.x) * Bt + i];
        g_s[threadIdx.x] = g[j0 + threadIdx.x];
      }
      __syncthreads();
      if (act) {
        for (int jj = 0; jj < rows; ++jj) {
          const float lvj = lv_s[jj * D + l];
          const float vh = fmaxf(expf(lvj), kVarEps);
          const float d = z_s[jj * D + l] - mi;
          const float lp = -(0.5f * (logf(vh) + d * d / vh) + kHalfLog2Pi);
          if (lp >= kFloor) {
            const float G = wq_s[jj] - g_s[jj] * expf(lp + log_iw(c, row_offset + j0 + jj, i) - ls_s[jj * D + l]);
            acc += G * d / vh;
          }
        }
      }
    }
    if (act) dmu_all[(size_t)i * D + l] = acc;
  }
}

// solvers/tc.py:104-109: log q(z_j|x_j) and log p(z_j) with the ops.py:24-29 density
__global__ __launch_bounds__(256) void diag_logdensity_kernel(const float* __restrict__ z,
                                                             const float* __restrict__ mu,
                                                             const float* __restrict__ logvar,
                                                             float* __restrict__ logq, float* __restrict__ logp,
                                                             int D) {
  __shared__ float red[4];
  const int j = blockIdx.x;
  float a = 0.f, b = 0.f;
  for (int l = threadIdx.x; l < D; l += 256) {
    const float zz = z[(size_t)j * D + l];
    a += fmaxf(logdens<false>(zz - mu[(size_t)j * D + l], logvar[(size_t)j * D + l]), kFloor);
    b += fmaxf(logdens<false>(zz, 0.f), kFloor);
  }
  a = block_sum(a, red);
  b = block_sum(b, red);
  if (threadIdx.x == 0) logq[j] = a, logp[j] = b;
}

__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                   const float* __restrict__ eps, float* __restrict__ z, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    z[i] = mu[i] + eps[i] * expf(0.5f * lv[i]);
}
__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ lv,
                                   const float* __restrict__ eps, float* __restrict__ dmu,
                                   float* __restrict__ dlv, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = dz[i];
    dmu[i] = d;
    dlv[i] = d * eps[i] * 0.5f * expf(0.5f * lv[i]);
  }
}
__global__ __launch_bounds__(256) void kl_rows_fwd_kernel(const float* __restrict__ lv, const float* __restrict__ mu,
                                                         float* __restrict__ kl, int D) {
  __shared__ float red[4];
  const int j = blockIdx.x;
  float a = 0.f;
  for (int l = threadIdx.x; l < D; l += 256) {
    const float v = lv[(size_t)j * D + l], m = mu[(size_t)j * D + l];
    a += 1.f + v - expf(v) - m * m;
  }
  a = block_sum(a, red);
  if (threadIdx.x == 0) kl[j] = -0.5f * a;
}
__global__ void kl_rows_bwd_kernel(const float* __restrict__ g, const float* __restrict__ lv,
                                   const float* __restrict__ mu, float* __restrict__ dlv, float* __restrict__ dmu,
                                   int D, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gj = g[i / D];
    dlv[i] = gj * (-0.5f) * (1.f - expf(lv[i]));
    dmu[i] = gj * mu[i];
  }
}

static int make_const(const char* name, int Bt, int64_t N, TcConst* c) {
  if (Bt < 2) return fail("%s: batch size must be >= 2 (M = B-1 divides the weights; ops.py:43-45)", name);
  if (N <= 0) return fail("%s: dataset_size must be positive", name);
  const double n = (double)N, m = (double)(Bt - 1);
  // the reference fills an fp32 matrix and takes its log (ops.py:45-49)
  c->lw_n = logf((float)(1.0 / n));
  c->lw_s = logf((float)((n - m) / (n * m)));
  c->lw_m = logf((float)(1.0 / m));
  c->log_bn = (float)log((double)Bt * n);
  c->M = Bt - 1;
  return 0;
}

static inline int ew_grid(size_t n) {
  size_t b = cdivz(n, 256);
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace itcv

using namespace itcv;

extern "C" {

int itcv_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, size_t n, void* stream) {
  ITCV_REQUIRE(mu && logvar && eps && z, "itcv_reparam_fwd");
  if (!n) return 0;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, S(stream), mu, logvar, eps, z, n);
  ITCV_CHECK_LAUNCH("itcv_reparam_fwd");
  return 0;
}
int itcv_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar, size_t n,
                     void* stream) {
  ITCV_REQUIRE(dz && logvar && eps && dmu && dlogvar, "itcv_reparam_bwd");
  if (!n) return 0;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, S(stream), dz, logvar, eps, dmu, dlogvar, n);
  ITCV_CHECK_LAUNCH("itcv_reparam_bwd");
  return 0;
}
int itcv_kl_rows_fwd(const float* logvar, const float* mu, float* kl, int B, int D, void* stream) {
  ITCV_REQUIRE(logvar && mu && kl && B > 0 && D > 0, "itcv_kl_rows_fwd");
  hipLaunchKernelGGL(kl_rows_fwd_kernel, dim3(B), dim3(256), 0, S(stream), logvar, mu, kl, D);
  ITCV_CHECK_LAUNCH("itcv_kl_rows_fwd");
  return 0;
}
int itcv_kl_rows_bwd(const float* g, const float* logvar, const float* mu, float* dlogvar, float* dmu, int B, int D,
                     void* stream) {
  ITCV_REQUIRE(g && logvar && mu && dlogvar && dmu && B > 0 && D > 0, "itcv_kl_rows_bwd");
  const size_t n = (size_t)B * D;
  hipLaunchKernelGGL(kl_rows_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, S(stream), g, logvar, mu, dlogvar, dmu, D, n);
  ITCV_CHECK_LAUNCH("itcv_kl_rows_bwd");
  return 0;
}

int itcv_tc_fwd(const float* z, const float* mu_all, const float* logvar, float* prodm, float* logqz, float* lse,
                int Bl, int Bt, int row_offset, int D, int64_t dataset_size, int flags, void* stream) {
  ITCV_REQUIRE(z && mu_all && logvar && prodm && logqz && lse && Bl > 0 && D > 0, "itcv_tc_fwd");
  ITCV_REQUIRE(row_offset >= 0 && row_offset + Bl <= Bt, "itcv_tc_fwd(rows must lie inside the global batch)");
  TcConst c;
  if (int e = make_const("itcv_tc_fwd", Bt, dataset_size, &c)) return e;
  const bool vrow = flags & ITCV_TC_VAR_FROM_ROW, eps = flags & ITCV_TC_EPS_DENSITY, mws = flags & ITCV_TC_WEIGHTED;
  const size_t lds = ((size_t)(kTcThreads / 64) * Bt + (size_t)(vrow ? 1 : 2) * tc_chunk(D) * D) * sizeof(float);
  if (lds > 128 * 1024) return fail("%s: global batch %lld / latent size too large for the LDS row buffers", "itcv_tc_fwd", Bt);
  dim3 grid(Bl), block(kTcThreads);
  hipStream_t st = S(stream);
#define ITCV_TC_LAUNCH(V, E, W)                                                                              \
  do {                                                                                                       \
    if (lds > 64 * 1024)                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tc_fwd_kernel<V, E, W>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    hipLaunchKernelGGL((tc_fwd_kernel<V, E, W>), grid, block, lds, st, z, mu_all, logvar, prodm, logqz, lse, \
                       Bt, row_offset, D, c);                                                                \
  } while (0)
  if (vrow && eps && !mws) ITCV_TC_LAUNCH(true, true, false);
  else if (vrow && eps && mws) ITCV_TC_LAUNCH(true, true, true);
  else if (vrow && !eps && !mws) ITCV_TC_LAUNCH(true, false, false);
  else if (vrow && !eps && mws) ITCV_TC_LAUNCH(true, false, true);
  else if (!vrow && eps && !mws) ITCV_TC_LAUNCH(false, true, false);
  else if (!vrow && eps && mws) ITCV_TC_LAUNCH(false, true, true);
  else if (!vrow && !eps && !mws) ITCV_TC_LAUNCH(false, false, false);
  else ITCV_TC_LAUNCH(false, false, true);
#undef ITCV_TC_LAUNCH
  ITCV_CHECK_LAUNCH("itcv_tc_fwd");
  return 0;
}

size_t itcv_tc_bwd_workspace(int Bl, int Bt) { return Bl > 0 && Bt > 0 ? (size_t)Bl * Bt * sizeof(float) : 0; }

int itcv_tc_bwd(const float* g, const float* z, const float* mu_all, const float* logvar, const float* logqz,
                const float* lse, float* dz, float* dmu_all, float* dlogvar, int Bl, int Bt, int row_offset, int D,
                int64_t dataset_size, int flags, void* ws, size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(g && z && mu_all && logvar && logqz && lse && dz && dmu_all && dlogvar && Bl > 0 && D > 0,
               "itcv_tc_bwd");
  ITCV_REQUIRE(row_offset >= 0 && row_offset + Bl <= Bt, "itcv_tc_bwd(rows must lie inside the global batch)");
  if (flags != ITCV_TC_LIVE)
    return fail("%s: gradients exist for the live estimator only (flags == ITCV_TC_LIVE)", "itcv_tc_bwd");
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)Bl * Bt * sizeof(float), "itcv_tc_bwd(workspace)");
  TcConst c;
  if (int e = make_const("itcv_tc_bwd", Bt, dataset_size, &c)) return e;
  const size_t lds = ((size_t)(kTcThreads / 64) * Bt + (size_t)tc_chunk(D) * D) * sizeof(float);
  const size_t lds_c = ((size_t)3 * tc_chunk(D) * D + 2 * tc_chunk(D)) * sizeof(float);
  if (lds > 128 * 1024 || lds_c > 128 * 1024)
    return fail("%s: global batch %lld / latent size too large for the LDS row buffers", "itcv_tc_bwd", Bt);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tc_bwd_rows_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (lds_c > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tc_bwd_cols_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
  float* wq = static_cast<float*>(ws);
  hipStream_t st = S(stream);
  hipLaunchKernelGGL(tc_bwd_rows_kernel, dim3(Bl), dim3(kTcThreads), lds, st, g, z, mu_all, logvar, logqz, lse, wq, dz,
                     dlogvar, Bt, row_offset, D, c);
  ITCV_CHECK_LAUNCH("itcv_tc_bwd(rows)");
  hipLaunchKernelGGL(tc_bwd_cols_kernel, dim3(Bt), dim3(kTcThreads), lds_c, st, g, z, mu_all, logvar, lse, wq, dmu_all, Bl,
                     Bt, row_offset, D, c);
  ITCV_CHECK_LAUNCH("itcv_tc_bwd(cols)");
  return 0;
}

int itcv_diag_logdensity_rows(const float* z, const float* mu, const float* logvar, float* logq_cx, float* logpz,
                              int B, int D, void* stream) {
  ITCV_REQUIRE(z && mu && logvar && logq_cx && logpz && B > 0 && D > 0, "itcv_diag_logdensity_rows");
  hipLaunchKernelGGL(diag_logdensity_kernel, dim3(B), dim3(256), 0, S(stream), z, mu, logvar, logq_cx, logpz, D);
  ITCV_CHECK_LAUNCH("itcv_diag_logdensity_rows");
  return 0;
}

}  // extern "C"
