// Latent-space math of the beta-TC-VAE term: reparameterisation, analytic KL and the fused
// O(B^2 D) pairwise Gaussian log-density + minibatch stratified / weighted sampling estimator.
// Replaces /root/reference/ops.py:15-29,32-49,52-115,136-185 and solvers/tc.py:104-121.
//
// The [B,B,D] log-density tensor of the reference is never materialised: a workgroup owns one
// sample row j and a chunk of columns i, lanes run over the latent dimension l (coalesced reads of
// mu[i][:]), the per-(j,l) logsumexp over i is assembled from per-chunk partials and the per-(j,i)
// sum over l is one wavefront reduction.  VALU/transcendental bound (one exp per element), not a GEMM:
// the clamp(min=-50) sits inside the sum over l and the per-dimension logsumexp needs every
// element, so the contraction cannot be moved to the matrix cores without changing results.
#include <math.h>

#include "common.h"

namespace itcv {

constexpr float kHalfLog2Pi = 0.9189385332046727f;  // 0.5*log(2*pi)
constexpr float kLog2Pi = 1.8378770664093453f;
constexpr float kVarEps = 1e-4f;   // ops.py:18
constexpr float kFloor = -50.f;    // ops.py:21,29

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

// ---- estimator kernels ------------------------------------------------------------------------------------
// Work decomposition (round 2): the first version ran one 256-thread block per sample row with the lanes over the
// latent dimension -- 64 blocks on 256 CUs, half of each block idle at D = 128, two passes of exp/log/divide per
// element: 43 us for a 64 x 64 x 128 problem and ~8x that for the 8-GPU configuration's 64 x 512 rows.  Now
//   * the grid is (row j) x (chunk of kTcIC columns i): >= 256 blocks at every size of interest;
//   * inside a block the four waves take different columns i and the lanes of a wave run over l (D/64 values per
//     lane), so sum_l lp[j,i,l] is ONE wave reduction owned by one wave and every lane is busy;
//   * the per-element transcendental work is hoisted: the density is lp = -0.5*(A + d*d*Bv) - C with
//     A = log(vhat), Bv = 1/vhat (ops.py:15-21) or A = logvar, Bv = exp(-logvar) (ops.py:24-29) computed once per
//     (variance row, l); what remains per element is one exp for the logsumexp;
//   * the logsumexp over i is assembled from per-chunk (max, sum exp) partials by a short second kernel, which also
//     finishes log q(z_j) from the joint terms S[j,i] = logW[j,i] + sum_l lp[j,i,l].  S is an OUTPUT (`sjoint`): the
//     backward reads it instead of recomputing the forward.
// Orders of summation are fixed (no atomics): results are bitwise reproducible.
constexpr int kTcIC = 16;                 // columns i per block: four per wave
constexpr int kTcRW = kTcIC / 4;
constexpr int kTcDLMax = 8;               // D <= 512

__host__ __device__ inline int tc_chunks(int Bt) { return (Bt + kTcIC - 1) / kTcIC; }

// A, Bv of the density for one (variance row, l)
template <bool EPS>
__device__ __forceinline__ void dens_coef(float lv, float& A, float& Bv) {
  if (EPS) {
    const float vh = fmaxf(expf(lv), kVarEps);
    A = logf(vh), Bv = 1.f / vh;
  } else {
    A = lv, Bv = expf(-lv);
  }
}

template <int DL, bool VROW, bool EPS, bool MWS>
__global__ __launch_bounds__(256) void tc_fwd_part_kernel(const float* __restrict__ z, const float* __restrict__ mu_all,
                                                          const float* __restrict__ logvar, float* __restrict__ pmax,
                                                          float* __restrict__ psum, float* __restrict__ sjoint, int Bt,
                                                          int row_offset, int D, int nch, TcConst c) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* mu_s = sm;                                   // [kTcIC][D]
  float* A_s = mu_s + kTcIC * D;                      // [kTcIC][D]  (!VROW only)
  float* B_s = A_s + (VROW ? 0 : kTcIC * D);          // [kTcIC][D]  (!VROW only)
  float* cm = B_s + (VROW ? 0 : kTcIC * D);           // [4][D] per-wave running max
  float* cs = cm + 4 * D;                             // [4][D] per-wave sum of exp
  const int j = blockIdx.x, ch = blockIdx.y, jg = row_offset + j;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i0 = ch * kTcIC, rows = min(kTcIC, Bt - i0);
  for (int e = tid; e < rows * D; e += 256) {
    mu_s[e] = mu_all[(size_t)i0 * D + e];
    if (!VROW) dens_coef<EPS>(logvar[(size_t)i0 * D + e], A_s[e], B_s[e]);
  }
  float zj[DL], Aj[DL], Bj[DL];
#pragma unroll
  for (int k = 0; k < DL; ++k) {
    const int l = lane + 64 * k;
    zj[k] = 0.f, Aj[k] = 0.f, Bj[k] = 0.f;
    if (l < D) {
      zj[k] = z[(size_t)j * D + l];
      if (VROW) dens_coef<EPS>(logvar[(size_t)j * D + l], Aj[k], Bj[k]);
    }
  }
  __syncthreads();
  float v[kTcRW][DL];
#pragma unroll
  for (int r = 0; r < kTcRW; ++r) {
    const int ii = wid + 4 * r;                        // wave-uniform
    const bool have = ii < rows;
    const float liw = (have && !MWS) ? log_iw(c, jg, i0 + ii) : 0.f;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < DL; ++k) {
      const int l = lane + 64 * k;
      v[r][k] = -INFINITY;
      if (have && l < D) {
        const float d = zj[k] - mu_s[ii * D + l];
        const float A = VROW ? Aj[k] : A_s[ii * D + l], Bv = VROW ? Bj[k] : B_s[ii * D + l];
        const float lp = fmaxf(-0.5f * (A + d * d * Bv) - kHalfLog2Pi, kFloor);
        part += lp;
        v[r][k] = lp + liw;
      }
    }
    if (have) {
      part = wave_sum(part);
      if (lane == 0) sjoint[(size_t)j * Bt + i0 + ii] = part + liw;
    }
  }
#pragma unroll
  for (int k = 0; k < DL; ++k) {
    const int l = lane + 64 * k;
    float m = v[0][k];
#pragma unroll
    for (int r = 1; r < kTcRW; ++r) m = fmaxf(m, v[r][k]);
    float se = 0.f;
    if (m > -INFINITY) {
#pragma unroll
      for (int r = 0; r < kTcRW; ++r) se += expf(v[r][k] - m);     // exp(-inf) = 0 for the rows this wave does not have
    }
    if (l < D) cm[wid * D + l] = m, cs[wid * D + l] = se;
  }
  __syncthreads();
  for (int l = tid; l < D; l += 256) {
    float m = fmaxf(fmaxf(cm[l], cm[D + l]), fmaxf(cm[2 * D + l], cm[3 * D + l]));
    float se = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = cm[w * D + l];
      if (mw > -INFINITY) se += cs[w * D + l] * expf(mw - m);
    }
    pmax[((size_t)j * nch + ch) * D + l] = m;
    psum[((size_t)j * nch + ch) * D + l] = se;
  }
}

// stage 2: block per row j -- lse[j][l], prodm[j], logqz[j]
// Fused hook tail (solvers/tc.py:80-89): with `rows` set the block also evaluates the analytic KL of its sample,
// kl_j = -0.5 sum_l (1 + lv - e^lv - mu^2) (ops.py:161-163; mu_loc / lv_loc: this rank's rows), and writes
// rows[j] = coef_tc * (logqz_j - prodm_j) + coef_kl * kl_j  ((beta - 1) * TC + KL per sample, times the hook's scale).
struct TcKlOut {
  const float* mu_loc;
  const float* lv_loc;
  float* rows;
  float coef_tc, coef_kl;
};

template <bool MWS>
__global__ __launch_bounds__(256) void tc_fwd_finish_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                            const float* __restrict__ sjoint, float* __restrict__ prodm,
                                                            float* __restrict__ logqz, float* __restrict__ lse, int Bt, int D,
                                                            int nch, TcConst c, TcKlOut kk) {
  __shared__ float red[4];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float prod_acc = 0.f;
  for (int l = tid; l < D; l += 256) {
    const float* pm = pmax + (size_t)j * nch * D + l;
    const float* ps = psum + (size_t)j * nch * D + l;
    float m = -INFINITY;
    for (int k = 0; k < nch; ++k) m = fmaxf(m, pm[(size_t)k * D]);
    float se = 0.f;
    for (int k = 0; k < nch; ++k) se += ps[(size_t)k * D] * expf(pm[(size_t)k * D] - m);
    float r = m + logf(se);
    lse[(size_t)j * D + l] = r;
    if (MWS) r -= c.log_bn;
    prod_acc += r;
  }
  float mx = -INFINITY;
  for (int i = tid; i < Bt; i += 256) mx = fmaxf(mx, sjoint[(size_t)j * Bt + i]);
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int i = tid; i < Bt; i += 256) se += expf(sjoint[(size_t)j * Bt + i] - mx);
  se = block_sum(se, red);
  const float pm = block_sum(prod_acc, red);
  float klj = 0.f;
  if (kk.rows) {
    float a = 0.f;
    for (int l = tid; l < D; l += 256) {
      const float v = kk.lv_loc[(size_t)j * D + l], m = kk.mu_loc[(size_t)j * D + l];
      a += 1.f + v - expf(v) - m * m;
    }
    klj = -0.5f * block_sum(a, red);
  }
  if (tid == 0) {
    const float lq = mx + logf(se) - (MWS ? c.log_bn : 0.f);
    logqz[j] = lq;
    prodm[j] = pm;
    if (kk.rows) kk.rows[j] = kk.coef_tc * (lq - pm) + kk.coef_kl * klj;
  }
}

// out[0] = scale * sum_j rows[j] (/ B for mean): the reduction of a per-sample loss vector in one launch
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ rows, float* __restrict__ out, int B,
                                                         int mean, float scale) {
  __shared__ double red[4];
  double a = 0.0;
  for (int j = threadIdx.x; j < B; j += 256) a += (double)rows[j];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = scale * (float)(mean ? a / (double)B : a);
}

// ---- backward of sum_j g[j]*(logqz[j]-prodm[j]), live path (VROW, EPS, MSS) ------------------
//   G[j,i,l] = [lp >= -50] * (wq[j,i] - g_j * exp(lp + logW[j,i] - lse[j,l])),   wq[j,i] = g_j * exp(S[j,i] - logqz[j])
//   dz[j,l] = -sum_i G d/vhat_j,  dlogvar[j,l] = -var_j * sum_i G * 0.5*(1/vhat_j - (d/vhat_j)^2)   (straight-through clamp)
//   dmu[i,l] = sum_j G d/vhat_j
// Row kernel: block (j, 64-wide slice of l); the four waves split the columns i; also writes wq for the column kernel.
// Upstream gradient of the fused op: per row g[j] (bcast == 0) or one scalar g[0] (a reduced output); row j's TC term
// receives coef_tc * that, its KL term coef_kl * that (the plain estimator: coef_tc = 1, coef_kl = 0).
struct TcGrad {
  const float* g;
  int bcast;
  float coef_tc, coef_kl;
};
__device__ __forceinline__ float tc_g(const TcGrad& t, int j) { return t.bcast ? t.g[0] : t.g[j]; }

__global__ __launch_bounds__(256) void tc_bwd_rows_kernel(
    TcGrad tg, const float* __restrict__ z, const float* __restrict__ mu_all,
    const float* __restrict__ logvar, const float* __restrict__ logqz, const float* __restrict__ lse,
    const float* __restrict__ sjoint, float* __restrict__ wq, float* __restrict__ dz, float* __restrict__ dlogvar, int Bt,
    int row_offset, int D, TcConst c) {
  __shared__ float raz[4][64], rav[4][64];
  const int j = blockIdx.x, jg = row_offset + j, l = blockIdx.y * 64 + (threadIdx.x & 63);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool act = l < D;
  const float gr = tc_g(tg, j), gj = tg.coef_tc * gr, gk = tg.coef_kl * gr, lq = logqz[j];
  const float zj = act ? z[(size_t)j * D + l] : 0.f, lvj = act ? logvar[(size_t)j * D + l] : 0.f;
  const float var = expf(lvj), vh = fmaxf(var, kVarEps), lvh = logf(vh), ivh = 1.f / vh;
  const float ls = act ? lse[(size_t)j * D + l] : 0.f;
  float az = 0.f, av = 0.f;
#pragma unroll 4
  for (int i = wid; i < Bt; i += 4) {                  // wave-uniform i: sjoint / log_iw are scalar work
    const float liw = log_iw(c, jg, i);
    const float q = gj * expf(sjoint[(size_t)j * Bt + i] - lq);
    if (blockIdx.y == 0 && lane == 0) wq[(size_t)j * Bt + i] = q;
    if (act) {
      const float d = zj - mu_all[(size_t)i * D + l];
      const float lp = -(0.5f * (lvh + d * d * ivh) + kHalfLog2Pi);
      if (lp >= kFloor) {                              // clamp(min=-50) passes the gradient where lp >= -50
        const float G = q - gj * expf(lp + liw - ls);
        const float dv = d * ivh;
        az -= G * dv;
        av -= G * 0.5f * (ivh - dv * dv);              // d lp / d var at the clamped value (straight-through)
      }
    }
  }
  raz[wid][lane] = az, rav[wid][lane] = av;
  __syncthreads();
  if (wid == 0 && act) {
    dz[(size_t)j * D + l] = (raz[0][lane] + raz[1][lane]) + (raz[2][lane] + raz[3][lane]);
    // + the analytic KL's own d/dlogvar = -0.5 (1 - e^lv) (ops.py:161-163), weighted by gk (0 for the plain estimator)
    const float dl = ((rav[0][lane] + rav[1][lane]) + (rav[2][lane] + rav[3][lane])) * var;
    dlogvar[(size_t)j * D + l] = gk != 0.f ? dl + gk * (-0.5f) * (1.f - var) : dl;
  }
}

// Column kernel: block (i, 64-wide slice of l); the four waves split the rows j.
__global__ __launch_bounds__(256) void tc_bwd_cols_kernel(
    TcGrad tg, const float* __restrict__ z, const float* __restrict__ mu_all,
    const float* __restrict__ logvar, const float* __restrict__ lse, const float* __restrict__ wq,
    float* __restrict__ dmu_all, int Bl, int Bt, int row_offset, int D, TcConst c) {
  __shared__ float racc[4][64];
  const int i = blockIdx.x, l = blockIdx.y * 64 + (threadIdx.x & 63);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool act = l < D;
  const float mi = act ? mu_all[(size_t)i * D + l] : 0.f;
  float acc = 0.f;
#pragma unroll 2
  for (int j = wid; j < Bl; j += 4) {
    const float gj = tg.coef_tc * tc_g(tg, j), q = wq[(size_t)j * Bt + i], liw = log_iw(c, row_offset + j, i);
    if (act) {
      const float vh = fmaxf(expf(logvar[(size_t)j * D + l]), kVarEps), ivh = 1.f / vh;
      const float d = z[(size_t)j * D + l] - mi;
      const float lp = -(0.5f * (logf(vh) + d * d * ivh) + kHalfLog2Pi);
      if (lp >= kFloor) acc += (q - gj * expf(lp + liw - lse[(size_t)j * D + l])) * d * ivh;
    }
  }
  racc[wid][lane] = acc;
  __syncthreads();
  if (wid == 0 && act) {
    float v = (racc[0][lane] + racc[1][lane]) + (racc[2][lane] + racc[3][lane]);
    // the analytic KL's d/dmu = mu for this rank's own rows (column i is local row i - row_offset)
    if (tg.coef_kl != 0.f && i >= row_offset && i < row_offset + Bl) v += tg.coef_kl * tc_g(tg, i - row_offset) * mi;
    dmu_all[(size_t)i * D + l] = v;
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

// ---- materialising forms of the reference's named helpers (ops.py:15-29, 92-123) ---------------------------
// The training step never builds the [B,B,D] tensor (kernels above); these serve callers that use the reference's
// pieces one by one (`from ops import gaussian_log_density, minibatch_stratified_sampling`, solvers/tc.py:5-11).
struct Bc3 {                  // three operands broadcast to a common [n0][n1][n2] shape: element strides, 0 = broadcast
  int n1, n2;
  long long sx[3], sm[3], sl[3];
};

template <bool EPS>
__global__ void gauss_logdensity_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                            const float* __restrict__ lv, float* __restrict__ out, Bc3 b, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const long long i2 = (long long)(e % b.n2), r = (long long)(e / b.n2), i1 = r % b.n1, i0 = r / b.n1;
    const float d = x[i0 * b.sx[0] + i1 * b.sx[1] + i2 * b.sx[2]] - mu[i0 * b.sm[0] + i1 * b.sm[1] + i2 * b.sm[2]];
    out[e] = fmaxf(logdens<EPS>(d, lv[i0 * b.sl[0] + i1 * b.sl[1] + i2 * b.sl[2]]), kFloor);
  }
}

// elementwise gradients at the broadcast shape: dx (= -dmu) and dlogvar; zero where the -50 clamp is active;
// EPS: derivative with respect to the variance taken at the clamped value and handed to exp(logvar) unchanged
// (F.gaussian_nll_loss clamps a detached copy: straight-through), ops.py:17-21
template <bool EPS>
__global__ void gauss_logdensity_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                            const float* __restrict__ mu, const float* __restrict__ lv,
                                            float* __restrict__ dx, float* __restrict__ dlv, Bc3 b, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const long long i2 = (long long)(e % b.n2), r = (long long)(e / b.n2), i1 = r % b.n1, i0 = r / b.n1;
    const float d = x[i0 * b.sx[0] + i1 * b.sx[1] + i2 * b.sx[2]] - mu[i0 * b.sm[0] + i1 * b.sm[1] + i2 * b.sm[2]];
    const float l = lv[i0 * b.sl[0] + i1 * b.sl[1] + i2 * b.sl[2]];
    const float gg = g[e];
    float gx = 0.f, gl = 0.f;
    if (logdens<EPS>(d, l) >= kFloor) {
      if (EPS) {
        const float var = expf(l), ivh = 1.f / fmaxf(var, kVarEps), dv = d * ivh;
        gx = -gg * dv;
        gl = -gg * 0.5f * (ivh - dv * dv) * var;
      } else {
        const float iv = expf(-l);
        gx = -gg * d * iv;
        gl = -gg * 0.5f * (1.f - d * d * iv);
      }
    }
    dx[e] = gx, dlv[e] = gl;
  }
}

// ops.py:92-115 on a materialised lp[B][B][D]: block per row j.  S[j][i] = logW + sum_l lp, lse[j][l] are kept for
// the backward; logqz_raw is log q(z_j) before the weighted sampler's constant.
template <bool MWS>
__global__ __launch_bounds__(256) void sampling_fwd_kernel(const float* __restrict__ lp, float* __restrict__ prodm,
                                                           float* __restrict__ logqz, float* __restrict__ lse,
                                                           float* __restrict__ sj, int B, int D, TcConst c) {
  __shared__ float red[4];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* row = lp + (size_t)j * B * D;
  for (int i = wid; i < B; i += 4) {
    float s = 0.f;
    for (int l = lane; l < D; l += 64) s += row[(size_t)i * D + l];
    s = wave_sum(s);
    if (lane == 0) sj[(size_t)j * B + i] = s + (MWS ? 0.f : log_iw(c, j, i));
  }
  float prod_acc = 0.f;
  for (int l = tid; l < D; l += 256) {
    float m = -INFINITY;
    for (int i = 0; i < B; ++i) m = fmaxf(m, row[(size_t)i * D + l] + (MWS ? 0.f : log_iw(c, j, i)));
    float se = 0.f;
    for (int i = 0; i < B; ++i) se += expf(row[(size_t)i * D + l] + (MWS ? 0.f : log_iw(c, j, i)) - m);
    const float r = m + logf(se);
    lse[(size_t)j * D + l] = r;
    prod_acc += MWS ? r - c.log_bn : r;
  }
  __syncthreads();   // the S row written above is read by the whole block
  float mx = -INFINITY;
  for (int i = tid; i < B; i += 256) mx = fmaxf(mx, sj[(size_t)j * B + i]);
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int i = tid; i < B; i += 256) se += expf(sj[(size_t)j * B + i] - mx);
  se = block_sum(se, red);
  const float pm = block_sum(prod_acc, red);
  if (tid == 0) {
    logqz[j] = mx + logf(se) - (MWS ? c.log_bn : 0.f);
    prodm[j] = pm;
  }
}

// d lp[j][i][l] = g_prodm[j] * softmax_i(logW + lp)[j][i][l] + g_logqz[j] * softmax_i(S)[j][i]
template <bool MWS>
__global__ __launch_bounds__(256) void sampling_bwd_kernel(const float* __restrict__ gp, const float* __restrict__ gq,
                                                           const float* __restrict__ lp, const float* __restrict__ lse,
                                                           const float* __restrict__ sj, const float* __restrict__ logqz,
                                                           float* __restrict__ dlp, int B, int D, TcConst c) {
  const int j = blockIdx.x, i = blockIdx.y;
  const float liw = MWS ? 0.f : log_iw(c, j, i);
  const float lq = logqz[j] + (MWS ? c.log_bn : 0.f);
  const float wq = gq[j] * expf(sj[(size_t)j * B + i] - lq), g1 = gp[j];
  const size_t base = ((size_t)j * B + i) * D;
  for (int l = threadIdx.x; l < D; l += 256) dlp[base + l] = g1 * expf(lp[base + l] + liw - lse[(size_t)j * D + l]) + wq;
}

// ops.py:118-122 for a 2-D x[m][n] (m == n, or m == 1): diag[k] = x[k][k]; off[a][i][k] = x[i][k] - (i == k) * x[a][i]
// (torch.diag_embed puts the LAST dimension of x on the diagonal of a new trailing pair and x broadcasts against it)
__global__ void on_off_diag_kernel(const float* __restrict__ x, float* __restrict__ diag, float* __restrict__ off, int m,
                                   int n) {
  const size_t total = (size_t)m * n * n;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % n), i = (int)((e / n) % n), a = (int)(e / ((size_t)n * n));
    const float xv = x[(size_t)(m == 1 ? 0 : i) * n + k];
    off[e] = xv - (i == k ? x[(size_t)a * n + i] : 0.f);
    if (e < (size_t)(m < n ? m : n)) diag[e] = x[e * n + e];
  }
}

// ops.kl_divergence + the hook's `beta *` (ops.py:136-163, solvers/vae.py:63-77) in one launch: the waves of ONE block
// walk the rows; reduction 0: out[j] = scale * kl_j, 1 / 2: out[0] = scale * sum_j kl_j [/ B]
__global__ __launch_bounds__(1024) void kl_loss_fwd_kernel(const float* __restrict__ lv, const float* __restrict__ mu,
                                                          float* __restrict__ out, int B, int D, int reduction, float scale) {
  __shared__ double part[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double acc = 0.0;
  for (int j = w; j < B; j += 16) {
    float a = 0.f;
    for (int l = lane; l < D; l += 64) {
      const float v = lv[(size_t)j * D + l], m = mu[(size_t)j * D + l];
      a += 1.f + v - expf(v) - m * m;
    }
    const float kl = -0.5f * wave_sum(a);
    if (reduction == 0) {
      if (lane == 0) out[j] = scale * kl;
    } else {
      acc += (double)kl;
    }
  }
  if (reduction == 0) return;
  if (lane == 0) part[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += part[k];
    out[0] = scale * (float)(reduction == 2 ? t / (double)B : t);
  }
}
__global__ void kl_loss_bwd_kernel(const float* __restrict__ g, const float* __restrict__ lv, const float* __restrict__ mu,
                                   float* __restrict__ dlv, float* __restrict__ dmu, int D, size_t n, int reduction,
                                   float coef) {
  const float g0 = reduction ? g[0] * coef : 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gj = reduction ? g0 : g[i / D] * coef;
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

size_t itcv_tc_fwd_workspace(int Bl, int Bt, int D) {
  return Bl > 0 && Bt > 0 && D > 0 ? (size_t)2 * Bl * tc_chunks(Bt) * D * sizeof(float) : 0;
}

}  // extern "C"
static int tc_fwd_impl(const float* z, const float* mu_all, const float* logvar, float* prodm, float* logqz, float* lse,
                       float* sjoint, int Bl, int Bt, int row_offset, int D, int64_t dataset_size, int flags, void* ws,
                       size_t ws_bytes, const TcKlOut& kk, void* stream) {
  ITCV_REQUIRE(z && mu_all && logvar && prodm && logqz && lse && sjoint && Bl > 0 && D > 0, "itcv_tc_fwd");
  ITCV_REQUIRE(row_offset >= 0 && row_offset + Bl <= Bt, "itcv_tc_fwd(rows must lie inside the global batch)");
  if (D > 64 * kTcDLMax) return fail("%s: latent size %lld > 512 is not supported", "itcv_tc_fwd", D);
  TcConst c;
  if (int e = make_const("itcv_tc_fwd", Bt, dataset_size, &c)) return e;
  const int nch = tc_chunks(Bt);
  ITCV_REQUIRE(ws && ws_bytes >= itcv_tc_fwd_workspace(Bl, Bt, D), "itcv_tc_fwd(workspace)");
  float* pmax = static_cast<float*>(ws);
  float* psum = pmax + (size_t)Bl * nch * D;
  const bool vrow = flags & ITCV_TC_VAR_FROM_ROW, eps = flags & ITCV_TC_EPS_DENSITY, mws = flags & ITCV_TC_WEIGHTED;
  const size_t lds = ((size_t)(vrow ? 1 : 3) * kTcIC * D + (size_t)8 * D) * sizeof(float);   // <= 112 KB at D = 512
  const int dl = D <= 64 ? 1 : (D <= 128 ? 2 : (D <= 256 ? 4 : 8));
  dim3 grid(Bl, nch), block(256);
  hipStream_t st = S(stream);
#define ITCV_TC_PART(DL, V, E, W)                                                                               \
  do {                                                                                                          \
    if (lds > 64 * 1024)                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tc_fwd_part_kernel<DL, V, E, W>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
    hipLaunchKernelGGL((tc_fwd_part_kernel<DL, V, E, W>), grid, block, lds, st, z, mu_all, logvar, pmax, psum,   \
                       sjoint, Bt, row_offset, D, nch, c);                                                      \
  } while (0)
#define ITCV_TC_DL(V, E, W)                \
  do {                                     \
    if (dl == 1) ITCV_TC_PART(1, V, E, W); \
    else if (dl == 2) ITCV_TC_PART(2, V, E, W); \
    else if (dl == 4) ITCV_TC_PART(4, V, E, W); \
    else ITCV_TC_PART(8, V, E, W);         \
  } while (0)
  if (vrow && eps && !mws) ITCV_TC_DL(true, true, false);
  else if (vrow && eps && mws) ITCV_TC_DL(true, true, true);
  else if (vrow && !eps && !mws) ITCV_TC_DL(true, false, false);
  else if (vrow && !eps && mws) ITCV_TC_DL(true, false, true);
  else if (!vrow && eps && !mws) ITCV_TC_DL(false, true, false);
  else if (!vrow && eps && mws) ITCV_TC_DL(false, true, true);
  else if (!vrow && !eps && !mws) ITCV_TC_DL(false, false, false);
  else ITCV_TC_DL(false, false, true);
#undef ITCV_TC_DL
#undef ITCV_TC_PART
  ITCV_CHECK_LAUNCH("itcv_tc_fwd(partials)");
  if (mws)
    hipLaunchKernelGGL(tc_fwd_finish_kernel<true>, dim3(Bl), block, 0, st, pmax, psum, sjoint, prodm, logqz, lse, Bt, D, nch, c, kk);
  else
    hipLaunchKernelGGL(tc_fwd_finish_kernel<false>, dim3(Bl), block, 0, st, pmax, psum, sjoint, prodm, logqz, lse, Bt, D, nch, c, kk);
  ITCV_CHECK_LAUNCH("itcv_tc_fwd(finish)");
  return 0;
}
extern "C" {
int itcv_tc_fwd(const float* z, const float* mu_all, const float* logvar, float* prodm, float* logqz, float* lse,
                float* sjoint, int Bl, int Bt, int row_offset, int D, int64_t dataset_size, int flags, void* ws,
                size_t ws_bytes, void* stream) {
  return tc_fwd_impl(z, mu_all, logvar, prodm, logqz, lse, sjoint, Bl, Bt, row_offset, D, dataset_size, flags, ws, ws_bytes,
                     TcKlOut{nullptr, nullptr, nullptr, 0.f, 0.f}, stream);
}

int itcv_tc_kl_fwd(const float* z, const float* mu_all, const float* logvar, float* out, float* rows, float* prodm,
                   float* logqz, float* lse, float* sjoint, int Bl, int Bt, int row_offset, int D, int64_t dataset_size,
                   float coef_tc, float coef_kl, int reduction, void* ws, size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(out && (reduction == 0 || rows) && reduction >= 0 && reduction <= 2, "itcv_tc_kl_fwd");
  ITCV_REQUIRE(mu_all && row_offset >= 0 && row_offset + Bl <= Bt, "itcv_tc_kl_fwd(rows must lie inside the global batch)");
  const TcKlOut kk{mu_all + (size_t)row_offset * D, logvar, reduction == 0 ? out : rows, coef_tc, coef_kl};
  if (int e = tc_fwd_impl(z, mu_all, logvar, prodm, logqz, lse, sjoint, Bl, Bt, row_offset, D, dataset_size, ITCV_TC_LIVE, ws,
                          ws_bytes, kk, stream))
    return e;
  if (reduction) {
    hipLaunchKernelGGL(rows_reduce_kernel, dim3(1), dim3(256), 0, S(stream), rows, out, Bl, reduction == 2 ? 1 : 0, 1.f);
    ITCV_CHECK_LAUNCH("itcv_tc_kl_fwd(reduce)");
  }
  return 0;
}

size_t itcv_tc_bwd_workspace(int Bl, int Bt) { return Bl > 0 && Bt > 0 ? (size_t)Bl * Bt * sizeof(float) : 0; }

}  // extern "C"
static int tc_bwd_impl(const TcGrad& tg, const float* z, const float* mu_all, const float* logvar, const float* logqz,
                       const float* lse, const float* sjoint, float* dz, float* dmu_all, float* dlogvar, int Bl, int Bt,
                       int row_offset, int D, int64_t dataset_size, int flags, void* ws, size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(tg.g && z && mu_all && logvar && logqz && lse && sjoint && dz && dmu_all && dlogvar && Bl > 0 && D > 0,
               "itcv_tc_bwd");
  ITCV_REQUIRE(row_offset >= 0 && row_offset + Bl <= Bt, "itcv_tc_bwd(rows must lie inside the global batch)");
  if (flags != ITCV_TC_LIVE)
    return fail("%s: gradients exist for the live estimator only (flags == ITCV_TC_LIVE)", "itcv_tc_bwd");
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)Bl * Bt * sizeof(float), "itcv_tc_bwd(workspace)");
  TcConst c;
  if (int e = make_const("itcv_tc_bwd", Bt, dataset_size, &c)) return e;
  float* wq = static_cast<float*>(ws);
  hipStream_t st = S(stream);
  const int lch = cdiv(D, 64);
  hipLaunchKernelGGL(tc_bwd_rows_kernel, dim3(Bl, lch), dim3(256), 0, st, tg, z, mu_all, logvar, logqz, lse, sjoint, wq, dz,
                     dlogvar, Bt, row_offset, D, c);
  ITCV_CHECK_LAUNCH("itcv_tc_bwd(rows)");
  hipLaunchKernelGGL(tc_bwd_cols_kernel, dim3(Bt, lch), dim3(256), 0, st, tg, z, mu_all, logvar, lse, wq, dmu_all, Bl, Bt,
                     row_offset, D, c);
  ITCV_CHECK_LAUNCH("itcv_tc_bwd(cols)");
  return 0;
}
extern "C" {
int itcv_tc_bwd(const float* g, const float* z, const float* mu_all, const float* logvar, const float* logqz,
                const float* lse, const float* sjoint, float* dz, float* dmu_all, float* dlogvar, int Bl, int Bt,
                int row_offset, int D, int64_t dataset_size, int flags, void* ws, size_t ws_bytes, void* stream) {
  return tc_bwd_impl(TcGrad{g, 0, 1.f, 0.f}, z, mu_all, logvar, logqz, lse, sjoint, dz, dmu_all, dlogvar, Bl, Bt, row_offset,
                     D, dataset_size, flags, ws, ws_bytes, stream);
}
int itcv_tc_kl_bwd(const float* g, const float* z, const float* mu_all, const float* logvar, const float* logqz,
                   const float* lse, const float* sjoint, float* dz, float* dmu_all, float* dlogvar, int Bl, int Bt,
                   int row_offset, int D, int64_t dataset_size, float coef_tc, float coef_kl, int reduction, void* ws,
                   size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(reduction >= 0 && reduction <= 2, "itcv_tc_kl_bwd");
  const float r = reduction == 2 ? 1.f / (float)Bl : 1.f;
  return tc_bwd_impl(TcGrad{g, reduction ? 1 : 0, coef_tc * r, coef_kl * r}, z, mu_all, logvar, logqz, lse, sjoint, dz, dmu_all,
                     dlogvar, Bl, Bt, row_offset, D, dataset_size, ITCV_TC_LIVE, ws, ws_bytes, stream);
}

int itcv_kl_loss_fwd(const float* logvar, const float* mu, float* out, int B, int D, int reduction, float scale, void* stream) {
  ITCV_REQUIRE(logvar && mu && out && B > 0 && D > 0 && reduction >= 0 && reduction <= 2, "itcv_kl_loss_fwd");
  hipLaunchKernelGGL(kl_loss_fwd_kernel, dim3(1), dim3(1024), 0, S(stream), logvar, mu, out, B, D, reduction, scale);
  ITCV_CHECK_LAUNCH("itcv_kl_loss_fwd");
  return 0;
}
int itcv_kl_loss_bwd(const float* g, const float* logvar, const float* mu, float* dlogvar, float* dmu, int B, int D,
                     int reduction, float scale, void* stream) {
  ITCV_REQUIRE(g && logvar && mu && dlogvar && dmu && B > 0 && D > 0 && reduction >= 0 && reduction <= 2, "itcv_kl_loss_bwd");
  const size_t n = (size_t)B * D;
  hipLaunchKernelGGL(kl_loss_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, S(stream), g, logvar, mu, dlogvar, dmu, D, n, reduction,
                     reduction == 2 ? scale / (float)B : scale);
  ITCV_CHECK_LAUNCH("itcv_kl_loss_bwd");
  return 0;
}

static Bc3 make_bc3(const int64_t* dims, const int64_t* sx, const int64_t* sm, const int64_t* sl) {
  Bc3 b;
  b.n1 = (int)dims[1], b.n2 = (int)dims[2];
  for (int k = 0; k < 3; ++k) b.sx[k] = sx[k], b.sm[k] = sm[k], b.sl[k] = sl[k];
  return b;
}

int itcv_gauss_logdensity_fwd(const float* x, const float* mu, const float* logvar, float* out, const int64_t* dims,
                              const int64_t* sx, const int64_t* sm, const int64_t* sl, int eps_density, void* stream) {
  ITCV_REQUIRE(x && mu && logvar && out && dims && sx && sm && sl, "itcv_gauss_logdensity_fwd");
  ITCV_REQUIRE(dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[1] < (1ll << 31) && dims[2] < (1ll << 31),
               "itcv_gauss_logdensity_fwd(shape)");
  const size_t n = (size_t)dims[0] * dims[1] * dims[2];
  const Bc3 b = make_bc3(dims, sx, sm, sl);
  if (eps_density)
    hipLaunchKernelGGL(gauss_logdensity_fwd_kernel<true>, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, mu, logvar, out, b, n);
  else
    hipLaunchKernelGGL(gauss_logdensity_fwd_kernel<false>, dim3(ew_grid(n)), dim3(256), 0, S(stream), x, mu, logvar, out, b, n);
  ITCV_CHECK_LAUNCH("itcv_gauss_logdensity_fwd");
  return 0;
}

int itcv_gauss_logdensity_bwd(const float* g, const float* x, const float* mu, const float* logvar, float* dx,
                              float* dlogvar, const int64_t* dims, const int64_t* sx, const int64_t* sm,
                              const int64_t* sl, int eps_density, void* stream) {
  ITCV_REQUIRE(g && x && mu && logvar && dx && dlogvar && dims && sx && sm && sl, "itcv_gauss_logdensity_bwd");
  ITCV_REQUIRE(dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[1] < (1ll << 31) && dims[2] < (1ll << 31),
               "itcv_gauss_logdensity_bwd(shape)");
  const size_t n = (size_t)dims[0] * dims[1] * dims[2];
  const Bc3 b = make_bc3(dims, sx, sm, sl);
  if (eps_density)
    hipLaunchKernelGGL(gauss_logdensity_bwd_kernel<true>, dim3(ew_grid(n)), dim3(256), 0, S(stream), g, x, mu, logvar, dx,
                       dlogvar, b, n);
  else
    hipLaunchKernelGGL(gauss_logdensity_bwd_kernel<false>, dim3(ew_grid(n)), dim3(256), 0, S(stream), g, x, mu, logvar, dx,
                       dlogvar, b, n);
  ITCV_CHECK_LAUNCH("itcv_gauss_logdensity_bwd");
  return 0;
}

int itcv_sampling_fwd(const float* lp, float* prodm, float* logqz, float* lse, float* sjoint, int B, int D,
                      int64_t dataset_size, int weighted, void* stream) {
  ITCV_REQUIRE(lp && prodm && logqz && lse && sjoint && B > 0 && D > 0, "itcv_sampling_fwd");
  TcConst c;
  if (int e = make_const("itcv_sampling_fwd", B, dataset_size, &c)) return e;
  if (weighted)
    hipLaunchKernelGGL(sampling_fwd_kernel<true>, dim3(B), dim3(256), 0, S(stream), lp, prodm, logqz, lse, sjoint, B, D, c);
  else
    hipLaunchKernelGGL(sampling_fwd_kernel<false>, dim3(B), dim3(256), 0, S(stream), lp, prodm, logqz, lse, sjoint, B, D, c);
  ITCV_CHECK_LAUNCH("itcv_sampling_fwd");
  return 0;
}

int itcv_sampling_bwd(const float* g_prodm, const float* g_logqz, const float* lp, const float* lse, const float* sjoint,
                      const float* logqz, float* dlp, int B, int D, int64_t dataset_size, int weighted, void* stream) {
  ITCV_REQUIRE(g_prodm && g_logqz && lp && lse && sjoint && logqz && dlp && B > 0 && D > 0, "itcv_sampling_bwd");
  ITCV_REQUIRE(B <= 65535, "itcv_sampling_bwd(batch)");
  TcConst c;
  if (int e = make_const("itcv_sampling_bwd", B, dataset_size, &c)) return e;
  if (weighted)
    hipLaunchKernelGGL(sampling_bwd_kernel<true>, dim3(B, B), dim3(256), 0, S(stream), g_prodm, g_logqz, lp, lse, sjoint,
                       logqz, dlp, B, D, c);
  else
    hipLaunchKernelGGL(sampling_bwd_kernel<false>, dim3(B, B), dim3(256), 0, S(stream), g_prodm, g_logqz, lp, lse, sjoint,
                       logqz, dlp, B, D, c);
  ITCV_CHECK_LAUNCH("itcv_sampling_bwd");
  return 0;
}

int itcv_on_off_diag(const float* x, float* diag, float* off, int m, int n, void* stream) {
  ITCV_REQUIRE(x && diag && off && m > 0 && n > 0 && (m == n || m == 1), "itcv_on_off_diag");
  hipLaunchKernelGGL(on_off_diag_kernel, dim3(ew_grid((size_t)m * n * n)), dim3(256), 0, S(stream), x, diag, off, m, n);
  ITCV_CHECK_LAUNCH("itcv_on_off_diag");
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
