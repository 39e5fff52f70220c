// Implicit-GEMM convolution / linear layers on the exact-fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain, so results track the fp32 CPU
// reference to rounding).  Replaces the ATen conv/linear forward + backward behind
// /root/reference/models.py:28-47,213,233,270,290.
//
//   forward / data-gradient:  C[M][N] = A[K][M]^T * im2col(X)[K][N]
//        M = output channels, N = B*H*W (pixel index, contiguous in NCHW),
//        K ordered TAP-MAJOR: k = tap*Cip + ci (Cip = channels padded to 16), so one 16-deep
//        K-tile is ONE filter tap x 16 consecutive channels: the im2col gather of a tile is a
//        single shifted window -- one bounds decision and one address per thread per tile, the
//        16 rows differ only by a scalar channel stride (buffer_load soffset).
//        A = packed weights (K-major, zero padded to the tile), read with unconditional float4s.
//   weight-gradient:          C[M][N] = sum_k dY[M][k] * im2col(X)[N][k]
//        M = Co, N = (tap, ci) tap-major, k = (b,h,w); split-K over k with fp32 slabs and a
//        deterministic reduce that also transposes to the OIHW layout of dW (no atomics).
//
// Padding and tile tails are zero-filled by the buffer unit: an invalid element gets a voffset
// >= 2^31 > num_records, which the hardware range check turns into 0 (no branches in the loop).
//
// Tiling: 256 threads = 4 waves; each wave owns TMxTN tiles of 32x32 accumulators (AGPRs); LDS
// tiles are K-major so every ds_read_b32 of an MFMA operand is 32 consecutive floats per
// half-wave (conflict free); operand fragments are double-buffered in registers across the
// K-steps.  Global->register->LDS software pipeline with two LDS buffers and one barrier per
// K-tile.  blockIdx -> tile mapping keeps tiles that share an im2col panel on one XCD.
#include "conv_shared.h"

namespace itcv {



constexpr uint32_t kOobBase = 0x80000000u;  // + any soffset < 2^31 stays out of range

__device__ __forceinline__ float buf_load_s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  // soffset must live in an SGPR: state the wave-uniformity explicitly or hipcc wraps the load in a
  // waterfall loop whenever register pressure parked the (loop-invariant) value in a VGPR
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, __builtin_amdgcn_readfirstlane(soff), 0));
}

__host__ __device__ inline int tile_rows_for(int M) { return M <= 32 ? 32 : (M <= 64 ? 64 : 128); }

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int B, Ci, H, W, Co;
  int Cip, Mp, N;
  int mt, nt;
  int ktiles, ktiles_per_split, cpt;  // cpt = K-tiles per tap = Cip/16
  uint32_t x_bytes;
  size_t slab_stride;
};

// One MFMA K-tile worth of work on LDS buffers As[BK][PA], Bs[BK][PB] (K-major)
template <int BK, int PA, int PB, int TM, int TN>
__device__ __forceinline__ void mfma_tile(const float* __restrict__ Ab, const float* __restrict__ Bb,
                                          f32x16 (&acc)[TM][TN]) {
  float av[2][TM], bv[2][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) av[0][i] = Ab[i * 32];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[0][j] = Bb[j * 32];
#pragma unroll
  for (int s = 0; s < BK / 2; ++s) {
    const int c = s & 1, nx = c ^ 1;
    if (s + 1 < BK / 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[nx][i] = Ab[(2 * s + 2) * PA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[nx][j] = Bb[(2 * s + 2) * PB + j * 32];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
  }
}

template <int KS, int BM, int BN, int WM, int WN, bool UP2, bool CI_TAIL>
__global__ __launch_bounds__(WM* WN * 64) void conv_fwd_kernel(ConvArgs a) {
  constexpr int NT = WM * WN * 64, BK = 16, P = KS / 2;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int PA = BM + 4, PB = BN + 4;
  static_assert(NT % BN == 0 && BN % 64 == 0, "B gather rows must be wave-uniform");
  __shared__ __attribute__((aligned(16))) float As[2][BK * PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * PB];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, half = lane >> 5;
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int tile_m = q % a.mt, tile_n = (q / a.mt) * 8 + xcd;
  if (tile_n >= a.nt) return;
  const int sk = blockIdx.y;
  const int kt0 = sk * a.ktiles_per_split;
  const int kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? W / 2 : W, HWs = Hs * Ws;

  // ---- per-thread im2col column: pixel n -> (image, h, w), tap validity mask ------------
  constexpr int BROWS = NT / BN, BL = BK / BROWS;
  const int nl = t % BN;
  const int kr0 = __builtin_amdgcn_readfirstlane(t / BN);  // wave-uniform: row offsets live in SGPRs
  const int n = n0 + nl;
  const bool nvalid = n < a.N;
  int bi = 0, h = 0, w = 0;
  if (nvalid) {
    bi = n / HW;
    const int hw = n - bi * HW;
    h = hw / W;
    w = hw - h * W;
  }
  uint32_t tapmask = 0;
#pragma unroll
  for (int tap = 0; tap < KS * KS; ++tap) {
    const int dh = tap / KS - P, dw = tap % KS - P;
    if (nvalid && (unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)W) tapmask |= 1u << tap;
  }
  const int img_base = bi * a.Ci * HWs;  // elements
  const int tb = img_base + (UP2 ? 0 : h * W + w);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const uint32_t row_stride4 = (uint32_t)(BROWS * HWs) * 4u, row0_4 = (uint32_t)(kr0 * HWs) * 4u;

  constexpr int AV = BK * BM / 4, AL = (AV + NT - 1) / NT, AROWS = NT / (BM / 4);
  // thread's first A element: row t/(BM/4), 4 columns at (t%(BM/4))*4; further loads are AROWS rows below
  const float* pa0 = a.wp + (size_t)(t / (BM / 4)) * a.Mp + m0 + (t % (BM / 4)) * 4;
  const size_t a_row_step = (size_t)AROWS * a.Mp;
  f32x4 areg[AL];
  float breg[BL];

  auto load_tile = [&](int kt, int tap, int cib) {
    const float* pk = pa0 + (size_t)kt * BK * a.Mp;
#pragma unroll
    for (int i = 0; i < AL; ++i)
      if (AV % NT == 0 || t + i * NT < AV) areg[i] = *reinterpret_cast<const f32x4*>(pk + i * a_row_step);
    const int dh = tap / KS - P, dw = tap - (tap / KS) * KS - P;  // scalar
    const bool valid = (tapmask >> tap) & 1u;
    int off;
    if (UP2)
      off = tb + ((h + dh) >> 1) * Ws + ((w + dw) >> 1) + cib * 16 * HWs;
    else
      off = tb + dh * W + dw + cib * 16 * HWs;
    const uint32_t voff = valid ? (uint32_t)off * 4u : kOobBase;
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      uint32_t v = voff;
      if (CI_TAIL) v = (cib * 16 + kr0 + i * BROWS < a.Ci) ? voff : kOobBase;  // padded channel rows (wave-uniform)
      breg[i] = buf_load_s(rx, v, row0_4 + (uint32_t)i * row_stride4);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int idx = t + i * NT;
      if (AV % NT == 0 || idx < AV)
        *reinterpret_cast<f32x4*>(&As[buf][(idx / (BM / 4)) * PA + (idx % (BM / 4)) * 4]) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) Bs[buf][(kr0 + i * BROWS) * PB + nl] = breg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt0 < kt1) {
    // K-tiles run channel-block outer, tap inner: the KS*KS shifted windows of one channel block are
    // gathered back to back and hit the vector L1 instead of re-reading the planes from L2 per tap
    constexpr int KKc = KS * KS;
    int cib = kt0 / KKc, tap = kt0 - cib * KKc;
    load_tile(kt0, tap, cib);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (++tap == KKc) tap = 0, ++cib;
      if (more) load_tile(kt + 1, tap, cib);
      mfma_tile<BK, PA, PB, TM, TN>(&As[cur][half * PA + wm * WTM + l31], &Bs[cur][half * PB + wn * WTN + l31], acc);
      if (more) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: C[m][n] -> y[b][m][h][w] (+bias); 32 consecutive pixels per half-wave -----
  float* out = a.y + (size_t)sk * a.slab_stride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nn = n0 + wn * WTN + j * 32 + l31;
    if (nn >= a.N) continue;
    const int b2 = nn / HW, hw2 = nn - b2 * HW;
    const size_t base = (size_t)b2 * a.Co * HW + hw2;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < a.Co) {
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[m];
          out[base + (size_t)m * HW] = v;
        }
      }
    }
  }
}

// y[idx] = sum_s slab[s][idx] (+ bias[channel])
__global__ void splitk_reduce_fwd(const float* __restrict__ slab, const float* __restrict__ bias,
                                  float* __restrict__ y, size_t total, size_t stride, int splits, int HW,
                                  int Co) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = fold_strided(0.f, slab + i, stride, splits);
    if (bias) s += bias[(i / HW) % Co];
    y[i] = s;
  }
}

// ------------------------------------------------------------------ small dense GEMMs (nn.Linear)
// The fc layers (models.py:233,270: 8192 <-> 2*zdim / zdim <-> 8192 at batch 64) are skinny GEMMs whose cost is
// reading the weight once.  C[M][N] = sum_k A(m,k) * B(k,n) with arbitrary element strides covers forward (x W^T),
// data-gradient (dy W) and weight-gradient (dy^T x); exact fp32 MFMA in every conv-math mode; 64x64 tiles, split-K
// over blockIdx.y into fp32 slabs (fixed-order reduce) when there are few tiles.
struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;   // indexed by n; only applied when the kernel writes C itself (no split-K)
  float* C;
  int M, N, K;
  long long sam, sak, sbk, sbn;
  int mt, nt, ktiles, ktiles_per_split;
  size_t slab_stride;
  int accumulate;
};

template <bool A_KC, bool B_KC>   // is the operand's k index the contiguous one in memory?
__global__ __launch_bounds__(256) void gemm64_kernel(GemmArgs a) {
  constexpr int BM = 64, BN = 64, BK = 32, PA = BM + 1, PB = BN + 1;
  __shared__ float As[2][BK * PA];
  __shared__ float Bs[2][BK * PB];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, half = lane >> 5;
  const int tile_m = blockIdx.x % a.mt, tile_n = blockIdx.x / a.mt, sk = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt0 = sk * a.ktiles_per_split, kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);

  float areg[8], breg[8];
  // Clamped (always valid) addresses, all 16 loads of a K-tile issued as one group, bounds applied afterwards by selects:
  // under their bounds checks the loads were 16 branches with a full wait each -- 16 dependent round trips per K-tile.
  auto load_tile = [&](int kt) {
    const int kb = kt * BK;
    const float* pa[8];
    const float* pb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = A_KC ? (t & 31) : (t >> 6) + 4 * i, m = A_KC ? (t >> 5) + 8 * i : (t & 63);
      pa[i] = a.A + (long long)min(m0 + m, a.M - 1) * a.sam + (long long)min(kb + k, a.K - 1) * a.sak;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = B_KC ? (t & 31) : (t >> 6) + 4 * i, n = B_KC ? (t >> 5) + 8 * i : (t & 63);
      pb[i] = a.B + (long long)min(kb + k, a.K - 1) * a.sbk + (long long)min(n0 + n, a.N - 1) * a.sbn;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) areg[i] = *pa[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) breg[i] = *pb[i];
    __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);     // the 16 VMEM reads together, ahead of everything else
  };
  auto store_tile = [&](int buf, int kt) {                  // (the bounds are applied here, when the values are first used)
    const int kb = kt * BK;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = A_KC ? (t & 31) : (t >> 6) + 4 * i, m = A_KC ? (t >> 5) + 8 * i : (t & 63);
      As[buf][k * PA + m] = (m0 + m < a.M && kb + k < a.K) ? areg[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = B_KC ? (t & 31) : (t >> 6) + 4 * i, n = B_KC ? (t >> 5) + 8 * i : (t & 63);
      Bs[buf][k * PB + n] = (n0 + n < a.N && kb + k < a.K) ? breg[i] : 0.f;
    }
  };
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0, kt0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (more) load_tile(kt + 1);
      mfma_tile<BK, PA, PB, 1, 1>(&As[cur][half * PA + wm * 32 + l31], &Bs[cur][half * PB + wn * 32 + l31], acc);
      if (more) store_tile(cur ^ 1, kt + 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  const int n = n0 + wn * 32 + l31;
  if (n >= a.N) return;
  float* out = a.C + (size_t)sk * a.slab_stride;
  const float bv = (a.bias && !a.slab_stride) ? a.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m < a.M) {
      float* q = out + (size_t)m * a.N + n;
      const float v = acc[0][0][r] + bv;
      *q = a.accumulate ? *q + v : v;
    }
  }
}

// C (+)= bias[n] + sum_s slab[s]
__global__ void gemm64_reduce(const float* __restrict__ slab, const float* __restrict__ bias, float* __restrict__ C,
                              size_t total, int N, int splits, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = fold_strided(0.f, slab + i, total, splits);
    if (bias) s += bias[i % N];
    C[i] = accumulate ? C[i] + s : s;
  }
}

// wp[k][m] with k = ((c/16)*KK + tap)*16 + c%16 (16-channel block outer, tap inner); for_dgrad swaps the
// channel roles and flips the taps
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int KK,
                                   int for_dgrad, int C, int M, int Cip, int Mp) {
  const size_t total = (size_t)KK * Cip * Mp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / Mp), m = (int)(i - (size_t)k * Mp);
    const int kt = k >> 4, cib = kt / KK, tap = kt - cib * KK, c = cib * 16 + (k & 15);
    float v = 0.f;
    if (c < C && m < M)
      v = for_dgrad ? w[((size_t)c * Ci + m) * KK + (KK - 1 - tap)] : w[((size_t)m * Ci + c) * KK + tap];
    wp[i] = v;
  }
}

// ------------------------------------------------------------------------------ split-bf16 forward
// Throughput variant of conv_fwd_kernel on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the
// per-instruction work of the fp32 MFMA).  Every fp32 operand is split into NS bf16 planes
// (x = x0 + x1 (+ x2), each plane the bf16 rounding of the remaining residual) and the product is
// accumulated in fp32 over the plane pairs whose weight is above fp32 rounding:
//   NS = 2 ("bf16x3"): x0y0 + x0y1 + x1y0            -> ~2^-16 relative per product
//   NS = 3 ("bf16x6"): + x0y2 + x2y0 + x1y1           -> ~2^-23, i.e. fp32-class
// Weights are split once at pack time; the gathered im2col operand is split on the way into LDS.
// Same tap-major K order (32 channels of one tap per K-tile), same tile/XCD mapping and epilogue as
// the fp32 kernel.  LDS holds 16-byte chunks [plane][k/8][row] so that every MFMA fragment is one
// conflict-free ds_read_b128.

struct ConvArgsB {
  const float* x;
  const u32x4* wp;
  const float* bias;
  float* y;
  int B, Ci, H, W, Co;
  int Mp, N;
  int mt, nt;
  int ktiles, ktiles_per_split, cpt;  // cpt = K-tiles per tap = Cip/32
  uint32_t x_bytes;
  size_t slab_stride;
#ifdef ITCV_DIAG
  int ablate;  // diagnostic only (ITCV_ABLATE): 1 no gathers, 2 no split, 4 no weight DMA, 8 no MFMA, 16 no B stores
#endif
};

// Wave-specialised: 512 threads = 4 consumer waves (one per SIMD) that only read fragments and issue MFMAs + 4
// producer waves that only gather / split / store the next tiles.  (The first form of this kernel had every wave do
// both: two identical 4-wave blocks sharing a SIMD ran in lockstep, collided on the matrix pipe and then idled
// together -- measured 36 % MFMA busy, 42 % issue stalls; removed.)  A producer wave next to a consumer wave uses the
// VALU / VMEM / LDS-store paths while the matrix pipe of the same SIMD stays fed.
template <int KS, int BM, int BN, int WM, int WN, bool UP2, int NS>
__global__ __launch_bounds__(512) void conv_fwd_bf16s_ws_kernel(ConvArgsB a) {
  constexpr int NP = 256, BK = 32, KC = BK / 8, P = KS / 2, KKc = KS * KS;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  static_assert(WM * WN == 4 && (BN == 128 || BN == 256) && BM % 64 == 0, "tile / wave layout");
  constexpr int ASZ = NS * KC * BM, BSZ = NS * KC * BN;
  __shared__ u32x4 As[2 * ASZ];
  __shared__ u32x4 Bs[2 * BSZ];

  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int tile_m = q % a.mt, tile_n = (q / a.mt) * 8 + xcd;
  if (tile_n >= a.nt) return;
  const int sk = blockIdx.y;
  const int kt0 = sk * a.ktiles_per_split;
  const int kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int H = a.H, W = a.W, HW = H * W;
  const int nk = kt1 - kt0;
  if (nk <= 0) return;

  if (wid >= 4) {
    // ------------------------------------------------------------------ producers
    const int p = t - 256;
    const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? W / 2 : W, HWs = Hs * Ws;
    constexpr int BROWS = NP / BN, RPT = BK / BROWS;
    const int nl = p % BN;
    const int kg = __builtin_amdgcn_readfirstlane(p / BN);
    const int n = n0 + nl;
    const bool nvalid = n < a.N;
    int bi = 0, h = 0, w = 0;
    if (nvalid) {
      bi = n / HW;
      const int hw = n - bi * HW;
      h = hw / W;
      w = hw - h * W;
    }
    uint32_t tapmask = 0;
#pragma unroll
    for (int tap = 0; tap < KKc; ++tap) {
      const int dh = tap / KS - P, dw = tap % KS - P;
      if (nvalid && (unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)W) tapmask |= 1u << tap;
    }
    const int tb = bi * a.Ci * HWs + (UP2 ? 0 : h * W + w);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const uint32_t hw4 = (uint32_t)HWs * 4u, row0_4 = (uint32_t)(kg * RPT) * hw4;
    constexpr int DEPTH = 3;                 // gathers run DEPTH-1 tiles ahead of the tile being stored
    float breg[DEPTH][RPT];

    auto load_B = [&](float (&dst)[RPT], int tap, int cib) {
      const int dh = tap / KS - P, dw = tap - (tap / KS) * KS - P;
      const bool valid = (tapmask >> tap) & 1u;
      int off;
      if (UP2)
        off = tb + ((h + dh) >> 1) * Ws + ((w + dw) >> 1) + cib * BK * HWs;
      else
        off = tb + dh * W + dw + cib * BK * HWs;
      const uint32_t voff = valid ? (uint32_t)off * 4u : kOobBase;
      if ITCV_ABL(a, 1) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) dst[i] = __builtin_bit_cast(float, voff + i);
        return;
      }
#pragma unroll
      for (int i = 0; i < RPT; ++i) dst[i] = buf_load_s(rx, voff, row0_4 + (uint32_t)i * hw4);
    };
    auto store_B = [&](const float (&src)[RPT], int buf) {
      u32x4* Bd = Bs + buf * BSZ;
#pragma unroll
      for (int c = 0; c < RPT / 8; ++c) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[c * 8 + j];
        u32x4 pl[NS];
        if ITCV_ABL(a, 2) {
#pragma unroll
          for (int pp = 0; pp < NS; ++pp)
            pl[pp] = u32x4{__builtin_bit_cast(unsigned, v[pp]), __builtin_bit_cast(unsigned, v[2 + pp]),
                           __builtin_bit_cast(unsigned, v[4 + pp]), __builtin_bit_cast(unsigned, v[6 + (pp & 1)])};
        } else {
          split8<NS>(v, pl);
        }
        const int kc = kg * (RPT / 8) + c;
        if ITCV_ABL(a, 16) {
          asm volatile("" ::"v"(pl[0][0]), "v"(pl[NS - 1][3]));
          continue;
        }
#pragma unroll
        for (int pp = 0; pp < NS; ++pp) Bd[(pp * KC + kc) * BN + nl] = pl[pp];
      }
    };
    int cib = kt0 / KKc, tap = kt0 - cib * KKc;
    auto advance = [&]() {
      if (++tap == KKc) tap = 0, ++cib;
    };
    // prologue: tile 0 -> LDS[0]; tiles 1 .. DEPTH-1 in flight (past the end the tap mask / hardware range
    // check turn the gathers into zeros, so no tail branches)
    load_B(breg[0], tap, cib);
    store_B(breg[0], 0);
#pragma unroll
    for (int d = 1; d < DEPTH; ++d) {
      advance();
      load_B(breg[d % DEPTH], tap, cib);
    }
    __syncthreads();
    int cur = 0;
    // iteration i: store tile i+1 (ring slot (i+1)%DEPTH), DMA weight tile i+1, gather tile i+DEPTH into the
    // slot just freed; unrolled by DEPTH so that every ring slot is a fixed register set
    for (int i0 = 0; i0 < nk; i0 += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int i = i0 + u;
        if (i < nk) {
          store_B(breg[(u + 1) % DEPTH], cur ^ 1);   // tile i+1 lives in ring slot (i+1) % DEPTH
          advance();
          load_B(breg[u], tap, cib);                  // tile i+DEPTH -> slot i % DEPTH (tile i's, now free)
          // publish the stored tile: only the ds_writes must have landed; the gathers stay in flight
          // across the barrier (vmcnt is per wave and the weight DMA lives in the consumer waves' queue)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          cur ^= 1;
        }
      }
    }
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, half = lane >> 5;
  // weight tiles (pre-split, already in LDS order) go global -> LDS by LDS-DMA from the consumer waves:
  // their vmcnt queue holds nothing else, so draining it at the barrier costs the gathers nothing
  constexpr int AL = ASZ / NP;
  static_assert(ASZ % NP == 0, "A tile chunks must divide evenly");
  auto dma_A = [&](int kt, int buf) {
    if ITCV_ABL(a, 4) return;
    const u32x4* wt = a.wp + (size_t)kt * NS * KC * a.Mp;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int idx = t + i * NP, pk = idx / BM, ml = idx - pk * BM;
      const int wave_chunk = __builtin_amdgcn_readfirstlane((t & ~63) + i * NP);
      lds_dma16(wt + (size_t)pk * a.Mp + m0 + ml, lds_addr(As + buf * ASZ + wave_chunk));
    }
  };
  dma_A(kt0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < nk; ++it) {
    dma_A(min(kt0 + it + 1, kt1 - 1), cur ^ 1);
    const u32x4* Ab = As + cur * ASZ;
    const u32x4* Bb = Bs + cur * BSZ;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if ITCV_ABL(a, 8) break;
      const int kc = ks * 2 + half;
      bf16x8 af[NS][TM], bfr[NS][TN];
#pragma unroll
      for (int pp = 0; pp < NS; ++pp) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[pp][i] = __builtin_bit_cast(bf16x8, Ab[(pp * KC + kc) * BM + wm * WTM + i * 32 + l31]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[pp][j] = __builtin_bit_cast(bf16x8, Bb[(pp * KC + kc) * BN + wn * WTN + j * 32 + l31]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];
          if constexpr (NS == 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[2][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bfr[0][j], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight DMA of tile it+1 has landed
    __syncthreads();
    cur ^= 1;
  }

  float* out = a.y + (size_t)sk * a.slab_stride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nn = n0 + wn * WTN + j * 32 + l31;
    if (nn >= a.N) continue;
    const int b2 = nn / HW, hw2 = nn - b2 * HW;
    const size_t base = (size_t)b2 * a.Co * HW + hw2;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < a.Co) {
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[m];
          if (ITCV_ABL(a, 32) && v != 12345.f) continue;
          out[base + (size_t)m * HW] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- pre-split ("planes") operand
// The im2col operand can also arrive ALREADY split: planes[p][b][c/8][h][w] = one 16-byte chunk holding
// bf16 plane p of channels c..c+7 of one pixel -- exactly one MFMA fragment row.  The producing pass
// (BatchNorm apply / its backward, or itcv_split_planes) writes it next to the fp32 tensor; the conv
// kernel then has no gather, no conversion and no ds_write at all: both operands go global -> LDS by
// LDS-DMA, one 1-KiB piece per wave instruction, issued by four loader waves that run NSTAGE-1 K-tiles
// ahead of the four MFMA waves (counted vmcnt + one barrier per K-tile).  Padding taps read a zero chunk.
// Arithmetic (split, product order, K order) is identical to conv_fwd_bf16s_*: results match bit for bit.
__device__ u32x4 g_zero_chunk = {0u, 0u, 0u, 0u};

struct ConvArgsP {
  const u32x4* xp;
  const u32x4* wp;
  const float* bias;
  float* y;
  int B, Ci, H, W, Co;
  int Mp, N;
  int mt, nt;
  int ktiles, ktiles_per_split;
  size_t slab_stride;
  size_t plane_stride;   // chunks per plane = B * (Ci/8) * Hs * Ws
  const ScaleRec* xscale;   // fp16 planes: the input's scale record
#ifdef ITCV_DIAG
  int ablate;            // diagnostic only (ITCV_ABLATE): 1 no B pieces, 4 no A pieces, 8 no MFMA
#endif
};

// WM x WN MFMA waves (4, or 8 = two per SIMD: one fills the other's barrier / LDS-latency bubbles) + 4 loader waves.
template <int KS, int BM, int BN, int WM, int WN, bool UP2, int NS, int NSTAGE, bool M16 = false, bool F16 = false>
__global__ __launch_bounds__(64 * (WM * WN + 4)) void conv_fwd_bf16p_kernel(ConvArgsP a) {
  static_assert(!M16 || NS == 2, "the 16x16x32 form is written for two planes");
  static_assert(!F16 || M16, "the fp16 planes form uses the 16x16x32 products");
  constexpr int BK = 32, KC = BK / 8, P = KS / 2, KKc = KS * KS, NMW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  static_assert((NMW == 4 || NMW == 8) && (BN == 128 || BN == 256) && (BM == 64 || BM == 128), "tile / wave layout");
  constexpr int ASZ = NS * KC * BM, BSZ = NS * KC * BN, SSZ = ASZ + BSZ;   // 16-byte chunks per stage
  extern __shared__ u32x4 smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int tile_m = q % a.mt, tile_n = (q / a.mt) * 8 + xcd;
  if (tile_n >= a.nt) return;
  const int sk = blockIdx.y;
  const int kt0 = sk * a.ktiles_per_split;
  const int kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int H = a.H, W = a.W, HW = H * W;
  const int nk = kt1 - kt0;
  if (nk <= 0) return;
  const uint32_t smem_base = lds_addr(smem);

  if (wid >= NMW) {
    // ------------------------------------------------------------------ loaders
    const int lw = wid - NMW;
    constexpr int NPA = BM / 64, NPB = BN / 64;            // 1-KiB pieces per (plane, k-chunk) row
    constexpr int PA = NS * KC * NPA / 4, PB = NS * KC * NPB / 4, PT = PA + PB;   // pieces per wave per tile
    static_assert((NS * KC * NPA) % 4 == 0 && (NPB == 2 || NPB == 4), "piece split");
    const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? W / 2 : W, HWs = Hs * Ws;
    const int C8 = a.Ci >> 3;
    // this wave's pixel column of the B tile is the same for all of its pieces
    const int nlc = lw % NPB;
    const int n = n0 + nlc * 64 + lane;
    const bool nvalid = n < a.N;
    int bi = 0, h = 0, w = 0;
    if (nvalid) {
      bi = n / HW;
      const int hw = n - bi * HW;
      h = hw / W;
      w = hw - h * W;
    }
    uint32_t tapmask = 0;
#pragma unroll
    for (int tap = 0; tap < KKc; ++tap) {
      const int dh = tap / KS - P, dw = tap % KS - P;
      if (nvalid && (unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)W) tapmask |= 1u << tap;
    }
    const long long cb = (long long)bi * C8 * HWs + (UP2 ? 0 : h * W + w);
    const u32x4* zero = &g_zero_chunk;

    auto issue = [&](int kt, int slot) {
      const int cib = kt / KKc, tap = kt - cib * KKc;
      const int dh = tap / KS - P, dw = tap - (tap / KS) * KS - P;
      const bool valid = (tapmask >> tap) & 1u;
      const int shift = UP2 ? ((h + dh) >> 1) * Ws + ((w + dw) >> 1) : dh * W + dw;
      const u32x4* src0 = a.xp + (cb + (long long)(cib * KC) * HWs + shift);
      const uint32_t sbase = smem_base + (uint32_t)(slot * SSZ) * 16u;
      const u32x4* wt = a.wp + (size_t)kt * NS * KC * a.Mp + m0 + lane;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        if ITCV_ABL(a, 4) break;
        const int piece = j * 4 + lw, mlc = piece % NPA, pk = piece / NPA;
        lds_dma16(wt + (size_t)pk * a.Mp + mlc * 64, sbase + (uint32_t)(pk * BM + mlc * 64) * 16u);
      }
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        if ITCV_ABL(a, 1) break;
        const int piece = j * 4 + lw, pk = piece / NPB, pl = pk / KC, kc = pk - pl * KC;
        const u32x4* src = src0 + ((size_t)pl * a.plane_stride + (size_t)kc * HWs);
        lds_dma16(valid ? src : zero, sbase + (uint32_t)(ASZ + pk * BN + nlc * 64) * 16u);
      }
    };
    // tiles 0 .. NSTAGE-2 in flight, tile 0 landed
#pragma unroll
    for (int d = 0; d < NSTAGE - 1; ++d) issue(min(kt0 + d, kt1 - 1), d);
    wait_vmcnt<PT*(NSTAGE - 2)>();
    __builtin_amdgcn_s_barrier();
    int slot = NSTAGE - 1;
    for (int i = 0; i < nk; ++i) {
      issue(min(kt0 + i + NSTAGE - 1, kt1 - 1), slot);   // into the slot tile i-1 has just left
      if (++slot == NSTAGE) slot = 0;
      wait_vmcnt<PT*(NSTAGE - 2)>();                      // tile i+1 has landed
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // -------------------------------------------------------------------- MFMA waves
  typedef BandMfma<M16, F16> MM;
  typedef typename MM::acc_t acc_t;
  constexpr int TS = MM::TS, TMx = WTM / TS, TNx = WTN / TS;
  const float oscale = F16 ? inv_scale_of(a.xscale) * (1.f / (float)(1 << kWeightScaleLog2)) : 1.f;   // exact: powers of two
  const int wm = wid / WN, wn = wid % WN, lr = lane & (TS - 1), kq = lane / TS;
  acc_t acc[TMx][TNx];
#pragma unroll
  for (int i = 0; i < TMx; ++i)
#pragma unroll
    for (int j = 0; j < TNx; ++j)
#pragma unroll
      for (int r = 0; r < MM::NR; ++r) acc[i][j][r] = 0.f;
  __builtin_amdgcn_s_barrier();
  int slot = 0;
  const long long dbg_c0 = ITCV_ABL(a, 64) ? clock64() : 0, dbg_w0 = ITCV_ABL(a, 64) ? wall_clock64() : 0;
  for (int it = 0; it < nk; ++it) {
    const u32x4* Ab = smem + slot * SSZ;
    const u32x4* Bb = Ab + ASZ;
    if (!ITCV_ABL(a, 8)) {
      if constexpr (!M16) {
        // both k-steps' fragments requested up front, reads interleaved behind the MFMAs (see conv_fwd_bf16p2_kernel)
        bf16x8 af[2][NS][TMx], bfr[2][NS][TNx];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
          const int kc = ks * 2 + kq;
#pragma unroll
          for (int pp = 0; pp < NS; ++pp) {
#pragma unroll
            for (int i = 0; i < TMx; ++i)
              af[ks][pp][i] = __builtin_bit_cast(bf16x8, Ab[(pp * KC + kc) * BM + wm * WTM + i * 32 + lr]);
#pragma unroll
            for (int j = 0; j < TNx; ++j)
              bfr[ks][pp][j] = __builtin_bit_cast(bf16x8, Bb[(pp * KC + kc) * BN + wn * WTN + j * 32 + lr]);
          }
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
          for (int i = 0; i < TMx; ++i)
#pragma unroll
            for (int j = 0; j < TNx; ++j) {
              acc_t c = acc[i][j];
              if constexpr (NS == 3) {
                c = MM::mma(af[ks][1][i], bfr[ks][1][j], c);
                c = MM::mma(af[ks][0][i], bfr[ks][2][j], c);
                c = MM::mma(af[ks][2][i], bfr[ks][0][j], c);
              }
              c = MM::mma(af[ks][0][i], bfr[ks][1][j], c);
              c = MM::mma(af[ks][1][i], bfr[ks][0][j], c);
              c = MM::mma(af[ks][0][i], bfr[ks][0][j], c);
              acc[i][j] = c;
            }
        if constexpr (NS == 2) {
          constexpr int RD = NS * (TMx + TNx), MF = TMx * TNx * 3;
          __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
          for (int r = 0; r < RD; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF - RD, 0);
        }
      } else {
        // 16x16x32: one MFMA per (tile, product) covers the 32-channel K-tile, lane group kq holds chunk kq
        bf16x8 af[NS][TMx], bfr[NS][TNx];
#pragma unroll
        for (int pp = 0; pp < NS; ++pp)
#pragma unroll
          for (int j = 0; j < TNx; ++j) bfr[pp][j] = __builtin_bit_cast(bf16x8, Bb[(pp * KC + kq) * BN + wn * WTN + j * 16 + lr]);
#pragma unroll
        for (int i = 0; i < TMx; ++i)
#pragma unroll
          for (int pp = 0; pp < NS; ++pp) af[pp][i] = __builtin_bit_cast(bf16x8, Ab[(pp * KC + kq) * BM + wm * WTM + i * 16 + lr]);
#pragma unroll
        for (int i = 0; i < TMx; ++i)
#pragma unroll
          for (int j = 0; j < TNx; ++j) {
            acc_t c = acc[i][j];
            c = MM::mma(af[0][i], bfr[1][j], c);
            c = MM::mma(af[1][i], bfr[0][j], c);
            c = MM::mma(af[0][i], bfr[0][j], c);
            acc[i][j] = c;
          }
        {
          constexpr int RD0 = NS * TNx + NS * (TMx / 2), RD1 = NS * (TMx - TMx / 2), MF = TMx * TNx * 3;
          __builtin_amdgcn_sched_group_barrier(0x100, RD0, 0);
#pragma unroll
          for (int r = 0; r < RD1; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, MF - RD1, 0);
        }
      }
    }
    if (++slot == NSTAGE) slot = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's fragments are in registers
    __builtin_amdgcn_s_barrier();
  }
  if (ITCV_ABL(a, 64) && t == 0 && (bid == 0 || bid == 300)) {   // diagnostic: main-loop shader cycles / 100 MHz ticks
    a.y[(bid ? 2 : 0) + 0] = (float)(clock64() - dbg_c0);
    a.y[(bid ? 2 : 0) + 1] = (float)(wall_clock64() - dbg_w0);
    return;
  }

  float* out = a.y + (size_t)sk * a.slab_stride;
#pragma unroll
  for (int j = 0; j < TNx; ++j) {
    const int nn = n0 + wn * WTN + j * TS + lr;
    if (nn >= a.N) continue;
    const int b2 = nn / HW, hw2 = nn - b2 * HW;
    const size_t base = (size_t)b2 * a.Co * HW + hw2;
#pragma unroll
    for (int i = 0; i < TMx; ++i) {
#pragma unroll
      for (int r = 0; r < MM::NR; ++r) {
        const int m = m0 + wm * WTM + i * TS + MM::row(r, kq);
        if (m < a.Co) {
          float v = acc[i][j][r];
          if (F16) v *= oscale;
          if (a.bias) v += a.bias[m];
          out[base + (size_t)m * HW] = v;
        }
      }
    }
  }
}

// fp32 NCHW -> planes[p][b][c/8][hw] (C % 8 == 0); one thread per chunk, coalesced over hw.
// F16: fp16 hi/lo planes of S*x; S comes from `amax` (kAbsmaxParts block maxima of |x|, itcv_absmax) or is 1 when amax is
// null (activations: O(1) values, see common.h); the record {S, 1/S} is written behind the second plane.
template <int NS, bool F16 = false>
__global__ void split_planes_kernel(const float* __restrict__ x, u32x4* __restrict__ planes, int B, int C8, int HW,
                                    const float* __restrict__ amax) {
  const size_t total = (size_t)B * C8 * HW;
  float scale = 1.f;
  if constexpr (F16) {
    if (amax) scale = scale_for_bound(wave_absmax_of(amax, kAbsmaxParts));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ScaleRec* rec = reinterpret_cast<ScaleRec*>(planes + (size_t)NS * total);
      rec->scale = scale, rec->inv = 1.f / scale, rec->pad[0] = rec->pad[1] = 0.f;
    }
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bc = i / HW;
    const int hw = (int)(i - bc * HW);
    const float* src = x + bc * 8 * HW + hw;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * HW];
    u32x4 pl[NS];
    split8<NS, F16>(v, pl, scale);
#pragma unroll
    for (int p = 0; p < NS; ++p) planes[(size_t)p * total + i] = pl[p];
  }
}

// out[blockIdx.x] = max |x| over the block's grid-stride share (kAbsmaxParts blocks; NaNs are ignored by fmaxf: they
// travel through the conversion itself)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// wp[kt = cib*KK + tap][plane][kc][Mp] 16-byte chunks of 8 bf16: channels cib*32 + kc*8 + j
// One block packs a tile of 32 rows (m) x 32 reduction channels (one cib) x all KK taps.  The source is read along
// its contiguous direction -- forward: the (c, tap) run of row m; data-gradient: the (m, tap) run of channel c -- into
// LDS, then every thread emits whole 16-byte chunks (8 channels of one (tap, m)) with m fastest across the lanes.
// Tiles of a layer: (Mp / 32) * cpt, tile = mt32 * cpt + cib.
template <int NS, int KK, bool F16 = false>
__device__ __forceinline__ void pack_weight_bf16s_body(const float* __restrict__ w, u32x4* __restrict__ wp, int Ci,
                                                       int for_dgrad, int C, int M, int cpt, int Mp, int tile) {
  constexpr int RUN = 32 * KK, PITCH = RUN + 1;      // odd pitch: the column reads below are conflict free
  constexpr int ITER = 32 * RUN / 256;               // source elements per thread: all loads are issued before the first use
  __shared__ float sw[32 * PITCH];
  const int mt32 = tile / cpt, cib = tile - mt32 * cpt;
  const int m0 = mt32 * 32, c0 = cib * 32;
  float ld[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = it * 256 + threadIdx.x, o = i / RUN, r = i - o * RUN;   // outer row, position inside its contiguous run
    float v = 0.f;
    if (for_dgrad) {                                 // outer = channel c0 + o, run = (m0 .. m0+31, tap)
      const int c = c0 + o, m = m0 + r / KK;
      if (c < C && m < M) v = w[((size_t)c * Ci + m0) * KK + r];
    } else {                                         // outer = row m0 + o, run = (c0 .. c0+31, tap)
      const int m = m0 + o, c = c0 + r / KK;
      if (m < M && c < C) v = w[((size_t)m * Ci + c0) * KK + r];
    }
    ld[it] = v;
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = it * 256 + threadIdx.x, o = i / RUN, r = i - o * RUN;
    sw[o * PITCH + r] = ld[it];
  }
  __syncthreads();
#pragma unroll
  for (int i0 = 0; i0 < KK * 128; i0 += 256) {
    const int i = i0 + threadIdx.x;
    if (i >= KK * 128) break;
    const int ml = i & 31, kc = (i >> 5) & 3, tap = i >> 7;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cl = kc * 8 + j;
      v[j] = for_dgrad ? sw[cl * PITCH + ml * KK + (KK - 1 - tap)] : sw[ml * PITCH + cl * KK + tap];
    }
    u32x4 pl[NS];
    split8<NS, F16>(v, pl, F16 ? (float)(1 << kWeightScaleLog2) : 1.f);
    const int kt = cib * KK + tap;
#pragma unroll
    for (int p = 0; p < NS; ++p) wp[(((size_t)kt * NS + p) * 4 + kc) * Mp + m0 + ml] = pl[p];
  }
}

template <int NS, bool F16 = false>
__global__ __launch_bounds__(256) void pack_weight_bf16s_kernel(const float* __restrict__ w, u32x4* __restrict__ wp, int Co,
                                                                int Ci, int KK, int for_dgrad, int C, int M, int cpt, int Mp) {
  if (KK == 9) pack_weight_bf16s_body<NS, 9, F16>(w, wp, Ci, for_dgrad, C, M, cpt, Mp, blockIdx.x);
  else pack_weight_bf16s_body<NS, 1, F16>(w, wp, Ci, for_dgrad, C, M, cpt, Mp, blockIdx.x);
}

// Many layers in one launch: a device-resident table of descriptors, every layer owns a contiguous block range.
struct PackDesc {
  const float* w;
  u32x4* wp;
  int Ci, KK, for_dgrad, C, M, cpt, Mp;
  int block0, nblocks;
  int pad_;
};
static_assert(sizeof(PackDesc) == 56, "PackDesc layout is part of the ABI (itcv_pack_desc_bytes)");

template <int NS, bool F16 = false>
__global__ __launch_bounds__(256) void pack_weights_bf16s_table_kernel(const PackDesc* __restrict__ tab, int n) {
  // the block's layer = the last descriptor whose first block is <= blockIdx.x: all descriptors are looked at in
  // parallel (a serial walk costs one memory round trip per layer)
  __shared__ int s_cnt[4];
  int mine = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mine += tab[i].block0 <= (int)blockIdx.x ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = mine;
  __syncthreads();
  const int e = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3] - 1;
  __syncthreads();
  const PackDesc d = tab[e];
  if (d.KK == 9)
    pack_weight_bf16s_body<NS, 9, F16>(d.w, d.wp, d.Ci, d.for_dgrad, d.C, d.M, d.cpt, d.Mp, blockIdx.x - d.block0);
  else
    pack_weight_bf16s_body<NS, 1, F16>(d.w, d.wp, d.Ci, d.for_dgrad, d.C, d.M, d.cpt, d.Mp, blockIdx.x - d.block0);
}

// ------------------------------------------------------------------------------ wgrad
struct WgradArgs {
  const float* x;
  const float* dy;
  float* out;
  int B, Ci, H, W, Co;
  int Cip, Np, Ktot;  // Np = padded slab row length (nt*128)
  int mt, nt, tiles;
  int ktiles, ktiles_per_split, splits;
  int w_shift, hw_shift;  // log2 when W / H*W are powers of two, else -1
  uint32_t x_bytes, dy_bytes;
  size_t slab_stride;
};

// CB = channels per 128-wide N tile (16/32/64/128); TPB = 128/CB taps per tile
template <int KS, int BM, int CB, int WM, int WN, bool UP2, int NBUF>
__global__ __launch_bounds__(WM* WN * 64) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = WM * WN * 64, BK = 32, BN = 128, KK = KS * KS, P = KS / 2, TPB = BN / CB;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int PA = BM + 1, PB = BN + 1;  // odd pitch: transposing ds_write_b32 is conflict free
  constexpr int RP = NT / 32, AL = BM / RP, BL = BN / RP;
  static_assert(NT == 256 && RP == 8, "loader mapping assumes 256 threads");
  __shared__ float As[NBUF][BK * PA];
  __shared__ float Bs[NBUF][BK * PB];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, half = lane >> 5;
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int tile = q % a.tiles, sk = (q / a.tiles) * 8 + xcd;
  if (sk >= a.splits) return;
  const int tile_m = tile % a.mt, tile_n = tile / a.mt;
  const int m0 = tile_m * BM;
  int tap0, ci0;
  if (CB == 128) {
    const int per_tap = a.Cip / 128;
    tap0 = tile_n / per_tap;
    ci0 = (tile_n - tap0 * per_tap) * 128;
  } else {
    tap0 = tile_n * TPB;
    ci0 = 0;
  }
  const int kt0 = sk * a.ktiles_per_split;
  const int kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);
  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? W / 2 : W, HWs = Hs * Ws;
  const int kl = t & 31, r0 = t >> 5;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.dy, a.dy_bytes);
  const uint32_t a_stride4 = (uint32_t)(RP * HW) * 4u, b_unit4 = (uint32_t)HWs * 4u;

  float areg[AL], breg[BL];
  auto load_tile = [&](int kt) {
    const int k = kt * BK + kl;
    const bool kvalid = k < a.Ktot;
    int bi, hw, h, w;
    if (a.hw_shift >= 0) {
      bi = k >> a.hw_shift;
      hw = k & (HW - 1);
    } else {
      bi = k / HW;
      hw = k - bi * HW;
    }
    if (a.w_shift >= 0) {
      h = hw >> a.w_shift;
      w = hw & (W - 1);
    } else {
      h = hw / W;
      w = hw - h * W;
    }
    // A rows m0+r0+8i of dY: rows past Co only feed accumulator rows that are never stored
    const uint32_t va = kvalid ? (uint32_t)((bi * a.Co + m0 + r0) * HW + hw) * 4u : kOobBase;
#pragma unroll
    for (int i = 0; i < AL; ++i) areg[i] = buf_load_s(rdy, va, (uint32_t)i * a_stride4);
    if constexpr (CB == 4) {
      // <= 4 reduction channels (the 3-channel stem, or the predict conv with its operands swapped): column
      // n = tap_local*4 + channel, this thread's columns r0 + 8i are taps 2i + (r0 >> 2) of channel r0 & 3
      const int tsel = r0 >> 2, cl = r0 & 3;
#pragma unroll
      for (int i = 0; i < BL; ++i) {
        const int tap = tap0 + 2 * i + tsel;
        const int hh = h + tap / KS - P, ww = w + tap % KS - P;
        const bool valid = kvalid && tap < KK && cl < a.Ci && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const int hs = UP2 ? hh >> 1 : hh, wsrc = UP2 ? ww >> 1 : ww;
        breg[i] = buf_load(rx, valid ? (uint32_t)(((bi * a.Ci + cl) * Hs + hs) * Ws + wsrc) * 4u : kOobBase);
      }
    } else {
      uint32_t vb[TPB];
#pragma unroll
      for (int tl = 0; tl < TPB; ++tl) {
        const int tap = tap0 + tl;
        const int hh = h + tap / KS - P, ww = w + tap % KS - P;
        const bool valid = kvalid && tap < KK && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const int hs = UP2 ? hh >> 1 : hh, wsrc = UP2 ? ww >> 1 : ww;
        vb[tl] = valid ? (uint32_t)(((bi * a.Ci + ci0 + r0) * Hs + hs) * Ws + wsrc) * 4u : kOobBase;
      }
#pragma unroll
      for (int i = 0; i < BL; ++i) {
        const int tl = (RP * i) / CB, cl = (RP * i) % CB;  // compile-time after unrolling
        breg[i] = buf_load_s(rx, vb[tl], (uint32_t)cl * b_unit4);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) As[buf][kl * PA + r0 + i * RP] = areg[i];
#pragma unroll
    for (int i = 0; i < BL; ++i) Bs[buf][kl * PB + r0 + i * RP] = breg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (more) load_tile(kt + 1);
      mfma_tile<BK, PA, PB, TM, TN>(&As[cur][half * PA + wm * WTM + l31], &Bs[cur][half * PB + wn * WTN + l31], acc);
      if (NBUF == 2) {
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
      } else {                      // one LDS buffer (half the LDS -> twice the resident blocks)
        __syncthreads();
        if (more) store_tile(0);
        __syncthreads();
      }
    }
  }

  // slab[sk][m][tile_n*128 + nl], nl = tap_local*CB + channel_local
  float* out = a.out + (size_t)sk * a.slab_stride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wn * WTN + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < a.Co) out[(size_t)m * a.Np + tile_n * BN + nl] = acc[i][j][r];
      }
  }
}

// ------------------------------------------------------------------------------ split-bf16 wgrad
// Weight gradient on the bf16 matrix cores (see conv_fwd_bf16s_kernel for the split arithmetic).
// Both operands are K-contiguous in memory (k = pixel index), which is exactly the MFMA fragment
// shape: 8 consecutive pixels of one dY row / one shifted X row are one 16-byte chunk per plane.
// dY chunks are aligned 2 x dwordx4 loads; X chunks are 8 dword loads at the tap-shifted address
// (per-dword hardware range check; the one pixel that wraps across an image-row edge is masked).
// Needs W % 8 == 0, KS in {1,3}, Ci % 32 == 0; other shapes stay on the fp32 kernel.
__device__ __forceinline__ f32x4 buf_load_x4(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

template <int KS, int BM, int CB, int WM, int WN, int NS>
__global__ __launch_bounds__(WM* WN * 64) void conv_wgrad_bf16s_kernel(WgradArgs a) {
  constexpr int NT = WM * WN * 64, BK = 32, KC = BK / 8, BN = 128, KK = KS * KS, P = KS / 2, TPB = BN / CB;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int AL = BM / 64, BL = BN / 64;   // 16-row blocks per thread: 4 waves x 16 rows per pass
  static_assert(NT == 256 && CB >= 32, "loader mapping");
  __shared__ u32x4 As[NS * KC * BM];
  __shared__ u32x4 Bs[NS * KC * BN];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, half = lane >> 5;
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int tile = q % a.tiles, sk = (q / a.tiles) * 8 + xcd;
  if (sk >= a.splits) return;
  const int tile_m = tile % a.mt, tile_n = tile / a.mt;
  const int m0 = tile_m * BM;
  int tap0, ci0;
  if (CB == 128) {
    const int per_tap = a.Cip / 128;
    tap0 = tile_n / per_tap;
    ci0 = (tile_n - tap0 * per_tap) * 128;
  } else {
    tap0 = tile_n * TPB;
    ci0 = 0;
  }
  const int kt0 = sk * a.ktiles_per_split;
  const int kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);
  const int H = a.H, W = a.W, HW = H * W;
  const int sub = lane & 15, kc_l = lane >> 4;          // this lane's row/column within a 16-block, its k chunk
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.dy, a.dy_bytes);

  float areg[AL][8], breg[BL][8];
  auto load_tile = [&](int kt) {
    const int k = kt * BK + kc_l * 8;                   // first pixel of this lane's chunk
    const bool kvalid = k < a.Ktot;
    int bi, hw, h, w0;
    if (a.hw_shift >= 0) {
      bi = k >> a.hw_shift;
      hw = k & (HW - 1);
    } else {
      bi = k / HW;
      hw = k - bi * HW;
    }
    if (a.w_shift >= 0) {
      h = hw >> a.w_shift;
      w0 = hw & (W - 1);
    } else {
      h = hw / W;
      w0 = hw - h * W;
    }
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int m = m0 + (wid + 4 * i) * 16 + sub;      // rows past Co only feed unstored accumulator rows
      const uint32_t va = kvalid ? (uint32_t)((bi * a.Co + m) * HW + hw) * 4u : kOobBase;
      const f32x4 lo = buf_load_x4(rdy, va), hi = buf_load_x4(rdy, va + 16u);
#pragma unroll
      for (int j = 0; j < 4; ++j) areg[i][j] = lo[j], areg[i][4 + j] = hi[j];
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const int nb = (wid + 4 * i) * 16;                // column block (wave-uniform): one tap, 16 channels
      const int tl = nb / CB, tap = tap0 + tl;
      const int ci = ci0 + (nb - tl * CB) + sub;
      const int dh = tap / KS - P, dw = tap - (tap / KS) * KS - P;
      const int hh = h + dh;
      const bool valid = kvalid && tap < KK && ci < a.Ci && (unsigned)hh < (unsigned)H;
      const uint32_t vb = valid ? (uint32_t)(((bi * a.Ci + ci) * H + hh) * W + w0 + dw) * 4u : kOobBase;
      // The pixel that would wrap across the image-row edge is padding.  Its load is redirected to the
      // neighbouring in-range element: hipcc merges consecutive dword loads into dwordx4, and a merged
      // load that STARTS at offset -4 (first row of the tensor, dw = -1) or ends one past the tensor
      // (last row, dw = +1) is range-checked as a whole -- the valid pixels beside it came back 0.
      const bool edge_l = dw < 0 && w0 == 0, edge_r = dw > 0 && w0 + 8 == W;
      const uint32_t v0 = edge_l ? vb + 4u : vb, v7 = edge_r ? vb + 24u : vb + 28u;
      breg[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, v0, 0, 0));
#pragma unroll
      for (int j = 1; j < 7; ++j)
        breg[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vb + 4u * j, 0, 0));
      breg[i][7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, v7, 0, 0));
      if (edge_l) breg[i][0] = 0.f;
      if (edge_r) breg[i][7] = 0.f;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      u32x4 pl[NS];
      split8<NS>(areg[i], pl);
#pragma unroll
      for (int p = 0; p < NS; ++p) As[(p * KC + kc_l) * BM + (wid + 4 * i) * 16 + sub] = pl[p];
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      u32x4 pl[NS];
      split8<NS>(breg[i], pl);
#pragma unroll
      for (int p = 0; p < NS; ++p) Bs[(p * KC + kc_l) * BN + (wid + 4 * i) * 16 + sub] = pl[p];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mfma_step = [&]() {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int kc = ks * 2 + half;
      bf16x8 af[NS][TM], bfr[NS][TN];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[p][i] = __builtin_bit_cast(bf16x8, As[(p * KC + kc) * BM + wm * WTM + i * 32 + l31]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[p][j] = __builtin_bit_cast(bf16x8, Bs[(p * KC + kc) * BN + wn * WTN + j * 32 + l31]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];
          if constexpr (NS == 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[2][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bfr[0][j], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  };

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile();
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (more) load_tile(kt + 1);
      mfma_step();
      __syncthreads();
      if (more) store_tile();
      __syncthreads();
    }
  }

  float* out = a.out + (size_t)sk * a.slab_stride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wn * WTN + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < a.Co) out[(size_t)m * a.Np + tile_n * BN + nl] = acc[i][j][r];
      }
  }
}

// ---------------------------------------------------------- weight gradient on pre-split planes
// dW[co][ci][dh][dw] = sum over pixels of dY[co][px] * X[ci][px + (dh, dw)], with BOTH operands given as the
// channel-blocked planes the forward / data-gradient kernels use (chunk = 8 channels of one pixel).  The
// reduction index is the pixel, so in LDS the operands are "K-major"; the MFMA fragments (8 consecutive
// pixels of one channel per lane) come out of ds_read_b64_tr_b16, the gfx950 transposing LDS read -- no
// second copy of the activations in a pixel-major layout is ever made.
//   block  = 128 output x 64 input channels (64 x 128 when Co <= 64) x the 3 taps of one filter row (dh), one K slice
//   step   = 64 consecutive pixels; X arrives once per step as a halo'd band (NR rows x (W+2) columns, the
//            rows already shifted by dh): the 3 taps read it at column offsets -1/0/+1
//   waves  = 8 MFMA waves (4 x 2, a 32x32 tile x 3 taps each) + 4 loader waves (LDS-DMA, two stages)
// Padding is a zero chunk at DMA time; the MFMA loop has no bounds logic.  Partial sums go to fp32 slabs
// [split][tap][co][ci], reduced in a fixed order by wgrad_p_reduce (bitwise reproducible).
struct WgradArgsP {
  const u32x4* xp;
  const u32x4* dyp;
  float* slab;
  int B, Ci, H, W, Co;
  int tiles_m, tiles_n;
  int steps, steps_per_split, splits;
  int h_shift;
  int wseg_shift;           // images wider than 64 (LOG2W = 6 instantiation): log2(W / 64) 64-pixel segments per row
  int groups;               // tiles_m * tiles_n * splits: (tile, K slice) pairs, three blocks (filter rows) each
  size_t xplane, dyplane;   // chunks per plane
  const ScaleRec* xscale;   // fp16 planes: scale records of x and dy (the slabs are multiplied by 1 / (Sx * Sdy))
  const ScaleRec* dyscale;
#ifdef ITCV_DIAG
  int debug;                // diagnostic only (ITCV_ABLATE & 64): block 0 reports main-loop shader cycles / steps in slab[0..1]
#endif
};

__device__ __forceinline__ void tr_read8(bf16x8& dst, uint32_t addr0, uint32_t addr1) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)addr0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)addr1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  dst = __builtin_bit_cast(bf16x8, v);
}

// KH = 2 (the 64 x 64 tile): the eight MFMA waves are two groups of four that take alternate halves of a step's four
// 16-pixel k-steps and keep separate slabs (slab index split*KH + kh) -- the tile is too small for eight waves otherwise.
// M16: the products are v_mfma_f32_16x16x32 (a 32 x 32 wave tile = 2 x 2 blocks, k-steps of 32 pixels: the four 16-lane
// groups of a transposing read take the four 8-pixel quarters of a k-step instead of two halves of a 16-pixel one) -- same
// LDS reads and MFMA cycles per FLOP as 32x32x16; the chip holds a higher clock on this shape (band kernels: +5-8 %).
template <int LOG2W, bool UP2, int BM, int BN, int NST, int NLW, int KH = 1, bool F16 = false, bool M16 = false>
__global__ __launch_bounds__(512 + 64 * NLW) void conv_wgrad_bf16p_kernel(WgradArgsP a) {
  constexpr int W = 1 << LOG2W, NR = 64 >> LOG2W, WP = W + 2, NP = NR * WP;   // band: NR rows x (W+2) columns
  constexpr int PXA = 68, PXB = ((NP + 11) / 16) * 16 + 4;                     // row strides = 4 (mod 16) chunks: conflict-free tr reads
  static_assert(PXB >= NP && PXB % 16 == 4, "band stride");
  constexpr int ACH = BM / 8, BCH = BN / 8;                                    // 128 x 128, 128 x 64 or 64 x 128 (co x ci)
  constexpr int WMn = BM / 32, WNn = (8 / KH) / WMn, TNw = BN / (32 * WNn), KPW = 4 / KH;   // KPW: k-steps per wave
  static_assert(KH == 1 || KH == 2, "k-halves");          // 32 x 32*TNw accumulators x 3 taps per wave
  static_assert(TNw >= 1 && BN == 32 * WNn * TNw, "wave tiling");
  constexpr int ASZ = 2 * ACH * PXA, BSZ = 2 * BCH * PXB, SSZ = ASZ + BSZ;     // chunks per stage
  extern __shared__ u32x4 smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  // The three filter-row blocks (dhi) of one (tile, K slice) read the SAME dY and X bytes: they get block ids 8 apart
  // inside a run of 24, i.e. (under the observed round-robin dealing of blocks to XCDs) the same XCD, and run at the same
  // time (the grid is at most one block per CU), so two of the three reads are served by that XCD's L2 -- measured
  // without this the 64-channel 64x64 layer moved 3x its operand bytes from HBM.  Speed only: nothing depends on it.
  const int run = blockIdx.x / 24, rr = blockIdx.x % 24, dhi = rr >> 3;
  int bid = run * 8 + (rr & 7);
  if (bid >= a.groups) return;
  const int split = bid % a.splits;
  bid /= a.splits;
  const int tn = bid % a.tiles_n, tm = bid / a.tiles_n;
  const int co0 = tm * BM, ci0 = tn * BN;
  const int s0 = split * a.steps_per_split, s1 = min(a.steps, s0 + a.steps_per_split);
  // Image width: the band (one step = 64 pixels = NR rows of W <= 64 columns) is the image row for LOG2W < 6; the
  // LOG2W = 6 instantiation also serves 128- and 256-wide images, one 64-column SEGMENT of a row per step
  // (the segment's halo columns come from the neighbouring segments instead of the zero padding)
  const int wsh = LOG2W == 6 ? a.wseg_shift : 0;
  const int Wi = W << wsh;
  const int H = a.H, HW = H * Wi;
  const uint32_t smem_base = lds_addr(smem);

  if (wid >= 8) {
    // ------------------------------------------------------------------ loaders
    if (s0 >= s1) return;
    const int lw = wid - 8;
    const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? Wi / 2 : Wi, HWs = Hs * Ws;
    const int Co8 = a.Co >> 3, Ci8 = a.Ci >> 3, dh = dhi - 1;
    const u32x4* zero = &g_zero_chunk;
    // this wave's band pixel (same for all of its X pieces): half = lw & 1
    const int hp = (lw & 1) * 64 + lane;
    const bool hp_active = hp < NP;
    const int R = hp / WP, w = hp - R * WP - 1;     // column inside the band: -1 .. W

    auto issue = [&](int st, int stage) {
      const uint32_t sbase = smem_base + (uint32_t)(stage * SSZ) * 16u;
      // dY: 64 consecutive pixels of the step, 16 channel chunks x 2 planes = 32 pieces; this wave: q = 4j + lw
      {
        const int g = st * 64 + lane, b = g / HW, pix = g - b * HW;
#pragma unroll
        for (int j = 0; j < 2 * ACH / NLW; ++j) {
          const int qq = j * NLW + lw, pl = qq / ACH, c8 = qq % ACH;
          const int gc8 = (co0 >> 3) + c8;
          const u32x4* src = a.dyp + ((size_t)pl * a.dyplane + ((size_t)b * Co8 + gc8) * HW + pix);
          lds_dma16(gc8 < Co8 ? src : zero, sbase + (uint32_t)((pl * ACH + c8) * PXA) * 16u);
        }
      }
      // X band: rows st*NR .. st*NR+NR-1 of the (batch x height) row index, shifted by dh; 2 pieces per (plane, chunk row)
      {
        const int gr = (st >> wsh) * NR + R, b = gr >> a.h_shift, h = gr & (H - 1), hh = h + dh;
        const int wi = ((st & ((1 << wsh) - 1)) << 6) + w;     // image column (wsh = 0: the band is the whole row)
        const bool valid = (unsigned)wi < (unsigned)Wi && (unsigned)hh < (unsigned)H;
        const int spix = UP2 ? (hh >> 1) * Ws + (wi >> 1) : hh * Wi + wi;
#pragma unroll
        for (int j = 0; j < 4 * BCH / NLW; ++j) {
          const int qq = (j * NLW + lw) >> 1, pl = qq / BCH, c8 = qq % BCH;
          const int gc8 = (ci0 >> 3) + c8;
          const u32x4* src = a.xp + ((size_t)pl * a.xplane + ((size_t)b * Ci8 + gc8) * HWs + spix);
          if (hp_active)
            lds_dma16((valid && gc8 < Ci8) ? src : zero,
                      sbase + (uint32_t)(ASZ + (pl * BCH + c8) * PXB + (lw & 1) * 64) * 16u);
        }
      }
    };
    // NST-deep ring: NST-1 steps in flight ahead of the one being multiplied (measured with two stages: a step took
    // ~4300 cycles against 2304 of MFMA work -- one step of lookahead does not cover the ~50 KB / 30 B/clk ingest
    // plus its latency).  Every piece below is issued by every loader wave, so the counted waits are exact.
    constexpr int PST = 2 * ACH / NLW + 4 * BCH / NLW;   // pieces per loader wave per step
    issue(s0, 0);
    if (NST == 3 && s0 + 1 < s1) {
      issue(s0 + 1, 1);
      wait_vmcnt<PST>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    int stage = NST - 1;
    for (int st = s0; st < s1; ++st) {
      if (st + NST - 1 < s1) {
        issue(st + NST - 1, stage);                  // into the slot step st-1 has just left
        if (NST == 3) wait_vmcnt<PST>();             // step st+1 has landed, st+2 may be in flight
        else wait_vmcnt<0>();
      } else {
        wait_vmcnt<0>();
      }
      if (++stage == NST) stage = 0;
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // -------------------------------------------------------------------- MFMA waves
  const int kh = wid / (8 / KH), w8 = wid % (8 / KH);
  const int wm = w8 / WNn, wn = w8 % WNn;
  const int G = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, half = G >> 1, rb = G & 1;
  const int l31 = lane & 31;
  if constexpr (M16) {
    typedef float f32x4m __attribute__((ext_vector_type(4)));
    constexpr int KP32 = 2 / KH;                      // 32-pixel k-steps per wave and 64-pixel step
    f32x4m acc[3][TNw][2][2];                         // [tap][32-column tile][16-row block][16-column block]
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int j = 0; j < TNw; ++j)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
          for (int bj = 0; bj < 2; ++bj) acc[tp][j][bi][bj] = f32x4m{0.f, 0.f, 0.f, 0.f};
    if (s0 < s1) {
      // this lane's transposed reads: 16-lane group G takes pixels 8G .. 8G+7 of the k-step; (q, pp) as above
      const uint32_t aoff = (uint32_t)(((4 * wm + (pp >> 1)) * PXA + G * 8 + q) * 16 + (pp & 1) * 8);
      const uint32_t boff = (uint32_t)((ASZ + (4 * wn * TNw + (pp >> 1)) * PXB) * 16 + (pp & 1) * 8);
      uint32_t hidx[2 * KP32];                        // band index of pixel k = kk*32 + G*8 + s*4 + q, centre tap
#pragma unroll
      for (int u = 0; u < 2 * KP32; ++u) {
        const int k = ((u >> 1) + kh * KP32) * 32 + G * 8 + (u & 1) * 4 + q;
        hidx[u] = (uint32_t)(((k >> LOG2W) * WP + (k & (W - 1)) + 1) * 16);
      }
      __builtin_amdgcn_s_barrier();
      int stage = 0;
      for (int st = s0; st < s1; ++st) {
        const uint32_t sb = smem_base + (uint32_t)(stage * SSZ) * 16u;
        if constexpr (TNw == 1) {
          // software pipeline over the step's KP32 * 6 units (k-step, tap, column block): the two B fragments of unit u+1
          // (and the four A fragments of the next k-step) are requested before the six MFMAs of unit u
          bf16x8 af[2][2][2], bfr[2][2];
          auto fetch_a = [&](int kk, int kb) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) {
                const uint32_t ad = sb + aoff + (uint32_t)((pl * ACH + 2 * bi) * PXA * 16 + (kh * KP32 + kk) * 512);
                tr_read8(af[kb][bi][pl], ad, ad + 64);
              }
          };
          auto fetch_b = [&](int u, int ub) {
            const int kk = u / 6, tp = (u % 6) >> 1, bj = u & 1;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              const uint32_t bd = sb + boff + (uint32_t)((pl * BCH + 2 * bj) * PXB * 16 + (tp - 1) * 16);
              tr_read8(bfr[ub][pl], bd + hidx[kk * 2], bd + hidx[kk * 2 + 1]);
            }
          };
          constexpr int NU = KP32 * 6;
          fetch_a(0, 0);
          fetch_b(0, 0);
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int kk = u / 6, tp = (u % 6) >> 1, bj = u & 1, ub = u & 1, kb = kk & 1;
            if (u + 1 < NU) {
              if ((u + 1) % 6 == 0) fetch_a(kk + 1, kb ^ 1);
              fetch_b(u + 1, ub ^ 1);
            }
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
              f32x4m c = acc[tp][0][bi][bj];
              c = mma16x16x32<F16>(af[kb][bi][0], bfr[ub][1], c);
              c = mma16x16x32<F16>(af[kb][bi][1], bfr[ub][0], c);
              c = mma16x16x32<F16>(af[kb][bi][0], bfr[ub][0], c);
              acc[tp][0][bi][bj] = c;
            }
          }
        } else {
#pragma unroll
        for (int kk = 0; kk < KP32; ++kk) {
          bf16x8 af[2][2];
#pragma unroll
          for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              const uint32_t ad = sb + aoff + (uint32_t)((pl * ACH + 2 * bi) * PXA * 16 + (kh * KP32 + kk) * 512);
              tr_read8(af[bi][pl], ad, ad + 64);
            }
#pragma unroll
          for (int tp = 0; tp < 3; ++tp)
#pragma unroll
            for (int j = 0; j < TNw; ++j)
#pragma unroll
              for (int bj = 0; bj < 2; ++bj) {
                bf16x8 bfr[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                  const uint32_t bd = sb + boff + (uint32_t)((pl * BCH + 4 * j + 2 * bj) * PXB * 16 + (tp - 1) * 16);
                  tr_read8(bfr[pl], bd + hidx[kk * 2], bd + hidx[kk * 2 + 1]);
                }
#pragma unroll
                for (int bi = 0; bi < 2; ++bi) {
                  f32x4m c = acc[tp][j][bi][bj];
                  c = mma16x16x32<F16>(af[bi][0], bfr[1], c);
                  c = mma16x16x32<F16>(af[bi][1], bfr[0], c);
                  c = mma16x16x32<F16>(af[bi][0], bfr[0], c);
                  acc[tp][j][bi][bj] = c;
                }
              }
        }
        }
        if (++stage == NST) stage = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    // slab[split][tap][co][ci]: lane (G, i16) holds rows 4G .. 4G+3 and column i16 of every 16 x 16 block
    const float oscale = F16 ? inv_scale_of(a.xscale) * inv_scale_of(a.dyscale) : 1.f;
#pragma unroll
    for (int j = 0; j < TNw; ++j)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int ci = ci0 + 32 * (wn * TNw + j) + 16 * bj + i16;
        if (ci >= a.Ci) continue;
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
          float* out = a.slab + ((size_t)((split * KH + kh) * 9 + dhi * 3 + tp) * a.Co) * a.Ci + ci;
#pragma unroll
          for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = co0 + 32 * wm + 16 * bi + 4 * G + r;
              if (co < a.Co) out[(size_t)co * a.Ci] = F16 ? acc[tp][j][bi][bj][r] * oscale : acc[tp][j][bi][bj][r];
            }
        }
      }
    return;
  }
  f32x16 acc[3][TNw];
#pragma unroll
  for (int tp = 0; tp < 3; ++tp)
#pragma unroll
    for (int j = 0; j < TNw; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tp][j][r] = 0.f;
  if (s0 < s1) {
    // byte offsets inside a stage of this lane's transposed reads (plane 0, k-step 0, first half)
    const uint32_t aoff = (uint32_t)(((4 * wm + 2 * rb + (pp >> 1)) * PXA + half * 8 + q) * 16 + (pp & 1) * 8);
    const uint32_t boff = (uint32_t)((ASZ + (4 * wn * TNw + 2 * rb + (pp >> 1)) * PXB) * 16 + (pp & 1) * 8);
    // band index of pixel k = kk*16 + half*8 + s*4 + q, for the 8 (kk, s) pairs, centre tap
    uint32_t hidx16[2 * KPW];
#pragma unroll
    for (int u = 0; u < 2 * KPW; ++u) {
      const int k = ((u >> 1) + kh * KPW) * 16 + half * 8 + (u & 1) * 4 + q;
      hidx16[u] = (uint32_t)(((k >> LOG2W) * WP + (k & (W - 1)) + 1) * 16);
    }
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    const long long dbg_c0 = ITCV_DBG(a) ? clock64() : 0;
    for (int st = s0; st < s1; ++st) {
      const uint32_t sb = smem_base + (uint32_t)(stage * SSZ) * 16u;
      if constexpr (TNw == 1) {
        // register double buffer over the four k-steps: the 16 transposed reads of k-step kk+1 are interleaved (two
        // behind each MFMA, by scheduler group hints) with the 9 MFMAs of k-step kk instead of sitting in front of
        // their own use
        bf16x8 af[2][2], bfr[2][3][2];
        auto fetch = [&](int kk, int bufi) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const uint32_t ad = sb + aoff + (uint32_t)(pl * ACH * PXA * 16 + (kh * KPW + kk) * 256);
            tr_read8(af[bufi][pl], ad, ad + 64);
          }
#pragma unroll
          for (int tp = 0; tp < 3; ++tp)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              const uint32_t bd = sb + boff + (uint32_t)(pl * BCH * PXB * 16 + (tp - 1) * 16);
              tr_read8(bfr[bufi][tp][pl], bd + hidx16[kk * 2], bd + hidx16[kk * 2 + 1]);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < KPW; ++kk) {
          const int cur = kk & 1;
          if (kk + 1 < KPW) fetch(kk + 1, cur ^ 1);
#pragma unroll
          for (int tp = 0; tp < 3; ++tp) {
            f32x16 c = acc[tp][0];
            c = mma32x32x16<F16>(af[cur][0], bfr[cur][tp][1], c);
            c = mma32x32x16<F16>(af[cur][1], bfr[cur][tp][0], c);
            c = mma32x32x16<F16>(af[cur][0], bfr[cur][tp][0], c);
            acc[tp][0] = c;
          }
          if (kk == 0) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // the first k-step's own reads
          if (kk + 1 < KPW) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
          }
        }
      } else {
      static_assert(TNw == 1 || KH == 1, "k-halves only with one accumulator column per wave");
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 af[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const uint32_t ad = sb + aoff + (uint32_t)(pl * ACH * PXA * 16 + kk * 256);
          tr_read8(af[pl], ad, ad + 64);
        }
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
          for (int j = 0; j < TNw; ++j) {
            bf16x8 bfr[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              const uint32_t bd = sb + boff + (uint32_t)((pl * BCH + 4 * j) * PXB * 16 + (tp - 1) * 16);
              tr_read8(bfr[pl], bd + hidx16[kk * 2], bd + hidx16[kk * 2 + 1]);
            }
            f32x16 c = acc[tp][j];
            c = mma32x32x16<F16>(af[0], bfr[1], c);
            c = mma32x32x16<F16>(af[1], bfr[0], c);
            c = mma32x32x16<F16>(af[0], bfr[0], c);
            acc[tp][j] = c;
          }
      }
      }
      if (++stage == NST) stage = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (ITCV_DBG(a) && t == 0 && blockIdx.x == 0) {
      a.slab[0] = (float)(clock64() - dbg_c0), a.slab[1] = (float)(s1 - s0);
      return;
    }
  }
  // slab[split][tap][co][ci]
  const float oscale = F16 ? inv_scale_of(a.xscale) * inv_scale_of(a.dyscale) : 1.f;   // exact: powers of two
#pragma unroll
  for (int j = 0; j < TNw; ++j) {
    const int ci = ci0 + 32 * (wn * TNw + j) + l31;
    if (ci >= a.Ci) continue;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      float* out = a.slab + ((size_t)((split * KH + kh) * 9 + dhi * 3 + tp) * a.Co) * a.Ci + ci;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < a.Co) out[(size_t)co * a.Ci] = F16 ? acc[tp][j][r] * oscale : acc[tp][j][r];
      }
    }
  }
}

// ---- 5x5 weight gradients with a 3-channel side (stem 3 -> 64, predict 64 -> 3), bf16x3, matrix cores ----------
// OUT[(cs, dw)][dh][cb] = sum_{b,h,w} S[b][cs][h + sg*(dh-2)][w + sg*(dw-2)] * T[b][cb][h][w]
//   S: the <= 3-channel tensor (fp32), T: the 64-channel tensor as planes; sg = +1 for the stem (S = x, T = dy) and
//   -1 for the predict conv (S = dy, T = x, the sum re-indexed over x's pixel).
// MFMA 16x16x32 rows are (cs, dw) -- 15 of 16 used; the reduction index is the pixel, so T's chunks (8 channels of a
// pixel) are staged K-major in LDS and read back through ds_read_b64_tr_b16; S fragments are 8 consecutive pixels of
// a (shifted) row, split in registers.  A wave owns a few image rows (T row staged once, used by the five S rows it
// pairs with) and all 5 x 64 (dh, cb) accumulators; its partial result goes to a slab, folded by wgrad5_reduce.
// F16: T is given as fp16 planes (scale record behind them); S is split in registers into fp16 planes of Ss*S with Ss from
// `sm_amax` (block maxima of |S|, itcv_absmax; null: Ss = 1 -- the stem's S is the input image).
template <int SG, bool F16 = false>
__global__ __launch_bounds__(256) void conv_wgrad5_planes_kernel(const float* __restrict__ sm, const u32x4* __restrict__ tp,
                                                                float* __restrict__ slab, int B, int CS, int H, int W,
                                                                int rows_per_job, int njobs, size_t plane_stride,
                                                                const float* __restrict__ sm_amax) {
  constexpr int PXS = 68;                          // row stride of the staged T row: 4 (mod 16) chunks, see conv_wgrad_bf16p_kernel
  constexpr int RSZ = 2 * 8 * PXS;                 // chunks per staged row ([plane][c8][px], up to 64 px)
  __shared__ u32x4 rows[4][RSZ];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int job = blockIdx.x * 4 + wv;
  if (job >= njobs) return;
  const int jobs_per_img = (H + rows_per_job - 1) / rows_per_job;
  const int b = job / jobs_per_img, h_lo = (job - b * jobs_per_img) * rows_per_job, h_hi = min(H, h_lo + rows_per_job);
  const size_t HW = (size_t)H * W;
  const int n = lane & 15, kg = lane >> 4;                 // A row m = n = (cs, dw); k group of 8 pixels
  const int cs = n / 5, dw = n - cs * 5;
  const bool row_used = cs < CS;
  const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;   // transposed-read lane roles
  const uint32_t rbase = lds_addr(rows[wv]);
  float sscale = 1.f, oscale = 1.f;
  if constexpr (F16) {
    if (sm_amax) sscale = scale_for_bound(wave_absmax_of(sm_amax, kAbsmaxParts));
    oscale = inv_scale_of(reinterpret_cast<const ScaleRec*>(tp + 2 * plane_stride)) / sscale;      // exact: powers of two
  }
  f32x4_t acc[5][4];
#pragma unroll
  for (int dh = 0; dh < 5; ++dh)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[dh][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int h = h_lo; h < h_hi; ++h) {
   for (int seg = 0; seg < W; seg += 64) {                  // rows wider than 64 pixels: one 64-pixel segment at a time
    const int segw = min(64, W - seg);
    // stage T row (b, h), columns seg .. seg+segw-1: 8 channel chunks x 2 planes; lane = pixel
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // previous segment's fragment reads are done (wave-private buffer)
    if (lane < segw) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8)
          rows[wv][(pl * 8 + c8) * PXS + lane] =
              tp[(size_t)pl * plane_stride + ((size_t)b * 8 + c8) * HW + (size_t)h * W + seg + lane];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int w0 = 0; w0 < segw; w0 += 32) {
      // B fragments: T^T, columns = 16 channels of N-tile nt, k = pixels w0 + 8*kg + {0..7}
      bf16x8 bfr[4][2];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const uint32_t ad = rbase + (uint32_t)(((pl * 8 + 2 * nt + (pp >> 1)) * PXS + w0 + 8 * kg + q) * 16 + (pp & 1) * 8);
          tr_read8(bfr[nt][pl], ad, ad + 64);
        }
      // A fragments: S[b][cs][h + SG*(dh-2)][w0 + 8*kg + j + SG*(dw-2)], j = 0..7.  The 40 loads of the five filter rows
      // go out together, from clamped (always valid) addresses, and are masked afterwards: loaded under their bounds
      // checks, filter row by filter row, they were five dependent L2 round trips per 32 pixels.
      float sv[5][8];
      const int ws0 = seg + w0 + 8 * kg + SG * (dw - 2);
      const float* spc = sm + ((size_t)b * CS + (row_used ? cs : 0)) * HW;
#pragma unroll
      for (int dh = 0; dh < 5; ++dh) {
        const int hs = h + SG * (dh - 2);
        const float* sp = spc + (size_t)min(max(hs, 0), H - 1) * W;
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[dh][j] = sp[min(max(ws0 + j, 0), W - 1)];
      }
#pragma unroll
      for (int dh = 0; dh < 5; ++dh) {
        const int hs = h + SG * (dh - 2);
        float v[8];
        const bool hok = row_used && (unsigned)hs < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (hok && (unsigned)(ws0 + j) < (unsigned)W) ? sv[dh][j] : 0.f;
        u32x4 ap[2];
        split8<2, F16>(v, ap, sscale);
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, ap[0]), a1 = __builtin_bit_cast(bf16x8, ap[1]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4_t c = acc[dh][nt];
          c = mma16x16x32<F16>(a0, bfr[nt][1], c);
          c = mma16x16x32<F16>(a1, bfr[nt][0], c);
          c = mma16x16x32<F16>(a0, bfr[nt][0], c);
          acc[dh][nt] = c;
        }
      }
    }
   }
  }
  // slab[job][m = (cs,dw) 0..15][dh][cb 0..63]; lane: column n -> cb = nt*16 + n, rows m = 4*kg + i
  float* out = slab + (size_t)job * 16 * 5 * 64;
#pragma unroll
  for (int dh = 0; dh < 5; ++dh)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[((4 * kg + i) * 5 + dh) * 64 + nt * 16 + n] = F16 ? acc[dh][nt][i] * oscale : acc[dh][nt][i];
}

// dw (+)= sum_jobs slab: element e = (m = cs*5 + dw, dh, cb); stem: dW[cb][cs][dh][dw], predict: dW[cs][cb][dh][dw]
// (16 threads per element walk the jobs, folded in a fixed order through LDS)
__global__ __launch_bounds__(256) void wgrad5_reduce(const float* __restrict__ slab, float* __restrict__ dwt, int CS,
                                                    int njobs, int stem, int accumulate) {
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int total = CS * 5 * 5 * 64, i = blockIdx.x * 16 + e;   // i over (cs, dw, dh, cb)
  float s = 0.f;
  int dst = 0;
  if (i < total) {
    const int cb = i & 63, r = i >> 6, dh = r % 5, m = r / 5, cs = m / 5, dwi = m - cs * 5;
    const float* p = slab + (size_t)((m * 5 + dh) * 64 + cb);
    // eight jobs' loads in flight (clamped index + select: no branch between the loads); job order per thread unchanged
    for (int k = g; k < njobs; k += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)min(k + 16 * u, njobs - 1) * (16 * 5 * 64)];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (k + 16 * u < njobs) ? v[u] : 0.f;
    }
    dst = stem ? ((cb * CS + cs) * 5 + dh) * 5 + dwi : ((cs * 64 + cb) * 5 + dh) * 5 + dwi;
  }
  part[g][e] = s;
  __syncthreads();
  if (g == 0 && i < total) {
    float t = accumulate ? dwt[dst] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][e];
    dwt[dst] = t;
  }
}

// dw[co][ci][tap] (+)= sum_s slab[s][tap][co][ci].  A block owns 64 consecutive (co, ci) elements x the 9 taps: the four
// waves take the four quarters of the slice range (lane = element, nine accumulators, the loads of four slices in
// flight), the quarters' partial sums meet in LDS and are folded in wave order, and the 576 results leave as ONE
// contiguous run of dw (dw is tap-minor: a thread per (tap, element) wrote 4 bytes every 36 -- PMC: 359 MB written for 46 MB
// of gradients).  Every order is fixed => bitwise reproducible.  Several slab sources (a weight used by several passes of
// a backward) are folded one after the other, each from zero: dw + S_a + S_b, which is what separate calls compute.
struct WgReduceDesc {
  const float* slab[4];
  float* dw;
  int splits[4];
  int nsrc, coci, accumulate, block0, nblocks, fine;   // fine: wgrad_reduce_tile_fine
};
static_assert(sizeof(WgReduceDesc) == 80, "WgReduceDesc layout is part of the ABI (itcv_wgrad_reduce_desc_bytes)");

// "Wide" tiles (layers with few slices): a thread owns FOUR consecutive (co, ci) elements x the 9 taps and walks all
// slices itself, two slices (18 float4 loads) in flight -- no LDS, no barrier, and its 36 results are one contiguous
// 144-byte run of dw.  The first form of this fold (64 elements x 9 taps per block, the four waves taking quarters of the
// slice range, exchange through LDS) had a few KB in flight per block and idle phases around its barriers: ~2 TB/s whatever
// the tiling (tools/reduce_bench.py).
__device__ __forceinline__ void wgrad_reduce_tile(const WgReduceDesc& d, int tile) {
  const int e = (tile * 256 + (int)threadIdx.x) * 4;
  if (e >= d.coci) return;                                   // Co * Ci % 4 == 0 (Ci % 32 == 0)
  const size_t plane = (size_t)d.coci, stride = 9 * plane;
  float* dwp = d.dw + (size_t)e * 9;                         // 36 consecutive floats: (element i, tap tp) at i*9 + tp
  const bool vec = ((size_t)dwp & 15) == 0;                  // (a gradient view inside a flat buffer may be 4-byte aligned only)
  float tot[36];
  if (d.accumulate) {
    if (vec) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float4 t = reinterpret_cast<const float4*>(dwp)[q];
        tot[4 * q] = t.x, tot[4 * q + 1] = t.y, tot[4 * q + 2] = t.z, tot[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 36; ++q) tot[q] = dwp[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 36; ++q) tot[q] = 0.f;
  }
  for (int k = 0; k < d.nsrc; ++k) {
    const float* p = d.slab[k] + e;
    float acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.f;
    int sl = 0;
    for (; sl + 2 <= d.splits[k]; sl += 2) {
      float4 v[2][9];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) v[u][tp] = *reinterpret_cast<const float4*>(p + (size_t)(sl + u) * stride + (size_t)tp * plane);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
          acc[tp] += v[u][tp].x, acc[9 + tp] += v[u][tp].y, acc[18 + tp] += v[u][tp].z, acc[27 + tp] += v[u][tp].w;
    }
    if (sl < d.splits[k]) {
      float4 v[9];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) v[tp] = *reinterpret_cast<const float4*>(p + (size_t)sl * stride + (size_t)tp * plane);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc[tp] += v[tp].x, acc[9 + tp] += v[tp].y, acc[18 + tp] += v[tp].z, acc[27 + tp] += v[tp].w;
    }
#pragma unroll
    for (int q = 0; q < 36; ++q) tot[q] += acc[q];
  }
  if (vec) {
#pragma unroll
    for (int q = 0; q < 9; ++q) reinterpret_cast<float4*>(dwp)[q] = make_float4(tot[4 * q], tot[4 * q + 1], tot[4 * q + 2], tot[4 * q + 3]);
  } else {
#pragma unroll
    for (int q = 0; q < 36; ++q) dwp[q] = tot[q];
  }
}

// Layers with MANY slices (the 64-channel layers: a 147-KB weight, 170 slices): a thread walking all slices of its elements
// is a chain of ~85 dependent memory round trips, the critical path of the whole launch (157 against 84 us).  "Fine" tiles
// own 16 elements x 9 taps and split the slice range SIXTEEN ways (lane = (slice group, element), 4 groups per wave);
// the partial sums meet in LDS and are folded in group order, the 144 results leave as one contiguous run.  Only for
// >= 64 slices: at 8 / 16 slices fine tiles took 195 / 126 us for the same launch (tools/reduce_bench.py).
constexpr int kWgFineSplits = 64;   // a descriptor is "fine" when any of its sources has at least this many slices
__device__ __forceinline__ void wgrad_reduce_tile_fine(const WgReduceDesc& d, int tile, float* part /* [16][9][16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, el = lane & 15, sg = wave * 4 + (lane >> 4);
  const int e0 = tile * 16, e = e0 + el;
  const bool live = e < d.coci;
  const size_t plane = (size_t)d.coci, stride = 9 * plane;
  const int o = threadIdx.x;                          // outputs 0..143 = (element, tap) of this tile
  const bool oo = o < 144 && e0 * 9 + o < d.coci * 9;
  float tot = (d.accumulate && oo) ? d.dw[(size_t)e0 * 9 + o] : 0.f;
  for (int k = 0; k < d.nsrc; ++k) {
    const int q = (d.splits[k] + 15) >> 4;
    const int s0 = sg * q, s1 = min(s0 + q, d.splits[k]);
    float acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) acc[tp] = 0.f;
    if (live) {
      const float* p = d.slab[k] + e;
      int sl = s0;
      for (; sl + 4 <= s1; sl += 4) {
        float v[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) v[u][tp] = p[(size_t)(sl + u) * stride + (size_t)tp * plane];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) acc[tp] += v[u][tp];
      }
      for (; sl < s1; ++sl) {
        float v[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) v[tp] = p[(size_t)sl * stride + (size_t)tp * plane];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] += v[tp];
      }
    }
    if (k) __syncthreads();
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) part[(sg * 9 + tp) * 16 + el] = acc[tp];
    __syncthreads();
    if (o < 144) {
      const int el2 = o / 9, tp = o - el2 * 9;
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) sum += part[(g * 9 + tp) * 16 + el2];
      tot += sum;
    }
  }
  if (oo) d.dw[(size_t)e0 * 9 + o] = tot;
}
static inline bool wgrad_reduce_fine(const int* splits, int nsrc) {
  for (int k = 0; k < nsrc; ++k)
    if (splits[k] >= kWgFineSplits) return true;
  return false;
}
static inline int wgrad_reduce_blocks(int coci, bool fine) { return fine ? cdiv(coci, 16) : cdiv(coci, 1024); }

__global__ __launch_bounds__(256) void wgrad_p_reduce(const float* __restrict__ slab, float* __restrict__ dw, int CoCi,
                                                     int splits, int accumulate, int fine) {
  __shared__ float part[4][9][64];
  WgReduceDesc d;
  d.slab[0] = slab, d.dw = dw, d.splits[0] = splits, d.nsrc = 1, d.coci = CoCi, d.accumulate = accumulate;
  if (fine) wgrad_reduce_tile_fine(d, blockIdx.x, &part[0][0][0]);
  else wgrad_reduce_tile(d, blockIdx.x);
}

// The same reduce for MANY layers in one launch (a backward pass's weight gradients, folded when the pass is over): a
// device-resident table of descriptors, every layer owns a contiguous range of tile numbers; a layer whose weight took
// part in several network passes of the backward lists their slabs in call order.  Persistent blocks: each copies the
// table into LDS once and walks tiles blockIdx.x, + gridDim.x, ... -- with one block per tile every block began with two
// dependent global round trips (find the descriptor, load it) in front of a tile that needs one.
constexpr int kWgMaxDesc = 96;
__global__ __launch_bounds__(256) void wgrad_p_reduce_many(const WgReduceDesc* __restrict__ tab, int n, int tile0, int tile1) {
  __shared__ float part[4][9][64];
  __shared__ WgReduceDesc s_desc[kWgMaxDesc];
  {
    const int* src = reinterpret_cast<const int*>(tab);
    int* dst = reinterpret_cast<int*>(s_desc);
    for (int i = threadIdx.x; i < n * (int)(sizeof(WgReduceDesc) / 4); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  int di = 0;
  for (int t = tile0 + (int)blockIdx.x; t < tile1; t += (int)gridDim.x) {
    while (di + 1 < n && s_desc[di + 1].block0 <= t) ++di;
    const WgReduceDesc d = s_desc[di];
    if (d.fine) {
      wgrad_reduce_tile_fine(d, t - d.block0, &part[0][0][0]);
      __syncthreads();                                // `part` is reused by the next tile
    } else {
      wgrad_reduce_tile(d, t - d.block0);
    }
  }
}

// Same reduce for SMALL weight tensors with many slices (the 5x5 layers: 4800 elements x 256 slices): one thread
// per element would walk the slices as one long latency chain, so 16 threads share an element (slice s -> thread
// s % 16, each in ascending order) and their partial sums are folded in a fixed order through LDS.
__global__ __launch_bounds__(256) void splitk_reduce_wgrad_small(const float* __restrict__ slab, float* __restrict__ dw,
                                                                 int Co, int Ci, int KK, int Cip, int cb, int Np,
                                                                 size_t slab_stride, int splits, int accumulate,
                                                                 int swapped) {
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const size_t total = (size_t)Co * Ci * KK, i = (size_t)blockIdx.x * 16 + e;
  float s = 0.f;
  if (i < total) {
    int tap = (int)(i % KK);
    const size_t r = i / KK;
    int ci = (int)(r % Ci), co = (int)(r / Ci);
    if (swapped) {
      const int tmp = ci;
      ci = co, co = tmp, tap = KK - 1 - tap;
    }
    int col;
    if (cb == 128) {
      const int per_tap = Cip / 128;
      col = (tap * per_tap + ci / 128) * 128 + (ci & 127);
    } else {
      const int tpb = 128 / cb;
      col = (tap / tpb) * 128 + (tap % tpb) * cb + ci;
    }
    const float* p = slab + (size_t)co * Np + col;
    for (int k = g; k < splits; k += 16) s += p[(size_t)k * slab_stride];
  }
  part[g][e] = s;
  __syncthreads();
  if (g == 0 && i < total) {
    float t = accumulate ? dw[i] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][e];
    dw[i] = t;
  }
}

// dw[co][ci][tap] (+)= sum_s slab[s][co][column(tap, ci)]
// swapped: the GEMM ran with the operands exchanged (rows = ci, columns = (flipped tap, co)); Co/Ci here are
// always those of dw
__global__ void splitk_reduce_wgrad(const float* __restrict__ slab, float* __restrict__ dw, int Co, int Ci, int KK,
                                    int Cip, int cb, int Np, size_t slab_stride, int splits, int accumulate,
                                    int swapped) {
  const size_t total = (size_t)Co * Ci * KK;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int tap = (int)(i % KK);
    const size_t r = i / KK;
    int ci = (int)(r % Ci), co = (int)(r / Ci);
    if (swapped) {
      const int tmp = ci;
      ci = co, co = tmp, tap = KK - 1 - tap;
    }
    int col;
    if (cb == 128) {
      const int per_tap = Cip / 128;
      col = (tap * per_tap + ci / 128) * 128 + (ci & 127);
    } else {
      const int tpb = 128 / cb;
      col = (tap / tpb) * 128 + (tap % tpb) * cb + ci;
    }
    dw[i] = fold_strided(accumulate ? dw[i] : 0.f, slab + (size_t)co * Np + col, slab_stride, splits);
  }
}

// db[c] (+)= sum over (b, hw) of dy[b][c][hw]: grid (C, splits) partials + fixed-order combine
__global__ __launch_bounds__(256) void bias_grad_partial(const float* __restrict__ dy, double* __restrict__ part, int B,
                                                         int C, int HW, int splits) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const size_t total = (size_t)B * HW;
  const size_t chunk = (total + splits - 1) / splits;
  const size_t beg = (size_t)s * chunk, end = beg + chunk < total ? beg + chunk : total;
  double acc = 0.0;
  for (size_t i = beg + threadIdx.x; i < end; i += 256) {
    const size_t b = i / HW, hw = i - b * HW;
    acc += (double)dy[(b * C + c) * HW + hw];
  }
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) part[(size_t)s * C + c] = acc;
}
__global__ void bias_grad_combine(const double* __restrict__ part, float* __restrict__ db, int C, int splits,
                                  int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double s = fold_strided(0.0, part + c, (size_t)C, splits);
  db[c] = (accumulate ? db[c] : 0.f) + (float)s;
}

// ------------------------------------------------------------------------------ host side
static inline int pad16(int c) { return (c + 15) & ~15; }

struct FwdPlan {
  int bm, bn, mt, nt, cip, ktiles, splits, kps;
};

static FwdPlan plan_fwd(int B, int Ci, int H, int W, int Co, int KS) {
  FwdPlan p;
  const long long N = (long long)B * H * W;
  p.bm = tile_rows_for(Co);
  p.bn = p.bm == 128 ? 128 : 256;
  p.mt = cdiv(Co, p.bm);
  p.nt = (int)((N + p.bn - 1) / p.bn);
  p.cip = pad16(Ci);
  p.ktiles = KS * KS * (p.cip / 16);
  const int tiles = p.mt * p.nt;
  int splits = 1;
  if (tiles < 192 && p.ktiles >= 8) {
    splits = cdiv(512, tiles);
    if (splits > p.ktiles / 4) splits = p.ktiles / 4;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  p.kps = cdiv(p.ktiles, splits);
  p.splits = cdiv(p.ktiles, p.kps);
  return p;
}

static inline bool wgrad_swapped(int Ci, int Co) { return Co <= 4 && Ci > 4; }

struct WgPlan {
  int bm, cb, cip, mt, nt, tiles, ktiles, splits, kps;
};

static WgPlan plan_wgrad(int B, int Ci, int H, int W, int Co, int KS) {
  WgPlan p;
  const long long Ktot = (long long)B * H * W;
  p.bm = tile_rows_for(Co);
  p.cb = Ci <= 4 ? 4 : (Ci <= 16 ? 16 : (Ci <= 32 ? 32 : (Ci <= 64 ? 64 : 128)));
  p.cip = p.cb < 128 ? p.cb : (Ci + 127) / 128 * 128;
  p.mt = cdiv(Co, p.bm);
  p.nt = p.cb == 128 ? KS * KS * (p.cip / 128) : cdiv(KS * KS, 128 / p.cb);
  p.tiles = p.mt * p.nt;
  p.ktiles = (int)((Ktot + 31) / 32);
  int splits = cdiv(768, p.tiles);
  if (p.cb == 4) splits = cdiv(256, p.tiles);   // one 128-column tile: a block per CU with long K slices; the slab reduce shrinks with the splits
  if (splits > p.ktiles / 8) splits = p.ktiles / 8;
  if (splits > 256) splits = 256;
  if (splits < 1) splits = 1;
  p.kps = cdiv(p.ktiles, splits);
  p.splits = cdiv(p.ktiles, p.kps);
  return p;
}

template <int KS, int BM, int BN, int WM, int WN>
static void launch_fwd_cfg(const ConvArgs& a, int splits, int up2, hipStream_t st) {
  dim3 grid(cdiv(a.nt, 8) * 8 * a.mt, splits), block(WM * WN * 64);
  const bool tail = (a.Ci & 15) != 0;
  if (up2 && tail)
    launch_timed((conv_fwd_kernel<KS, BM, BN, WM, WN, true, true>), grid, block, 0, st, a);
  else if (up2)
    launch_timed((conv_fwd_kernel<KS, BM, BN, WM, WN, true, false>), grid, block, 0, st, a);
  else if (tail)
    launch_timed((conv_fwd_kernel<KS, BM, BN, WM, WN, false, true>), grid, block, 0, st, a);
  else
    launch_timed((conv_fwd_kernel<KS, BM, BN, WM, WN, false, false>), grid, block, 0, st, a);
}

template <int KS>
static void launch_fwd(const ConvArgs& a, int bm, int splits, int up2, hipStream_t st) {
  if (bm == 32)
    launch_fwd_cfg<KS, 32, 256, 1, 4>(a, splits, up2, st);
  else if (bm == 64)
    launch_fwd_cfg<KS, 64, 256, 1, 4>(a, splits, up2, st);
  else
    launch_fwd_cfg<KS, 128, 128, 2, 2>(a, splits, up2, st);
}

template <int KS, int CB, bool UP2>
static void launch_wgrad_bm(const WgradArgs& a, int bm, hipStream_t st) {
  dim3 grid(cdiv(a.splits, 8) * 8 * a.tiles);
  if (bm == 32)
    launch_timed((conv_wgrad_kernel<KS, 32, CB, 1, 4, UP2, 1>), grid, dim3(256), 0, st, a);
  else if (bm == 64)
    launch_timed((conv_wgrad_kernel<KS, 64, CB, 1, 4, UP2, 1>), grid, dim3(256), 0, st, a);
  else
    launch_timed((conv_wgrad_kernel<KS, 128, CB, 2, 2, UP2, 1>), grid, dim3(256), 0, st, a);
}
template <int KS, bool UP2>
static void launch_wgrad_cb(const WgradArgs& a, int bm, int cb, hipStream_t st) {
  if (cb == 4)
    launch_wgrad_bm<KS, 4, UP2>(a, bm, st);
  else if (cb == 16)
    launch_wgrad_bm<KS, 16, UP2>(a, bm, st);
  else if (cb == 32)
    launch_wgrad_bm<KS, 32, UP2>(a, bm, st);
  else if (cb == 64)
    launch_wgrad_bm<KS, 64, UP2>(a, bm, st);
  else
    launch_wgrad_bm<KS, 128, UP2>(a, bm, st);
}
template <int KS>
static void launch_wgrad(const WgradArgs& a, int bm, int cb, int up2, hipStream_t st) {
  if (up2)
    launch_wgrad_cb<KS, true>(a, bm, cb, st);
  else
    launch_wgrad_cb<KS, false>(a, bm, cb, st);
}

static inline int pad32(int c) { return (c + 31) & ~31; }

struct FwdPlanB {
  int bm, bn, mt, nt, cip, ktiles, splits, kps;
};
// blocks aimed at when K is split: one block per CU -- half the slabs and prologues of a 512-block split (measured +2 % step)
constexpr int kFwdBSplitTarget = 256;
static FwdPlanB plan_fwd_b(int B, int Ci, int H, int W, int Co, int KS) {
  FwdPlanB p;
  const long long N = (long long)B * H * W;
  p.bm = Co <= 64 ? 64 : 128;
  p.bn = p.bm == 128 ? 128 : 256;
  p.mt = cdiv(Co, p.bm);
  p.nt = (int)((N + p.bn - 1) / p.bn);
  p.cip = pad32(Ci);
  p.ktiles = KS * KS * (p.cip / 32);
  const int tiles = p.mt * p.nt;
  int splits = 1;
  if (tiles < 192 && p.ktiles >= 8) {
    splits = cdiv(kFwdBSplitTarget, tiles);
    if (splits > p.ktiles / 4) splits = p.ktiles / 4;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  p.kps = cdiv(p.ktiles, splits);
  p.splits = cdiv(p.ktiles, p.kps);
  return p;
}

template <int KS, int NS>
static void launch_fwd_b(const ConvArgsB& a, int bm, int splits, int up2, hipStream_t st) {
  dim3 grid(cdiv(a.nt, 8) * 8 * a.mt, splits), blk(512);
  if (bm == 64) {
    if (up2) launch_timed((conv_fwd_bf16s_ws_kernel<KS, 64, 256, 1, 4, true, NS>), grid, blk, 0, st, a);
    else launch_timed((conv_fwd_bf16s_ws_kernel<KS, 64, 256, 1, 4, false, NS>), grid, blk, 0, st, a);
  } else {
    if (up2) launch_timed((conv_fwd_bf16s_ws_kernel<KS, 128, 128, 2, 2, true, NS>), grid, blk, 0, st, a);
    else launch_timed((conv_fwd_bf16s_ws_kernel<KS, 128, 128, 2, 2, false, NS>), grid, blk, 0, st, a);
  }
}

template <int KS, int BM, int BN, int WM, int WN, bool UP2, int NS, int NSTAGE, bool F16>
static void launch_fwd_p_cfg(const ConvArgsP& a, int splits, hipStream_t st) {
  constexpr size_t lds = (size_t)NSTAGE * NS * 4 * (BM + BN) * 16;
  static_assert(lds <= 160 * 1024, "LDS ring too large");
  dim3 grid(cdiv(a.nt, 8) * 8 * a.mt, splits), blk(64 * (WM * WN + 4));
  if constexpr (NS == 2) {
    if (F16 || band_m16()) {
      auto k16 = conv_fwd_bf16p_kernel<KS, BM, BN, WM, WN, UP2, NS, NSTAGE, true, F16>;
      static bool attr16 = false;
      if (!attr16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr16 = true;
      }
      launch_timed(k16, grid, blk, lds, st, a);
      return;
    }
  }
  if constexpr (!F16) {
    auto kern = conv_fwd_bf16p_kernel<KS, BM, BN, WM, WN, UP2, NS, NSTAGE>;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    launch_timed(kern, grid, blk, lds, st, a);
  }
}
template <int KS, int NS, int NSTAGE, bool F16>
static void launch_fwd_p_st(const ConvArgsP& a, int bm, int splits, int up2, hipStream_t st) {
  if (bm == 64) {
    if (up2) launch_fwd_p_cfg<KS, 64, 256, 1, 4, true, NS, NSTAGE, F16>(a, splits, st);
    else launch_fwd_p_cfg<KS, 64, 256, 1, 4, false, NS, NSTAGE, F16>(a, splits, st);
  } else if (NS == 2 && g_opt.planes_mfma_waves == 8 && (F16 || band_m16())) {
    if constexpr (NS == 2) {   // eight MFMA waves of 64 x 32 (16x16x32 form)
      if (up2) launch_fwd_p_cfg<KS, 128, 128, 2, 4, true, NS, NSTAGE, F16>(a, splits, st);
      else launch_fwd_p_cfg<KS, 128, 128, 2, 4, false, NS, NSTAGE, F16>(a, splits, st);
    }
  } else {
    if (up2) launch_fwd_p_cfg<KS, 128, 128, 2, 2, true, NS, NSTAGE, F16>(a, splits, st);
    else launch_fwd_p_cfg<KS, 128, 128, 2, 2, false, NS, NSTAGE, F16>(a, splits, st);
  }
}
template <int KS, int NS, bool F16 = false>
static void launch_fwd_p(const ConvArgsP& a, int bm, int splits, int up2, hipStream_t st) {
  launch_fwd_p_st<KS, NS, NS == 2 ? 3 : 2, F16>(a, bm, splits, up2, st);   // ring depth: three stages where LDS allows (two planes)
}

struct WgPlanP {
  int bm, bn, tiles_m, tiles_n, steps, splits, sps;
  int kh;   // slabs per K slice (2 for the 64 x 64 tile: its wave groups keep separate slabs)
};
static WgPlanP plan_wgrad_p(int B, int Ci, int H, int W, int Co) {
  WgPlanP p;
  p.bm = (Co <= 64 && Ci >= 128) ? 64 : 128;    // 64 x 128 (co x ci) tiles when the output side is narrow
  p.bn = 8192 / p.bm;
  p.kh = 1;
  // 64 -> 64 channels: a 128-row tile would multiply 64 rows of zeros; 64 x 64 with the waves split over the k-steps
  if (Co <= 64 && Ci <= 64) p.bm = 64, p.bn = 64, p.kh = 2;
  // 128 x 128 tiles halve the operand bytes per MFMA (the 128 x 64 form runs near the ~30 B/clk/CU ingest limit) but
  // double the slab a block writes: measured faster only for the widest layers, where few K slices are needed
  if (p.bm == 128 && Ci >= 512 && Co >= 256 && W >= 8) p.bn = 128;
  p.tiles_m = cdiv(Co, p.bm), p.tiles_n = cdiv(Ci, p.bn);
  p.steps = (int)(((long long)B * H * W) / 64);
  const int T = p.tiles_m * p.tiles_n * 3;
  constexpr int target = 256;   // blocks aimed at: one 768-thread block per CU
  // blocks are launched in runs of 24 (8 (tile, K slice) groups x 3 filter rows, see the kernel): keep the padded grid
  // within one block per CU
  int splits = (8 * (target / 24)) / (T / 3);
  if (splits > p.steps / 2) splits = p.steps / 2;
  if (splits < 1) splits = 1;
  p.sps = cdiv(p.steps, splits);
  p.splits = cdiv(p.steps, p.sps);
  return p;
}

template <int LOG2W, bool UP2, int BM, int BN, int KH, bool F16>
static void launch_wgrad_p_cfg(const WgradArgsP& a, int blocks, hipStream_t st) {
  constexpr int W = 1 << LOG2W, NP = (64 >> LOG2W) * (W + 2), PXB = ((NP + 11) / 16) * 16 + 4;
  constexpr size_t stage_bytes = (size_t)(2 * (BM / 8) * 68 + 2 * (BN / 8) * PXB) * 16;
  constexpr int NST = 3 * stage_bytes <= 160 * 1024 ? 3 : 2;     // three stages where LDS allows (W = 32 / 64 bands)
  constexpr size_t lds = NST * stage_bytes;
  if constexpr (lds <= 160 * 1024) {
    constexpr int NLW = 4;   // loader waves (eight were measured no faster, and spill in the 128 x 128 form)
    // 16x16x32 by default (tools/wgrad_m16_bench.py, c2 shapes, f16x3: -7.8 % summed over the layers; +14 % on the
    // 128-column tiles, +9 % on the 4x4 layers, within -3.5 % on two shapes)
    const bool m16 = g_opt.wgrad_m16 != 0;
    if (m16) {
      auto kern = conv_wgrad_bf16p_kernel<LOG2W, UP2, BM, BN, NST, NLW, KH, F16, true>;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
      }
      launch_timed(kern, dim3(blocks), dim3(512 + 64 * NLW), lds, st, a);
      return;
    }
    auto kern = conv_wgrad_bf16p_kernel<LOG2W, UP2, BM, BN, NST, NLW, KH, F16>;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    launch_timed(kern, dim3(blocks), dim3(512 + 64 * NLW), lds, st, a);
  }
}
template <int LOG2W, bool F16>
static void launch_wgrad_p_t(const WgradArgsP& a, int bm, int bn, int up2, int blocks, hipStream_t st) {
  if (bm == 64 && bn == 64) {
    if (up2) launch_wgrad_p_cfg<LOG2W, true, 64, 64, 2, F16>(a, blocks, st);
    else launch_wgrad_p_cfg<LOG2W, false, 64, 64, 2, F16>(a, blocks, st);
  } else if (bm == 64) {
    if (up2) launch_wgrad_p_cfg<LOG2W, true, 64, 128, 1, F16>(a, blocks, st);
    else launch_wgrad_p_cfg<LOG2W, false, 64, 128, 1, F16>(a, blocks, st);
  } else if (bn == 128) {
    if (up2) launch_wgrad_p_cfg<LOG2W, true, 128, 128, 1, F16>(a, blocks, st);
    else launch_wgrad_p_cfg<LOG2W, false, 128, 128, 1, F16>(a, blocks, st);
  } else {
    if (up2) launch_wgrad_p_cfg<LOG2W, true, 128, 64, 1, F16>(a, blocks, st);
    else launch_wgrad_p_cfg<LOG2W, false, 128, 64, 1, F16>(a, blocks, st);
  }
}
template <int LOG2W>
static void launch_wgrad_p(const WgradArgsP& a, int bm, int bn, int up2, int blocks, int f16, hipStream_t st) {
  if (f16) launch_wgrad_p_t<LOG2W, true>(a, bm, bn, up2, blocks, st);
  else launch_wgrad_p_t<LOG2W, false>(a, bm, bn, up2, blocks, st);
}

template <int KS, int CB, int NS>
static void launch_wgrad_b_bm(const WgradArgs& a, int bm, hipStream_t st) {
  dim3 grid(cdiv(a.splits, 8) * 8 * a.tiles);
  if (bm == 64)
    launch_timed((conv_wgrad_bf16s_kernel<KS, 64, CB, 1, 4, NS>), grid, dim3(256), 0, st, a);
  else
    launch_timed((conv_wgrad_bf16s_kernel<KS, 128, CB, 2, 2, NS>), grid, dim3(256), 0, st, a);
}
template <int KS, int NS>
static void launch_wgrad_b(const WgradArgs& a, int bm, int cb, hipStream_t st) {
  if (cb == 32)
    launch_wgrad_b_bm<KS, 32, NS>(a, bm, st);
  else if (cb == 64)
    launch_wgrad_b_bm<KS, 64, NS>(a, bm, st);
  else
    launch_wgrad_b_bm<KS, 128, NS>(a, bm, st);
}

static int check_dims(const char* name, int B, int Ci, int H, int W, int Co, int KS) {
  if (!(KS == 1 || KS == 3 || KS == 5)) return fail("%s: kernel size must be 1, 3 or 5 (got %lld)", name, KS);
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return fail("%s: empty or negative dimension", name);
  const long long in_elems = (long long)B * pad16(Ci) * H * W, out_elems = (long long)B * Co * H * W;
  if (in_elems >= (1LL << 29) || out_elems >= (1LL << 29))
    return fail("%s: tensor too large for 31-bit buffer offsets (%lld / %lld elements)", name, in_elems, out_elems);
  return 0;
}

constexpr int kBiasSplitsMax = 256;
static inline int bias_splits(int B, int C, int HW) {
  int s = cdiv(1024, C);
  const size_t maxs = cdivz((size_t)B * HW, 2048);
  if ((size_t)s > maxs) s = (int)maxs;
  if (s > kBiasSplitsMax) s = kBiasSplitsMax;
  return s < 1 ? 1 : s;
}

}  // namespace itcv

using namespace itcv;

extern "C" {

size_t itcv_conv2d_packed_weight_elems(int Co, int Ci, int KS, int for_dgrad) {
  const int M = for_dgrad ? Ci : Co, C = for_dgrad ? Co : Ci;
  const int bm = tile_rows_for(M);
  return (size_t)KS * KS * pad16(C) * (cdiv(M, bm) * bm);
}

int itcv_conv2d_pack_weight(const float* w, float* wp, int Co, int Ci, int KS, int for_dgrad, void* stream) {
  ITCV_REQUIRE(w && wp && Co > 0 && Ci > 0 && (KS == 1 || KS == 3 || KS == 5), "itcv_conv2d_pack_weight");
  const int M = for_dgrad ? Ci : Co, C = for_dgrad ? Co : Ci;
  const int bm = tile_rows_for(M), Mp = cdiv(M, bm) * bm, Cip = pad16(C);
  const size_t total = (size_t)KS * KS * Cip * Mp;
  const int blocks = (int)(cdivz(total, 256) < 4096 ? cdivz(total, 256) : 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, S(stream), w, wp, Co, Ci, KS * KS, for_dgrad,
                     C, M, Cip, Mp);
  ITCV_CHECK_LAUNCH("itcv_conv2d_pack_weight");
  return 0;
}

size_t itcv_conv2d_fwd_workspace(int B, int Ci, int H, int W, int Co, int KS) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return 0;
  const FwdPlan p = plan_fwd(B, Ci, H, W, Co, KS);
  return p.splits > 1 ? (size_t)p.splits * B * Co * H * W * sizeof(float) : 0;
}

int itcv_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, int B, int Ci, int H, int W,
                    int Co, int KS, int up2, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_dims("itcv_conv2d_fwd", B, Ci, H, W, Co, KS)) return e;
  ITCV_REQUIRE(x && wp && y, "itcv_conv2d_fwd");
  if (up2) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_conv2d_fwd(up2)");
  const FwdPlan p = plan_fwd(B, Ci, H, W, Co, KS);
  const size_t out_elems = (size_t)B * Co * H * W;
  if (p.splits > 1 && (!ws || ws_bytes < (size_t)p.splits * out_elems * sizeof(float)))
    return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_fwd",
                (long long)((size_t)p.splits * out_elems * sizeof(float)));
  ConvArgs a;
  a.x = x;
  a.wp = wp;
  a.bias = p.splits > 1 ? nullptr : bias;
  a.y = p.splits > 1 ? static_cast<float*>(ws) : y;
  a.B = B, a.Ci = Ci, a.H = H, a.W = W, a.Co = Co;
  a.Cip = p.cip;
  a.Mp = p.mt * p.bm;
  a.N = B * H * W;
  a.mt = p.mt, a.nt = p.nt, a.ktiles = p.ktiles, a.ktiles_per_split = p.kps, a.cpt = p.cip / 16;
  a.x_bytes = (uint32_t)((size_t)B * Ci * (up2 ? (H / 2) * (W / 2) : H * W) * sizeof(float));
  a.slab_stride = p.splits > 1 ? out_elems : 0;
  hipStream_t st = S(stream);
  {
    ProfScope prof(st, 0, KS, p.bm, up2 ? 1 : 0, 0, 2.0 * B * H * W * (double)Co * Ci * KS * KS);
    if (KS == 1)
      launch_fwd<1>(a, p.bm, p.splits, up2, st);
    else if (KS == 3)
      launch_fwd<3>(a, p.bm, p.splits, up2, st);
    else
      launch_fwd<5>(a, p.bm, p.splits, up2, st);
  }
  ITCV_CHECK_LAUNCH("itcv_conv2d_fwd");
  if (p.splits > 1) {
    const int blocks = (int)(cdivz(out_elems, 256) < 2048 ? cdivz(out_elems, 256) : 2048);
    hipLaunchKernelGGL(splitk_reduce_fwd, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), bias, y,
                       out_elems, out_elems, p.splits, H * W, Co);
    ITCV_CHECK_LAUNCH("itcv_conv2d_fwd(reduce)");
  }
  return 0;
}

// ---- split-bf16 forward / data-gradient (throughput mode) -----------------------------------
int itcv_conv2d_bf16s_supported(int Ci, int Co, int KS) {
  return (KS == 1 || KS == 3) && Ci >= 32 && Ci % 32 == 0 && Co >= 33;
}

static inline int planes_of_fmt(int ns) { return ns == ITCV_PLANES_F16X2 ? 2 : ns; }   // planes per tensor of a format code
static inline bool fmt_ok(int ns) { return ns == 2 || ns == 3 || ns == ITCV_PLANES_F16X2; }

size_t itcv_conv2d_packed_weight_bytes_bf16s(int Co, int Ci, int KS, int for_dgrad, int ns) {
  const int M = for_dgrad ? Ci : Co, C = for_dgrad ? Co : Ci;
  const int bm = M <= 64 ? 64 : 128;
  return (size_t)KS * KS * (pad32(C) / 32) * planes_of_fmt(ns) * 4 * (cdiv(M, bm) * bm) * 16;
}

int itcv_conv2d_pack_weight_bf16s(const float* w, void* wp, int Co, int Ci, int KS, int for_dgrad, int ns,
                                  void* stream) {
  ITCV_REQUIRE(w && wp && Co > 0 && Ci > 0 && (KS == 1 || KS == 3) && fmt_ok(ns), "itcv_conv2d_pack_weight_bf16s");
  const int M = for_dgrad ? Ci : Co, C = for_dgrad ? Co : Ci;
  const int bm = M <= 64 ? 64 : 128, Mp = cdiv(M, bm) * bm, cpt = pad32(C) / 32;
  const int blocks = (Mp / 32) * cpt;   // one block per 32 x 32-channel tile
  if (ns == ITCV_PLANES_F16X2)
    hipLaunchKernelGGL((pack_weight_bf16s_kernel<2, true>), dim3(blocks), dim3(256), 0, S(stream), w, static_cast<u32x4*>(wp),
                       Co, Ci, KS * KS, for_dgrad, C, M, cpt, Mp);
  else if (ns == 2)
    hipLaunchKernelGGL(pack_weight_bf16s_kernel<2>, dim3(blocks), dim3(256), 0, S(stream), w, static_cast<u32x4*>(wp),
                       Co, Ci, KS * KS, for_dgrad, C, M, cpt, Mp);
  else
    hipLaunchKernelGGL(pack_weight_bf16s_kernel<3>, dim3(blocks), dim3(256), 0, S(stream), w, static_cast<u32x4*>(wp),
                       Co, Ci, KS * KS, for_dgrad, C, M, cpt, Mp);
  ITCV_CHECK_LAUNCH("itcv_conv2d_pack_weight_bf16s");
  return 0;
}

size_t itcv_pack_desc_bytes(void) { return sizeof(PackDesc); }

int itcv_conv2d_pack_desc_bf16s(void* host_desc, const float* w, void* wp, int Co, int Ci, int KS, int for_dgrad,
                                int ns, int block0) {
  if (!(host_desc && w && wp && Co > 0 && Ci > 0 && (KS == 1 || KS == 3) && fmt_ok(ns) && block0 >= 0)) {
    fail("%s: bad argument (pointers, Co/Ci > 0, KS in {1,3}, ns in {2,3,4}, block0 >= 0)", "itcv_conv2d_pack_desc_bf16s");
    return -1;     // the success value is a block count
  }
  const int M = for_dgrad ? Ci : Co, C = for_dgrad ? Co : Ci;
  const int bm = M <= 64 ? 64 : 128, Mp = cdiv(M, bm) * bm, cpt = pad32(C) / 32;
  PackDesc d;
  memset(&d, 0, sizeof(d));
  d.w = w, d.wp = static_cast<u32x4*>(wp);
  d.Ci = Ci, d.KK = KS * KS, d.for_dgrad = for_dgrad, d.C = C, d.M = M, d.cpt = cpt, d.Mp = Mp;
  d.block0 = block0;
  d.nblocks = (Mp / 32) * cpt;          // one block per 32 x 32-channel tile
  memcpy(host_desc, &d, sizeof(d));
  return d.nblocks;      // > 0: the number of blocks this layer adds to the launch
}

int itcv_conv2d_pack_weights_bf16s(const void* dev_table, int n, int total_blocks, int ns, void* stream) {
  ITCV_REQUIRE(dev_table && n > 0 && total_blocks > 0 && fmt_ok(ns), "itcv_conv2d_pack_weights_bf16s");
  const PackDesc* tab = static_cast<const PackDesc*>(dev_table);
  if (ns == ITCV_PLANES_F16X2)
    hipLaunchKernelGGL((pack_weights_bf16s_table_kernel<2, true>), dim3(total_blocks), dim3(256), 0, S(stream), tab, n);
  else if (ns == 2)
    hipLaunchKernelGGL(pack_weights_bf16s_table_kernel<2>, dim3(total_blocks), dim3(256), 0, S(stream), tab, n);
  else
    hipLaunchKernelGGL(pack_weights_bf16s_table_kernel<3>, dim3(total_blocks), dim3(256), 0, S(stream), tab, n);
  ITCV_CHECK_LAUNCH("itcv_conv2d_pack_weights_bf16s");
  return 0;
}

size_t itcv_conv2d_fwd_bf16s_workspace(int B, int Ci, int H, int W, int Co, int KS) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return 0;
  const FwdPlanB p = plan_fwd_b(B, Ci, H, W, Co, KS);
  return p.splits > 1 ? (size_t)p.splits * B * Co * H * W * sizeof(float) : 0;
}

size_t itcv_conv2d_fwd_bf16p_workspace(int B, int Ci, int H, int W, int Co, int KS, int ns) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return 0;
  const FwdPlanP2 p2 = plan_fwd_p2(B, Ci, H, W, Co, KS, planes_of_fmt(ns));
  if (p2.ok) return p2.splits > 1 ? (size_t)p2.splits * B * Co * H * W * sizeof(float) : 0;
  return itcv_conv2d_fwd_bf16s_workspace(B, Ci, H, W, Co, KS);
}

int itcv_conv2d_fwd_bf16s(const float* x, const void* wp, const float* bias, float* y, int B, int Ci, int H, int W,
                          int Co, int KS, int up2, int ns, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_dims("itcv_conv2d_fwd_bf16s", B, Ci, H, W, Co, KS)) return e;
  ITCV_REQUIRE(x && wp && y && (ns == 2 || ns == 3), "itcv_conv2d_fwd_bf16s");
  if (!itcv_conv2d_bf16s_supported(Ci, Co, KS))
    return fail("%s: shape not supported by the split-bf16 kernel (Ci %% 32, Co > 32, KS 1/3)", "itcv_conv2d_fwd_bf16s");
  if (up2) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_conv2d_fwd_bf16s(up2)");
  const FwdPlanB p = plan_fwd_b(B, Ci, H, W, Co, KS);
  const size_t out_elems = (size_t)B * Co * H * W;
  if (p.splits > 1 && (!ws || ws_bytes < (size_t)p.splits * out_elems * sizeof(float)))
    return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_fwd_bf16s",
                (long long)((size_t)p.splits * out_elems * sizeof(float)));
  ConvArgsB a;
  a.x = x;
  a.wp = static_cast<const u32x4*>(wp);
  a.bias = p.splits > 1 ? nullptr : bias;
  a.y = p.splits > 1 ? static_cast<float*>(ws) : y;
  a.B = B, a.Ci = Ci, a.H = H, a.W = W, a.Co = Co;
  a.Mp = p.mt * p.bm;
  a.N = B * H * W;
  a.mt = p.mt, a.nt = p.nt, a.ktiles = p.ktiles, a.ktiles_per_split = p.kps, a.cpt = p.cip / 32;
  a.x_bytes = (uint32_t)((size_t)B * Ci * (up2 ? (H / 2) * (W / 2) : H * W) * sizeof(float));
  a.slab_stride = p.splits > 1 ? out_elems : 0;
#ifdef ITCV_DIAG
  a.ablate = diag_ablate();
#endif
  hipStream_t st = S(stream);
  {
    ProfScope prof(st, 1, KS, p.bm, up2 ? 1 : 0, ns, 2.0 * B * H * W * (double)Co * Ci * KS * KS);
    if (KS == 1) {
      if (ns == 2) launch_fwd_b<1, 2>(a, p.bm, p.splits, up2, st);
      else launch_fwd_b<1, 3>(a, p.bm, p.splits, up2, st);
    } else {
      if (ns == 2) launch_fwd_b<3, 2>(a, p.bm, p.splits, up2, st);
      else launch_fwd_b<3, 3>(a, p.bm, p.splits, up2, st);
    }
  }
  ITCV_CHECK_LAUNCH("itcv_conv2d_fwd_bf16s");
  if (p.splits > 1) {
    const int blocks = (int)(cdivz(out_elems, 256) < 2048 ? cdivz(out_elems, 256) : 2048);
    hipLaunchKernelGGL(splitk_reduce_fwd, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), bias, y,
                       out_elems, out_elems, p.splits, H * W, Co);
    ITCV_CHECK_LAUNCH("itcv_conv2d_fwd_bf16s(reduce)");
  }
  return 0;
}

size_t itcv_planes_bytes(int B, int C, int HW, int ns) {
  if (B <= 0 || C <= 0 || HW <= 0 || (C & 7) || !fmt_ok(ns)) return 0;
  // fp16 planes carry their scale record {S, 1/S} behind the last plane
  return (size_t)planes_of_fmt(ns) * B * (C / 8) * HW * 16 + (ns == ITCV_PLANES_F16X2 ? sizeof(ScaleRec) : 0);
}

int itcv_absmax(const float* x, size_t n, float* parts, void* stream) {
  ITCV_REQUIRE(x && parts && n > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "itcv_absmax");
  hipLaunchKernelGGL(absmax_kernel, dim3(kAbsmaxParts), dim3(256), 0, S(stream), x, n, parts);
  ITCV_CHECK_LAUNCH("itcv_absmax");
  return 0;
}

int itcv_split_planes_scaled(const float* x, void* planes, int B, int C, int HW, int ns, const float* amax, void* stream) {
  ITCV_REQUIRE(x && planes && B > 0 && C > 0 && HW > 0 && (C & 7) == 0 && fmt_ok(ns), "itcv_split_planes");
  ITCV_REQUIRE(!amax || ns == ITCV_PLANES_F16X2, "itcv_split_planes(a scale only applies to fp16 planes)");
  const size_t total = (size_t)B * (C / 8) * HW;
  const int blocks = (int)(cdivz(total, 256) < 8192 ? cdivz(total, 256) : 8192);
  u32x4* pl = static_cast<u32x4*>(planes);
  if (ns == ITCV_PLANES_F16X2)
    hipLaunchKernelGGL((split_planes_kernel<2, true>), dim3(blocks), dim3(256), 0, S(stream), x, pl, B, C / 8, HW, amax);
  else if (ns == 2)
    hipLaunchKernelGGL(split_planes_kernel<2>, dim3(blocks), dim3(256), 0, S(stream), x, pl, B, C / 8, HW, amax);
  else
    hipLaunchKernelGGL(split_planes_kernel<3>, dim3(blocks), dim3(256), 0, S(stream), x, pl, B, C / 8, HW, amax);
  ITCV_CHECK_LAUNCH("itcv_split_planes");
  return 0;
}
int itcv_split_planes(const float* x, void* planes, int B, int C, int HW, int ns, void* stream) {
  return itcv_split_planes_scaled(x, planes, B, C, HW, ns, nullptr, stream);
}

// Same contract as itcv_conv2d_fwd_bf16s, with the input given as pre-split planes
// ([ns][B][Ci/8][Hs][Ws] 16-byte chunks; Hs,Ws = H/2,W/2 when up2).  Workspace: itcv_conv2d_fwd_bf16s_workspace.
int itcv_conv2d_fwd_bf16p(const void* xplanes, const void* wp, const float* bias, float* y, int B, int Ci, int H,
                          int W, int Co, int KS, int up2, int ns, void* ws, size_t ws_bytes, void* stream) {
  return itcv_conv2d_fwd_bf16p_st(xplanes, wp, bias, y, B, Ci, H, W, Co, KS, up2, ns, nullptr, ws, ws_bytes, stream);
}

// Number of pixel tiles for which itcv_conv2d_fwd_bf16p_st reports per-channel output sums (0: the kernel chosen for
// this shape cannot -- split-K slabs, or a tile form without the staged epilogue).
int itcv_conv2d_fwd_bf16p_stat_tiles(int B, int Ci, int H, int W, int Co, int KS, int ns) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0 || !itcv_conv2d_bf16s_supported(Ci, Co, KS)) return 0;
  if (ns == ITCV_PLANES_F16X2) return 0;     // the staged epilogue exists for the bf16 32x32x16 form only
  const FwdPlanP2 p2 = plan_fwd_p2(B, Ci, H, W, Co, KS, ns);
  return (p2.ok && p2.splits == 1 && p2.bn == 256) ? p2.nt : 0;
}

int itcv_conv2d_fwd_bf16p_st(const void* xplanes, const void* wp, const float* bias, float* y, int B, int Ci, int H,
                             int W, int Co, int KS, int up2, int ns, float* tile_stats, void* ws, size_t ws_bytes,
                             void* stream) {
  if (int e = check_dims("itcv_conv2d_fwd_bf16p", B, Ci, H, W, Co, KS)) return e;
  ITCV_REQUIRE(xplanes && wp && y && fmt_ok(ns), "itcv_conv2d_fwd_bf16p");
  const int f16 = ns == ITCV_PLANES_F16X2;
  ns = planes_of_fmt(ns);
  if (tile_stats && (f16 || !itcv_conv2d_fwd_bf16p_stat_tiles(B, Ci, H, W, Co, KS, ns)))
    return fail("%s: tile statistics are not available for this shape / format (see itcv_conv2d_fwd_bf16p_stat_tiles)",
                "itcv_conv2d_fwd_bf16p_st");
  const size_t in_plane = (size_t)B * (Ci / 8) * (up2 ? (H / 2) * (W / 2) : H * W);
  const ScaleRec* xrec = f16 ? reinterpret_cast<const ScaleRec*>(static_cast<const u32x4*>(xplanes) + 2 * in_plane) : nullptr;
  if (!itcv_conv2d_bf16s_supported(Ci, Co, KS))
    return fail("%s: shape not supported by the split-bf16 kernel (Ci %% 32, Co > 32, KS 1/3)", "itcv_conv2d_fwd_bf16p");
  if (up2) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_conv2d_fwd_bf16p(up2)");
  const size_t out_elems = (size_t)B * Co * H * W;
  const FwdPlanP2 p2 = plan_fwd_p2(B, Ci, H, W, Co, KS, ns);
  if (p2.ok) {   // band kernel: tap reuse through LDS
    if (p2.splits > 1 && (!ws || ws_bytes < (size_t)p2.splits * out_elems * sizeof(float)))
      return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_fwd_bf16p",
                  (long long)((size_t)p2.splits * out_elems * sizeof(float)));
    ConvArgsP2 a;
    a.xp = static_cast<const u32x4*>(xplanes);
    a.wp = static_cast<const u32x4*>(wp);
    a.bias = p2.splits > 1 ? nullptr : bias;
    a.y = p2.splits > 1 ? static_cast<float*>(ws) : y;
    a.B = B, a.Ci = Ci, a.H = H, a.Co = Co;
    a.Mp = p2.mt * p2.bm;
    a.N = B * H * W;
    a.mt = p2.mt, a.nt = p2.nt, a.cpt = p2.cpt, a.cpt_per_split = p2.cps;
    a.SR = p2.SR, a.NSEG = p2.NSEG, a.NP = p2.NP, a.NPC = p2.NPC, a.PXB = p2.PXB;
    a.h_shift = log2_exact(H);
    a.stats = tile_stats, a.stat_T = p2.nt;
    a.xscale = xrec;
    a.slab_stride = p2.splits > 1 ? out_elems : 0;
    a.plane_stride = (size_t)B * (Ci / 8) * (up2 ? (H / 2) * (W / 2) : H * W);
#ifdef ITCV_DIAG
    a.debug = (diag_ablate() & 64) ? 1 : 0;
#endif
    hipStream_t st = S(stream);
    {
      ProfScope prof(st, band_is_persistent(a, p2) ? 9 : 8, log2_exact(W), p2.bm, up2 ? 1 : 0, f16 ? ITCV_PLANES_F16X2 : ns,
                     2.0 * B * H * W * (double)Co * Ci * KS * KS);
      launch_fwd_p2(a, p2, W, up2, f16, st);
    }
    ITCV_CHECK_LAUNCH("itcv_conv2d_fwd_bf16p(band)");
    if (p2.splits > 1) {
      const int blocks = (int)(cdivz(out_elems, 256) < 2048 ? cdivz(out_elems, 256) : 2048);
      hipLaunchKernelGGL(splitk_reduce_fwd, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), bias, y,
                         out_elems, out_elems, p2.splits, H * W, Co);
      ITCV_CHECK_LAUNCH("itcv_conv2d_fwd_bf16p(reduce)");
    }
    return 0;
  }
  const FwdPlanB p = plan_fwd_b(B, Ci, H, W, Co, KS);
  if (p.splits > 1 && (!ws || ws_bytes < (size_t)p.splits * out_elems * sizeof(float)))
    return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_fwd_bf16p",
                (long long)((size_t)p.splits * out_elems * sizeof(float)));
  ConvArgsP a;
  a.xp = static_cast<const u32x4*>(xplanes);
  a.wp = static_cast<const u32x4*>(wp);
  a.bias = p.splits > 1 ? nullptr : bias;
  a.y = p.splits > 1 ? static_cast<float*>(ws) : y;
  a.B = B, a.Ci = Ci, a.H = H, a.W = W, a.Co = Co;
  a.Mp = p.mt * p.bm;
  a.N = B * H * W;
  a.mt = p.mt, a.nt = p.nt, a.ktiles = p.ktiles, a.ktiles_per_split = p.kps;
  a.slab_stride = p.splits > 1 ? out_elems : 0;
  a.plane_stride = in_plane;
  a.xscale = xrec;
#ifdef ITCV_DIAG
  a.ablate = diag_ablate();
#endif
  hipStream_t st = S(stream);
  {
    ProfScope prof(st, 6, KS, p.bm, up2 ? 1 : 0, f16 ? ITCV_PLANES_F16X2 : ns, 2.0 * B * H * W * (double)Co * Ci * KS * KS);
    if (KS == 1) {
      if (f16) launch_fwd_p<1, 2, true>(a, p.bm, p.splits, up2, st);
      else if (ns == 2) launch_fwd_p<1, 2>(a, p.bm, p.splits, up2, st);
      else launch_fwd_p<1, 3>(a, p.bm, p.splits, up2, st);
    } else {
      if (f16) launch_fwd_p<3, 2, true>(a, p.bm, p.splits, up2, st);
      else if (ns == 2) launch_fwd_p<3, 2>(a, p.bm, p.splits, up2, st);
      else launch_fwd_p<3, 3>(a, p.bm, p.splits, up2, st);
    }
  }
  ITCV_CHECK_LAUNCH("itcv_conv2d_fwd_bf16p");
  if (p.splits > 1) {
    const int blocks = (int)(cdivz(out_elems, 256) < 2048 ? cdivz(out_elems, 256) : 2048);
    hipLaunchKernelGGL(splitk_reduce_fwd, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), bias, y,
                       out_elems, out_elems, p.splits, H * W, Co);
    ITCV_CHECK_LAUNCH("itcv_conv2d_fwd_bf16p(reduce)");
  }
  return 0;
}

size_t itcv_conv2d_wgrad_workspace(int B, int Ci, int H, int W, int Co, int KS) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return 0;
  const WgPlan p = plan_wgrad(B, Ci, H, W, Co, KS);
  size_t n = (size_t)p.splits * Co * p.nt * 128 * sizeof(float);
  if (wgrad_swapped(Ci, Co)) {   // the call may run with its operands exchanged (see itcv_conv2d_wgrad)
    const WgPlan q = plan_wgrad(B, Co, H, W, Ci, KS);
    const size_t m = (size_t)q.splits * Ci * q.nt * 128 * sizeof(float);
    if (m > n) n = m;
  }
  return n;
}

int itcv_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int Ci, int H, int W, int Co, int KS,
                      int up2, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_dims("itcv_conv2d_wgrad", B, Ci, H, W, Co, KS)) return e;
  ITCV_REQUIRE(x && dy && dw, "itcv_conv2d_wgrad");
  if (up2) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_conv2d_wgrad(up2)");
  // Few output channels (the predict conv, 64 -> 3): exchange the operands so that the wide tensor is the
  // GEMM's row side and the 3-channel one the gathered side (4-channel column groups):
  //   dW[co][ci][tap] = sum_q x[ci][q] * dy[co][q - tap]  =  wgrad(x' = dy, dy' = x)[ci][co][KK-1-tap]
  const bool swapped = !up2 && wgrad_swapped(Ci, Co);
  const int Co_dw = Co, Ci_dw = Ci;
  if (swapped) {
    const float* tp = x;
    x = dy, dy = tp;
    const int tc = Ci;
    Ci = Co, Co = tc;
  }
  const WgPlan p = plan_wgrad(B, Ci, H, W, Co, KS);
  const size_t slab = (size_t)Co * p.nt * 128;
  if (!ws || ws_bytes < (size_t)p.splits * slab * sizeof(float))
    return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_wgrad",
                (long long)((size_t)p.splits * slab * sizeof(float)));
  WgradArgs a;
  a.x = x, a.dy = dy;
  a.out = static_cast<float*>(ws);
  a.B = B, a.Ci = Ci, a.H = H, a.W = W, a.Co = Co;
  a.Cip = p.cip;
  a.Np = p.nt * 128;
  a.Ktot = B * H * W;
  a.mt = p.mt, a.nt = p.nt, a.tiles = p.tiles;
  a.ktiles = p.ktiles, a.ktiles_per_split = p.kps, a.splits = p.splits;
  a.w_shift = log2_exact(W), a.hw_shift = log2_exact(H * W);
  a.x_bytes = (uint32_t)((size_t)B * Ci * (up2 ? (H / 2) * (W / 2) : H * W) * sizeof(float));
  a.dy_bytes = (uint32_t)((size_t)B * Co * H * W * sizeof(float));
  a.slab_stride = slab;
  hipStream_t st = S(stream);
  {
    ProfScope prof(st, 2, KS, p.bm, up2 ? 1 : 0, 0, 2.0 * B * H * W * (double)Co * Ci * KS * KS);
    if (KS == 1)
      launch_wgrad<1>(a, p.bm, p.cb, up2, st);
    else if (KS == 3)
      launch_wgrad<3>(a, p.bm, p.cb, up2, st);
    else
      launch_wgrad<5>(a, p.bm, p.cb, up2, st);
  }
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad");
  const size_t dw_elems = (size_t)Co * Ci * KS * KS;
  const int blocks = (int)(cdivz(dw_elems, 256) < 2048 ? cdivz(dw_elems, 256) : 2048);
  if (dw_elems <= 65536 && p.splits >= 32)
    hipLaunchKernelGGL(splitk_reduce_wgrad_small, dim3((int)cdivz(dw_elems, 16)), dim3(256), 0, st,
                       static_cast<const float*>(ws), dw, Co_dw, Ci_dw, KS * KS, p.cip, p.cb, a.Np, slab, p.splits,
                       accumulate, swapped ? 1 : 0);
  else
    hipLaunchKernelGGL(splitk_reduce_wgrad, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), dw, Co_dw,
                       Ci_dw, KS * KS, p.cip, p.cb, a.Np, slab, p.splits, accumulate, swapped ? 1 : 0);
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad(reduce)");
  return 0;
}

// ---- split-bf16 weight gradient ---------------------------------------------------------------
int itcv_conv2d_wgrad_bf16s_supported(int Ci, int H, int W, int Co, int KS) {
  return (KS == 1 || KS == 3) && Ci % 32 == 0 && W % 8 == 0 && Co >= 33 && H > 0;
}

int itcv_conv2d_wgrad_bf16s(const float* x, const float* dy, float* dw, int B, int Ci, int H, int W, int Co, int KS,
                            int ns, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_dims("itcv_conv2d_wgrad_bf16s", B, Ci, H, W, Co, KS)) return e;
  ITCV_REQUIRE(x && dy && dw && (ns == 2 || ns == 3), "itcv_conv2d_wgrad_bf16s");
  if (!itcv_conv2d_wgrad_bf16s_supported(Ci, H, W, Co, KS))
    return fail("%s: shape not supported by the split-bf16 kernel (Ci %% 32, W %% 8, Co > 32, KS 1/3)",
                "itcv_conv2d_wgrad_bf16s");
  WgPlan p = plan_wgrad(B, Ci, H, W, Co, KS);     // same tiling / slab layout as the fp32 kernel
  if (p.bm < 64) p.bm = 64, p.mt = cdiv(Co, 64), p.tiles = p.mt * p.nt;
  const size_t slab = (size_t)Co * p.nt * 128;
  if (!ws || ws_bytes < (size_t)p.splits * slab * sizeof(float))
    return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_wgrad_bf16s",
                (long long)((size_t)p.splits * slab * sizeof(float)));
  WgradArgs a;
  a.x = x, a.dy = dy;
  a.out = static_cast<float*>(ws);
  a.B = B, a.Ci = Ci, a.H = H, a.W = W, a.Co = Co;
  a.Cip = p.cip;
  a.Np = p.nt * 128;
  a.Ktot = B * H * W;
  a.mt = p.mt, a.nt = p.nt, a.tiles = p.tiles;
  a.ktiles = p.ktiles, a.ktiles_per_split = p.kps, a.splits = p.splits;
  a.w_shift = log2_exact(W), a.hw_shift = log2_exact(H * W);
  a.x_bytes = (uint32_t)((size_t)B * Ci * H * W * sizeof(float));
  a.dy_bytes = (uint32_t)((size_t)B * Co * H * W * sizeof(float));
  a.slab_stride = slab;
  hipStream_t st = S(stream);
  {
    ProfScope prof(st, 3, KS, p.bm, 0, ns, 2.0 * B * H * W * (double)Co * Ci * KS * KS);
    if (KS == 1) {
      if (ns == 2) launch_wgrad_b<1, 2>(a, p.bm, p.cb, st);
      else launch_wgrad_b<1, 3>(a, p.bm, p.cb, st);
    } else {
      if (ns == 2) launch_wgrad_b<3, 2>(a, p.bm, p.cb, st);
      else launch_wgrad_b<3, 3>(a, p.bm, p.cb, st);
    }
  }
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad_bf16s");
  const size_t dw_elems = (size_t)Co * Ci * KS * KS;
  const int blocks = (int)(cdivz(dw_elems, 256) < 2048 ? cdivz(dw_elems, 256) : 2048);
  hipLaunchKernelGGL(splitk_reduce_wgrad, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), dw, Co, Ci,
                     KS * KS, p.cip, p.cb, a.Np, slab, p.splits, accumulate, 0);
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad_bf16s(reduce)");
  return 0;
}

// ---- weight gradient on pre-split planes ------------------------------------------------------
int itcv_conv2d_wgrad_bf16p_supported(int B, int Ci, int H, int W, int Co, int KS) {
  if (KS != 3 || B <= 0 || Ci < 8 || Co < 8 || (Ci & 7) || (Co & 7)) return 0;
  if (log2_exact(W) < 2 || W > 256 || log2_exact(H) < 0) return 0;   // W > 64: 64-column segments of a row per step
  const long long px = (long long)B * H * W;
  if (px % 64) return 0;
  if (H * W < 64 && 64 % (H * W)) return 0;
  return 1;
}

size_t itcv_conv2d_wgrad_bf16p_workspace(int B, int Ci, int H, int W, int Co, int KS) {
  if (!itcv_conv2d_wgrad_bf16p_supported(B, Ci, H, W, Co, KS)) return 0;
  const WgPlanP p = plan_wgrad_p(B, Ci, H, W, Co);
  return (size_t)p.splits * p.kh * 9 * Co * Ci * sizeof(float);
}

// dw[Co][Ci][3][3] (+)= conv weight gradient from the pre-split planes of x ([2][B][Ci/8][Hs][Ws]; Hs,Ws =
// H/2,W/2 with up2) and dy ([2][B][Co/8][H][W]); bf16x3 arithmetic (two planes).
int itcv_conv2d_wgrad_bf16p(const void* xplanes, const void* dyplanes, float* dw, int B, int Ci, int H, int W, int Co,
                            int KS, int up2, int ns, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_dims("itcv_conv2d_wgrad_bf16p", B, Ci, H, W, Co, KS)) return e;
  ITCV_REQUIRE(xplanes && dyplanes && dw && (ns == 2 || ns == ITCV_PLANES_F16X2), "itcv_conv2d_wgrad_bf16p");
  const int f16 = ns == ITCV_PLANES_F16X2;
  if (!itcv_conv2d_wgrad_bf16p_supported(B, Ci, H, W, Co, KS))
    return fail("%s: shape not supported (KS 3, W a power of two in 4..256, H a power of two, C %% 8)", "itcv_conv2d_wgrad_bf16p");
  if (up2) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_conv2d_wgrad_bf16p(up2)");
  const WgPlanP p = plan_wgrad_p(B, Ci, H, W, Co);
  const size_t need = (size_t)p.splits * p.kh * 9 * Co * Ci * sizeof(float);
  if (!ws || ws_bytes < need) return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_wgrad_bf16p", (long long)need);
  WgradArgsP a;
  a.xp = static_cast<const u32x4*>(xplanes), a.dyp = static_cast<const u32x4*>(dyplanes);
  a.slab = static_cast<float*>(ws);
  a.B = B, a.Ci = Ci, a.H = H, a.W = W, a.Co = Co;
  a.tiles_m = p.tiles_m, a.tiles_n = p.tiles_n;
  a.steps = p.steps, a.steps_per_split = p.sps, a.splits = p.splits;
  a.h_shift = log2_exact(H);
  a.wseg_shift = W > 64 ? log2_exact(W) - 6 : 0;
  a.xplane = (size_t)B * (Ci / 8) * (up2 ? (H / 2) * (W / 2) : H * W);
  a.dyplane = (size_t)B * (Co / 8) * H * W;
  a.xscale = f16 ? reinterpret_cast<const ScaleRec*>(a.xp + 2 * a.xplane) : nullptr;
  a.dyscale = f16 ? reinterpret_cast<const ScaleRec*>(a.dyp + 2 * a.dyplane) : nullptr;
#ifdef ITCV_DIAG
  a.debug = (diag_ablate() & 64) ? 1 : 0;
#endif
  hipStream_t st = S(stream);
  a.groups = p.tiles_m * p.tiles_n * p.splits;
  const int blocks = cdiv(a.groups, 8) * 24;
  {
    ProfScope prof(st, 7, log2_exact(W), p.bm, up2 ? 1 : 0, ns, 2.0 * B * H * W * (double)Co * Ci * KS * KS);
    switch (log2_exact(W)) {
      case 2: launch_wgrad_p<2>(a, p.bm, p.bn, up2, blocks, f16, st); break;
      case 3: launch_wgrad_p<3>(a, p.bm, p.bn, up2, blocks, f16, st); break;
      case 4: launch_wgrad_p<4>(a, p.bm, p.bn, up2, blocks, f16, st); break;
      case 5: launch_wgrad_p<5>(a, p.bm, p.bn, up2, blocks, f16, st); break;
      default: launch_wgrad_p<6>(a, p.bm, p.bn, up2, blocks, f16, st); break;
    }
  }
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad_bf16p");
  if (accumulate == 2) return 0;     // deferred: the slabs stay in `ws` for itcv_wgrad_reduce_many
  const int coci = Co * Ci;
  const int rsplits = p.splits * p.kh;
  const bool fine = wgrad_reduce_fine(&rsplits, 1);
  hipLaunchKernelGGL(wgrad_p_reduce, dim3(wgrad_reduce_blocks(coci, fine)), dim3(256), 0, st, static_cast<const float*>(ws),
                     dw, coci, rsplits, accumulate, fine ? 1 : 0);
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad_bf16p(reduce)");
  return 0;
}

// Deferred reduce of itcv_conv2d_wgrad_bf16p(accumulate = 2) calls: slabs per call = itcv_conv2d_wgrad_bf16p_slabs.
int itcv_conv2d_wgrad_bf16p_slabs(int B, int Ci, int H, int W, int Co, int KS) {
  if (!itcv_conv2d_wgrad_bf16p_supported(B, Ci, H, W, Co, KS)) return 0;
  const WgPlanP p = plan_wgrad_p(B, Ci, H, W, Co);
  return p.splits * p.kh;
}
size_t itcv_wgrad_reduce_desc_bytes(void) { return sizeof(WgReduceDesc); }
int itcv_wgrad_reduce_desc(void* host_desc, const float* const* slabs, const int* splits, int nsrc, float* dw, int Co, int Ci,
                           int accumulate, int block0) {
  if (!(host_desc && slabs && splits && nsrc >= 1 && nsrc <= 4 && dw && Co > 0 && Ci > 0 && block0 >= 0)) {
    fail("%s: bad argument (1..4 slab sources, dw, Co/Ci > 0, block0 >= 0)", "itcv_wgrad_reduce_desc");
    return -1;
  }
  WgReduceDesc d;
  memset(&d, 0, sizeof(d));
  for (int k = 0; k < nsrc; ++k) {
    if (!slabs[k] || splits[k] < 1) {
      fail("%s: null slab or non-positive slice count", "itcv_wgrad_reduce_desc");
      return -1;
    }
    d.slab[k] = slabs[k], d.splits[k] = splits[k];
  }
  d.nsrc = nsrc, d.dw = dw, d.coci = Co * Ci, d.accumulate = accumulate ? 1 : 0, d.block0 = block0;
  d.fine = wgrad_reduce_fine(d.splits, nsrc) ? 1 : 0;
  d.nblocks = wgrad_reduce_blocks(d.coci, d.fine != 0);
  memcpy(host_desc, &d, sizeof(d));
  return d.nblocks;
}
int itcv_wgrad_reduce_max_descs(void) { return kWgMaxDesc; }
int itcv_wgrad_reduce_many(const void* dev_table, int n, int total_blocks, void* stream) {
  ITCV_REQUIRE(dev_table && n > 0 && total_blocks > 0, "itcv_wgrad_reduce_many");
  if (n > kWgMaxDesc) return fail("%s: at most %d descriptors per table (itcv_wgrad_reduce_max_descs)", "itcv_wgrad_reduce_many", kWgMaxDesc);
  const int grid = total_blocks < 2048 ? total_blocks : 2048;            // 8 blocks per CU, persistent
  hipLaunchKernelGGL(wgrad_p_reduce_many, dim3(grid), dim3(256), 0, S(stream), static_cast<const WgReduceDesc*>(dev_table), n, 0,
                     total_blocks);
  ITCV_CHECK_LAUNCH("itcv_wgrad_reduce_many");
  return 0;
}

// ---- nn.Linear as skinny fp32 GEMMs ----------------------------------------------------------------
}  // extern "C"

namespace itcv {
struct GemmPlan {
  int mt, nt, ktiles, splits, kps;
};
static GemmPlan plan_gemm64(int M, int N, int K) {
  GemmPlan p;
  p.mt = cdiv(M, 64), p.nt = cdiv(N, 64), p.ktiles = cdiv(K, 32);
  const int tiles = p.mt * p.nt;
  int splits = 1;
  if (tiles < 128 && p.ktiles >= 4) {
    splits = cdiv(256, tiles);
    if (splits > 32) splits = 32;             // the slab reduce is a serial chain over the splits
    if (splits > p.ktiles / 2) splits = p.ktiles / 2;
    if (splits < 1) splits = 1;
  }
  p.kps = cdiv(p.ktiles, splits);
  p.splits = cdiv(p.ktiles, p.kps);
  return p;
}
static size_t gemm64_ws(int M, int N, int K) {
  const GemmPlan p = plan_gemm64(M, N, K);
  return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}
static int run_gemm64(const char* name, const float* A, const float* B, const float* bias, float* C, int M, int N,
                      int K, long long sam, long long sak, long long sbk, long long sbn, int accumulate, void* ws,
                      size_t ws_bytes, hipStream_t st) {
  const GemmPlan p = plan_gemm64(M, N, K);
  const size_t need = p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
  if (need && (!ws || ws_bytes < need)) return fail("%s: workspace too small (need %lld bytes)", name, (long long)need);
  GemmArgs a;
  a.A = A, a.B = B, a.bias = bias;
  a.C = p.splits > 1 ? static_cast<float*>(ws) : C;
  a.M = M, a.N = N, a.K = K;
  a.sam = sam, a.sak = sak, a.sbk = sbk, a.sbn = sbn;
  a.mt = p.mt, a.nt = p.nt, a.ktiles = p.ktiles, a.ktiles_per_split = p.kps;
  a.slab_stride = p.splits > 1 ? (size_t)M * N : 0;
  a.accumulate = p.splits > 1 ? 0 : accumulate;
  dim3 grid(p.mt * p.nt, p.splits), blk(256);
  const bool akc = sak == 1, bkc = sbk == 1;
  if (akc && bkc) hipLaunchKernelGGL((gemm64_kernel<true, true>), grid, blk, 0, st, a);
  else if (akc) hipLaunchKernelGGL((gemm64_kernel<true, false>), grid, blk, 0, st, a);
  else if (bkc) hipLaunchKernelGGL((gemm64_kernel<false, true>), grid, blk, 0, st, a);
  else hipLaunchKernelGGL((gemm64_kernel<false, false>), grid, blk, 0, st, a);
  ITCV_CHECK_LAUNCH(name);
  if (p.splits > 1) {
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(gemm64_reduce, dim3((int)(cdivz(total, 256) < 1024 ? cdivz(total, 256) : 1024)), dim3(256), 0, st,
                       static_cast<const float*>(ws), bias, C, total, N, p.splits, accumulate);
    ITCV_CHECK_LAUNCH(name);
  }
  return 0;
}
}  // namespace itcv

extern "C" {

// nn.Linear(K -> N) at batch B: y[B][N] = x[B][K] w[N][K]^T + bias; dx[B][K] = dy[B][N] w; dw[N][K] (+)= dy^T x.
// One workspace size serves all three.
size_t itcv_linear_workspace(int B, int K, int N) {
  if (B <= 0 || K <= 0 || N <= 0) return 0;
  size_t n = gemm64_ws(B, N, K);
  const size_t d = gemm64_ws(B, K, N), w = gemm64_ws(N, K, B);
  if (d > n) n = d;
  if (w > n) n = w;
  return n;
}
int itcv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, void* ws,
                    size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(x && w && y && B > 0 && K > 0 && N > 0, "itcv_linear_fwd");
  return run_gemm64("itcv_linear_fwd", x, w, bias, y, B, N, K, K, 1, 1, K, 0, ws, ws_bytes, S(stream));
}
int itcv_linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, void* ws, size_t ws_bytes,
                      void* stream) {
  ITCV_REQUIRE(dy && w && dx && B > 0 && K > 0 && N > 0, "itcv_linear_dgrad");
  return run_gemm64("itcv_linear_dgrad", dy, w, nullptr, dx, B, K, N, N, 1, K, 1, 0, ws, ws_bytes, S(stream));
}
int itcv_linear_wgrad(const float* dy, const float* x, float* dw, int B, int K, int N, int accumulate, void* ws,
                      size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(dy && x && dw && B > 0 && K > 0 && N > 0, "itcv_linear_wgrad");
  return run_gemm64("itcv_linear_wgrad", dy, x, nullptr, dw, N, K, B, 1, N, K, 1, accumulate, ws, ws_bytes, S(stream));
}

// ---- 5x5 weight gradient with a 3-channel side, from planes of the 64-channel side (bf16x3) -------------------
// stem = 1: dw[64][Cs][5][5] for x[B][Cs][H][W] (fp32, `small`) and the planes of dy[B][64][H][W] (`big_planes`);
// stem = 0: dw[Cs][64][5][5] for dy[B][Cs][H][W] (fp32, `small`) and the planes of x[B][64][H][W].
int itcv_conv2d_wgrad5_bf16p_supported(int Cs, int Cb, int H, int W) {
  return Cs >= 1 && Cs <= 3 && Cb == 64 && H > 0 && W >= 32 && W <= 256 && W % 32 == 0;
}
static inline int wgrad5_rows_per_job(int B, int H) {
  int r = cdiv(B * H, 1024);      // ~1024 wave jobs (256 blocks of 4)
  return r < 1 ? 1 : r;
}
size_t itcv_conv2d_wgrad5_bf16p_workspace(int B, int H) {
  if (B <= 0 || H <= 0) return 0;
  const int rpj = wgrad5_rows_per_job(B, H);
  return (size_t)B * cdiv(H, rpj) * 16 * 5 * 64 * sizeof(float);
}
int itcv_conv2d_wgrad5_bf16p(const float* small, const void* big_planes, float* dw, int B, int Cs, int H, int W,
                             int stem, int ns, const float* small_amax, int accumulate, void* ws, size_t ws_bytes,
                             void* stream) {
  ITCV_REQUIRE(small && big_planes && dw && B > 0 && (ns == 2 || ns == ITCV_PLANES_F16X2), "itcv_conv2d_wgrad5_bf16p");
  ITCV_REQUIRE(!small_amax || ns == ITCV_PLANES_F16X2, "itcv_conv2d_wgrad5_bf16p(a scale only applies to fp16 planes)");
  const int f16 = ns == ITCV_PLANES_F16X2;
  if (!itcv_conv2d_wgrad5_bf16p_supported(Cs, 64, H, W))
    return fail("%s: needs Cs <= 3, 64 channels on the other side, W in {32, 64}", "itcv_conv2d_wgrad5_bf16p");
  const int rpj = wgrad5_rows_per_job(B, H), njobs = B * cdiv(H, rpj);
  const size_t need = (size_t)njobs * 16 * 5 * 64 * sizeof(float);
  if (!ws || ws_bytes < need) return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_wgrad5_bf16p", (long long)need);
  hipStream_t st = S(stream);
  const size_t plane_stride = (size_t)B * 8 * H * W;
  {
    ProfScope prof(st, 12, 5, Cs, stem ? 1 : 0, ns, 2.0 * B * H * W * 64.0 * Cs * 25);
    const dim3 grid(cdiv(njobs, 4)), blk(256);
    const u32x4* bp = static_cast<const u32x4*>(big_planes);
    float* slab = static_cast<float*>(ws);
    if (stem && f16)
      launch_timed((conv_wgrad5_planes_kernel<1, true>), grid, blk, 0, st, small, bp, slab, B, Cs, H, W, rpj, njobs, plane_stride, small_amax);
    else if (stem)
      launch_timed((conv_wgrad5_planes_kernel<1, false>), grid, blk, 0, st, small, bp, slab, B, Cs, H, W, rpj, njobs, plane_stride, small_amax);
    else if (f16)
      launch_timed((conv_wgrad5_planes_kernel<-1, true>), grid, blk, 0, st, small, bp, slab, B, Cs, H, W, rpj, njobs, plane_stride, small_amax);
    else
      launch_timed((conv_wgrad5_planes_kernel<-1, false>), grid, blk, 0, st, small, bp, slab, B, Cs, H, W, rpj, njobs, plane_stride, small_amax);
  }
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad5_bf16p");
  const int total = Cs * 25 * 64;
  hipLaunchKernelGGL(wgrad5_reduce, dim3(cdiv(total, 16)), dim3(256), 0, st, static_cast<const float*>(ws), dw, Cs, njobs,
                     stem ? 1 : 0, accumulate);
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad5_bf16p(reduce)");
  return 0;
}

// Which kernel instantiation / decomposition a call resolves to (for profiling buckets):
//   bits 0-7 block rows BM, 8-15 KS, 16 up2, 20-31 split-K factor
int itcv_conv2d_fwd_variant(int B, int Ci, int H, int W, int Co, int KS, int up2) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return -1;
  const FwdPlan p = plan_fwd(B, Ci, H, W, Co, KS);
  return p.bm | (KS << 8) | ((up2 ? 1 : 0) << 16) | (p.splits << 20);
}
int itcv_conv2d_wgrad_variant(int B, int Ci, int H, int W, int Co, int KS, int up2) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return -1;
  const WgPlan p = plan_wgrad(B, Ci, H, W, Co, KS);
  return p.bm | (KS << 8) | ((up2 ? 1 : 0) << 16) | (p.splits << 20);
}

size_t itcv_bias_grad_workspace(int B, int C, int HW) {
  return B > 0 && C > 0 && HW > 0 ? (size_t)bias_splits(B, C, HW) * C * sizeof(double) : 0;
}

int itcv_bias_grad(const float* dy, float* db, int B, int C, int HW, int accumulate, void* ws, size_t ws_bytes,
                   void* stream) {
  ITCV_REQUIRE(dy && db && B > 0 && C > 0 && HW > 0, "itcv_bias_grad");
  const int splits = bias_splits(B, C, HW);
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)splits * C * sizeof(double), "itcv_bias_grad(workspace)");
  double* part = static_cast<double*>(ws);
  hipLaunchKernelGGL(bias_grad_partial, dim3(C, splits), dim3(256), 0, S(stream), dy, part, B, C, HW, splits);
  ITCV_CHECK_LAUNCH("itcv_bias_grad");
  hipLaunchKernelGGL(bias_grad_combine, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), part, db, C, splits, accumulate);
  ITCV_CHECK_LAUNCH("itcv_bias_grad(combine)");
  return 0;
}

}  // extern "C"
