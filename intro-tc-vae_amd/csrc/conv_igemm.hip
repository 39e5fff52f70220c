// Implicit-GEMM convolution / linear layers on the exact-fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain, so results track the fp32 CPU
// reference to rounding).  Replaces the ATen conv/linear forward + backward behind
// /root/reference/models.py:28-47,213,233,270,290.
//
//   forward / data-gradient:  C[M][N] = A[K][M]^T * im2col(X)[K][N]
//        M = output channels, N = B*H*W (pixel index, contiguous in NCHW), K = Cin*KS*KS
//        A = packed weights (K-major, zero padded), gathered B operand read with hardware
//        bounds-checked buffer loads (padding and tile tails come back as 0).
//   weight-gradient:          C[M][N] = sum_k dY[M][k] * im2col(X)[N][k]
//        M = Co, N = Ci*KS*KS (== the OIHW layout of dW), k = (b,h,w); split-K over k with
//        fp32 slabs and a deterministic reduce (bitwise reproducible, no atomics).
//
// Tiling: 256 threads = 4 waves; each wave owns TMxTN tiles of 32x32 accumulators (AGPRs);
// LDS tiles are K-major so every ds_read_b32 of an MFMA operand is 32 consecutive floats per
// half-wave (conflict free).  Global->register->LDS software pipeline with two LDS buffers and
// one barrier per K-tile.  blockIdx -> tile mapping keeps tiles that share an im2col panel on
// one XCD (blocks b and b+8 share an XCD's L2).
#include "common.h"

namespace itcv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int B, Ci, H, W, Co;
  int Mp, K, N;
  int mt, nt;
  int ktiles, ktiles_per_split;
  uint32_t x_bytes;
  size_t slab_stride;
};

template <int KS, int BM, int BN, int WM, int WN, bool UP2>
__global__ __launch_bounds__(WM* WN * 64) void conv_fwd_kernel(ConvArgs a) {
  constexpr int NT = WM * WN * 64, BK = 16, KK = KS * KS, P = KS / 2;
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
  const int nl = t % BN, kr0 = t / BN;
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
  for (int tap = 0; tap < KK; ++tap) {
    const int dh = tap / KS - P, dw = tap % KS - P;
    if (nvalid && (unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)W) tapmask |= 1u << tap;
  }
  const int tbase = bi * a.Ci * HWs + (UP2 ? 0 : h * W + w);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);

  constexpr int AV = BK * BM / 4, AL = (AV + NT - 1) / NT;
  float4 areg[AL];
  float breg[BL];

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int idx = t + i * NT;
      if (AV % NT == 0 || idx < AV) {
        const int kr = idx / (BM / 4), m = m0 + (idx % (BM / 4)) * 4;
        areg[i] = (m < a.Mp) ? *reinterpret_cast<const float4*>(a.wp + (size_t)(k0 + kr) * a.Mp + m)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const int k = k0 + kr0 + i * BROWS;
      const int ci = k / KK, tap = k - ci * KK;
      const int dh = tap / KS - P, dw = tap % KS - P;
      const bool valid = (k < a.K) && ((tapmask >> tap) & 1u);
      int off;
      if (UP2)
        off = tbase + ci * HWs + ((h + dh) >> 1) * Ws + ((w + dw) >> 1);
      else
        off = tbase + ci * HW + dh * W + dw;
      breg[i] = buf_load(rx, valid ? (uint32_t)off * 4u : kOOB);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int idx = t + i * NT;
      if (AV % NT == 0 || idx < AV) {
        const int kr = idx / (BM / 4), m4 = (idx % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(&As[buf][kr * PA + m4]) = areg[i];
      }
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

  auto compute = [&](int buf) {
    const float* Ab = &As[buf][half * PA + wm * WTM + l31];
    const float* Bb = &Bs[buf][half * PB + wn * WTN + l31];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = Ab[kk * PA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bb[kk * PB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (more) load_tile(kt + 1);
      compute(cur);
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
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(size_t)k * stride + i];
    if (bias) s += bias[(i / HW) % Co];
    y[i] = s;
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int KK,
                                   int for_dgrad, int K, int M, int Kp, int Mp) {
  const size_t total = (size_t)Kp * Mp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / Mp), m = (int)(i - (size_t)k * Mp);
    float v = 0.f;
    if (k < K && m < M) {
      const int c = k / KK, tap = k - c * KK;
      v = for_dgrad ? w[((size_t)c * Ci + m) * KK + (KK - 1 - tap)] : w[((size_t)m * Ci + c) * KK + tap];
    }
    wp[i] = v;
  }
}

// ------------------------------------------------------------------------------ wgrad
struct WgradArgs {
  const float* x;
  const float* dy;
  float* out;  // slab base (or dw when splits == 1 and !accumulate)
  int B, Ci, H, W, Co;
  int Ntot, Ktot;
  int mt, nt, tiles;
  int ktiles, ktiles_per_split, splits;
  uint32_t x_bytes, dy_bytes;
  size_t slab_stride;
};

template <int KS, int BM, int BN, int WM, int WN, bool UP2>
__global__ __launch_bounds__(WM* WN * 64) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = WM * WN * 64, BK = 32, KK = KS * KS, P = KS / 2;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int PA = BM + 1, PB = BN + 1;  // odd pitch: transposing ds_write_b32 is conflict free
  constexpr int RP = NT / 32, AL = BM / RP, BL = BN / RP;
  __shared__ float As[2][BK * PA];
  __shared__ float Bs[2][BK * PB];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, half = lane >> 5;
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int tile = q % a.tiles, sk = (q / a.tiles) * 8 + xcd;
  if (sk >= a.splits) return;
  const int tile_m = tile % a.mt, tile_n = tile / a.mt;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt0 = sk * a.ktiles_per_split;
  const int kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);
  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? W / 2 : W;
  const int kl = t & 31, r0 = t >> 5;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.dy, a.dy_bytes);

  float areg[AL], breg[BL];
  auto load_tile = [&](int kt) {
    const int k = kt * BK + kl;
    const bool kvalid = k < a.Ktot;
    const int bi = k / HW, hw = k - bi * HW;
    const int h = hw / W, w = hw - h * W;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int m = m0 + r0 + i * RP;
      const bool valid = kvalid && m < a.Co;
      areg[i] = buf_load(rdy, valid ? (uint32_t)((bi * a.Co + m) * HW + hw) * 4u : kOOB);
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const int nn = n0 + r0 + i * RP;
      const int ci = nn / KK, tap = nn - ci * KK;
      const int hh = h + tap / KS - P, ww = w + tap % KS - P;
      const bool valid = kvalid && nn < a.Ntot && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
      const int hs = UP2 ? hh >> 1 : hh, wsrc = UP2 ? ww >> 1 : ww;
      breg[i] = buf_load(rx, valid ? (uint32_t)(((bi * a.Ci + ci) * Hs + hs) * Ws + wsrc) * 4u : kOOB);
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

  auto compute = [&](int buf) {
    const float* Ab = &As[buf][half * PA + wm * WTM + l31];
    const float* Bb = &Bs[buf][half * PB + wn * WTN + l31];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = Ab[kk * PA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bb[kk * PB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (more) load_tile(kt + 1);
      compute(cur);
      if (more) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  float* out = a.out + (size_t)sk * a.slab_stride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nn = n0 + wn * WTN + j * 32 + l31;
    if (nn >= a.Ntot) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < a.Co) out[(size_t)m * a.Ntot + nn] = acc[i][j][r];
      }
  }
}

__global__ void splitk_reduce_wgrad(const float* __restrict__ slab, float* __restrict__ dw, size_t total,
                                    int splits, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = accumulate ? dw[i] : 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(size_t)k * total + i];
    dw[i] = s;
  }
}

// db[c] (+)= sum over (b, hw) of dy[b][c][hw]; one block per channel, fixed-order reduction
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int B,
                                                        int C, int HW, int accumulate) {
  __shared__ double scratch[4];
  const int c = blockIdx.x;
  double s = 0.0;
  const size_t total = (size_t)B * HW;
  for (size_t i = threadIdx.x; i < total; i += blockDim.x) {
    const size_t b = i / HW, hw = i - b * HW;
    s += (double)dy[(b * C + c) * HW + hw];
  }
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) db[c] = (accumulate ? db[c] : 0.f) + (float)s;
}

// ------------------------------------------------------------------------------ host side
struct FwdPlan {
  int bm, bn, mt, nt, ktiles, splits, kps;
};

static FwdPlan plan_fwd(int B, int Ci, int H, int W, int Co, int KS) {
  FwdPlan p;
  const int K = Ci * KS * KS;
  const long long N = (long long)B * H * W;
  p.bm = Co <= 32 ? 32 : (Co <= 64 ? 64 : 128);
  p.bn = p.bm == 128 ? 128 : 256;
  p.mt = cdiv(Co, p.bm);
  p.nt = (int)((N + p.bn - 1) / p.bn);
  p.ktiles = cdiv(K, 16);
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

struct WgPlan {
  int bm, mt, nt, tiles, ktiles, splits, kps;
};

static WgPlan plan_wgrad(int B, int Ci, int H, int W, int Co, int KS) {
  WgPlan p;
  const int Ntot = Ci * KS * KS;
  const long long Ktot = (long long)B * H * W;
  p.bm = Co <= 32 ? 32 : (Co <= 64 ? 64 : 128);
  p.mt = cdiv(Co, p.bm);
  p.nt = cdiv(Ntot, 128);
  p.tiles = p.mt * p.nt;
  p.ktiles = (int)((Ktot + 31) / 32);
  int splits = cdiv(768, p.tiles);
  if (splits > p.ktiles / 8) splits = p.ktiles / 8;
  if (splits > 256) splits = 256;
  if (splits < 1) splits = 1;
  p.kps = cdiv(p.ktiles, splits);
  p.splits = cdiv(p.ktiles, p.kps);
  return p;
}

template <int KS, int BM, int BN, int WM, int WN>
static void launch_fwd_cfg(const ConvArgs& a, int splits, int up2, hipStream_t st) {
  dim3 grid(cdiv(a.nt, 8) * 8 * a.mt, splits);
  if (up2)
    hipLaunchKernelGGL((conv_fwd_kernel<KS, BM, BN, WM, WN, true>), grid, dim3(WM * WN * 64), 0, st, a);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<KS, BM, BN, WM, WN, false>), grid, dim3(WM * WN * 64), 0, st, a);
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

template <int KS, bool UP2>
static void launch_wgrad_up(const WgradArgs& a, int bm, hipStream_t st) {
  dim3 grid(cdiv(a.splits, 8) * 8 * a.tiles);
  if (bm == 32)
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, 32, 128, 1, 4, UP2>), grid, dim3(256), 0, st, a);
  else if (bm == 64)
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, 64, 128, 1, 4, UP2>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, 128, 128, 2, 2, UP2>), grid, dim3(256), 0, st, a);
}
template <int KS>
static void launch_wgrad(const WgradArgs& a, int bm, int up2, hipStream_t st) {
  if (up2)
    launch_wgrad_up<KS, true>(a, bm, st);
  else
    launch_wgrad_up<KS, false>(a, bm, st);
}

static int check_dims(const char* name, int B, int Ci, int H, int W, int Co, int KS) {
  if (!(KS == 1 || KS == 3 || KS == 5)) return fail("%s: kernel size must be 1, 3 or 5 (got %lld)", name, KS);
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return fail("%s: empty or negative dimension", name);
  const long long in_elems = (long long)B * Ci * H * W, out_elems = (long long)B * Co * H * W;
  if (in_elems >= (1LL << 30) || out_elems >= (1LL << 30))
    return fail("%s: tensor too large for 32-bit buffer offsets (%lld / %lld elements)", name, in_elems, out_elems);
  return 0;
}

}  // namespace itcv

using namespace itcv;

extern "C" {

size_t itcv_conv2d_packed_weight_elems(int Co, int Ci, int KS, int for_dgrad) {
  const int M = for_dgrad ? Ci : Co, K = (for_dgrad ? Co : Ci) * KS * KS;
  return align_up(K, 16) * align_up(M, 32);
}

int itcv_conv2d_pack_weight(const float* w, float* wp, int Co, int Ci, int KS, int for_dgrad, void* stream) {
  ITCV_REQUIRE(w && wp && Co > 0 && Ci > 0 && (KS == 1 || KS == 3 || KS == 5), "itcv_conv2d_pack_weight");
  const int M = for_dgrad ? Ci : Co, K = (for_dgrad ? Co : Ci) * KS * KS;
  const int Kp = (int)align_up(K, 16), Mp = (int)align_up(M, 32);
  const size_t total = (size_t)Kp * Mp;
  const int blocks = (int)(cdivz(total, 256) < 4096 ? cdivz(total, 256) : 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, S(stream), w, wp, Co, Ci, KS * KS, for_dgrad,
                     K, M, Kp, Mp);
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
  a.Mp = (int)align_up(Co, 32);
  a.K = Ci * KS * KS;
  a.N = B * H * W;
  a.mt = p.mt, a.nt = p.nt, a.ktiles = p.ktiles, a.ktiles_per_split = p.kps;
  a.x_bytes = (uint32_t)((size_t)B * Ci * (up2 ? (H / 2) * (W / 2) : H * W) * sizeof(float));
  a.slab_stride = p.splits > 1 ? out_elems : 0;
  hipStream_t st = S(stream);
  if (KS == 1)
    launch_fwd<1>(a, p.bm, p.splits, up2, st);
  else if (KS == 3)
    launch_fwd<3>(a, p.bm, p.splits, up2, st);
  else
    launch_fwd<5>(a, p.bm, p.splits, up2, st);
  ITCV_CHECK_LAUNCH("itcv_conv2d_fwd");
  if (p.splits > 1) {
    const int blocks = (int)(cdivz(out_elems, 256) < 2048 ? cdivz(out_elems, 256) : 2048);
    hipLaunchKernelGGL(splitk_reduce_fwd, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), bias, y,
                       out_elems, out_elems, p.splits, H * W, Co);
    ITCV_CHECK_LAUNCH("itcv_conv2d_fwd(reduce)");
  }
  return 0;
}

size_t itcv_conv2d_wgrad_workspace(int B, int Ci, int H, int W, int Co, int KS) {
  if (B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0) return 0;
  const WgPlan p = plan_wgrad(B, Ci, H, W, Co, KS);
  return (size_t)p.splits * Co * Ci * KS * KS * sizeof(float);
}

int itcv_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int Ci, int H, int W, int Co, int KS,
                      int up2, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_dims("itcv_conv2d_wgrad", B, Ci, H, W, Co, KS)) return e;
  ITCV_REQUIRE(x && dy && dw, "itcv_conv2d_wgrad");
  if (up2) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_conv2d_wgrad(up2)");
  const WgPlan p = plan_wgrad(B, Ci, H, W, Co, KS);
  const size_t dw_elems = (size_t)Co * Ci * KS * KS;
  const bool direct = p.splits == 1 && !accumulate;
  if (!direct && (!ws || ws_bytes < (size_t)p.splits * dw_elems * sizeof(float)))
    return fail("%s: workspace too small (need %lld bytes)", "itcv_conv2d_wgrad",
                (long long)((size_t)p.splits * dw_elems * sizeof(float)));
  WgradArgs a;
  a.x = x, a.dy = dy;
  a.out = direct ? dw : static_cast<float*>(ws);
  a.B = B, a.Ci = Ci, a.H = H, a.W = W, a.Co = Co;
  a.Ntot = Ci * KS * KS;
  a.Ktot = B * H * W;
  a.mt = p.mt, a.nt = p.nt, a.tiles = p.tiles;
  a.ktiles = p.ktiles, a.ktiles_per_split = p.kps, a.splits = p.splits;
  a.x_bytes = (uint32_t)((size_t)B * Ci * (up2 ? (H / 2) * (W / 2) : H * W) * sizeof(float));
  a.dy_bytes = (uint32_t)((size_t)B * Co * H * W * sizeof(float));
  a.slab_stride = dw_elems;
  hipStream_t st = S(stream);
  if (KS == 1)
    launch_wgrad<1>(a, p.bm, up2, st);
  else if (KS == 3)
    launch_wgrad<3>(a, p.bm, up2, st);
  else
    launch_wgrad<5>(a, p.bm, up2, st);
  ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad");
  if (!direct) {
    const int blocks = (int)(cdivz(dw_elems, 256) < 2048 ? cdivz(dw_elems, 256) : 2048);
    hipLaunchKernelGGL(splitk_reduce_wgrad, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), dw,
                       dw_elems, p.splits, accumulate);
    ITCV_CHECK_LAUNCH("itcv_conv2d_wgrad(reduce)");
  }
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

int itcv_bias_grad(const float* dy, float* db, int B, int C, int HW, int accumulate, void* stream) {
  ITCV_REQUIRE(dy && db && B > 0 && C > 0 && HW > 0, "itcv_bias_grad");
  hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, S(stream), dy, db, B, C, HW, accumulate);
  ITCV_CHECK_LAUNCH("itcv_bias_grad");
  return 0;
}

}  // extern "C"
