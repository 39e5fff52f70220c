// BatchNorm2d(train) + LeakyReLU (+ AvgPool2d(2) / nearest-Upsample(2) adjoints), and the
// pointwise / resampling kernels of the encoder/decoder stacks.  All HBM-bound: one coalesced
// pass per tensor, fp64 channel statistics with a deterministic two-stage reduction.
// Replaces /root/reference/models.py:37-38,48-49,214-216,225,271,284-286,291.
#include <stdlib.h>

#include "common.h"

namespace itcv {

constexpr int kRedThreads = 256;

// division by a runtime constant that is usually a power of two (sh = log2 or -1)
__device__ __forceinline__ uint32_t fdiv(uint32_t a, uint32_t d, int sh) { return sh >= 0 ? (a >> sh) : (a / d); }

// ------------------------------------------------------------------ channel moments (fp64)
// grid (C, splits): block (c, s) reduces its slice of channel c's B*HW values
// BatchNorm groups stacked in one tensor (see itcv_bn_train_fwd): element / chunk distances between consecutive groups.
// Zero-initialised = one group.  Apply kernels take the group from blockIdx.z; the fused statistics kernels walk the
// groups in order inside the block (the running buffers are advanced group by group).
struct BnGrp {
  size_t xs;    // x, skip, dx, dskip: elements per group
  size_t os;    // forward output y: elements per group
  size_t dys;   // upstream gradient dy: elements per group (depends on pool / up2)
  size_t ps;    // planes: chunks per group inside one plane
  int cs;       // mean / rstd: C per group (dsums: 2C)
  int G;        // groups walked by the fused statistics kernels (0 or 1 = one)
};

struct BnFinal {   // arguments of the fused single-launch path (splits == 1)
  double count;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  int64_t* nbt;
  float* mean;
  float* rstd;
};

template <bool FUSED>
__global__ __launch_bounds__(kRedThreads) void bn_moments_partial(const float* __restrict__ x,
                                                                  double* __restrict__ part, int B, int C, int HW,
                                                                  int splits, int hw_shift, BnFinal f, BnGrp grp) {
  __shared__ double scratch[kRedThreads / 64];
  const uint32_t c = blockIdx.x, s = blockIdx.y, hw_n = HW, total = (uint32_t)B * hw_n;
  const uint32_t chunk = ((total + splits - 1) / splits + 3) & ~3u;  // multiple of 4: float4 stays aligned
  const uint32_t beg = s * chunk, end = min(beg + chunk, total);
  const int ngroups = (FUSED && grp.G > 1) ? grp.G : 1;
  if (!FUSED) {   // sliced form: the group comes from blockIdx.z (gridDim.z = 1 and a zero `grp` otherwise)
    x += (size_t)blockIdx.z * grp.xs;
    part += (size_t)blockIdx.z * splits * 2 * C;
  }
 for (int gi = 0; gi < ngroups; ++gi, x += grp.xs, f.mean += grp.cs, f.rstd += grp.cs) {
  double s1 = 0.0, s2 = 0.0;
  if ((HW & 3) == 0) {
#pragma unroll 4
    for (uint32_t i = beg + threadIdx.x * 4; i < end; i += kRedThreads * 4) {
      const uint32_t b = fdiv(i, hw_n, hw_shift), hw = i - b * hw_n;
      const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * C + c) * hw_n + hw);
      s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
  } else {
    for (uint32_t i = beg + threadIdx.x; i < end; i += kRedThreads) {
      const uint32_t b = fdiv(i, hw_n, hw_shift), hw = i - b * hw_n;
      const float v = x[((size_t)b * C + c) * hw_n + hw];
      s1 += (double)v;
      s2 += (double)v * v;
    }
  }
  s1 = block_sum(s1, scratch);
  s2 = block_sum(s2, scratch);
  if (threadIdx.x == 0) {
    if (FUSED) {   // one block owns the whole channel: finalise here, no second launch
      const double m = s1 / f.count;
      double var = s2 / f.count - m * m;
      if (var < 0.0) var = 0.0;
      f.mean[c] = (float)m;
      f.rstd[c] = (float)(1.0 / sqrt(var + (double)f.eps));
      if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)m;
      if (f.running_var) {
        const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
      }
      if (c == 0 && f.nbt) f.nbt[0] += 1;
    } else {
      part[((size_t)s * 2 + 0) * C + c] = s1;
      part[((size_t)s * 2 + 1) * C + c] = s2;
    }
  }
 }
}

// sums[k][c] = sum_s part[s][k][c]   (k in {0,1}), fixed order
__global__ void combine_partials(const double* __restrict__ part, double* __restrict__ sums, int C2, int splits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C2) return;
  sums[i] = fold_strided(0.0, part + i, (size_t)C2, splits);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, float eps, float momentum,
                                   float* running_mean, float* running_var, int64_t* nbt, float* mean, float* rstd,
                                   int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += 1;
  if (c >= C) return;
  const double m = sums[c] / count;
  double var = sums[C + c] / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// statistics from the per-tile sums a conv epilogue wrote (fp32 per 256-value tile, folded here in fp64): block per channel
__global__ __launch_bounds__(256) void bn_tile_stats_finalize_kernel(const float* __restrict__ ts, int tiles, int pitch,
                                                                    double count, float eps, float momentum,
                                                                    float* running_mean, float* running_var, int64_t* nbt,
                                                                    float* mean, float* rstd, int C) {
  __shared__ double scratch[4];
  const int c = blockIdx.x;
  const float* p1 = ts + (size_t)c * pitch;
  const float* p2 = ts + ((size_t)C + c) * pitch;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < tiles; t += 256) s1 += (double)p1[t], s2 += (double)p2[t];
  s1 = block_sum(s1, scratch);
  s2 = block_sum(s2, scratch);
  if (threadIdx.x == 0) {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    if (c == 0 && nbt) nbt[0] += 1;
  }
}

// single-rank fast path: fold the partial sums and finalise in one launch
__global__ void bn_combine_finalize_kernel(const double* __restrict__ part, int splits, double count, float eps,
                                           float momentum, float* running_mean, float* running_var, int64_t* nbt,
                                           float* mean, float* rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += 1;
  if (c >= C) return;
  const double s1 = fold_strided(0.0, part + c, (size_t)2 * C, splits);
  const double s2 = fold_strided(0.0, part + C + c, (size_t)2 * C, splits);
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// BatchNorm groups, sliced statistics: fold every group's partial sums and finalise the groups IN ORDER (running buffers)
__global__ void bn_combine_finalize_groups_kernel(const double* __restrict__ part, int splits, int G, double count, float eps,
                                                  float momentum, float* running_mean, float* running_var, int64_t* nbt,
                                                  float* mean, float* rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += G;
  if (c >= C) return;
  for (int g = 0; g < G; ++g) {
    const double* pg = part + (size_t)g * splits * 2 * C;
    const double s1 = fold_strided(0.0, pg + c, (size_t)2 * C, splits);
    const double s2 = fold_strided(0.0, pg + C + c, (size_t)2 * C, splits);
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[(size_t)g * C + c] = (float)m;
    rstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}
__global__ void bn_combine_param_groups_kernel(const double* __restrict__ part, double* __restrict__ dsums, int C, int splits,
                                               int G, float* dgamma, float* dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float db = (accumulate && dbeta) ? dbeta[c] : 0.f, dg = (accumulate && dgamma) ? dgamma[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    const double* pg = part + (size_t)g * splits * 2 * C;
    const double s1 = fold_strided(0.0, pg + c, (size_t)2 * C, splits);
    const double s2 = fold_strided(0.0, pg + C + c, (size_t)2 * C, splits);
    dsums[(size_t)g * 2 * C + c] = s1;
    dsums[(size_t)g * 2 * C + C + c] = s2;
    db += (float)s1, dg += (float)s2;
  }
  if (dbeta) dbeta[c] = db;
  if (dgamma) dgamma[c] = dg;
}

// backward: fold the partial sums into dsums and (optionally) the parameter gradients
__global__ void bn_combine_param_kernel(const double* __restrict__ part, double* __restrict__ dsums, int C,
                                        int splits, float* dgamma, float* dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double s1 = fold_strided(0.0, part + c, (size_t)2 * C, splits);
  const double s2 = fold_strided(0.0, part + C + c, (size_t)2 * C, splits);
  dsums[c] = s1;
  dsums[C + c] = s2;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
}

__global__ void bn_eval_stats_kernel(const float* rm, const float* rv, float eps, float* mean, float* rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  rstd[c] = 1.f / sqrtf(rv[c] + eps);
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ------------------------------------------------------------------ forward apply
// y = pool(lrelu(gamma*(x-mean)*rstd + beta (+skip)))
template <int POOL>
__global__ void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ skip,
                                  float* __restrict__ y, int C, int H, int W, size_t nout, float slope,
                                  int hw_shift, int c_mask) {
  const int HW = H * W;
  if (POOL == 0) {
    // 4 consecutive pixels per thread (HW % 4 == 0 is checked on the host)
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < nout;
         i += (size_t)gridDim.x * blockDim.x * 4) {
      const uint32_t pl = fdiv((uint32_t)i, HW, hw_shift);
      const int c = c_mask >= 0 ? (int)(pl & (uint32_t)c_mask) : (int)(pl % (uint32_t)C);
      const float sc = gamma[c] * rstd[c], sh = beta[c] - mean[c] * sc;
      float4 v = *reinterpret_cast<const float4*>(x + i);
      v.x = v.x * sc + sh, v.y = v.y * sc + sh, v.z = v.z * sc + sh, v.w = v.w * sc + sh;
      if (skip) {
        const float4 k = *reinterpret_cast<const float4*>(skip + i);
        v.x += k.x, v.y += k.y, v.z += k.z, v.w += k.w;
      }
      v.x = lrelu(v.x, slope), v.y = lrelu(v.y, slope), v.z = lrelu(v.z, slope), v.w = lrelu(v.w, slope);
      *reinterpret_cast<float4*>(y + i) = v;
    }
  } else {
    const int Ho = H / 2, Wo = W / 2, HWo = Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (size_t)gridDim.x * blockDim.x) {
      const uint32_t bc = hw_shift >= 2 ? ((uint32_t)i >> (hw_shift - 2)) : (uint32_t)i / (uint32_t)HWo;
      const int r = (int)((uint32_t)i - bc * HWo), ho = r / Wo, wo = r - ho * Wo;
      const int c = c_mask >= 0 ? (int)(bc & (uint32_t)c_mask) : (int)(bc % (uint32_t)C);
      const float sc = gamma[c] * rstd[c], sh = beta[c] - mean[c] * sc;
      const size_t src = (size_t)bc * HW + (size_t)(2 * ho) * W + 2 * wo;
      const float2 a = *reinterpret_cast<const float2*>(x + src);
      const float2 b = *reinterpret_cast<const float2*>(x + src + W);
      float v0 = a.x * sc + sh, v1 = a.y * sc + sh, v2 = b.x * sc + sh, v3 = b.y * sc + sh;
      if (skip) {
        const float2 ka = *reinterpret_cast<const float2*>(skip + src);
        const float2 kb = *reinterpret_cast<const float2*>(skip + src + W);
        v0 += ka.x, v1 += ka.y, v2 += kb.x, v3 += kb.y;
      }
      y[i] = 0.25f * (lrelu(v0, slope) + lrelu(v1, slope) + lrelu(v2, slope) + lrelu(v3, slope));
    }
  }
}

// Same pass, additionally emitting the output as pre-split bf16 planes for the consumer conv
// (planes[p][b][c/8][h][w], see include/itcv_hip.h): a thread owns 8 channels x 4 output pixels (POOL = 0) or
// x 2 pooled pixels (POOL = 1), so each pixel's 8 channel values meet in one thread.  The fp32 output is
// computed with exactly the expressions of bn_act_fwd_kernel (bitwise the same tensor).
// When `st.part` is set the launch also FINALISES the batch statistics from the per-slice partial sums of
// bn_moments_partial (one launch less per layer): the (<= kStatCh) channels a block touches are folded in slice
// order by one thread each -- the same fixed-order sum in every block -- and the block that holds the first pixels
// of image 0 of a channel group records mean / rstd / running statistics for the backward pass.
constexpr int kStatCh = 64;   // channels per block iteration (8 groups of 8); larger footprints take the two-launch path
struct BnStatsIn {
  const double* part;   // [splits][2][C]
  int splits;
  double count;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  int64_t* nbt;
  float* mean_out;
  float* rstd_out;
};

// Plane stores of a wave: every thread holds PX consecutive 16-byte chunks (its PX pixels) per plane.  Stored straight
// from the registers, one wave instruction writes 64 pieces of 16 bytes that lie PX*16 bytes apart -- every 128-byte line
// is touched by PX instructions, two lanes each.  Instead the wave's PX*64 chunks pass through a wave-private LDS strip
// in pixel order and every store instruction writes 1 KB of consecutive chunks.  `cidx0`: the wave's first thread index
// (idx of lane 0); thread index -> (bc8, pp) as in the callers: chunk n of the wave belongs to thread cidx0 + n / PX.
template <int PX, int NS>
__device__ __forceinline__ void store_planes_wave(u32x4* __restrict__ planes, size_t plane_stride, uint32_t per_plane,
                                                  uint32_t HWo, uint32_t cidx0, const u32x4 (&ch)[PX][NS],
                                                  u32x4* __restrict__ strip) {
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int p = 0; p < NS; ++p) {
#pragma unroll
    for (int s = 0; s < PX; ++s) strip[lane * PX + s] = ch[s][p];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      const uint32_t n = k * 64 + lane, idx2 = cidx0 + n / PX, s2 = n % PX;
      const uint32_t bc8 = idx2 / per_plane, pp = idx2 - bc8 * per_plane;
      planes[(size_t)p * plane_stride + (size_t)bc8 * HWo + (size_t)pp * PX + s2] = strip[n];
    }
  }
}

// F16: the planes are fp16 hi / lo of the output with scale 1 (activations are O(1); common.h), record behind plane 1.
template <int POOL, int NS, bool STATS, bool STRIP = true, bool F16 = false>
__global__ __launch_bounds__(256) void bn_act_fwd_planes_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ skip,
    float* __restrict__ y, u32x4* __restrict__ planes, int B, int C, int H, int W, float slope, BnStatsIn st,
    size_t plane_stride, BnGrp grp) {
  if constexpr (F16) {
    if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
      ScaleRec* rec = reinterpret_cast<ScaleRec*>(planes + (size_t)NS * plane_stride);
      rec->scale = 1.f, rec->inv = 1.f, rec->pad[0] = rec->pad[1] = 0.f;
    }
  }
  {
    const size_t g = blockIdx.z;     // BatchNorm group (gridDim.z = 1 and a zero `grp` otherwise)
    x += g * grp.xs, planes += g * grp.ps, mean += g * grp.cs, rstd += g * grp.cs;
    if (skip) skip += g * grp.xs;
    if (y) y += g * grp.os;
  }
  __shared__ float s_mean[STATS ? kStatCh : 1], s_rstd[STATS ? kStatCh : 1];
  constexpr int PX = POOL ? 2 : 4;                      // output pixels per thread
  __shared__ u32x4 strips[STRIP ? 4 : 1][STRIP ? 64 * PX : 1];
  const int HW = H * W, C8 = C >> 3;
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W, HWo = Ho * Wo;
  const uint32_t per_plane = (uint32_t)HWo / PX, total = (uint32_t)B * C8 * per_plane;
  for (uint32_t base = blockIdx.x * blockDim.x; base < total; base += gridDim.x * blockDim.x) {
    const uint32_t idx = base + threadIdx.x;
    const uint32_t g0 = base / per_plane;                // first (image, channel-group) of this block iteration
    if (STATS) {
      __syncthreads();
      const uint32_t last = min(base + blockDim.x, total) - 1, ng = last / per_plane - g0 + 1;
      if (threadIdx.x < ng * 8) {
        // BatchNorm groups (blockIdx.z): every group folds its own partial sums; the running buffers advance group by
        // group, so the recording thread of group 0 folds the later groups as well and applies them in order
        const uint32_t gz = blockIdx.z, G = grp.G > 1 ? grp.G : 1;
        const uint32_t g = g0 + threadIdx.x / 8, c = (g % C8) * 8 + (threadIdx.x & 7);
        const double* part = st.part + (size_t)gz * st.splits * 2 * C;
        const double s1 = fold_strided(0.0, part + c, (size_t)2 * C, st.splits);
        const double s2 = fold_strided(0.0, part + C + c, (size_t)2 * C, st.splits);
        const double m = s1 / st.count;
        double var = s2 / st.count - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)st.eps));
        s_mean[threadIdx.x] = mf, s_rstd[threadIdx.x] = rf;
        if (g < (uint32_t)C8 && g * per_plane >= base) {   // image 0, and the group starts inside this block
          st.mean_out[(size_t)gz * C + c] = mf, st.rstd_out[(size_t)gz * C + c] = rf;
          if (gz == 0) {
            double mg = m, vg = var;
            for (uint32_t gg = 0; gg < G; ++gg) {
              if (gg) {
                const double* pg = st.part + (size_t)gg * st.splits * 2 * C;
                const double t1 = fold_strided(0.0, pg + c, (size_t)2 * C, st.splits);
                const double t2 = fold_strided(0.0, pg + C + c, (size_t)2 * C, st.splits);
                mg = t1 / st.count;
                vg = t2 / st.count - mg * mg;
                if (vg < 0.0) vg = 0.0;
              }
              if (st.running_mean) st.running_mean[c] = (1.f - st.momentum) * st.running_mean[c] + st.momentum * (float)mg;
              if (st.running_var) {
                const double unbiased = st.count > 1.0 ? vg * st.count / (st.count - 1.0) : vg;
                st.running_var[c] = (1.f - st.momentum) * st.running_var[c] + st.momentum * (float)unbiased;
              }
            }
            if (c == 0 && st.nbt) st.nbt[0] += G;
          }
        }
      }
      __syncthreads();
    }
    if (idx >= total) continue;
    const uint32_t bc8 = idx / per_plane, pp = idx - bc8 * per_plane;
    const uint32_t b = bc8 / C8, c8 = bc8 - b * C8;
    float o[8][PX];
    // POOL == 0: the loads of the thread's 8 channels (tensor, skip, parameters) are all issued before the first is
    // consumed (-3 % on the large layers, -13 % on the small ones; the pooled form, which would hold 16 + 16 float4,
    // lost 19 % to its register count and keeps the channel-by-channel loop)
    float4 xin[POOL ? 1 : 8];
    float p_ga[8], p_be[8], p_mu[8], p_rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c8 * 8 + j;
      if (POOL == 0) xin[POOL ? 0 : j] = *reinterpret_cast<const float4*>(x + ((size_t)b * C + c) * HW + (size_t)pp * 4);
      p_ga[j] = gamma[c], p_be[j] = beta[c];
      p_mu[j] = STATS ? s_mean[(bc8 - g0) * 8 + j] : mean[c], p_rs[j] = STATS ? s_rstd[(bc8 - g0) * 8 + j] : rstd[c];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c8 * 8 + j;
      const float mu = p_mu[j], rs = p_rs[j];
      const float sc = p_ga[j] * rs, sh = p_be[j] - mu * sc;
      const size_t bc = (size_t)b * C + c;
      if (POOL == 0) {
        const size_t i = bc * HW + (size_t)pp * 4;
        float4 v = xin[POOL ? 0 : j];
        v.x = v.x * sc + sh, v.y = v.y * sc + sh, v.z = v.z * sc + sh, v.w = v.w * sc + sh;
        if (skip) {          // (residual blocks only: loaded here, channel by channel)
          const float4 k = *reinterpret_cast<const float4*>(skip + i);
          v.x += k.x, v.y += k.y, v.z += k.z, v.w += k.w;
        }
        v.x = lrelu(v.x, slope), v.y = lrelu(v.y, slope), v.z = lrelu(v.z, slope), v.w = lrelu(v.w, slope);
        if (y) *reinterpret_cast<float4*>(y + i) = v;
        o[j][0] = v.x, o[j][1] = v.y, o[j][POOL ? 0 : 2] = v.z, o[j][POOL ? 1 : 3] = v.w;
      } else {
        const uint32_t ho = (pp * 2) / Wo, wo = pp * 2 - ho * Wo;
        const size_t src = bc * HW + (size_t)(2 * ho) * W + 2 * wo;
        const float4 a = *reinterpret_cast<const float4*>(x + src);
        const float4 bb = *reinterpret_cast<const float4*>(x + src + W);
        float v0 = a.x * sc + sh, v1 = a.y * sc + sh, v2 = bb.x * sc + sh, v3 = bb.y * sc + sh;
        float w0 = a.z * sc + sh, w1 = a.w * sc + sh, w2 = bb.z * sc + sh, w3 = bb.w * sc + sh;
        if (skip) {
          const float4 ka = *reinterpret_cast<const float4*>(skip + src);
          const float4 kb = *reinterpret_cast<const float4*>(skip + src + W);
          v0 += ka.x, v1 += ka.y, v2 += kb.x, v3 += kb.y;
          w0 += ka.z, w1 += ka.w, w2 += kb.z, w3 += kb.w;
        }
        const float r0 = 0.25f * (lrelu(v0, slope) + lrelu(v1, slope) + lrelu(v2, slope) + lrelu(v3, slope));
        const float r1 = 0.25f * (lrelu(w0, slope) + lrelu(w1, slope) + lrelu(w2, slope) + lrelu(w3, slope));
        if (y) *reinterpret_cast<float2*>(y + bc * HWo + (size_t)pp * 2) = make_float2(r0, r1);
        o[j][0] = r0, o[j][1] = r1;
      }
    }
    u32x4 chk[PX][NS];
#pragma unroll
    for (int px = 0; px < PX; ++px) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = o[j][px];
      split8<NS, F16>(v8, chk[px]);
    }
    const uint32_t cidx0 = base + (threadIdx.x & ~63u);
    if (STRIP && cidx0 + 64 <= total) {   // whole wave in range (wave-uniform): coalesced plane stores
      store_planes_wave<PX, NS>(planes, plane_stride, per_plane, (uint32_t)HWo, cidx0, chk, strips[STRIP ? threadIdx.x >> 6 : 0]);
    } else {
#pragma unroll
      for (int px = 0; px < PX; ++px)
#pragma unroll
        for (int p = 0; p < NS; ++p) planes[(size_t)p * plane_stride + (size_t)bc8 * HWo + (size_t)pp * PX + px] = chk[px][p];
    }
  }
}

// ------------------------------------------------------------------ backward
// upstream gradient of the pre-activation u = bn(x) (+skip) at pixel (bc, h, w):
//   MODE 0: dy same shape;  MODE 1: dy is the gradient of the 2x2-average-pooled output;
//   MODE 2: dy is the gradient of the nearest x2 upsampled output (sum of its 4 children).
template <int MODE>
__device__ __forceinline__ float upstream(const float* __restrict__ dy, size_t bc, int h, int w, int H, int W) {
  if (MODE == 0) return dy[bc * H * W + (size_t)h * W + w];
  if (MODE == 1) return 0.25f * dy[bc * (H / 2) * (W / 2) + (size_t)(h >> 1) * (W / 2) + (w >> 1)];
  const size_t base = bc * (size_t)(4 * H * W) + (size_t)(2 * h) * (2 * W) + 2 * w;
  const float2 a = *reinterpret_cast<const float2*>(dy + base);
  const float2 b = *reinterpret_cast<const float2*>(dy + base + 2 * W);
  return (a.x + a.y) + (b.x + b.y);
}

template <int MODE>
__global__ __launch_bounds__(kRedThreads) void bn_bwd_partial(const float* __restrict__ x,
                                                              const float* __restrict__ dy,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ skip,
                                                              double* __restrict__ part, int B, int C, int H, int W,
                                                              float slope, int splits) {
  __shared__ double scratch[kRedThreads / 64];
  const int c = blockIdx.x, s = blockIdx.y, HW = H * W;
  const size_t total = (size_t)B * HW;
  const size_t chunk = (total + splits - 1) / splits;
  const size_t beg = (size_t)s * chunk, end = beg + chunk < total ? beg + chunk : total;
  const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
  double s1 = 0.0, s2 = 0.0;
  for (size_t i = beg + threadIdx.x; i < end; i += kRedThreads) {
    const size_t b = i / HW;
    const int hw = (int)(i - b * HW), h = hw / W, w = hw - h * W;
    const size_t bc = b * C + c;
    const float xh = (x[bc * HW + hw] - mu) * rs;
    float u = xh * ga + be;
    if (skip) u += skip[bc * HW + hw];
    float g = upstream<MODE>(dy, bc, h, w, H, W);
    if (!(u > 0.f)) g *= slope;
    s1 += (double)g;
    s2 += (double)g * (double)xh;
  }
  s1 = block_sum(s1, scratch);
  s2 = block_sum(s2, scratch);
  if (threadIdx.x == 0) {
    part[((size_t)s * 2 + 0) * C + c] = s1;
    part[((size_t)s * 2 + 1) * C + c] = s2;
  }
}

template <int MODE>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ skip, const double* __restrict__ dsums, double count,
                                    float* __restrict__ dx, float* __restrict__ dskip, int C, int H, int W,
                                    size_t n, float slope) {
  const int HW = H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bc = i / HW;
    const int hw = (int)(i - bc * HW), h = hw / W, w = hw - h * W;
    const int c = (int)(bc % C);
    const float mu = mean[c], rs = rstd[c], ga = gamma[c];
    const float xh = (x[i] - mu) * rs;
    float u = xh * ga + beta[c];
    if (skip) u += skip[i];
    float g = upstream<MODE>(dy, bc, h, w, H, W);
    if (!(u > 0.f)) g *= slope;
    const float m1 = (float)(dsums[c] / count), m2 = (float)(dsums[C + c] / count);
    dx[i] = ga * rs * (g - m1 - xh * m2);
    if (dskip) dskip[i] = g;
  }
}


// ---- vectorised (float4) backward kernels: W % 4 == 0, tensors < 2^31 elements ------------------

// upstream gradient of the 4 consecutive pixels (h, w..w+3) of plane bc (see upstream<MODE>)
template <int MODE>
__device__ __forceinline__ float4 upstream4(const float* __restrict__ dy, uint32_t bc, uint32_t h, uint32_t w,
                                            uint32_t H, uint32_t W) {
  if (MODE == 0) return *reinterpret_cast<const float4*>(dy + (size_t)bc * H * W + h * W + w);
  if (MODE == 1) {
    const float2 d = *reinterpret_cast<const float2*>(dy + (size_t)bc * (H / 2) * (W / 2) + (h >> 1) * (W / 2) + (w >> 1));
    return make_float4(0.25f * d.x, 0.25f * d.x, 0.25f * d.y, 0.25f * d.y);
  }
  const float* p = dy + (size_t)bc * (4 * H * W) + (size_t)(2 * h) * (2 * W) + 2 * w;
  const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(p + 2 * W), b1 = *reinterpret_cast<const float4*>(p + 2 * W + 4);
  return make_float4((a0.x + a0.y) + (b0.x + b0.y), (a0.z + a0.w) + (b0.z + b0.w), (a1.x + a1.y) + (b1.x + b1.y),
                     (a1.z + a1.w) + (b1.z + b1.w));
}

struct BnBwdFinal {   // fused single-launch path (splits == 1): the block writes dsums and the parameter grads
  double* dsums;
  float* dgamma;
  float* dbeta;
  int accumulate;
};

// MX (fp16 gradient planes): the block also records u = |gamma rstd| max|g| and v = max|xhat| over its share, at
// mx[idx] / mx[nmx + idx], idx = (group * splits + slice) * C + c.  The apply pass turns their maxima U, V into the
// tensor's scale: |dx| = |gamma rstd| |g - m1 - xhat m2| <= |gamma rstd| G_c (2 + X_c) <= U (2 + V), because
// |m1| = |mean g| <= G_c and |m2| = |mean g xhat| <= G_c sqrt(mean xhat^2) <= G_c.
template <int MODE, bool FUSED, bool MX = false>
__global__ __launch_bounds__(kRedThreads) void bn_bwd_partial_v4(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ skip, double* __restrict__ part, int B, int C, int H, int W, float slope, int splits,
    int w_shift, int hw_shift, BnBwdFinal f, BnGrp grp, float* __restrict__ mx = nullptr, int nmx = 0) {
  __shared__ double scratch[kRedThreads / 64];
  __shared__ float mscratch[MX ? 2 * (kRedThreads / 64) : 1];
  const uint32_t gz = FUSED ? 0u : blockIdx.z;
  const uint32_t c = blockIdx.x, s = blockIdx.y, HW = H * W, total = (uint32_t)B * HW;
  const uint32_t chunk = ((total + splits - 1) / splits + 3) & ~3u;
  const uint32_t beg = s * chunk, end = min(beg + chunk, total);
  const float ga = gamma[c], be = beta[c];
  const int ngroups = (FUSED && grp.G > 1) ? grp.G : 1;
  if (!FUSED) {   // sliced form: the group comes from blockIdx.z
    const size_t g = blockIdx.z;
    x += g * grp.xs, dy += g * grp.dys, mean += g * grp.cs, rstd += g * grp.cs, part += g * splits * 2 * C;
    if (skip) skip += g * grp.xs;
  }
  float acc_db = 0.f, acc_dg = 0.f;      // parameter gradients add up over the groups (thread 0)
  if (FUSED && threadIdx.x == 0 && f.accumulate) {
    if (f.dbeta) acc_db = f.dbeta[c];
    if (f.dgamma) acc_dg = f.dgamma[c];
  }
 for (int gi = 0; gi < ngroups; ++gi, x += grp.xs, dy += grp.dys, mean += grp.cs, rstd += grp.cs, f.dsums += 2 * grp.cs) {
  if (skip && gi) skip += grp.xs;
  const float mu = mean[c], rs = rstd[c];
  double s1 = 0.0, s2 = 0.0;
  float gmax = 0.f, xmax = 0.f;
  // Four iterations' loads are issued before the first is consumed (the accumulation order is that of the plain loop:
  // bitwise the same sums); out-of-range iterations load a valid address and add zeros.  -9..12 % on this kernel.
  constexpr int U = 4;
  for (uint32_t i0 = beg + threadIdx.x * 4; i0 < end; i0 += kRedThreads * 4 * U) {
    float4 xv4[U], g4[U], k4[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t iu = i0 + (uint32_t)u * kRedThreads * 4;
      ok[u] = iu < end;
      const uint32_t i = ok[u] ? iu : i0;
      const uint32_t b = fdiv(i, HW, hw_shift), hw = i - b * HW;
      const uint32_t h = fdiv(hw, W, w_shift), w = hw - h * W;
      const uint32_t bc = b * C + c;
      xv4[u] = *reinterpret_cast<const float4*>(x + (size_t)bc * HW + hw);
      g4[u] = upstream4<MODE>(dy, bc, h, w, H, W);
    }
    if (skip) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t iu = i0 + (uint32_t)u * kRedThreads * 4, i = iu < end ? iu : i0;
        const uint32_t b = fdiv(i, HW, hw_shift), hw = i - b * HW;
        k4[u] = *reinterpret_cast<const float4*>(skip + (size_t)(b * C + c) * HW + hw);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) k4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4 xv = xv4[u];
      float4 g = g4[u];
      const float xh0 = (xv.x - mu) * rs, xh1 = (xv.y - mu) * rs, xh2 = (xv.z - mu) * rs, xh3 = (xv.w - mu) * rs;
      float u0 = xh0 * ga + be, u1 = xh1 * ga + be, u2 = xh2 * ga + be, u3 = xh3 * ga + be;
      if (skip) u0 += k4[u].x, u1 += k4[u].y, u2 += k4[u].z, u3 += k4[u].w;
      if (!(u0 > 0.f)) g.x *= slope;
      if (!(u1 > 0.f)) g.y *= slope;
      if (!(u2 > 0.f)) g.z *= slope;
      if (!(u3 > 0.f)) g.w *= slope;
      const double a1 = ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
      const double a2 = ((double)g.x * xh0 + (double)g.y * xh1) + ((double)g.z * xh2 + (double)g.w * xh3);
      s1 += ok[u] ? a1 : 0.0;
      s2 += ok[u] ? a2 : 0.0;
      if constexpr (MX) {
        const float gm = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));
        const float xm = fmaxf(fmaxf(fabsf(xh0), fabsf(xh1)), fmaxf(fabsf(xh2), fabsf(xh3)));
        gmax = fmaxf(gmax, ok[u] ? gm : 0.f);
        xmax = fmaxf(xmax, ok[u] ? xm : 0.f);
      }
    }
  }
  s1 = block_sum(s1, scratch);
  s2 = block_sum(s2, scratch);
  if constexpr (MX) {
    gmax = wave_max(gmax), xmax = wave_max(xmax);
    if ((threadIdx.x & 63) == 0) mscratch[threadIdx.x >> 6] = gmax, mscratch[kRedThreads / 64 + (threadIdx.x >> 6)] = xmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float gm = 0.f, xm = 0.f;
      for (int k = 0; k < kRedThreads / 64; ++k) gm = fmaxf(gm, mscratch[k]), xm = fmaxf(xm, mscratch[kRedThreads / 64 + k]);
      const size_t idx = ((size_t)(FUSED ? (uint32_t)gi : gz) * splits + s) * C + c;
      mx[idx] = fabsf(ga * rs) * gm, mx[(size_t)nmx + idx] = xm;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (FUSED) {
      f.dsums[c] = s1;
      f.dsums[C + c] = s2;
      acc_db += (float)s1, acc_dg += (float)s2;
    } else {
      part[((size_t)s * 2 + 0) * C + c] = s1;
      part[((size_t)s * 2 + 1) * C + c] = s2;
    }
  }
 }
  if (FUSED && threadIdx.x == 0) {
    if (f.dbeta) f.dbeta[c] = acc_db;
    if (f.dgamma) f.dgamma[c] = acc_dg;
  }
}

template <int MODE>
__global__ void bn_bwd_apply_v4(const float* __restrict__ x, const float* __restrict__ dy,
                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ skip, const double* __restrict__ dsums, double count,
                                float* __restrict__ dx, float* __restrict__ dskip, int C, int H, int W, uint32_t n4,
                                float slope, int w_shift, int hw_shift, int c_mask) {
  const uint32_t HW = H * W;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += gridDim.x * blockDim.x) {
    const uint32_t i = j * 4;
    const uint32_t bc = fdiv(i, HW, hw_shift), hw = i - bc * HW;
    const uint32_t h = fdiv(hw, W, w_shift), w = hw - h * W;
    const uint32_t c = c_mask >= 0 ? (bc & (uint32_t)c_mask) : (bc % (uint32_t)C);
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    const float m1 = (float)(dsums[c] / count), m2 = (float)(dsums[C + c] / count), gr = ga * rs;
    const float4 xv = *reinterpret_cast<const float4*>(x + i);
    float4 g = upstream4<MODE>(dy, bc, h, w, H, W);
    const float xh0 = (xv.x - mu) * rs, xh1 = (xv.y - mu) * rs, xh2 = (xv.z - mu) * rs, xh3 = (xv.w - mu) * rs;
    float u0 = xh0 * ga + be, u1 = xh1 * ga + be, u2 = xh2 * ga + be, u3 = xh3 * ga + be;
    if (skip) {
      const float4 k = *reinterpret_cast<const float4*>(skip + i);
      u0 += k.x, u1 += k.y, u2 += k.z, u3 += k.w;
    }
    if (!(u0 > 0.f)) g.x *= slope;
    if (!(u1 > 0.f)) g.y *= slope;
    if (!(u2 > 0.f)) g.z *= slope;
    if (!(u3 > 0.f)) g.w *= slope;
    *reinterpret_cast<float4*>(dx + i) = make_float4(gr * (g.x - m1 - xh0 * m2), gr * (g.y - m1 - xh1 * m2),
                                                     gr * (g.z - m1 - xh2 * m2), gr * (g.w - m1 - xh3 * m2));
    if (dskip) *reinterpret_cast<float4*>(dskip + i) = g;
  }
}

// bn_bwd_apply_v4 that also emits dx as pre-split planes (for the data- and weight-gradient GEMMs of the conv
// below): a thread owns 8 channels x 4 pixels; dx is computed with the same expressions (bitwise equal).
// `sums.part` set: the launch folds the per-slice partial sums of bn_bwd_partial_v4 itself (see BnStatsIn above)
// and the block holding a channel group's first pixels of image 0 writes dsums and the parameter gradients.
struct BnBwdSumsIn {
  const double* part;   // [splits][2][C]
  int splits;
  double* dsums_out;
  float* dgamma;
  float* dbeta;
  int accumulate;
};

// F16: dx goes out as fp16 hi / lo planes of S dx.  S is the same in every block of every group of the call: each block
// takes the maxima U, V of the partial pass's mx arrays (see bn_bwd_partial_v4) and maps the bound U (2 + V) just under
// 2^15; block 0 of group 0 records {S, 1/S} behind plane 1.
template <int MODE, int NS, bool SUMS, bool STRIP = true, bool F16 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_planes(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ skip, const double* __restrict__ dsums, double count, float* __restrict__ dx,
    float* __restrict__ dskip, u32x4* __restrict__ planes, int B, int C, int H, int W, float slope, int w_shift,
    BnBwdSumsIn sm, size_t plane_stride, BnGrp grp, const float* __restrict__ mx = nullptr, int nmx = 0) {
  float pscale = 1.f;
  if constexpr (F16) {
    __shared__ float s_uv[8];
    float u = 0.f, v = 0.f;
    // nmx = groups * slices * C (C % 8 == 0): float4 loads, four of them in flight per array -- a plain scalar loop is
    // one exposed L2 round trip per element
    const float4* mu4 = reinterpret_cast<const float4*>(mx);
    const float4* mv4 = reinterpret_cast<const float4*>(mx + nmx);
    const int n4 = nmx >> 2;
    for (int i = threadIdx.x; i < n4; i += 4 * 256) {
      float4 tu[4], tv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = i + k * 256;
        tu[k] = idx < n4 ? mu4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        tv[k] = idx < n4 ? mv4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u = fmaxf(fmaxf(u, fmaxf(tu[k].x, tu[k].y)), fmaxf(tu[k].z, tu[k].w));
        v = fmaxf(fmaxf(v, fmaxf(tv[k].x, tv[k].y)), fmaxf(tv[k].z, tv[k].w));
      }
    }
    u = wave_max(u), v = wave_max(v);
    if ((threadIdx.x & 63) == 0) s_uv[threadIdx.x >> 6] = u, s_uv[4 + (threadIdx.x >> 6)] = v;
    __syncthreads();
    u = fmaxf(fmaxf(s_uv[0], s_uv[1]), fmaxf(s_uv[2], s_uv[3]));
    v = fmaxf(fmaxf(s_uv[4], s_uv[5]), fmaxf(s_uv[6], s_uv[7]));
    pscale = scale_for_bound(u * (2.f + v));
    if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
      ScaleRec* rec = reinterpret_cast<ScaleRec*>(planes + (size_t)NS * plane_stride);
      rec->scale = pscale, rec->inv = 1.f / pscale, rec->pad[0] = rec->pad[1] = 0.f;
    }
  }
  // Walk the tensor BACKWARDS (last group, last images first): the partial pass in front of this launch read (x, dy) in
  // ascending order, and on layers whose two tensors exceed the 256 MiB Infinity Cache an ascending second pass finds
  // exactly the lines that were evicted first; descending, it starts on the most recently read ones (c3: +4 % on this
  // kernel; c2's layers fit the cache either way).
  const uint32_t zg = gridDim.z - 1 - blockIdx.z;     // BatchNorm group (gridDim.z = 1 and a zero `grp` otherwise)
  {
    const size_t g = zg;
    x += g * grp.xs, dy += g * grp.dys, planes += g * grp.ps, mean += g * grp.cs, rstd += g * grp.cs;
    if (dsums) dsums += g * 2 * grp.cs;
    if (skip) skip += g * grp.xs;
    if (dx) dx += g * grp.xs;
    if (dskip) dskip += g * grp.xs;
  }
  __shared__ float s_m1[SUMS ? kStatCh : 1], s_m2[SUMS ? kStatCh : 1];
  __shared__ u32x4 strips[STRIP ? 4 : 1][STRIP ? 64 * 4 : 1];
  const uint32_t HW = H * W, C8 = C >> 3, per_plane = HW / 4, total = (uint32_t)B * C8 * per_plane;
  const uint32_t sweep = gridDim.x * blockDim.x, rbase = (gridDim.x - 1 - blockIdx.x) * blockDim.x;
  for (int it = (int)((total + sweep - 1) / sweep) - 1; it >= 0; --it) {
    const uint32_t base = (uint32_t)it * sweep + rbase;
    if (base >= total) continue;                      // block-uniform
    const uint32_t idx = base + threadIdx.x;
    const uint32_t g0 = base / per_plane;
    if (SUMS) {
      __syncthreads();
      const uint32_t last = min(base + blockDim.x, total) - 1, ng = last / per_plane - g0 + 1;
      if (threadIdx.x < ng * 8) {
        // BatchNorm groups (blockIdx.z): own partial sums per group; the parameter gradients add up over the groups in
        // order, done by the recording thread of group 0
        const uint32_t gz = zg, G = grp.G > 1 ? grp.G : 1;
        const uint32_t g = g0 + threadIdx.x / 8, c = (g % C8) * 8 + (threadIdx.x & 7);
        const double* part = sm.part + (size_t)gz * sm.splits * 2 * C;
        const double s1 = fold_strided(0.0, part + c, (size_t)2 * C, sm.splits);
        const double s2 = fold_strided(0.0, part + C + c, (size_t)2 * C, sm.splits);
        s_m1[threadIdx.x] = (float)(s1 / count), s_m2[threadIdx.x] = (float)(s2 / count);
        if (g < C8 && g * per_plane >= base) {
          sm.dsums_out[(size_t)gz * 2 * C + c] = s1, sm.dsums_out[(size_t)gz * 2 * C + C + c] = s2;
          if (gz == 0) {
            float db = (sm.accumulate && sm.dbeta) ? sm.dbeta[c] : 0.f, dg = (sm.accumulate && sm.dgamma) ? sm.dgamma[c] : 0.f;
            db += (float)s1, dg += (float)s2;
            for (uint32_t gg = 1; gg < G; ++gg) {
              const double* pg = sm.part + (size_t)gg * sm.splits * 2 * C;
              db += (float)fold_strided(0.0, pg + c, (size_t)2 * C, sm.splits);
              dg += (float)fold_strided(0.0, pg + C + c, (size_t)2 * C, sm.splits);
            }
            if (sm.dbeta) sm.dbeta[c] = db;
            if (sm.dgamma) sm.dgamma[c] = dg;
          }
        }
      }
      __syncthreads();
    }
    if (idx >= total) continue;
    const uint32_t bc8 = idx / per_plane, p4 = idx - bc8 * per_plane;
    const uint32_t b = bc8 / C8, c8 = bc8 - b * C8;
    const uint32_t hw = p4 * 4, h = fdiv(hw, W, w_shift), w = hw - h * W;
    float o[8][4];
    // four channels at a time, every load of the four (tensors and per-channel parameters) issued before the first is
    // consumed: channel by channel the loop was a chain of eight dependent round trips with two loads in flight
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += 4) {
      float4 xv4[4], g4[4], k4[4];
      float pm[4][4];
      double ds4[4][2];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const uint32_t c = c8 * 8 + j0 + jj, bc = b * C + c;
        xv4[jj] = *reinterpret_cast<const float4*>(x + bc * HW + hw);
        g4[jj] = upstream4<MODE>(dy, bc, h, w, H, W);
        pm[jj][0] = mean[c], pm[jj][1] = rstd[c], pm[jj][2] = gamma[c], pm[jj][3] = beta[c];
        if (!SUMS) ds4[jj][0] = dsums[c], ds4[jj][1] = dsums[C + c];
      }
      if (skip) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) k4[jj] = *reinterpret_cast<const float4*>(skip + (b * C + c8 * 8 + j0 + jj) * HW + hw);
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) k4[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        const uint32_t c = c8 * 8 + j, bc = b * C + c;
        const uint32_t i = bc * HW + hw;
        const float mu = pm[jj][0], rs = pm[jj][1], ga = pm[jj][2], be = pm[jj][3];
        const float m1 = SUMS ? s_m1[(bc8 - g0) * 8 + j] : (float)(ds4[jj][0] / count);
        const float m2 = SUMS ? s_m2[(bc8 - g0) * 8 + j] : (float)(ds4[jj][1] / count), gr = ga * rs;
        const float4 xv = xv4[jj];
        float4 g = g4[jj];
        const float xh0 = (xv.x - mu) * rs, xh1 = (xv.y - mu) * rs, xh2 = (xv.z - mu) * rs, xh3 = (xv.w - mu) * rs;
        float u0 = xh0 * ga + be, u1 = xh1 * ga + be, u2 = xh2 * ga + be, u3 = xh3 * ga + be;
        if (skip) {
          const float4 k = k4[jj];
          u0 += k.x, u1 += k.y, u2 += k.z, u3 += k.w;
        }
        if (!(u0 > 0.f)) g.x *= slope;
        if (!(u1 > 0.f)) g.y *= slope;
        if (!(u2 > 0.f)) g.z *= slope;
        if (!(u3 > 0.f)) g.w *= slope;
        const float4 d = make_float4(gr * (g.x - m1 - xh0 * m2), gr * (g.y - m1 - xh1 * m2), gr * (g.z - m1 - xh2 * m2),
                                     gr * (g.w - m1 - xh3 * m2));
        if (dx) *reinterpret_cast<float4*>(dx + i) = d;
        if (dskip) *reinterpret_cast<float4*>(dskip + i) = g;
        o[j][0] = d.x, o[j][1] = d.y, o[j][2] = d.z, o[j][3] = d.w;
      }
    }
    u32x4 chk[4][NS];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = o[j][px];
      split8<NS, F16>(v8, chk[px], pscale);
    }
    const uint32_t cidx0 = base + (threadIdx.x & ~63u);
    if (STRIP && cidx0 + 64 <= total) {   // whole wave in range (wave-uniform): coalesced plane stores
      store_planes_wave<4, NS>(planes, plane_stride, per_plane, HW, cidx0, chk, strips[STRIP ? threadIdx.x >> 6 : 0]);
    } else {
#pragma unroll
      for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int p = 0; p < NS; ++p) planes[(size_t)p * plane_stride + (size_t)bc8 * HW + hw + px] = chk[px][p];
    }
  }
}

__global__ void bn_param_grad_kernel(const double* __restrict__ local, float* dgamma, float* dbeta, int C,
                                     int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)local[c];
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)local[C + c];
}

// ------------------------------------------------------------------ pointwise / resampling
__global__ void lrelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float slope) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = lrelu(x[i], slope);
}
__global__ void lrelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                 size_t n, float slope) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = x[i] > 0.f ? dy[i] : dy[i] * slope;
}
__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                   float* __restrict__ dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float s = y[i];
    dx[i] = dy[i] * (1.f - s) * s;
  }
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                           size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    o[i] = a[i] + b[i];
}
// y[bc][ho][wo] = mean of the 2x2 window
__global__ void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, size_t nout) {
  const int Ho = H / 2, Wo = W / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bc = i / ((size_t)Ho * Wo);
    const int r = (int)(i - bc * Ho * Wo), ho = r / Wo, wo = r - ho * Wo;
    const size_t src = bc * H * W + (size_t)(2 * ho) * W + 2 * wo;
    const float2 a = *reinterpret_cast<const float2*>(x + src);
    const float2 b = *reinterpret_cast<const float2*>(x + src + W);
    y[i] = 0.25f * ((a.x + a.y) + (b.x + b.y));
  }
}
// dx[bc][h][w] = 0.25 * dy[bc][h/2][w/2]       (n = elements of dx)
__global__ void avgpool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bc = i / ((size_t)H * W);
    const int r = (int)(i - bc * H * W), h = r / W, w = r - h * W;
    dx[i] = 0.25f * dy[bc * (H / 2) * (W / 2) + (size_t)(h >> 1) * (W / 2) + (w >> 1)];
  }
}
// y[bc][2h+a][2w+b] = x[bc][h][w]             (n = elements of y, H/W = input dims)
__global__ void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, size_t n) {
  const int Ho = 2 * H, Wo = 2 * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bc = i / ((size_t)Ho * Wo);
    const int r = (int)(i - bc * Ho * Wo), h = r / Wo, w = r - h * Wo;
    y[i] = x[bc * H * W + (size_t)(h >> 1) * W + (w >> 1)];
  }
}
// dx[bc][h][w] = sum of the 4 children        (n = elements of dx, H/W = dims of dx)
__global__ void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bc = i / ((size_t)H * W);
    const int r = (int)(i - bc * H * W), h = r / W, w = r - h * W;
    dx[i] = upstream<2>(dy, bc, h, w, H, W);
  }
}

static inline int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int sh = 0;
  while ((1 << sh) < v) ++sh;
  return sh;
}

static inline int grid_for(size_t n, int per_thread = 1) {
  size_t b = cdivz(cdivz(n, per_thread), 256);
  if (b > 256 * 8) b = 256 * 8;  // grid-stride the rest (>= 8 blocks per CU resident)
  return b < 1 ? 1 : (int)b;
}

// Plane stores of the apply kernels always go through the wave-private LDS strip (store_planes_wave): measured on one
// box, whole c2 step, against direct stores: 19.75 -> 19.30 ms.  All BatchNorm groups of a layer are issued together and
// the apply pass folds the per-slice partial sums itself where a thread block covers whole channel groups (the one-launch-
// per-group and separate-finalize forms of round 2 were diagnostics and are gone).
static inline bool bn_fmt_ok(int ns) { return ns == 2 || ns == 3 || ns == ITCV_PLANES_F16X2; }
static inline int bn_splits(int B, int C, int HW) {
  const size_t total = (size_t)B * HW;
  constexpr int target = 1024;   // blocks aimed at by the sliced reductions
  int s = cdiv(target, C);
  const size_t maxs = cdivz(total, 1024);
  if ((size_t)s > maxs) s = (int)maxs;
  if (s < 1) s = 1;
  // wide layers: one block per channel fills the chip on its own and lets the statistics (and the
  // backward sums) be finalised in the same launch
  if (C >= 256 && total <= 65536) s = 1;
  return s;
}

}  // namespace itcv

using namespace itcv;

extern "C" {

size_t itcv_bn_workspace(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  // per-slice partial sums (fp64) + the per-slice maxima the backward of the fp16 planes format records (two floats)
  return (size_t)bn_splits(B, C, HW) * 2 * C * (sizeof(double) + sizeof(float));
}

int itcv_bn_moments(const float* x, double* sums, int B, int C, int HW, void* ws, size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(x && sums && B > 0 && C > 0 && HW > 0, "itcv_bn_moments");
  const int splits = bn_splits(B, C, HW);
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)splits * 2 * C * sizeof(double), "itcv_bn_moments(workspace)");
  double* part = static_cast<double*>(ws);
  hipLaunchKernelGGL(bn_moments_partial<false>, dim3(C, splits), dim3(kRedThreads), 0, S(stream), x, part, B, C, HW,
                     splits, ilog2_exact(HW), BnFinal{}, BnGrp{});
  ITCV_CHECK_LAUNCH("itcv_bn_moments");
  hipLaunchKernelGGL(combine_partials, dim3(cdiv(2 * C, 256)), dim3(256), 0, S(stream), part, sums, 2 * C, splits);
  ITCV_CHECK_LAUNCH("itcv_bn_moments(combine)");
  return 0;
}

int itcv_bn_train_stats(const float* x, int B, int C, int HW, float eps, float momentum, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd, void* ws,
                        size_t ws_bytes, void* stream) {
  ITCV_REQUIRE(x && mean && rstd && B > 0 && C > 0 && HW > 0, "itcv_bn_train_stats");
  const int splits = bn_splits(B, C, HW);
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)splits * 2 * C * sizeof(double), "itcv_bn_train_stats(workspace)");
  double* part = static_cast<double*>(ws);
  if (splits == 1) {
    const BnFinal f{(double)B * HW, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd};
    hipLaunchKernelGGL(bn_moments_partial<true>, dim3(C, 1), dim3(kRedThreads), 0, S(stream), x, part, B, C, HW, 1,
                       ilog2_exact(HW), f, BnGrp{});
    ITCV_CHECK_LAUNCH("itcv_bn_train_stats(fused)");
    return 0;
  }
  hipLaunchKernelGGL(bn_moments_partial<false>, dim3(C, splits), dim3(kRedThreads), 0, S(stream), x, part, B, C, HW,
                     splits, ilog2_exact(HW), BnFinal{}, BnGrp{});
  ITCV_CHECK_LAUNCH("itcv_bn_train_stats");
  hipLaunchKernelGGL(bn_combine_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), part, splits,
                     (double)B * HW, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd, C);
  ITCV_CHECK_LAUNCH("itcv_bn_train_stats(finalize)");
  return 0;
}

int itcv_bn_finalize(const double* sums, double count, float eps, float momentum, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd, int C,
                     void* stream) {
  ITCV_REQUIRE(sums && mean && rstd && C > 0 && count > 0, "itcv_bn_finalize");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), sums, count, eps, momentum,
                     running_mean, running_var, num_batches_tracked, mean, rstd, C);
  ITCV_CHECK_LAUNCH("itcv_bn_finalize");
  return 0;
}

int itcv_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* rstd,
                       int C, void* stream) {
  ITCV_REQUIRE(running_mean && running_var && mean && rstd && C > 0, "itcv_bn_eval_stats");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), running_mean, running_var,
                     eps, mean, rstd, C);
  ITCV_CHECK_LAUNCH("itcv_bn_eval_stats");
  return 0;
}

int itcv_bn_act_planes_supported(int C, int H, int W, int pool) {
  return C > 0 && (C & 7) == 0 && H > 0 && W > 0 && W % 4 == 0 && (!pool || H % 2 == 0);
}

int itcv_bn_act_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* skip, float* y, int B, int C, int H, int W, float slope, int pool, void* planes,
                    int ns, size_t plane_stride, void* stream) {
  ITCV_REQUIRE(x && mean && rstd && gamma && beta && (y || planes) && B > 0 && C > 0 && H > 0 && W > 0, "itcv_bn_act_fwd");
  ITCV_REQUIRE((size_t)B * C * H * W < (1ull << 31), "itcv_bn_act_fwd(tensor < 2^31 elements)");
  if (planes) {
    ITCV_REQUIRE(bn_fmt_ok(ns) && itcv_bn_act_planes_supported(C, H, W, pool), "itcv_bn_act_fwd(planes)");
    const size_t threads = (size_t)B * (C / 8) * ((pool ? (H / 2) * (W / 2) : H * W) / (pool ? 2 : 4));
    const dim3 grid(grid_for(threads)), blk(256);
    u32x4* pl = static_cast<u32x4*>(planes);
    const size_t pstride = plane_stride ? plane_stride : (size_t)B * (C / 8) * (pool ? (H / 2) * (W / 2) : H * W);
#define ITCV_FWD_PLANES(POOL_, NS_, F_)                                                                               \
  do {                                                                                                                \
      launch_timed((bn_act_fwd_planes_kernel<POOL_, NS_, false, true, F_>), grid, blk, 0, S(stream), x, mean, rstd, gamma, \
                         beta, skip, y, pl, B, C, H, W, slope, BnStatsIn{}, pstride, BnGrp{});                                   \
  } while (0)
#define ITCV_FWD_PLANES_NS(POOL_)                                   \
  do {                                                              \
    if (ns == ITCV_PLANES_F16X2) ITCV_FWD_PLANES(POOL_, 2, true);   \
    else if (ns == 2) ITCV_FWD_PLANES(POOL_, 2, false);             \
    else ITCV_FWD_PLANES(POOL_, 3, false);                          \
  } while (0)
    if (pool) ITCV_FWD_PLANES_NS(1);
    else ITCV_FWD_PLANES_NS(0);
#undef ITCV_FWD_PLANES_NS
#undef ITCV_FWD_PLANES
    ITCV_CHECK_LAUNCH("itcv_bn_act_fwd(planes)");
    return 0;
  }
  if (pool) {
    ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_bn_act_fwd(pool)");
    const size_t nout = (size_t)B * C * (H / 2) * (W / 2);
    hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(grid_for(nout)), dim3(256), 0, S(stream), x, mean, rstd, gamma,
                       beta, skip, y, C, H, W, nout, slope, ilog2_exact(H * W), ilog2_exact(C) >= 0 ? C - 1 : -1);
  } else {
    ITCV_REQUIRE((H * W) % 4 == 0, "itcv_bn_act_fwd(H*W % 4)");
    const size_t nout = (size_t)B * C * H * W;
    hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(grid_for(nout, 4)), dim3(256), 0, S(stream), x, mean, rstd, gamma,
                       beta, skip, y, C, H, W, nout, slope, ilog2_exact(H * W), ilog2_exact(C) >= 0 ? C - 1 : -1);
  }
  ITCV_CHECK_LAUNCH("itcv_bn_act_fwd");
  return 0;
}

}  // extern "C"
// mx != NULL (fp16 gradient planes): also record the per-slice maxima, mx[2][splits * C] (see bn_bwd_partial_v4)
static int bwd_reduce_impl(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, const float* skip, double* dsums, float* dgamma, float* dbeta,
                           int accumulate, int B, int C, int H, int W, float slope, int pool, int up2, void* ws,
                           size_t ws_bytes, float* mx, void* stream) {
  ITCV_REQUIRE(x && dy && mean && rstd && gamma && beta && dsums && B > 0 && C > 0, "itcv_bn_act_bwd_reduce");
  ITCV_REQUIRE(!(pool && up2), "itcv_bn_act_bwd_reduce(pool and up2 are exclusive)");
  if (pool) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_bn_act_bwd_reduce(pool)");
  const int splits = bn_splits(B, C, H * W);
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)splits * 2 * C * sizeof(double), "itcv_bn_act_bwd_reduce(workspace)");
  double* part = static_cast<double*>(ws);
  dim3 grid(C, splits);
  const bool vec = (W % 4 == 0) && ((size_t)B * C * H * W < (1ull << 31));
  const int wsh = ilog2_exact(W), hwsh = ilog2_exact(H * W);
  hipStream_t st = S(stream);
  ITCV_REQUIRE(!mx || vec, "itcv_bn_train_bwd(fp16 planes need W % 4 == 0)");
  const int nmx = splits * C;
#define ITCV_BWD_PARTIAL(MODE)                                                                                   \
  do {                                                                                                           \
    if (mx && splits == 1)                                                                                       \
      hipLaunchKernelGGL((bn_bwd_partial_v4<MODE, true, true>), grid, dim3(kRedThreads), 0, st, x, dy, mean, rstd, \
                         gamma, beta, skip, part, B, C, H, W, slope, 1, wsh, hwsh,                               \
                         BnBwdFinal{dsums, dgamma, dbeta, accumulate}, BnGrp{}, mx, nmx);                        \
    else if (mx)                                                                                                 \
      hipLaunchKernelGGL((bn_bwd_partial_v4<MODE, false, true>), grid, dim3(kRedThreads), 0, st, x, dy, mean, rstd, \
                         gamma, beta, skip, part, B, C, H, W, slope, splits, wsh, hwsh, BnBwdFinal{}, BnGrp{}, mx, nmx); \
    else if (vec && splits == 1)                                                                                 \
      hipLaunchKernelGGL((bn_bwd_partial_v4<MODE, true>), grid, dim3(kRedThreads), 0, st, x, dy, mean, rstd,      \
                         gamma, beta, skip, part, B, C, H, W, slope, 1, wsh, hwsh,                               \
                         BnBwdFinal{dsums, dgamma, dbeta, accumulate}, BnGrp{});                                          \
    else if (vec)                                                                                                \
      hipLaunchKernelGGL((bn_bwd_partial_v4<MODE, false>), grid, dim3(kRedThreads), 0, st, x, dy, mean, rstd,     \
                         gamma, beta, skip, part, B, C, H, W, slope, splits, wsh, hwsh, BnBwdFinal{}, BnGrp{});           \
    else                                                                                                         \
      hipLaunchKernelGGL(bn_bwd_partial<MODE>, grid, dim3(kRedThreads), 0, st, x, dy, mean, rstd, gamma, beta,    \
                         skip, part, B, C, H, W, slope, splits);                                                 \
  } while (0)
  if (pool)
    ITCV_BWD_PARTIAL(1);
  else if (up2)
    ITCV_BWD_PARTIAL(2);
  else
    ITCV_BWD_PARTIAL(0);
#undef ITCV_BWD_PARTIAL
  ITCV_CHECK_LAUNCH("itcv_bn_act_bwd_reduce");
  if (vec && splits == 1) return 0;   // the fused kernel already wrote dsums and the parameter gradients
  hipLaunchKernelGGL(bn_combine_param_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), part, dsums, C, splits,
                     dgamma, dbeta, accumulate);
  ITCV_CHECK_LAUNCH("itcv_bn_act_bwd_reduce(combine)");
  return 0;
}

static int bwd_apply_impl(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, const float* skip, const double* dsums, const double* local_dsums,
                          double count, float* dx, float* dskip, float* dgamma, float* dbeta, int accumulate, int B,
                          int C, int H, int W, float slope, int pool, int up2, void* dx_planes, int ns,
                          size_t plane_stride, const float* mx, int nmx, void* stream) {
  ITCV_REQUIRE(x && dy && mean && rstd && gamma && beta && dsums && (dx || dx_planes) && B > 0 && C > 0 && count > 0,
               "itcv_bn_act_bwd_apply");
  const size_t pstride = plane_stride ? plane_stride : (size_t)B * (C / 8) * H * W;
  ITCV_REQUIRE(!(pool && up2), "itcv_bn_act_bwd_apply(pool and up2 are exclusive)");
  const size_t n = (size_t)B * C * H * W;
  const bool vec = (W % 4 == 0) && (n < (1ull << 31));
  const int wsh = ilog2_exact(W), hwsh = ilog2_exact(H * W), cmask = ilog2_exact(C) >= 0 ? C - 1 : -1;
  hipStream_t st = S(stream);
  if (dx_planes) {
    ITCV_REQUIRE(bn_fmt_ok(ns) && vec && itcv_bn_act_planes_supported(C, H, W, 0), "itcv_bn_act_bwd_apply(planes)");
    if (ns == ITCV_PLANES_F16X2 && !mx)
      return fail("%s: fp16 gradient planes take their scale from the maxima of the reduce pass: use itcv_bn_train_bwd",
                  "itcv_bn_act_bwd_apply");
    const dim3 grid(grid_for(n / 32)), blk(256);
    u32x4* pl = static_cast<u32x4*>(dx_planes);
#define ITCV_BWD_PLANES(MODE_, NS_, F_)                                                                          \
  do {                                                                                                           \
      launch_timed((bn_bwd_apply_planes<MODE_, NS_, false, true, F_>), grid, blk, 0, st, x, dy, mean, rstd, gamma, beta, \
                         skip, dsums, count, dx, dskip, pl, B, C, H, W, slope, wsh, BnBwdSumsIn{}, pstride, BnGrp{}, mx, nmx); \
  } while (0)
#define ITCV_BWD_PLANES_NS(MODE_)                                   \
  do {                                                              \
    if (ns == ITCV_PLANES_F16X2) ITCV_BWD_PLANES(MODE_, 2, true);   \
    else if (ns == 2) ITCV_BWD_PLANES(MODE_, 2, false);             \
    else ITCV_BWD_PLANES(MODE_, 3, false);                          \
  } while (0)
    if (pool) ITCV_BWD_PLANES_NS(1);
    else if (up2) ITCV_BWD_PLANES_NS(2);
    else ITCV_BWD_PLANES_NS(0);
#undef ITCV_BWD_PLANES_NS
#undef ITCV_BWD_PLANES
    ITCV_CHECK_LAUNCH("itcv_bn_act_bwd_apply(planes)");
    if (dgamma || dbeta) {
      hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, local_dsums ? local_dsums : dsums,
                         dgamma, dbeta, C, accumulate);
      ITCV_CHECK_LAUNCH("itcv_bn_act_bwd_apply(param grads)");
    }
    return 0;
  }
#define ITCV_BWD_APPLY(MODE)                                                                                      \
  do {                                                                                                            \
    if (vec)                                                                                                      \
      hipLaunchKernelGGL(bn_bwd_apply_v4<MODE>, dim3(grid_for(n, 4)), dim3(256), 0, st, x, dy, mean, rstd, gamma,  \
                         beta, skip, dsums, count, dx, dskip, C, H, W, (uint32_t)(n / 4), slope, wsh, hwsh, cmask); \
    else                                                                                                          \
      hipLaunchKernelGGL(bn_bwd_apply_kernel<MODE>, dim3(grid_for(n)), dim3(256), 0, st, x, dy, mean, rstd, gamma, \
                         beta, skip, dsums, count, dx, dskip, C, H, W, n, slope);                                 \
  } while (0)
  if (pool)
    ITCV_BWD_APPLY(1);
  else if (up2)
    ITCV_BWD_APPLY(2);
  else
    ITCV_BWD_APPLY(0);
#undef ITCV_BWD_APPLY
  ITCV_CHECK_LAUNCH("itcv_bn_act_bwd_apply");
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream),
                       local_dsums ? local_dsums : dsums, dgamma, dbeta, C, accumulate);
    ITCV_CHECK_LAUNCH("itcv_bn_act_bwd_apply(param grads)");
  }
  return 0;
}

extern "C" {
int itcv_bn_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, const float* skip, double* dsums, float* dgamma, float* dbeta,
                           int accumulate, int B, int C, int H, int W, float slope, int pool, int up2, void* ws,
                           size_t ws_bytes, void* stream) {
  return bwd_reduce_impl(x, dy, mean, rstd, gamma, beta, skip, dsums, dgamma, dbeta, accumulate, B, C, H, W, slope, pool, up2,
                         ws, ws_bytes, nullptr, stream);
}
int itcv_bn_act_bwd_apply(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, const float* skip, const double* dsums, const double* local_dsums,
                          double count, float* dx, float* dskip, float* dgamma, float* dbeta, int accumulate, int B,
                          int C, int H, int W, float slope, int pool, int up2, void* dx_planes, int ns,
                          size_t plane_stride, void* stream) {
  return bwd_apply_impl(x, dy, mean, rstd, gamma, beta, skip, dsums, local_dsums, count, dx, dskip, dgamma, dbeta, accumulate,
                        B, C, H, W, slope, pool, up2, dx_planes, ns, plane_stride, nullptr, 0, stream);
}

// ---- single-rank training forms: statistics + apply (forward), sums + apply (backward) ----------------------
// Same results as itcv_bn_train_stats + itcv_bn_act_fwd (resp. itcv_bn_act_bwd_reduce + _apply); where the planes
// kernels apply and the reduction is sliced, the apply launch folds the slices itself: two launches instead of three.
int itcv_bn_train_fwd(const float* x, const float* gamma, const float* beta, const float* skip, float* y, void* planes,
                      int ns, int B, int C, int H, int W, float slope, int pool, float eps, float momentum,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd,
                      void* ws, size_t ws_bytes, size_t plane_stride, const float* tile_stats, int tiles, int tile_pitch,
                      int groups, void* stream) {
  ITCV_REQUIRE(x && gamma && beta && mean && rstd && (y || planes) && B > 0 && C > 0 && H > 0 && W > 0, "itcv_bn_train_fwd");
  // roofline leg of bench.py: one event pair around the apply kernel (HBM-bound: its "work" is algorithmic bytes -- read
  // x once more [+ skip], write the planes [+ the fp32 output])
  const double px_out = (double)(groups > 1 ? groups : 1) * B * C * (pool ? H * W / 4 : H * W);
  ProfScope bn_prof(S(stream), 13, ilog2_exact(W) >= 0 ? ilog2_exact(W) : 0, C / 8, pool ? 1 : 0, planes ? ns : 0,
                    (double)(groups > 1 ? groups : 1) * B * C * H * W * 4.0 * (skip ? 2 : 1) + px_out * ((planes ? 4.0 : 0.0) + (y ? 4.0 : 0.0)));
  if (groups > 1) {
    // `groups` BatchNorm groups of B images each, stacked along the batch dimension of x / y / planes; mean / rstd are
    // [groups][C].  Small layers (one block per channel computes the statistics): ONE statistics launch that walks the
    // groups in order (the running buffers advance group by group) and ONE apply launch with the group in blockIdx.z.
    // Other shapes: the groups are issued one after the other.
    ITCV_REQUIRE(!planes || plane_stride, "itcv_bn_train_fwd(groups need the plane stride of the whole tensor)");
    const int HWg = H * W, HWo = pool ? HWg / 4 : HWg;
    const size_t xs = (size_t)B * C * HWg, os = (size_t)B * C * HWo, ps = (size_t)B * (C / 8) * HWo;
    const int gsplits = bn_splits(B, C, HWg);
    const bool mergeable = !tile_stats && planes && bn_fmt_ok(ns) && itcv_bn_act_planes_supported(C, H, W, pool) &&
                           xs < (1ull << 31);
    const bool merged = mergeable && gsplits == 1;
    if (mergeable && gsplits > 1 && ws && ws_bytes >= (size_t)groups * gsplits * 2 * C * sizeof(double)) {
      // large layers: sliced statistics of all groups in one launch (group = blockIdx.z), a fold that finalises the groups
      // in order, one apply launch for all groups
      const BnGrp grp{xs, os, 0, ps, C, groups};
      double* part = static_cast<double*>(ws);
      hipLaunchKernelGGL(bn_moments_partial<false>, dim3(C, gsplits, groups), dim3(kRedThreads), 0, S(stream), x, part, B, C,
                         HWg, gsplits, ilog2_exact(HWg), BnFinal{}, grp);
      ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(grouped partials)");
      const int per_plane_g = pool ? (HWg / 4) / 2 : HWg / 4;
      const bool fold_in_apply = per_plane_g >= 64;   // the apply pass folds the partial sums itself
      if (!fold_in_apply) {
        hipLaunchKernelGGL(bn_combine_finalize_groups_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), part, gsplits,
                           groups, (double)B * HWg, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd,
                           C);
        ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(grouped finalize)");
      }
      const BnStatsIn stg{part, gsplits, (double)B * HWg, eps, momentum, running_mean, running_var, num_batches_tracked, mean,
                          rstd};
      const size_t threads = (size_t)B * (C / 8) * (HWo / (pool ? 2 : 4));
      const dim3 grid(grid_for(threads), 1, groups), blk(256);
      u32x4* pl = static_cast<u32x4*>(planes);
#define ITCV_FWD_GRP2_K(POOL_, NS_, ST_, F_)                                                                              \
  launch_timed((bn_act_fwd_planes_kernel<POOL_, NS_, ST_, true, F_>), grid, blk, 0, S(stream), x, mean, rstd, gamma, beta, \
                     skip, y, pl, B, C, H, W, slope, (ST_) ? stg : BnStatsIn{}, plane_stride, grp)
#define ITCV_FWD_GRP2(POOL_, NS_, F_)                                  \
  do {                                                                 \
    if (fold_in_apply) ITCV_FWD_GRP2_K(POOL_, NS_, true, F_);          \
    else ITCV_FWD_GRP2_K(POOL_, NS_, false, F_);                       \
  } while (0)
#define ITCV_FWD_GRP2_NS(POOL_)                                     \
  do {                                                              \
    if (ns == ITCV_PLANES_F16X2) ITCV_FWD_GRP2(POOL_, 2, true);     \
    else if (ns == 2) ITCV_FWD_GRP2(POOL_, 2, false);               \
    else ITCV_FWD_GRP2(POOL_, 3, false);                            \
  } while (0)
      if (pool) ITCV_FWD_GRP2_NS(1);
      else ITCV_FWD_GRP2_NS(0);
#undef ITCV_FWD_GRP2_NS
#undef ITCV_FWD_GRP2_K
#undef ITCV_FWD_GRP2
      ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(grouped apply)");
      return 0;
    }
    if (!merged) {
      for (int g = 0; g < groups; ++g)
        if (int e = itcv_bn_train_fwd(x + g * xs, gamma, beta, skip ? skip + g * xs : nullptr, y ? y + g * os : nullptr,
                                      planes ? static_cast<u32x4*>(planes) + g * ps : nullptr, ns, B, C, H, W, slope, pool,
                                      eps, momentum, running_mean, running_var, num_batches_tracked, mean + (size_t)g * C,
                                      rstd + (size_t)g * C, ws, ws_bytes, plane_stride,
                                      tile_stats ? tile_stats + (size_t)g * tiles : nullptr, tiles, tile_pitch, 1, stream))
          return e;
      return 0;
    }
    const BnGrp grp{xs, os, 0, ps, C, groups};
    const BnFinal f{(double)B * HWg, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd};
    hipLaunchKernelGGL(bn_moments_partial<true>, dim3(C, 1), dim3(kRedThreads), 0, S(stream), x, static_cast<double*>(nullptr),
                       B, C, HWg, 1, ilog2_exact(HWg), f, grp);
    ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(grouped statistics)");
    const size_t threads = (size_t)B * (C / 8) * (HWo / (pool ? 2 : 4));
    const dim3 grid(grid_for(threads), 1, groups), blk(256);
    u32x4* pl = static_cast<u32x4*>(planes);
#define ITCV_FWD_GRP(POOL_, NS_, F_)                                                                                      \
  do {                                                                                                                    \
      launch_timed((bn_act_fwd_planes_kernel<POOL_, NS_, false, true, F_>), grid, blk, 0, S(stream), x, mean, rstd, gamma, \
                         beta, skip, y, pl, B, C, H, W, slope, BnStatsIn{}, plane_stride, grp);                             \
  } while (0)
#define ITCV_FWD_GRP_NS(POOL_)                                     \
  do {                                                             \
    if (ns == ITCV_PLANES_F16X2) ITCV_FWD_GRP(POOL_, 2, true);     \
    else if (ns == 2) ITCV_FWD_GRP(POOL_, 2, false);               \
    else ITCV_FWD_GRP(POOL_, 3, false);                            \
  } while (0)
    if (pool) ITCV_FWD_GRP_NS(1);
    else ITCV_FWD_GRP_NS(0);
#undef ITCV_FWD_GRP_NS
#undef ITCV_FWD_GRP
    ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(grouped apply)");
    return 0;
  }
  const size_t pstride = plane_stride ? plane_stride : (size_t)B * (C / 8) * (pool ? (H / 2) * (W / 2) : H * W);
  if (tile_stats) {   // statistics come from the producing conv's epilogue: fold the tiles, then one apply pass
    ITCV_REQUIRE(tiles > 0 && tile_pitch >= tiles, "itcv_bn_train_fwd(tile statistics)");
    hipLaunchKernelGGL(bn_tile_stats_finalize_kernel, dim3(C), dim3(256), 0, S(stream), tile_stats, tiles, tile_pitch,
                       (double)B * H * W, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd, C);
    ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(tile statistics)");
    return itcv_bn_act_fwd(x, mean, rstd, gamma, beta, skip, y, B, C, H, W, slope, pool, planes, ns, plane_stride, stream);
  }
  const int HW = H * W, splits = bn_splits(B, C, HW);
  const int per_plane = pool ? (HW / 4) / 2 : HW / 4;
  const bool fusable = planes && bn_fmt_ok(ns) && itcv_bn_act_planes_supported(C, H, W, pool) && splits > 1 &&
                       per_plane >= 64 && (size_t)B * C * HW < (1ull << 31);
  if (!fusable) {
    if (int e = itcv_bn_train_stats(x, B, C, HW, eps, momentum, running_mean, running_var, num_batches_tracked, mean,
                                    rstd, ws, ws_bytes, stream))
      return e;
    return itcv_bn_act_fwd(x, mean, rstd, gamma, beta, skip, y, B, C, H, W, slope, pool, planes, ns, plane_stride, stream);
  }
  ITCV_REQUIRE(ws && ws_bytes >= (size_t)splits * 2 * C * sizeof(double), "itcv_bn_train_fwd(workspace)");
  double* part = static_cast<double*>(ws);
  hipLaunchKernelGGL(bn_moments_partial<false>, dim3(C, splits), dim3(kRedThreads), 0, S(stream), x, part, B, C, HW,
                     splits, ilog2_exact(HW), BnFinal{}, BnGrp{});
  ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(partials)");
  const BnStatsIn st{part, splits, (double)B * HW, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd};
  const size_t threads = (size_t)B * (C / 8) * per_plane;
  const dim3 grid(grid_for(threads)), blk(256);
  u32x4* pl = static_cast<u32x4*>(planes);
#define ITCV_FWD_FUSED(POOL_, NS_, F_)                                                                               \
  do {                                                                                                                \
      launch_timed((bn_act_fwd_planes_kernel<POOL_, NS_, true, true, F_>), grid, blk, 0, S(stream), x, mean, rstd, gamma, \
                         beta, skip, y, pl, B, C, H, W, slope, st, pstride, BnGrp{});                                            \
  } while (0)
#define ITCV_FWD_FUSED_NS(POOL_)                                     \
  do {                                                               \
    if (ns == ITCV_PLANES_F16X2) ITCV_FWD_FUSED(POOL_, 2, true);     \
    else if (ns == 2) ITCV_FWD_FUSED(POOL_, 2, false);               \
    else ITCV_FWD_FUSED(POOL_, 3, false);                            \
  } while (0)
  if (pool) ITCV_FWD_FUSED_NS(1);
  else ITCV_FWD_FUSED_NS(0);
#undef ITCV_FWD_FUSED_NS
#undef ITCV_FWD_FUSED
  ITCV_CHECK_LAUNCH("itcv_bn_train_fwd(apply)");
  return 0;
}

int itcv_bn_train_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const float* skip, double* dsums, float* dx, float* dskip, void* dx_planes,
                      int ns, float* dgamma, float* dbeta, int accumulate, int B, int C, int H, int W, float slope,
                      int pool, int up2, void* ws, size_t ws_bytes, size_t plane_stride, int groups, void* stream) {
  ITCV_REQUIRE(x && dy && mean && rstd && gamma && beta && dsums && (dx || dx_planes) && B > 0 && C > 0,
               "itcv_bn_train_bwd");
  ITCV_REQUIRE(!(pool && up2), "itcv_bn_train_bwd(pool and up2 are exclusive)");
  // roofline leg of bench.py: the apply kernel reads x and dy once more and writes the planes [+ fp32 dx, + dskip]
  const double bn_el = (double)(groups > 1 ? groups : 1) * B * C * H * W;
  ProfScope bn_prof(S(stream), 14, ilog2_exact(W) >= 0 ? ilog2_exact(W) : 0, C / 8, pool ? 1 : (up2 ? 2 : 0), dx_planes ? ns : 0,
                    bn_el * 4.0 * (1.0 + (pool ? 0.25 : (up2 ? 4.0 : 1.0)) + (skip ? 1.0 : 0.0)) +
                        bn_el * ((dx_planes ? 4.0 : 0.0) + (dx ? 4.0 : 0.0) + (dskip ? 4.0 : 0.0)));
  if (groups > 1) {   // see itcv_bn_train_fwd; dsums is [groups][2C], the parameter gradients add up over the groups
    ITCV_REQUIRE(!dx_planes || plane_stride, "itcv_bn_train_bwd(groups need the plane stride of the whole tensor)");
    const int HWg = H * W;
    const size_t xs = (size_t)B * C * HWg, dys = pool ? xs / 4 : (up2 ? xs * 4 : xs), ps = (size_t)B * (C / 8) * HWg;
    const bool vecg = (W % 4 == 0) && xs < (1ull << 31);
    const int gsplits = bn_splits(B, C, HWg);
    const bool mergeable = vecg && dx_planes && bn_fmt_ok(ns) && itcv_bn_act_planes_supported(C, H, W, 0);
    const bool merged = mergeable && gsplits == 1;
    const bool f16 = ns == ITCV_PLANES_F16X2;
    // fp16 planes: the maxima of all groups ([2][groups * gsplits * C] floats) follow the partial sums in the workspace
    const int nmx = groups * gsplits * C;
    const size_t ws_need = (size_t)groups * gsplits * 2 * C * sizeof(double) + (f16 ? (size_t)2 * nmx * sizeof(float) : 0);
    float* mx = f16 && ws ? reinterpret_cast<float*>(static_cast<double*>(ws) + (size_t)groups * gsplits * 2 * C) : nullptr;
    if (f16 && !(mergeable && ws && ws_bytes >= ws_need))
      return fail("%s: fp16 gradient planes need the merged group path (W %% 4 == 0, planes, a workspace of itcv_bn_workspace * groups)",
                  "itcv_bn_train_bwd");
    if (mergeable && gsplits > 1 && ws && ws_bytes >= ws_need) {
      if (pool) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_bn_train_bwd(pool)");
      const BnGrp grp{xs, 0, dys, ps, C, groups};
      hipStream_t st = S(stream);
      const int wsh = ilog2_exact(W), hwsh = ilog2_exact(HWg);
      double* part = static_cast<double*>(ws);
      const dim3 rgrid(C, gsplits, groups), agrid(grid_for(xs / 32), 1, groups), blk(256);
      u32x4* pl = static_cast<u32x4*>(dx_planes);
      const double count = (double)B * HWg;
      const bool fold_in_apply = HWg / 4 >= 64;   // the apply pass folds the partial sums itself
      const BnBwdSumsIn smg{part, gsplits, dsums, dgamma, dbeta, accumulate};
#define ITCV_BWD_GRP2_K(MODE_, NS_, SUMS_, F_)                                                                           \
  launch_timed((bn_bwd_apply_planes<MODE_, NS_, SUMS_, true, F_>), agrid, blk, 0, st, x, dy, mean, rstd, gamma, beta, \
                     skip, (SUMS_) ? static_cast<const double*>(nullptr) : dsums, count, dx, dskip, pl, B, C, H, W, slope, \
                     wsh, (SUMS_) ? smg : BnBwdSumsIn{}, plane_stride, grp, mx, nmx)
#define ITCV_BWD_GRP2(MODE_, NS_, F_)                                                                                    \
  do {                                                                                                                   \
    hipLaunchKernelGGL((bn_bwd_partial_v4<MODE_, false, F_>), rgrid, dim3(kRedThreads), 0, st, x, dy, mean, rstd, gamma, beta, \
                       skip, part, B, C, H, W, slope, gsplits, wsh, hwsh, BnBwdFinal{}, grp, mx, nmx);                   \
    if (fold_in_apply) {                                                                                                 \
      ITCV_BWD_GRP2_K(MODE_, NS_, true, F_);                                                                             \
    } else {                                                                                                             \
      hipLaunchKernelGGL(bn_combine_param_groups_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, part, dsums, C, gsplits,   \
                         groups, dgamma, dbeta, accumulate);                                                             \
      ITCV_BWD_GRP2_K(MODE_, NS_, false, F_);                                                                            \
    }                                                                                                                    \
  } while (0)
#define ITCV_BWD_GRP2_NS(MODE_)                          \
  do {                                                   \
    if (f16) ITCV_BWD_GRP2(MODE_, 2, true);              \
    else if (ns == 2) ITCV_BWD_GRP2(MODE_, 2, false);    \
    else ITCV_BWD_GRP2(MODE_, 3, false);                 \
  } while (0)
      if (pool) ITCV_BWD_GRP2_NS(1);
      else if (up2) ITCV_BWD_GRP2_NS(2);
      else ITCV_BWD_GRP2_NS(0);
#undef ITCV_BWD_GRP2_NS
#undef ITCV_BWD_GRP2
#undef ITCV_BWD_GRP2_K
      ITCV_CHECK_LAUNCH("itcv_bn_train_bwd(grouped, sliced)");
      return 0;
    }
    if (!merged) {
      for (int g = 0; g < groups; ++g)
        if (int e = itcv_bn_train_bwd(x + g * xs, dy + g * dys, mean + (size_t)g * C, rstd + (size_t)g * C, gamma, beta,
                                      skip ? skip + g * xs : nullptr, dsums + (size_t)g * 2 * C, dx ? dx + g * xs : nullptr,
                                      dskip ? dskip + g * xs : nullptr,
                                      dx_planes ? static_cast<u32x4*>(dx_planes) + g * ps : nullptr, ns, dgamma, dbeta,
                                      (accumulate || g > 0) ? 1 : 0, B, C, H, W, slope, pool, up2, ws, ws_bytes, plane_stride,
                                      1, stream))
          return e;
      return 0;
    }
    if (pool) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_bn_train_bwd(pool)");
    const BnGrp grp{xs, 0, dys, ps, C, groups};
    hipStream_t st = S(stream);
    const int wsh = ilog2_exact(W), hwsh = ilog2_exact(HWg);
    const BnBwdFinal bf{dsums, dgamma, dbeta, accumulate};
    const dim3 rgrid(C, 1), agrid(grid_for(xs / 32), 1, groups), blk(256);
    u32x4* pl = static_cast<u32x4*>(dx_planes);
    const double count = (double)B * HWg;
#define ITCV_BWD_GRP(MODE_, NS_, F_)                                                                                     \
  do {                                                                                                                   \
    hipLaunchKernelGGL((bn_bwd_partial_v4<MODE_, true, F_>), rgrid, dim3(kRedThreads), 0, st, x, dy, mean, rstd, gamma, beta, \
                       skip, static_cast<double*>(nullptr), B, C, H, W, slope, 1, wsh, hwsh, bf, grp, mx, nmx);          \
      launch_timed((bn_bwd_apply_planes<MODE_, NS_, false, true, F_>), agrid, blk, 0, st, x, dy, mean, rstd, gamma, beta, \
                         skip, dsums, count, dx, dskip, pl, B, C, H, W, slope, wsh, BnBwdSumsIn{}, plane_stride, grp, mx, nmx); \
  } while (0)
#define ITCV_BWD_GRP_NS(MODE_)                          \
  do {                                                  \
    if (f16) ITCV_BWD_GRP(MODE_, 2, true);              \
    else if (ns == 2) ITCV_BWD_GRP(MODE_, 2, false);    \
    else ITCV_BWD_GRP(MODE_, 3, false);                 \
  } while (0)
    if (pool) ITCV_BWD_GRP_NS(1);
    else if (up2) ITCV_BWD_GRP_NS(2);
    else ITCV_BWD_GRP_NS(0);
#undef ITCV_BWD_GRP_NS
#undef ITCV_BWD_GRP
    ITCV_CHECK_LAUNCH("itcv_bn_train_bwd(grouped)");
    return 0;
  }
  const int HW = H * W, splits = bn_splits(B, C, HW);
  const size_t n = (size_t)B * C * HW;
  const bool vec = (W % 4 == 0) && n < (1ull << 31);
  const bool fusable = dx_planes && bn_fmt_ok(ns) && vec && itcv_bn_act_planes_supported(C, H, W, 0) &&
                       splits > 1 && HW / 4 >= 64;
  const bool f16 = dx_planes && ns == ITCV_PLANES_F16X2;
  const int nmx = splits * C;
  const size_t ws_need = (size_t)splits * 2 * C * sizeof(double) + (f16 ? (size_t)2 * nmx * sizeof(float) : 0);
  if (f16) ITCV_REQUIRE(ws && ws_bytes >= ws_need, "itcv_bn_train_bwd(workspace, fp16 planes)");
  float* mx = f16 ? reinterpret_cast<float*>(static_cast<double*>(ws) + (size_t)splits * 2 * C) : nullptr;
  if (!fusable) {
    if (int e = bwd_reduce_impl(x, dy, mean, rstd, gamma, beta, skip, dsums, dgamma, dbeta, accumulate, B, C, H, W, slope,
                                pool, up2, ws, ws_bytes, mx, stream))
      return e;
    return bwd_apply_impl(x, dy, mean, rstd, gamma, beta, skip, dsums, nullptr, (double)B * HW, dx, dskip, nullptr, nullptr,
                          0, B, C, H, W, slope, pool, up2, dx_planes, ns, plane_stride, mx, nmx, stream);
  }
  if (pool) ITCV_REQUIRE(H % 2 == 0 && W % 2 == 0, "itcv_bn_train_bwd(pool)");
  ITCV_REQUIRE(ws && ws_bytes >= ws_need, "itcv_bn_train_bwd(workspace)");
  double* part = static_cast<double*>(ws);
  hipStream_t st = S(stream);
  const int wsh = ilog2_exact(W), hwsh = ilog2_exact(HW);
  const BnBwdSumsIn sm{part, splits, dsums, dgamma, dbeta, accumulate};
  const dim3 rgrid(C, splits), agrid(grid_for(n / 32)), blk(256);
  u32x4* pl = static_cast<u32x4*>(dx_planes);
  const double count = (double)B * HW;
  const size_t pstride = plane_stride ? plane_stride : (size_t)B * (C / 8) * HW;
#define ITCV_BWD_FUSED(MODE_, NS_, F_)                                                                                \
  do {                                                                                                                \
    hipLaunchKernelGGL((bn_bwd_partial_v4<MODE_, false, F_>), rgrid, dim3(kRedThreads), 0, st, x, dy, mean, rstd, gamma, \
                       beta, skip, part, B, C, H, W, slope, splits, wsh, hwsh, BnBwdFinal{}, BnGrp{}, mx, nmx);       \
      launch_timed((bn_bwd_apply_planes<MODE_, NS_, true, true, F_>), agrid, blk, 0, st, x, dy, mean, rstd, gamma, beta, \
                         skip, static_cast<const double*>(nullptr), count, dx, dskip, pl, B, C, H, W, slope, wsh, sm,   \
                         pstride, BnGrp{}, mx, nmx);                                                                  \
  } while (0)
#define ITCV_BWD_FUSED_NS(MODE_)                          \
  do {                                                    \
    if (f16) ITCV_BWD_FUSED(MODE_, 2, true);              \
    else if (ns == 2) ITCV_BWD_FUSED(MODE_, 2, false);    \
    else ITCV_BWD_FUSED(MODE_, 3, false);                 \
  } while (0)
  if (pool) ITCV_BWD_FUSED_NS(1);
  else if (up2) ITCV_BWD_FUSED_NS(2);
  else ITCV_BWD_FUSED_NS(0);
#undef ITCV_BWD_FUSED_NS
#undef ITCV_BWD_FUSED
  ITCV_CHECK_LAUNCH("itcv_bn_train_bwd");
  return 0;
}

#define ITCV_POINTWISE(NAME, KERNEL, N, ...)                                                        \
  do {                                                                                              \
    if ((N) == 0) return 0;                                                                         \
    hipLaunchKernelGGL(KERNEL, dim3(grid_for(N)), dim3(256), 0, S(stream), __VA_ARGS__);            \
    ITCV_CHECK_LAUNCH(NAME);                                                                        \
    return 0;                                                                                       \
  } while (0)

int itcv_lrelu_fwd(const float* x, float* y, size_t n, float slope, void* stream) {
  ITCV_REQUIRE(x && y, "itcv_lrelu_fwd");
  ITCV_POINTWISE("itcv_lrelu_fwd", lrelu_fwd_kernel, n, x, y, n, slope);
}
int itcv_lrelu_bwd(const float* x, const float* dy, float* dx, size_t n, float slope, void* stream) {
  ITCV_REQUIRE(x && dy && dx, "itcv_lrelu_bwd");
  ITCV_POINTWISE("itcv_lrelu_bwd", lrelu_bwd_kernel, n, x, dy, dx, n, slope);
}
int itcv_sigmoid_fwd(const float* x, float* y, size_t n, void* stream) {
  ITCV_REQUIRE(x && y, "itcv_sigmoid_fwd");
  ITCV_POINTWISE("itcv_sigmoid_fwd", sigmoid_fwd_kernel, n, x, y, n);
}
int itcv_sigmoid_bwd(const float* y, const float* dy, float* dx, size_t n, void* stream) {
  ITCV_REQUIRE(y && dy && dx, "itcv_sigmoid_bwd");
  ITCV_POINTWISE("itcv_sigmoid_bwd", sigmoid_bwd_kernel, n, y, dy, dx, n);
}
int itcv_add(const float* a, const float* b, float* out, size_t n, void* stream) {
  ITCV_REQUIRE(a && b && out, "itcv_add");
  ITCV_POINTWISE("itcv_add", add_kernel, n, a, b, out, n);
}
int itcv_avgpool2_fwd(const float* x, float* y, int BC, int H, int W, void* stream) {
  ITCV_REQUIRE(x && y && BC > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "itcv_avgpool2_fwd");
  const size_t n = (size_t)BC * (H / 2) * (W / 2);
  ITCV_POINTWISE("itcv_avgpool2_fwd", avgpool2_fwd_kernel, n, x, y, H, W, n);
}
int itcv_avgpool2_bwd(const float* dy, float* dx, int BC, int H, int W, void* stream) {
  ITCV_REQUIRE(dy && dx && BC > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "itcv_avgpool2_bwd");
  const size_t n = (size_t)BC * H * W;
  ITCV_POINTWISE("itcv_avgpool2_bwd", avgpool2_bwd_kernel, n, dy, dx, H, W, n);
}
int itcv_upsample2_fwd(const float* x, float* y, int BC, int H, int W, void* stream) {
  ITCV_REQUIRE(x && y && BC > 0 && H > 0 && W > 0, "itcv_upsample2_fwd");
  const size_t n = (size_t)BC * H * W * 4;
  ITCV_POINTWISE("itcv_upsample2_fwd", upsample2_fwd_kernel, n, x, y, H, W, n);
}
int itcv_upsample2_bwd(const float* dy, float* dx, int BC, int H, int W, void* stream) {
  ITCV_REQUIRE(dy && dx && BC > 0 && H > 0 && W > 0, "itcv_upsample2_bwd");
  const size_t n = (size_t)BC * H * W;
  ITCV_POINTWISE("itcv_upsample2_bwd", upsample2_bwd_kernel, n, dy, dx, H, W, n);
}

}  // extern "C"
