// Direct convolution for layers with at most 4 OUTPUT channels (the 5x5 "predict" conv 64->3 of the
// decoder, models.py:290, and the data-gradient of the 5x5 stem 3<-64, models.py:213).
//
// With M = 3 a 32-row MFMA tile is 91 % padding (the implicit-GEMM kernel ran these layers at
// ~10 TFLOP/s); they are done on the vector ALUs instead: one workgroup owns a 16x16 output tile of
// one image, stages the (16+KS-1)^2 input patch of 8 channels at a time in LDS (zero-filled halo),
// and every thread accumulates its pixel's CO outputs with the weights as wave-uniform scalar
// operands (s_load) -- 3 FMAs per LDS read.  fp32 throughout (exact, like the fp32 MFMA path).
#include "common.h"

#include <algorithm>

namespace itcv {

constexpr int kTile = 16, kChunk = 8;

// weight of (output channel m, reduction channel c, tap): forward reads w[m][c][tap] of the OIHW
// tensor, the data-gradient reads the transposed, flipped filter w[c][m][KK-1-tap].
template <int KS, int CO, bool DGRAD>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int C, int H, int W, int Cw, int tiles_x, int tiles_y) {
  constexpr int KK = KS * KS, P = KS / 2, PW = kTile + KS - 1, PSZ = PW * PW;
  __shared__ float patch[kChunk * PSZ];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  int bid = blockIdx.x;
  const int tile_x = bid % tiles_x;
  bid /= tiles_x;
  const int tile_y = bid % tiles_y, b = bid / tiles_y;
  const int h0 = tile_y * kTile, w0 = tile_x * kTile;
  const float* xb = x + (size_t)b * C * H * W;

  float acc[CO];
#pragma unroll
  for (int m = 0; m < CO; ++m) acc[m] = 0.f;

  for (int c0 = 0; c0 < C; c0 += kChunk) {
    __syncthreads();
    for (int i = t; i < kChunk * PSZ; i += 256) {
      const int c = i / PSZ, r = i - c * PSZ, ph = r / PW, pw = r - ph * PW;
      const int hh = h0 + ph - P, ww = w0 + pw - P;
      float v = 0.f;
      if (c0 + c < C && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
        v = xb[((size_t)(c0 + c) * H + hh) * W + ww];
      patch[i] = v;
    }
    __syncthreads();
    const int nc = min(kChunk, C - c0);
    for (int c = 0; c < nc; ++c) {
      const float* pc = patch + c * PSZ + ty * PW + tx;
      // wave-uniform weight row(s) for this reduction channel
      const float* wc[CO];
#pragma unroll
      for (int m = 0; m < CO; ++m)
        wc[m] = DGRAD ? w + ((size_t)(c0 + c) * Cw + m) * KK : w + ((size_t)m * Cw + (c0 + c)) * KK;
#pragma unroll
      for (int kh = 0; kh < KS; ++kh)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const float xv = pc[kh * PW + kw];
          const int tap = DGRAD ? KK - 1 - (kh * KS + kw) : kh * KS + kw;
#pragma unroll
          for (int m = 0; m < CO; ++m) acc[m] = fmaf(wc[m][tap], xv, acc[m]);
        }
    }
  }
  const int h = h0 + ty, ww = w0 + tx;
  if (h < H && ww < W) {
#pragma unroll
    for (int m = 0; m < CO; ++m)
      y[(((size_t)b * CO + m) * H + h) * W + ww] = acc[m] + (bias ? bias[m] : 0.f);
  }
}

// ---- at most 4 REDUCTION channels (the 5x5 stem 3->64 forward, and the data-gradient of the predict
// conv 3->64): every thread keeps its pixel's CI*KS*KS input window in registers and walks the output
// channels with wave-uniform scalar weights -- one FMA per weight, no LDS traffic in the inner loop.
template <int KS, int CI, bool DGRAD>
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int M, int H, int W, int tiles_x, int tiles_y) {
  constexpr int KK = KS * KS, P = KS / 2, PW = kTile + KS - 1, PSZ = PW * PW;
  __shared__ float patch[CI * PSZ];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  int bid = blockIdx.x;
  const int tile_x = bid % tiles_x;
  bid /= tiles_x;
  const int tile_y = bid % tiles_y, b = bid / tiles_y;
  const int h0 = tile_y * kTile, w0 = tile_x * kTile;
  const float* xb = x + (size_t)b * CI * H * W;
  for (int i = t; i < CI * PSZ; i += 256) {
    const int c = i / PSZ, r = i - c * PSZ, ph = r / PW, pw = r - ph * PW;
    const int hh = h0 + ph - P, ww = w0 + pw - P;
    patch[i] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? xb[((size_t)c * H + hh) * W + ww] : 0.f;
  }
  __syncthreads();
  float xv[CI * KK];
#pragma unroll
  for (int c = 0; c < CI; ++c)
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) xv[c * KK + kh * KS + kw] = patch[c * PSZ + (ty + kh) * PW + tx + kw];
  const int h = h0 + ty, ww = w0 + tx;
  const bool inside = h < H && ww < W;
  float* yb = y + ((size_t)b * M * H + h) * W + ww;
  for (int m = 0; m < M; ++m) {
    float acc = bias ? bias[m] : 0.f;
#pragma unroll
    for (int c = 0; c < CI; ++c)
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) {
        // forward: w[m][c][tap] of [M][CI][KK]; data-gradient: w[c][m][KK-1-tap] of [CI][M][KK]
        const float wv = DGRAD ? w[((size_t)c * M + m) * KK + (KK - 1 - tap)] : w[((size_t)m * CI + c) * KK + tap];
        acc = fmaf(wv, xv[c * KK + tap], acc);
      }
    if (inside) yb[(size_t)m * H * W] = acc;
  }
}

// ---- <= 3 output channels from pre-split planes, on the bf16 matrix cores (bf16x3) -----------------------------
// y[co][h][w] = sum_{ci,dh,dw} W[co][ci][dh][dw] x[ci][h+dh][w+dw] with 64 input channels given as planes.  The
// MFMA rows are (co, dw) -- 15 of 16 used -- and its reduction index (dh, ci): Z[(co,dw)][h][w'] = sum_{dh,ci} W x[ci][h+dh][w']
// needs no shift inside the GEMM, every B fragment is ONE 16-byte chunk of the planes (8 channels of a pixel) loaded
// straight from global memory, and the weights (16 x 320 x 2 planes) live in registers.  A wave owns a strip of 16
// input columns and walks down the rows: each input row (4 chunk loads per lane) feeds the five output rows it
// touches (rolling accumulators), y[co][h][w] = sum_dw Z[(co,dw)][h][w+dw-2] is folded per finished row through a
// 1-KB wave-private LDS tile; strips overlap by 4 columns (12 outputs per 16-column strip).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// F16: fp16 planes (scale record behind them), the register-resident weights split as fp16 planes of 2^8 w.
template <bool DGRAD, bool F16 = false>
__global__ __launch_bounds__(256) void conv_small_cout_planes_kernel(const u32x4* __restrict__ xp,
                                                                    const float* __restrict__ w,
                                                                    const float* __restrict__ bias, float* __restrict__ y,
                                                                    int B, int H, int W, int CO, int strips,
                                                                    int row_blocks, int RB, size_t plane_stride,
                                                                    int njobs) {
  constexpr int KS = 5, KK = 25, C = 64, C8 = 8;
  __shared__ float zs[4][16 * 17];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);   // wave-uniform: addresses go to SGPRs
  if (job >= njobs) return;
  const int s = job % strips, rbk = (job / strips) % row_blocks, b = job / (strips * row_blocks);
  const int n = lane & 15, kg = lane >> 4;
  const int h0 = rbk * RB, c0 = s * 12 - 2, wc = c0 + n;
  const bool col_ok = (unsigned)wc < (unsigned)W;
  const size_t HW = (size_t)H * W;

  // A fragments: row m = (co, dw), k = 8 input channels (half*32 + kg*8 + j) of filter row dh; two bf16 planes
  bf16x8 af[5][2][2];
  {
    const int m = n, co = m / 5, dw = m - co * 5;
#pragma unroll
    for (int dh = 0; dh < 5; ++dh)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // unconditional load from a clamped row, then a select: under a branch the compiler keeps the fp16 scale
          // multiply with the load and the 80 loads of a wave become 80 dependent round trips
          const int c = hf * 32 + kg * 8 + j, tap = dh * KS + dw, coc = co < CO ? co : 0;
          const float t = DGRAD ? w[((size_t)c * CO + coc) * KK + (KK - 1 - tap)] : w[((size_t)coc * C + c) * KK + tap];
          v[j] = co < CO ? t : 0.f;
        }
        u32x4 pl[2];
        split8<2, F16>(v, pl, F16 ? (float)(1 << kWeightScaleLog2) : 1.f);
        af[dh][hf][0] = __builtin_bit_cast(bf16x8, pl[0]);
        af[dh][hf][1] = __builtin_bit_cast(bf16x8, pl[1]);
      }
  }
  const float oscale = F16 ? inv_scale_of(reinterpret_cast<const ScaleRec*>(xp + 2 * plane_stride)) * (1.f / (float)(1 << kWeightScaleLog2)) : 1.f;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  // B fragments of one input row: [half][plane], chunk (b, c8 = half*4 + kg, row, wc)
  // Every load is unconditional, from a clamped (always valid) pixel, and masked when it is USED: a load under a bounds
  // check is a branch, and with branches between them the compiler cannot count the loads still in flight (it then
  // waits for all of them once per trip of the loop).
  const int wcc = min(max(wc, 0), W - 1);
  auto load_row = [&](int hr, u32x4 (&dst)[2][2]) -> bool {
    const int hc = min(max(hr, 0), H - 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        dst[hf][p] = xp[(size_t)p * plane_stride + ((size_t)b * C8 + hf * 4 + kg) * HW + (size_t)hc * W + wcc];
    return col_ok && (unsigned)hr < (unsigned)H;
  };
  f32x4_t acc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nrows = min(RB, H - h0);       // output rows of this job
  // A ring of five input rows, one slot per unrolled iteration (so the slots are registers with static names and the
  // compiler's counted vmcnt waits stay exact): four rows are in flight behind the one being multiplied -- a row's 30
  // MFMAs (~0.25 us) cover an eighth of a global-memory round trip, and with the three rotating buffers of the first
  // form the copies between them made every third row wait for everything outstanding.
  u32x4 ring[5][2][2];
  bool rok[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) rok[q] = load_row(h0 - 2 + q, ring[q]);
  float* zt = zs[wv];
  const float bias_l = (bias && lane < 36 && lane / 12 < CO) ? bias[lane / 12] : 0.f;   // the output channel of this lane's fold
  for (int i0 = 0; i0 < nrows + 4; i0 += 5) {
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int i = i0 + r;                // input row h0 - 2 + i sits in ring[r]
      if (i >= nrows + 4) break;
      // output row o = i - kh (kh = 0..4 = dh + 2) takes filter row kh; its accumulator slot is o mod 5
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) {
        const int slot = ((r - kh) % 5 + 5) % 5;
        f32x4_t c = acc[slot];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const bf16x8 b0 = __builtin_bit_cast(bf16x8, rok[r] ? ring[r][hf][0] : zero),
                       b1 = __builtin_bit_cast(bf16x8, rok[r] ? ring[r][hf][1] : zero);
          c = mma16x16x32<F16>(af[kh][hf][0], b1, c);
          c = mma16x16x32<F16>(af[kh][hf][1], b0, c);
          c = mma16x16x32<F16>(af[kh][hf][0], b0, c);
        }
        acc[slot] = c;
      }
      rok[r] = load_row(h0 - 2 + i + 5, ring[r]);   // refill the slot with the row five further down
      // output row o = i - 4 is complete (slot (r - 4) mod 5 = (r + 1) mod 5)
      const int o = i - 4, slot_done = (r + 1) % 5;
      if (o >= 0) {
        const f32x4_t z = acc[slot_done];
#pragma unroll
        for (int q = 0; q < 4; ++q) zt[(4 * kg + q) * 17 + n] = z[q];   // Z[m = 4*kg + q][column n]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 36) {
          const int co = lane / 12, nn = lane - co * 12, wo = s * 12 + nn;
          if (co < CO && wo < W) {
            float v = F16 ? 0.f : bias_l;
#pragma unroll
            for (int dw = 0; dw < 5; ++dw) v += zt[(co * 5 + dw) * 17 + nn + dw];
            if (F16) v = v * oscale + bias_l;
            y[(((size_t)b * CO + co) * H + (h0 + o)) * W + wo] = v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      acc[slot_done] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// ---- <= 3 reduction channels, 64 outputs, 5x5, on the bf16 matrix cores (bf16x3) ----------------------------------
// The stem layer (image -> 64 channels) and the data-gradient of the 64 -> 3 prediction layer.  MFMA rows are the 64
// output channels, columns 32 pixels of an image row, and the reduction index of one 32x32x16 product is (dw, ci) --
// 5*CI <= 15 of 16 slots -- for ONE filter row dh; the five filter rows accumulate into the same tile.  A lane's B
// fragment of an input row is 8 fp32 pixels read straight from the image (column n + dw - 2 of channel ci), split into
// two bf16 planes once and kept for the five output rows that touch it (a ring of six fragments while the wave walks
// down its strip), the weights (5 x 2 x 2 fragments) stay in registers: per output row a wave issues 8 loads,
// 30 MFMAs and 32 128-byte store segments.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// F16: both operands split in registers as fp16 planes: the weights of 2^8 w, the input of Sx x with Sx from `x_amax`
// (block maxima of |x|, itcv_absmax; null: Sx = 1 -- the stem's input is an image in [0, 1]).
template <int CI, bool DGRAD, int AHEAD, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_small_cin_mfma_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 int H, int W, int strips, int row_blocks, int RB,
                                                                 int njobs, const float* __restrict__ x_amax) {
  constexpr int KS = 5, KK = 25, M = 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);   // wave-uniform: addresses go to SGPRs
  if (job >= njobs) return;
  const int s = job % strips, rbk = (job / strips) % row_blocks, b = job / (strips * row_blocks);
  const int n = lane & 31, kg = lane >> 5;
  const int h0 = rbk * RB, w0 = s * 32;
  const int nrows = min(RB, H - h0);
  const size_t HW = (size_t)H * W;
  float xscale = 1.f, oscale = 1.f;
  if constexpr (F16) {
    if (x_amax) xscale = scale_for_bound(wave_absmax_of(x_amax, kAbsmaxParts));
    oscale = (1.f / (float)(1 << kWeightScaleLog2)) / xscale;      // exact: powers of two
  }

  // this lane's 8 reduction slots k = 8*kg + j -> (dw, ci); slots >= 5*CI are padding
  int k_dw[8], k_ci[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kg * 8 + j;
    k_dw[j] = k / CI;
    k_ci[j] = k - k_dw[j] * CI;
  }
  // A fragments: row m = output channel (mt*32 + n), two bf16 planes, per filter row dh
  bf16x8 af[5][2][2];
#pragma unroll
  for (int dh = 0; dh < 5; ++dh)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = mt * 32 + n;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool used = k_dw[j] < KS;              // padding slots: load tap 0 of channel 0 and drop it (no branch)
        const int tap = used ? dh * KS + k_dw[j] : 0, ci = used ? k_ci[j] : 0;
        const float t = DGRAD ? w[((size_t)ci * M + m) * KK + (KK - 1 - tap)] : w[((size_t)m * CI + ci) * KK + tap];
        v[j] = used ? t : 0.f;
      }
      u32x4 pl[2];
      split8<2, F16>(v, pl, F16 ? (float)(1 << kWeightScaleLog2) : 1.f);
      af[dh][mt][0] = __builtin_bit_cast(bf16x8, pl[0]);
      af[dh][mt][1] = __builtin_bit_cast(bf16x8, pl[1]);
    }
  const float* xb = x + (size_t)b * CI * HW;
  // element offset of slot j inside the image at row 0 (row-independent), -1 = padding slot or column outside
  int offj[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = w0 + n + k_dw[j] - 2;
    offj[j] = (k_dw[j] < KS && (unsigned)col < (unsigned)W) ? k_ci[j] * (int)HW + col : -1;
  }
  auto load_row = [&](int hr, u32x4 (&dst)[2]) {
    float v[8];
    const bool rok = (unsigned)hr < (unsigned)H;
    const float* xr = xb + hr * W;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = (rok && offj[j] >= 0) ? xr[offj[j]] : 0.f;
      v[j] = t;
    }
    split8<2, F16>(v, dst, xscale);
  };
  // ring of R = 5 + AHEAD input-row fragments: slot q holds input row h0 - 2 + i with i % R == q; AHEAD = 1 requests the
  // row that completes output row o + 1 while row o is multiplied
  constexpr int R = 5 + AHEAD;
  u32x4 ring[R][2];
#pragma unroll
  for (int q = 0; q < R - 1; ++q) load_row(h0 - 2 + q, ring[q]);
  float* yb = y + (size_t)b * M * HW;            // uniform
  const int lane_off = 4 * kg * (int)HW + w0 + n;
  // this lane's 32 bias values, loaded once (left in the store loop the compiler reloads one per stored element)
  float bv[2][16];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[mt][r] = bias ? bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] : 0.f;
  for (int o0 = 0; o0 < nrows; o0 += R) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int o = o0 + u;                      // output row h0 + o needs input rows i = o .. o + 4
      if (o >= nrows) break;
      load_row(h0 - 2 + o + R - 1, ring[(u + R - 1) % R]);
      f32x16_t acc[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
      for (int dh = 0; dh < 5; ++dh) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, ring[(u + dh) % R][0]),
                     b1 = __builtin_bit_cast(bf16x8, ring[(u + dh) % R][1]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[mt] = mma32x32x16<F16>(af[dh][mt][0], b1, acc[mt]);
          acc[mt] = mma32x32x16<F16>(af[dh][mt][1], b0, acc[mt]);
          acc[mt] = mma32x32x16<F16>(af[dh][mt][0], b0, acc[mt]);
        }
      }
      float* yr = yb + (size_t)(h0 + o) * W;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mu = mt * 32 + (r & 3) + 8 * (r >> 2);   // + 4*kg inside lane_off
          yr[(size_t)mu * HW + lane_off] = (F16 ? acc[mt][r] * oscale : acc[mt][r]) + bv[mt][r];
        }
    }
  }
}

}  // namespace itcv

using namespace itcv;

extern "C" {

int itcv_conv2d_small_cout_supported(int Co, int KS) { return Co >= 1 && Co <= 4 && (KS == 3 || KS == 5); }

// y[B][Co][H][W] = conv(x[B][C][H][W], w) (+bias), Co <= 4.  for_dgrad = 0: w is [Co][C][KS][KS];
// for_dgrad = 1: w is the forward layer's [C][Co][KS][KS] and the transposed, flipped filter is applied
// (x is then the output gradient, y the input gradient).
int itcv_conv2d_small_cout_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int H,
                               int W, int Co, int KS, int for_dgrad, void* stream) {
  ITCV_REQUIRE(x && w && y && B > 0 && C > 0 && H > 0 && W > 0, "itcv_conv2d_small_cout_fwd");
  if (!itcv_conv2d_small_cout_supported(Co, KS))
    return fail("%s: needs 1 <= Co <= 4 and KS in {3,5}", "itcv_conv2d_small_cout_fwd");
  const int tx = cdiv(W, kTile), ty = cdiv(H, kTile);
  dim3 grid(B * tx * ty), block(256);
  hipStream_t st = S(stream);
  const int Cw = for_dgrad ? Co : C;   // inner dimension of the weight tensor as indexed by the kernel
#define ITCV_SMALL(KS_, CO_)                                                                                       \
  do {                                                                                                             \
    if (for_dgrad)                                                                                                 \
      launch_timed((conv_small_cout_kernel<KS_, CO_, true>), grid, block, 0, st, x, w, bias, y, C, H, W, Cw, \
                         tx, ty);                                                                                  \
    else                                                                                                           \
      launch_timed((conv_small_cout_kernel<KS_, CO_, false>), grid, block, 0, st, x, w, bias, y, C, H, W,    \
                         Cw, tx, ty);                                                                              \
  } while (0)
#define ITCV_SMALL_KS(KS_)                    \
  do {                                        \
    if (Co == 1) ITCV_SMALL(KS_, 1);          \
    else if (Co == 2) ITCV_SMALL(KS_, 2);     \
    else if (Co == 3) ITCV_SMALL(KS_, 3);     \
    else ITCV_SMALL(KS_, 4);                  \
  } while (0)
  {
    ProfScope prof(st, 4, KS, Co, 0, 0, 2.0 * B * H * W * (double)Co * C * KS * KS);
    if (KS == 3)
      ITCV_SMALL_KS(3);
    else
      ITCV_SMALL_KS(5);
  }
#undef ITCV_SMALL_KS
#undef ITCV_SMALL
  ITCV_CHECK_LAUNCH("itcv_conv2d_small_cout_fwd");
  return 0;
}

// bf16x3 form of the <= 3-output 5x5 conv on pre-split planes of a 64-channel input ([2][B][8][H][W] chunks); same
// result tensor as itcv_conv2d_small_cout_fwd up to the 2^-16-per-product rounding of the split.
int itcv_conv2d_small_cout_bf16p_supported(int C, int Co, int KS) { return C == 64 && Co >= 1 && Co <= 3 && KS == 5; }

int itcv_conv2d_small_cout_fwd_bf16p(const void* xplanes, const float* w, const float* bias, float* y, int B, int C,
                                     int H, int W, int Co, int KS, int for_dgrad, int ns, void* stream) {
  ITCV_REQUIRE(xplanes && w && y && B > 0 && H > 0 && W > 0 && (ns == 2 || ns == ITCV_PLANES_F16X2),
               "itcv_conv2d_small_cout_fwd_bf16p");
  const bool f16 = ns == ITCV_PLANES_F16X2;
  if (!itcv_conv2d_small_cout_bf16p_supported(C, Co, KS))
    return fail("%s: needs C == 64, Co <= 3, KS == 5", "itcv_conv2d_small_cout_fwd_bf16p");
  const int strips = cdiv(W, 12), RB = H >= 16 ? 16 : H, row_blocks = cdiv(H, RB);
  const int njobs = B * row_blocks * strips;
  hipStream_t st = S(stream);
  const size_t plane_stride = (size_t)B * 8 * H * W;
  const u32x4* xp = static_cast<const u32x4*>(xplanes);
  ProfScope prof(st, 10, KS, Co, 0, ns, 2.0 * B * H * W * (double)Co * C * KS * KS);
  const dim3 grid(cdiv(njobs, 4)), blk(256);
#define ITCV_SCOUT_P(DG_, F_)                                                                                      \
  launch_timed((conv_small_cout_planes_kernel<DG_, F_>), grid, blk, 0, st, xp, w, bias, y, B, H, W, Co, strips, \
               row_blocks, RB, plane_stride, njobs)
  if (for_dgrad) {
    if (f16) ITCV_SCOUT_P(true, true);
    else ITCV_SCOUT_P(true, false);
  } else {
    if (f16) ITCV_SCOUT_P(false, true);
    else ITCV_SCOUT_P(false, false);
  }
#undef ITCV_SCOUT_P
  ITCV_CHECK_LAUNCH("itcv_conv2d_small_cout_fwd_bf16p");
  return 0;
}

int itcv_conv2d_small_cin_supported(int C, int KS) { return C >= 1 && C <= 4 && (KS == 3 || KS == 5); }

// y[B][Co][H][W] = conv(x[B][C][H][W], w) (+bias), C <= 4 reduction channels.  for_dgrad = 0: w is
// [Co][C][KS][KS]; for_dgrad = 1: w is the forward layer's [C][Co][KS][KS] (x = dy of that layer).
int itcv_conv2d_small_cin_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int H,
                              int W, int Co, int KS, int for_dgrad, void* stream) {
  ITCV_REQUIRE(x && w && y && B > 0 && Co > 0 && H > 0 && W > 0, "itcv_conv2d_small_cin_fwd");
  if (!itcv_conv2d_small_cin_supported(C, KS))
    return fail("%s: needs 1 <= C <= 4 and KS in {3,5}", "itcv_conv2d_small_cin_fwd");
  const int tx = cdiv(W, kTile), ty = cdiv(H, kTile);
  dim3 grid(B * tx * ty), block(256);
  hipStream_t st = S(stream);
#define ITCV_SCIN(KS_, CI_)                                                                                      \
  do {                                                                                                           \
    if (for_dgrad)                                                                                               \
      launch_timed((conv_small_cin_kernel<KS_, CI_, true>), grid, block, 0, st, x, w, bias, y, Co, H, W, tx, \
                         ty);                                                                                    \
    else                                                                                                         \
      launch_timed((conv_small_cin_kernel<KS_, CI_, false>), grid, block, 0, st, x, w, bias, y, Co, H, W,  \
                         tx, ty);                                                                                \
  } while (0)
#define ITCV_SCIN_KS(KS_)                 \
  do {                                    \
    if (C == 1) ITCV_SCIN(KS_, 1);        \
    else if (C == 2) ITCV_SCIN(KS_, 2);   \
    else if (C == 3) ITCV_SCIN(KS_, 3);   \
    else ITCV_SCIN(KS_, 4);               \
  } while (0)
  {
    ProfScope prof(st, 5, KS, C, 0, 0, 2.0 * B * H * W * (double)Co * C * KS * KS);
    if (KS == 3)
      ITCV_SCIN_KS(3);
    else
      ITCV_SCIN_KS(5);
  }
#undef ITCV_SCIN_KS
#undef ITCV_SCIN
  ITCV_CHECK_LAUNCH("itcv_conv2d_small_cin_fwd");
  return 0;
}

// bf16x3 form of the <= 3-channel -> 64-channel 5x5 conv (itcv_conv2d_small_cin_fwd's hot case) on the matrix cores;
// same tensors and for_dgrad meaning, results equal up to the 2^-16-per-product rounding of the split.
int itcv_conv2d_small_cin_bf16x3_supported(int C, int Co, int KS, int W) {
  return C >= 1 && C <= 3 && Co == 64 && KS == 5 && W >= 32 && W % 32 == 0;
}

int itcv_conv2d_small_cin_fwd_bf16x3(const float* x, const float* w, const float* bias, float* y, int B, int C, int H,
                                     int W, int Co, int KS, int for_dgrad, int ns, const float* x_amax, void* stream) {
  ITCV_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && (ns == 2 || ns == ITCV_PLANES_F16X2),
               "itcv_conv2d_small_cin_fwd_bf16x3");
  ITCV_REQUIRE(!x_amax || ns == ITCV_PLANES_F16X2, "itcv_conv2d_small_cin_fwd_bf16x3(a scale only applies to fp16 planes)");
  const bool f16 = ns == ITCV_PLANES_F16X2;
  if (!itcv_conv2d_small_cin_bf16x3_supported(C, Co, KS, W))
    return fail("%s: needs C <= 3, Co == 64, KS == 5, W %% 32 == 0", "itcv_conv2d_small_cin_fwd_bf16x3");
  const int strips = W / 32;
  // >= 2048 wave jobs where the image allows it (one wave per SIMD on 256 CUs, two rounds), >= 8 rows per job so the
  // 4-row halo stays a small part of the reads
  int row_blocks = cdiv(2048, B * strips);
  row_blocks = std::max(1, std::min(row_blocks, std::max(1, H / 8)));
  const int RB = cdiv(H, row_blocks);
  row_blocks = cdiv(H, RB);
  const int njobs = B * row_blocks * strips;
  hipStream_t st = S(stream);
  ProfScope prof(st, 11, KS, C, 0, ns, 2.0 * B * H * W * (double)Co * C * KS * KS);
  // (Requesting each input row one output row ahead of its use -- AHEAD = 1 -- was measured slower, 37 -> 43 us at
  // 128 x 3 x 64 x 64: the kernel is bound by its 134 MB of stores, and the longer ring costs registers.)
#define ITCV_SCIN_K(CI_, DG_, F_)                                                                                       \
  launch_timed((conv_small_cin_mfma_kernel<CI_, DG_, 0, F_>), dim3(cdiv(njobs, 4)), dim3(256), 0, st, x, w, bias, y, H, W, \
               strips, row_blocks, RB, njobs, x_amax)
#define ITCV_SCIN_M(CI_)                             \
  do {                                               \
    if (for_dgrad) {                                 \
      if (f16) ITCV_SCIN_K(CI_, true, true);         \
      else ITCV_SCIN_K(CI_, true, false);            \
    } else {                                         \
      if (f16) ITCV_SCIN_K(CI_, false, true);        \
      else ITCV_SCIN_K(CI_, false, false);           \
    }                                                \
  } while (0)
  if (C == 1) ITCV_SCIN_M(1);
  else if (C == 2) ITCV_SCIN_M(2);
  else ITCV_SCIN_M(3);
#undef ITCV_SCIN_K
#undef ITCV_SCIN_M
  ITCV_CHECK_LAUNCH("itcv_conv2d_small_cin_fwd_bf16x3");
  return 0;
}

}  // extern "C"
