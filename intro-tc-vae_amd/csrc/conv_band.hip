// Band form of the planes convolution (forward and data-gradient of the 3x3 layers with W <= 64): see the comment
// in front of the kernel.  Split out of conv_igemm.hip so that the two translation units compile in parallel.
#include "conv_shared.h"

namespace itcv {

static __device__ u32x4 g_zero_chunk = {0u, 0u, 0u, 0u};

// ---- second form of the planes kernel: tap reuse through LDS ---------------------------------------
// conv_fwd_bf16p_kernel fetches every input chunk nine times (once per tap) and its weight tile once per
// 128 pixels; measured, the L2 -> LDS ingest (~30 B/clk/CU) then takes as long as the MFMAs.  Here a block
// owns 256 consecutive pixels (a band of whole image rows) x BM output channels:
//   * per 32-channel group the band arrives ONCE, with its halo (rows -1..NR, columns -1..W, zero chunks
//     outside the image), double-buffered; the nine taps read it at fragment addresses shifted by
//     dh*(W+2)+dw chunks -- no per-tap traffic at all;
//   * the weight tile of each (group, tap) streams through a 3-slot ring and serves twice the pixels;
//   * 8 MFMA waves (two per SIMD, so one fills the other's barrier / LDS-latency bubbles) + 4 loader waves.
// Ingest per MFMA drops ~2.8x.  Same arithmetic and K order as the other split-bf16 kernels (bit-identical).

template <int LOG2W, int BM, bool UP2, bool M16 = false, bool F16 = false>
__global__ __launch_bounds__(768) void conv_fwd_bf16p2_kernel(ConvArgsP2 a) {
  static_assert(!F16 || M16, "the fp16 planes form uses the 16x16x32 products");
  constexpr int W = 1 << LOG2W, WP = W + 2, BN = 256, KC = 4, NS = 2;
  constexpr int WM = BM / 64, WN = 8 / WM, WTN = BN / WN, TM = 2, TN = WTN / 32;   // 128: 2x4 waves of 64x64; 64: 1x8 of 64x32
  constexpr int ASZ = NS * KC * BM;                // chunks per weight tile
  constexpr int PA = NS * KC * BM / 64 / 4;        // weight pieces per loader wave per K-tile
  // G K-tiles (taps) per barrier: the 64-row tiles do half the MFMA work per K-tile, so they take two taps per
  // stage (the block-wide barrier and the LDS latency behind it were ~40 % of their K-tile time)
  constexpr int G = BM == 64 ? 2 : 1, NSTG = (9 + G - 1) / G;
  extern __shared__ u32x4 smem[];                  // [3][G][ASZ] weight ring, then [2][NS*KC*PXB] bands

  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int tile_m = q % a.mt, tile_n = (q / a.mt) * 8 + xcd;
  if (tile_n >= a.nt) return;
  const int sk = blockIdx.y;
  const int c0 = sk * a.cpt_per_split, c1 = min(a.cpt, c0 + a.cpt_per_split);
  if (c0 >= c1) return;
  const int nk = (c1 - c0) * 9;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int H = a.H, HW = H << LOG2W;
  const int PXB = a.PXB, BSZ = NS * KC * PXB;
  const uint32_t smem_base = lds_addr(smem);
  const uint32_t band_base = smem_base + 3u * G * ASZ * 16u;

  if (wid >= 8) {
    // ------------------------------------------------------------------ loaders
    const int lw = wid - 8;
    const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? W / 2 : W, HWs = Hs * Ws;
    const int C8 = a.Ci >> 3;
    const u32x4* zero = &g_zero_chunk;
    // band pieces of this wave: rows pk = lw (plane 0, kc = lw) and lw + 4 (plane 1), halo pixels q*64 + lane
    const int R0 = n0 >> LOG2W, seg_px = (a.SR + 2) * WP;
    long long soff[8];
    uint32_t vmask = 0;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      const int hp = qq * 64 + lane;   // lanes past NP land in the row's padding (PXB = NPC*64) and read zeros
      soff[qq] = 0;
      if (qq < a.NPC && hp < a.NP) {
        const int seg = hp / seg_px, rem = hp - seg * seg_px, hr = rem / WP, hc = rem - hr * WP;
        const int grow = R0 + seg * a.SR, b = grow >> a.h_shift, h = (grow & (H - 1)) + hr - 1, wv = hc - 1;
        if (b < a.B && (unsigned)h < (unsigned)H && (unsigned)wv < (unsigned)W) {
          vmask |= 1u << qq;
          soff[qq] = (long long)b * C8 * HWs + (UP2 ? (h >> 1) * Ws + (wv >> 1) : h * W + wv);
        }
      }
    }
    // every wave instruction below is issued by all 64 lanes (never skipped): the counted vmcnt waits rely on it
    auto issue_band_piece = [&](int cib, int buf, long long so, bool valid, int qq) {
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) {
        const u32x4* src = a.xp + ((size_t)pl * a.plane_stride + (size_t)(cib * KC + lw) * HWs + so);
        lds_dma16(valid ? src : zero, band_base + (uint32_t)(buf * BSZ + (pl * KC + lw) * PXB + qq * 64) * 16u);
      }
    };
    auto issue_A = [&](int i, int slot) {   // weight tile of K-tile i of this slice
      const int kt = c0 * 9 + i;
      const u32x4* wt = a.wp + (size_t)kt * NS * KC * a.Mp + m0 + lane;
      const uint32_t sbase = smem_base + (uint32_t)(slot * ASZ) * 16u;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const int piece = j * 4 + lw, mlc = piece % (BM / 64), pk = piece / (BM / 64);
        lds_dma16(wt + (size_t)pk * a.Mp + mlc * 64, sbase + (uint32_t)(pk * BM + mlc * 64) * 16u);
      }
    };
    // weight tiles of one stage (taps st*G .. of group cib) -> ring slot `slot`; past the end of the slice the last
    // K-tile is fetched again (same instruction count: the counted waits stay valid)
    auto issue_stage_A = [&](int cib, int st, int slot) {
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (st * G + u < 9) issue_A(min((cib - c0) * 9 + st * G + u, nk - 1), slot * G + u);
    };
    auto stage_taps = [](int st) { return (st * G + G <= 9) ? G : 9 - st * G; };
#pragma unroll
    for (int qq = 0; qq < 8; ++qq)
      if (qq < a.NPC) issue_band_piece(c0, 0, soff[qq], (vmask >> qq) & 1u, qq);
    issue_stage_A(c0, 0, 0);
    issue_stage_A(c0, 1, 1);
    wait_vmcnt<PA * (NSTG > 1 ? ((1 * G + G <= 9) ? G : 9 - G) : 0)>();   // band and stage 0 landed; stage 1 may be in flight
    __builtin_amdgcn_s_barrier();
    int buf = 0, slot = 2;
    for (int cib = c0; cib < c1; ++cib) {
#pragma unroll
      for (int st = 0; st < NSTG; ++st) {
        const int st2 = (st + 2) % NSTG, cib2 = cib + (st + 2) / NSTG;
        issue_stage_A(cib2, st2, slot);                       // into the slot stage S-1 has just left
        if (++slot == 3) slot = 0;
        // next group's band, one piece (x 2 planes) per tap of this stage
        int nb = 0;
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const int tap = st * G + u;
          if (tap < 8 && tap < a.NPC && cib + 1 < c1) {
            issue_band_piece(cib + 1, buf ^ 1, soff[tap < 8 ? tap : 0], (vmask >> tap) & 1u, tap);
            ++nb;
          }
        }
        // stage S+1's weights (and everything older) have landed: only this iteration's loads may be in flight
        const int na = stage_taps(st2);
        if (na == G) {
          if (nb == 0) wait_vmcnt<PA * G>();
          else if (nb == 1) wait_vmcnt<PA * G + NS>();
          else wait_vmcnt<PA * G + 2 * NS>();
        } else {
          if (nb == 0) wait_vmcnt<PA*(9 % G ? 9 % G : G)>();
          else if (nb == 1) wait_vmcnt<PA*(9 % G ? 9 % G : G) + NS>();
          else wait_vmcnt<PA*(9 % G ? 9 % G : G) + 2 * NS>();
        }
        // last stage of the slice: the (re-fetched) weight tiles still in flight must have landed before the MFMA waves
        // reuse the whole LDS allocation as the epilogue's staging tile
        if (st == NSTG - 1 && cib + 1 == c1) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
      buf ^= 1;
    }
    return;
  }

  // -------------------------------------------------------------------- MFMA waves
  typedef BandMfma<M16, F16> MM;
  typedef typename MM::acc_t acc_t;
  constexpr int TS = MM::TS, KSN = MM::KSN, TMx = 32 * TM / TS, TNx = WTN / TS;   // MFMA tiles of a wave's 64 x WTN block
  const float oscale = F16 ? inv_scale_of(a.xscale) * (1.f / (float)(1 << kWeightScaleLog2)) : 1.f;   // exact: powers of two
  const int wm = wid / WN, wn = wid % WN, lr = lane & (TS - 1), kq = lane / TS;
  // band index of this lane's pixel of N-tile j (centre tap)
  uint32_t hoff[TNx];
#pragma unroll
  for (int j = 0; j < TNx; ++j) {
    const int nl = wn * WTN + j * TS + lr, R = nl >> LOG2W, w = nl & (W - 1);
    const int seg = R / a.SR, rr = R - seg * a.SR;
    hoff[j] = (uint32_t)((seg * (a.SR + 2) + rr + 1) * WP + w + 1) * 16u;
  }
  const uint32_t aoff = (uint32_t)(wm * 64 + lr) * 16u;
  acc_t acc[TMx][TNx];
#pragma unroll
  for (int i = 0; i < TMx; ++i)
#pragma unroll
    for (int j = 0; j < TNx; ++j)
#pragma unroll
      for (int r = 0; r < MM::NR; ++r) acc[i][j][r] = 0.f;
  const long long dbg_c00 = ITCV_DBG(a) ? clock64() : 0;
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  const long long dbg_c0 = ITCV_DBG(a) ? clock64() : 0, dbg_w0 = ITCV_DBG(a) ? wall_clock64() : 0;
  int slot = 0;
  for (int cib = c0; cib < c1; ++cib) {
    const uint32_t bb = band_base + (uint32_t)(buf * BSZ) * 16u;
#pragma unroll
    for (int st = 0; st < NSTG; ++st) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int tap = st * G + u;
        if (tap >= 9) continue;
        const int tapoff = ((tap / 3 - 1) * WP + (tap % 3 - 1)) * 16;
        const uint32_t ab = smem_base + (uint32_t)((slot * G + u) * ASZ) * 16u + aoff;
        bf16x8 af[KSN][NS][TMx], bfr[KSN][NS][TNx];
        if constexpr (!M16) {
          // all fragments of the K-tile (both 16-wide k-steps) are requested up front and the scheduler is told to
          // interleave: the first k-step's reads, then one read of the second k-step behind each of the first MFMAs --
          // left alone it parks most reads directly in front of their use (`s_waitcnt lgkmcnt(0)` before the MFMA)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int kc = ks * 2 + kq;
#pragma unroll
            for (int pp = 0; pp < NS; ++pp) {
#pragma unroll
              for (int i = 0; i < TMx; ++i)
                af[ks][pp][i] = __builtin_bit_cast(
                    bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(ab + (uint32_t)(((pp * KC + kc) * BM + i * 32) * 16)));
#pragma unroll
              for (int j = 0; j < TNx; ++j)
                bfr[ks][pp][j] = __builtin_bit_cast(
                    bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(bb + (uint32_t)((pp * KC + kc) * PXB) * 16u + hoff[j] + tapoff));
            }
          }
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TMx; ++i)
#pragma unroll
              for (int j = 0; j < TNx; ++j) {
                acc_t c = acc[i][j];
                c = MM::mma(af[ks][0][i], bfr[ks][1][j], c);
                c = MM::mma(af[ks][1][i], bfr[ks][0][j], c);
                c = MM::mma(af[ks][0][i], bfr[ks][0][j], c);
                acc[i][j] = c;
              }
          {
            constexpr int RD = NS * (TMx + TNx), MF = TMx * TNx * 3;   // LDS reads / MFMAs per k-step
            __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
            for (int r = 0; r < RD; ++r) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF - RD, 0);
          }
        } else {
          // 16x16x32: one MFMA per (tile, product) covers the 32-channel group (lane group kq holds chunk kq); the pixel
          // fragments and the first half of the weight rows are requested up front, the rest one behind each early MFMA
#pragma unroll
          for (int pp = 0; pp < NS; ++pp)
#pragma unroll
            for (int j = 0; j < TNx; ++j)
              bfr[0][pp][j] = __builtin_bit_cast(
                  bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(bb + (uint32_t)((pp * KC + kq) * PXB) * 16u + hoff[j] + tapoff));
#pragma unroll
          for (int i = 0; i < TMx; ++i)
#pragma unroll
            for (int pp = 0; pp < NS; ++pp)
              af[0][pp][i] = __builtin_bit_cast(
                  bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(ab + (uint32_t)(((pp * KC + kq) * BM + i * 16) * 16)));
#pragma unroll
          for (int i = 0; i < TMx; ++i)
#pragma unroll
            for (int j = 0; j < TNx; ++j) {
              acc_t c = acc[i][j];
              c = MM::mma(af[0][0][i], bfr[0][1][j], c);
              c = MM::mma(af[0][1][i], bfr[0][0][j], c);
              c = MM::mma(af[0][0][i], bfr[0][0][j], c);
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
      if (++slot == 3) slot = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    buf ^= 1;
  }
  if (ITCV_DBG(a) && t == 0 && (bid == 0 || bid == 100)) {
    float* d = a.y + (bid ? 4 : 0);
    d[0] = (float)(clock64() - dbg_c0), d[1] = (float)(wall_clock64() - dbg_w0), d[2] = (float)(dbg_c0 - dbg_c00), d[3] = (float)nk;
    return;
  }

  float* out = a.y + (size_t)sk * a.slab_stride;
  if (M16 || !a.stats) {     // (the 16x16x32 instantiation is never launched with statistics)
    // ---- epilogue, direct form: every lane stores its accumulator elements (one pixel of NR channels per MFMA tile:
    // 128-byte -- 64-byte for 16x16 tiles -- row pieces per lane group)
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
          const int m = m0 + wm * 64 + i * TS + MM::row(r, kq);
          if (m < a.Co) {
            float v = acc[i][j][r];
            if (F16) v *= oscale;
            if (a.bias) v += a.bias[m];
            out[base + (size_t)m * HW] = v;
          }
        }
      }
    }
    return;
  }
  if constexpr (!M16) {
  const int l31 = lr, half = kq;
  // ---- epilogue, staged form (when the consumer BatchNorm's statistics are wanted from this launch): the tile goes
  // through LDS once, every global store is a 16-byte-per-lane / 1-KB-per-wave piece of one channel row, and the row's
  // sum and sum of squares fall out of the same registers.  Measured (64 -> 64 @ 64x64, 128 images): 146 us against 135 us
  // for the direct form -- more than the 8.6-us statistics pass it replaces saves (that pass reads the tensor out of the
  // Infinity Cache right behind this kernel), which is why the solvers leave it off (ITCV_FUSE_BN_STATS=1 turns it on).
  // The rings and bands are dead: every wave has passed the slice's last barrier with all LDS-DMA landed.
  constexpr int EP = BN + 4;                                   // floats per staged channel row (1040 B: 16-byte aligned)
  float* stg = reinterpret_cast<float*>(smem);                 // [BM][EP]
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stg[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * EP + wn * WTN + j * 32 + l31] = acc[i][j][r];
  __builtin_amdgcn_s_barrier();                                // the eight MFMA waves (the loader waves have ended)
  const int nn = n0 + 4 * lane;                                // this lane's 4 consecutive pixels of every row
  const int b2 = nn / HW, hw2 = nn - b2 * HW;                  // HW % 4 == 0: the four stay inside one image
  const size_t base = (size_t)b2 * a.Co * HW + hw2;
  const bool live = nn < a.N;
#pragma unroll 4
  for (int k = 0; k < BM / 8; ++k) {
    const int ch = k * 8 + wid, m = m0 + ch;
    if (m >= a.Co) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(stg + ch * EP + 4 * lane);
    if (a.bias) {
      const float bv = a.bias[m];
      v[0] += bv, v[1] += bv, v[2] += bv, v[3] += bv;
    }
    if (live) *reinterpret_cast<f32x4*>(out + base + (size_t)m * HW) = v;
    // this tile's sum and sum of squares of channel m (fp32 over 256 values; the per-channel fold over tiles runs in
    // fp64, norm_act.hip).  Fixed order: reproducible.
    float s1 = live ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
    float s2 = live ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
    s1 = wave_sum(s1), s2 = wave_sum(s2);
    if (lane == 0) {
      a.stats[(size_t)m * a.stat_T + tile_n] = s1;
      a.stats[((size_t)a.Co + m) * a.stat_T + tile_n] = s2;
    }
  }
  }
}

// ---- persistent form of the band kernel -------------------------------------------------------------------
// conv_fwd_bf16p2_kernel runs one tile per block, one block per CU (its LDS fills the CU), so a CU's timeline is
// [band + weights arrive | MFMA | stores leave] and all 256 CUs go through those phases together: on the 64-channel
// layers (K = 576) the matrix cores sit idle about half of the kernel.  Here a block walks a LIST of tiles
// (tile = blockIdx.x, += gridDim.x) and treats (tile, channel group) as one continuous stream of groups: during the
// last group of a tile the loader waves already fetch the FIRST band of the next tile into the band buffer that has just
// become free (and the weight ring simply wraps to the next tile's first taps), and the MFMA waves' result stores of
// tile t drain while tile t+1 is multiplied.  Barrier structure, counted waits, K order and arithmetic are those of the
// one-tile kernel (bit-identical results).
// (Four MFMA waves of a 64x64 block each instead of eight of 64x32 -- a third fewer LDS fragment reads per MFMA -- were
// built and measured SLOWER, 64 -> 64 @ 64x64: 100.8 -> 110.4 us on one box: with one MFMA wave per SIMD nothing fills the
// barrier / LDS-latency bubbles.  Removed.)
// ROWS2 (128- and 256-wide images): the 128-pixel tile is 2 rows x 64 columns instead of (half of) one row -- the band is
// 4 x 66 chunks per plane row instead of 3 x 130, a third less ingest per tile.
template <int LOG2W, int BM, bool UP2, int BN, bool M16 = false, bool ROWS2 = false, bool F16 = false>
__global__ __launch_bounds__(768) void conv_fwd_bf16p3_kernel(ConvArgsP2 a) {
  static_assert(!F16 || M16, "the fp16 planes form uses the 16x16x32 products");
  constexpr int NMW = 8;
  // BN = 256 pixels per tile for W <= 64 (whole rows); BN = 128 for 128- and 256-wide images: one row, or one half of a
  // row whose band then takes its halo columns from the neighbouring half instead of the zero padding
  constexpr int W = 1 << LOG2W, WB = ROWS2 ? 64 : (W < BN ? W : BN), LOG2WB = ROWS2 ? 6 : (LOG2W < 7 ? LOG2W : 7), WP = WB + 2, KC = 4, NS = 2;
  static_assert(WB == (1 << LOG2WB) && (BN == 256 || BN == 128) && (!ROWS2 || (BN == 128 && LOG2W >= 7)), "band width");
  constexpr int WM = BN == 256 ? BM / 64 : 2, WN = NMW / WM, WTN = BN / WN, TM = BM / (32 * WM), TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTN == 32 * TN, "wave tiling");
  constexpr int ASZ = NS * KC * BM;
  constexpr int PA = NS * KC * BM / 64 / 4;
  constexpr int G = band_taps_per_stage(BM, BN, LOG2W), NSTG = (9 + G - 1) / G;
  extern __shared__ u32x4 smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int sk = blockIdx.y;
  const int c0 = sk * a.cpt_per_split, c1 = min(a.cpt, c0 + a.cpt_per_split);
  if (c0 >= c1) return;
  const int nk = (c1 - c0) * 9;
  const int H = a.H, HW = H << LOG2W;
  const int PXB = a.PXB, BSZ = NS * KC * PXB;
  const uint32_t smem_base = lds_addr(smem);
  const uint32_t band_base = smem_base + 3u * G * ASZ * 16u;
  // tile ids of this block: id = blockIdx.x + k * gridDim.x over the padded id space of the one-tile kernel
  // (id & 7 = XCD slot, so a block keeps its XCD); ids whose N tile lies past the end are skipped -- by every wave alike
  // N tiles are dealt to the XCD slots in CONTIGUOUS ranges (slot x owns tiles x*R .. x*R+R-1): the blocks of one XCD
  // then work on neighbouring row bands at any time and the halo rows two neighbouring tiles share are served by that
  // XCD's L2 instead of being fetched once per tile from the Infinity Cache / HBM (speed only: nothing depends on
  // where a block runs)
  const int xr = (a.nt + 7) >> 3;
  const int nids = xr * 8 * a.mt, stride = gridDim.x;
  auto tile_n_of = [&](int id) { return (id & 7) * xr + (id >> 3) / a.mt; };
  auto tile_m_of = [&](int id) { return (id >> 3) % a.mt; };
  auto next_tile = [&](int id) {
    id += stride;
    while (id < nids && tile_n_of(id) >= a.nt) id += stride;
    return id;
  };
  // first (batch x height) row and first column of N tile tn
  auto tile_origin = [&](int tn, int& R0, int& col0) {
    if constexpr (ROWS2) {
      constexpr int CBS = W / 64;
      const int rp = tn / CBS;
      R0 = 2 * rp, col0 = 64 * (tn - rp * CBS);
    } else {
      const int pix0 = tn * BN;
      R0 = pix0 >> LOG2W, col0 = pix0 & (W - 1);           // col0 != 0 only for W > BN
    }
  };
  int tcur = (int)blockIdx.x - stride;
  tcur = next_tile(tcur);
  if (tcur >= nids) return;

  if (wid >= NMW) {
    // ------------------------------------------------------------------ loaders
    const int lw = wid - NMW;
    const int Hs = UP2 ? H / 2 : H, Ws = UP2 ? W / 2 : W, HWs = Hs * Ws;
    const int C8 = a.Ci >> 3;
    const u32x4* zero = &g_zero_chunk;
    const int seg_px = (a.SR + 2) * WP;
    long long soff[8], soff_n[8];
    uint32_t vmask = 0, vmask_n = 0;
    auto band_offsets = [&](int id, long long (&so)[8], uint32_t& vm) {
      int R0, col0;
      tile_origin(tile_n_of(id), R0, col0);
      vm = 0;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const int hp = qq * 64 + lane;
        so[qq] = 0;
        if (qq < a.NPC && hp < a.NP) {
          const int seg = hp / seg_px, rem = hp - seg * seg_px, hr = rem / WP, hc = rem - hr * WP;
          const int grow = R0 + seg * a.SR, b = grow >> a.h_shift, h = (grow & (H - 1)) + hr - 1, wv = col0 + hc - 1;
          if (b < a.B && (unsigned)h < (unsigned)H && (unsigned)wv < (unsigned)W) {
            vm |= 1u << qq;
            so[qq] = (long long)b * C8 * HWs + (UP2 ? (h >> 1) * Ws + (wv >> 1) : h * W + wv);
          }
        }
      }
    };
    // every wave instruction below is issued by all 64 lanes (never skipped): the counted vmcnt waits rely on it
    auto issue_band_piece = [&](int cib, int buf, long long so, bool valid, int qq) {
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) {
        const u32x4* src = a.xp + ((size_t)pl * a.plane_stride + (size_t)(cib * KC + lw) * HWs + so);
        lds_dma16(valid ? src : zero, band_base + (uint32_t)(buf * BSZ + (pl * KC + lw) * PXB + qq * 64) * 16u);
      }
    };
    int m0 = tile_m_of(tcur) * BM, m0_n = m0;      // weight rows of the current / the next tile
    auto issue_A = [&](int i, int slot, int mrow) {   // weight tile of K-tile i of this slice
      const int kt = c0 * 9 + i;
      const u32x4* wt = a.wp + (size_t)kt * NS * KC * a.Mp + mrow + lane;
      const uint32_t sbase = smem_base + (uint32_t)(slot * ASZ) * 16u;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const int piece = j * 4 + lw, mlc = piece % (BM / 64), pk = piece / (BM / 64);
        lds_dma16(wt + (size_t)pk * a.Mp + mlc * 64, sbase + (uint32_t)(pk * BM + mlc * 64) * 16u);
      }
    };
    // weight tiles of one stage (taps st*G .. of group cib) -> ring slot `slot`.  Past the end of the tile's slice the
    // ring wraps to the first taps of the NEXT tile (`more`), or -- after the last tile -- fetches the last K-tile again
    // (same instruction count: the counted waits stay valid)
    auto issue_stage_A = [&](int cib, int st, int slot, bool more) {
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (st * G + u < 9) {
          int i = (cib - c0) * 9 + st * G + u, mrow = m0;
          if (i >= nk) {
            if (more) i -= nk, mrow = m0_n;
            else i = nk - 1;
          }
          issue_A(i, slot * G + u, mrow);
        }
    };
    auto stage_taps = [](int st) { return (st * G + G <= 9) ? G : 9 - st * G; };
    band_offsets(tcur, soff, vmask);
    int tnext = next_tile(tcur);
    bool more = tnext < nids;
    if (more) m0_n = tile_m_of(tnext) * BM;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq)
      if (qq < a.NPC) issue_band_piece(c0, 0, soff[qq], (vmask >> qq) & 1u, qq);
    issue_stage_A(c0, 0, 0, more);
    issue_stage_A(c0, 1, 1, more);
    wait_vmcnt<PA * (NSTG > 1 ? ((1 * G + G <= 9) ? G : 9 - G) : 0)>();   // band and stage 0 landed; stage 1 may be in flight
    __builtin_amdgcn_s_barrier();
    int buf = 0, slot = 2;
    while (true) {
      for (int cib = c0; cib < c1; ++cib) {
        const bool last_group = cib + 1 == c1;
        if (last_group && more) band_offsets(tnext, soff_n, vmask_n);    // the next tile's band goes out during this group
        const bool pre = !last_group || more;                           // is there a following group to prefetch?
#pragma unroll
        for (int st = 0; st < NSTG; ++st) {
          const int st2 = (st + 2) % NSTG, cib2 = cib + (st + 2) / NSTG;
          issue_stage_A(cib2, st2, slot, more);                 // into the slot stage S-1 has just left
          if (++slot == 3) slot = 0;
          // the following group's band, one piece (x 2 planes) per tap of this stage
          int nb = 0;
#pragma unroll
          for (int u = 0; u < G; ++u) {
            const int tap = st * G + u;
            if (tap < 8 && tap < a.NPC && pre) {
              if (last_group) issue_band_piece(c0, buf ^ 1, soff_n[tap < 8 ? tap : 0], (vmask_n >> tap) & 1u, tap);
              else issue_band_piece(cib + 1, buf ^ 1, soff[tap < 8 ? tap : 0], (vmask >> tap) & 1u, tap);
              ++nb;
            }
          }
          // stage S+1's weights (and everything older) have landed: only this iteration's loads may be in flight
          const int na = stage_taps(st2);
          if (na == G) {
            if (nb == 0) wait_vmcnt<PA * G>();
            else if (nb == 1) wait_vmcnt<PA * G + NS>();
            else wait_vmcnt<PA * G + 2 * NS>();
          } else {
            if (nb == 0) wait_vmcnt<PA*(9 % G ? 9 % G : G)>();
            else if (nb == 1) wait_vmcnt<PA*(9 % G ? 9 % G : G) + NS>();
            else wait_vmcnt<PA*(9 % G ? 9 % G : G) + 2 * NS>();
          }
          __builtin_amdgcn_s_barrier();
        }
        buf ^= 1;
      }
      if (!more) break;
      tcur = tnext;
      m0 = m0_n;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) soff[qq] = soff_n[qq];
      vmask = vmask_n;
      tnext = next_tile(tcur);
      more = tnext < nids;
      if (more) m0_n = tile_m_of(tnext) * BM;
    }
    return;
  }

  // -------------------------------------------------------------------- MFMA waves
  typedef BandMfma<M16, F16> MM;
  typedef typename MM::acc_t acc_t;
  constexpr int TS = MM::TS, KSN = MM::KSN, TMx = 32 * TM / TS, TNx = WTN / TS;   // MFMA tiles of a wave's 32*TM x WTN block
  const float oscale = F16 ? inv_scale_of(a.xscale) * (1.f / (float)(1 << kWeightScaleLog2)) : 1.f;   // exact: powers of two
  const int wm = wid / WN, wn = wid % WN, lr = lane & (TS - 1), kq = lane / TS;
  uint32_t hoff[TNx];
#pragma unroll
  for (int j = 0; j < TNx; ++j) {
    const int nl = wn * WTN + j * TS + lr, R = nl >> LOG2WB, w = nl & (WB - 1);
    const int seg = R / a.SR, rr = R - seg * a.SR;
    hoff[j] = (uint32_t)((seg * (a.SR + 2) + rr + 1) * WP + w + 1) * 16u;
  }
  const uint32_t aoff = (uint32_t)(wm * 32 * TM + lr) * 16u;
  __builtin_amdgcn_s_barrier();
  int buf = 0, slot = 0;
  while (tcur < nids) {
    acc_t acc[TMx][TNx];
#pragma unroll
    for (int i = 0; i < TMx; ++i)
#pragma unroll
      for (int j = 0; j < TNx; ++j)
#pragma unroll
        for (int r = 0; r < MM::NR; ++r) acc[i][j][r] = 0.f;
    for (int cib = c0; cib < c1; ++cib) {
      const uint32_t bb = band_base + (uint32_t)(buf * BSZ) * 16u;
#pragma unroll
      for (int st = 0; st < NSTG; ++st) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const int tap = st * G + u;
          if (tap >= 9) continue;
          const int tapoff = ((tap / 3 - 1) * WP + (tap % 3 - 1)) * 16;
          const uint32_t ab = smem_base + (uint32_t)((slot * G + u) * ASZ) * 16u + aoff;
          bf16x8 af[KSN][NS][TMx], bfr[KSN][NS][TNx];
          if constexpr (!M16) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const int kc = ks * 2 + kq;
#pragma unroll
              for (int pp = 0; pp < NS; ++pp) {
#pragma unroll
                for (int i = 0; i < TMx; ++i)
                  af[ks][pp][i] = __builtin_bit_cast(
                      bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(ab + (uint32_t)(((pp * KC + kc) * BM + i * 32) * 16)));
#pragma unroll
                for (int j = 0; j < TNx; ++j)
                  bfr[ks][pp][j] = __builtin_bit_cast(
                      bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(bb + (uint32_t)((pp * KC + kc) * PXB) * 16u + hoff[j] + tapoff));
              }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int i = 0; i < TMx; ++i)
#pragma unroll
                for (int j = 0; j < TNx; ++j) {
                  acc_t c = acc[i][j];
                  c = MM::mma(af[ks][0][i], bfr[ks][1][j], c);
                  c = MM::mma(af[ks][1][i], bfr[ks][0][j], c);
                  c = MM::mma(af[ks][0][i], bfr[ks][0][j], c);
                  acc[i][j] = c;
                }
            {
              constexpr int RD = NS * (TMx + TNx), MF = TMx * TNx * 3;   // LDS reads / MFMAs per k-step
              __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
              for (int r = 0; r < RD; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF - RD, 0);
            }
          } else {
            // one MFMA per (tile, product) covers the 32-channel group: lane group kq holds chunk kq.  The pixel fragments
            // and the first half of the weight rows are requested up front, the second half's reads go one behind each of
            // the first MFMAs
#pragma unroll
            for (int pp = 0; pp < NS; ++pp)
#pragma unroll
              for (int j = 0; j < TNx; ++j)
                bfr[0][pp][j] = __builtin_bit_cast(
                    bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(bb + (uint32_t)((pp * KC + kq) * PXB) * 16u + hoff[j] + tapoff));
#pragma unroll
            for (int i = 0; i < TMx; ++i)
#pragma unroll
              for (int pp = 0; pp < NS; ++pp)
                af[0][pp][i] = __builtin_bit_cast(
                    bf16x8, *(const __attribute__((address_space(3))) u32x4*)(size_t)(ab + (uint32_t)(((pp * KC + kq) * BM + i * 16) * 16)));
#pragma unroll
            for (int i = 0; i < TMx; ++i)
#pragma unroll
              for (int j = 0; j < TNx; ++j) {
                acc_t c = acc[i][j];
                c = MM::mma(af[0][0][i], bfr[0][1][j], c);
                c = MM::mma(af[0][1][i], bfr[0][0][j], c);
                c = MM::mma(af[0][0][i], bfr[0][0][j], c);
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
        if (++slot == 3) slot = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      buf ^= 1;
    }
    // result of this tile: plain stores, left to drain under the next tile's MFMAs
    const int m0 = tile_m_of(tcur) * BM, n0 = tile_n_of(tcur) * BN;
    int R0e = 0, col0e = 0;
    if constexpr (ROWS2) tile_origin(tile_n_of(tcur), R0e, col0e);
    float* out = a.y + (size_t)sk * a.slab_stride;
#pragma unroll
    for (int j = 0; j < TNx; ++j) {
      const int nl = wn * WTN + j * TS + lr;
      const int nn = ROWS2 ? (R0e + (nl >> 6)) * W + col0e + (nl & 63) : n0 + nl;   // global pixel index b*HW + h*W + w
      if (nn >= a.N) continue;
      const int b2 = nn / HW, hw2 = nn - b2 * HW;
      const size_t base = (size_t)b2 * a.Co * HW + hw2;
#pragma unroll
      for (int i = 0; i < TMx; ++i) {
#pragma unroll
        for (int r = 0; r < MM::NR; ++r) {
          const int m = m0 + wm * 32 * TM + i * TS + MM::row(r, kq);
          if (m < a.Co) {
            float v = acc[i][j][r];
            if (F16) v *= oscale;
            if (a.bias) v += a.bias[m];
            out[base + (size_t)m * HW] = v;
          }
        }
      }
    }
    tcur = next_tile(tcur);
  }
}

FwdPlanP2 plan_fwd_p2(int B, int Ci, int H, int W, int Co, int KS, int ns) {
  FwdPlanP2 p;
  memset(&p, 0, sizeof(p));
  const int lw = log2_exact(W), lh = log2_exact(H);
  // (The band form for 4-pixel-wide images -- the 4x4 layers -- was built, verified and measured no faster than the
  // 128-pixel planes kernel, 512 -> 512 @ 4x4 x 128 images: 302 vs 312 TFLOP/s: those launches are bound by streaming
  // 9.4 MB of weights per 128-pixel tile, not by the 9x activation re-reads the band form removes.  Removed.)
  if (KS != 3 || ns != 2 || lw < 3 || lw > 8 || lh < 0 || Ci % 32 || Co < 33) return p;
  // 128- and 256-wide images: 128-pixel tiles (two rows x 64 columns), persistent kernel only
  p.bn = lw > 6 ? 128 : 256;
  p.rows2 = (lw > 6 && H >= 2) ? 1 : 0;
  const int WBh = p.rows2 ? 64 : (W < p.bn ? W : p.bn), NR = p.bn / WBh;
  p.SR = NR < H ? NR : H;
  p.NSEG = NR / p.SR;
  p.NP = p.NSEG * (p.SR + 2) * (WBh + 2);
  p.NPC = cdiv(p.NP, 64);
  p.PXB = p.NPC * 64;
  if (p.NPC > 7) return p;
  p.nt = (int)(((long long)B * H * W + p.bn - 1) / p.bn);
  p.bm = Co <= 64 ? 64 : 128;
  // mid-sized layers: 64-row tiles when that fills the chip without split-K and 128-row tiles would not
  if (p.bm == 128 && cdiv(Co, 128) * p.nt < 192 && cdiv(Co, 64) * p.nt >= 128) p.bm = 64;
  p.lds = ((size_t)3 * band_taps_per_stage(p.bm, p.bn, lw) * 2 * 4 * p.bm + (size_t)2 * 2 * 4 * p.PXB) * 16;   // [3][G] weight ring + 2 bands
  const size_t stage_bytes = p.bn == 256 ? (size_t)p.bm * (256 + 4) * sizeof(float) : 0;   // the staged epilogue's tile reuses the allocation
  if (p.lds < stage_bytes) p.lds = stage_bytes;
  if (p.lds > 160 * 1024) return p;
  p.mt = cdiv(Co, p.bm);
  p.cpt = Ci / 32;
  const int tiles = p.mt * p.nt;
  // mid-sized layers: 128-pixel tiles (conv_fwd_bf16p_kernel) already fill the chip without split-K, 256-pixel
  // bands would not -- measured faster there
  if (tiles < 192 && cdiv(Co, 128) * (int)(((long long)B * H * W + 127) / 128) >= 192 && Co > 64) {
    p.ok = 0;
    return p;
  }
  int splits = 1;
  constexpr int target = 256;   // blocks aimed at when K is split: one per CU
  if (tiles < 192 && p.cpt >= 2) {
    splits = target / tiles;
    if (splits > p.cpt) splits = p.cpt;
    if (splits < 1) splits = 1;
  }
  p.cps = cdiv(p.cpt, splits);
  p.splits = cdiv(p.cpt, p.cps);
  p.ok = 1;
  return p;
}

static int band_persistent_blocks() { return g_opt.band_persist_blocks; }   // itcv_set_option("band_persist_blocks")

template <typename K>
static void set_lds(K kern, size_t& have, size_t lds) {
  if (have < lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    have = lds;
  }
}

// W <= 64: 256-pixel tiles.  More tiles than CUs (and no tile statistics wanted): persistent blocks (one per CU) that
// prefetch the next tile's band under the current MFMAs; else one tile per block.
template <int LOG2W, int BM, bool UP2, bool F16>
static void launch_fwd_p2_cfg(const ConvArgsP2& a, int splits, size_t lds, hipStream_t st) {
  const int ids = cdiv(a.nt, 8) * 8 * a.mt, pb = band_persistent_blocks();
  const bool persistent = !a.stats && pb > 0 && ids > pb;
  const bool m16 = F16 || (band_m16() && !a.stats);    // launches that produce tile statistics keep the 32x32x16 form
  if (persistent) {
    if (m16) {
      auto pk = conv_fwd_bf16p3_kernel<LOG2W, BM, UP2, 256, true, false, F16>;
      static size_t have = 0;
      set_lds(pk, have, lds);
      launch_timed(pk, dim3(pb, splits), dim3(768), lds, st, a);
    } else if constexpr (!F16) {
      auto pk = conv_fwd_bf16p3_kernel<LOG2W, BM, UP2, 256, false, false, false>;
      static size_t have = 0;
      set_lds(pk, have, lds);
      launch_timed(pk, dim3(pb, splits), dim3(768), lds, st, a);
    }
    return;
  }
  const dim3 grid(ids, splits);
  if (m16) {
    auto k16 = conv_fwd_bf16p2_kernel<LOG2W, BM, UP2, true, F16>;
    static size_t have = 0;
    set_lds(k16, have, lds);
    launch_timed(k16, grid, dim3(768), lds, st, a);
  } else if constexpr (!F16) {
    auto kern = conv_fwd_bf16p2_kernel<LOG2W, BM, UP2, false, false>;
    static size_t have = 0;
    set_lds(kern, have, lds);
    launch_timed(kern, grid, dim3(768), lds, st, a);
  }
}
template <int LOG2W, bool F16>
static void launch_fwd_p2_w(const ConvArgsP2& a, int bm, int up2, int splits, size_t lds, hipStream_t st) {
  if (bm == 64) {
    if (up2) launch_fwd_p2_cfg<LOG2W, 64, true, F16>(a, splits, lds, st);
    else launch_fwd_p2_cfg<LOG2W, 64, false, F16>(a, splits, lds, st);
  } else {
    if (up2) launch_fwd_p2_cfg<LOG2W, 128, true, F16>(a, splits, lds, st);
    else launch_fwd_p2_cfg<LOG2W, 128, false, F16>(a, splits, lds, st);
  }
}

// 128- / 256-wide images: the persistent kernel with 128-pixel tiles (2 rows x 64 columns)
template <int LOG2W, int BM, bool UP2, bool M16, bool ROWS2, bool F16>
static void launch_fwd_p3_wide_k(const ConvArgsP2& a, int splits, size_t lds, hipStream_t st) {
  auto pk = conv_fwd_bf16p3_kernel<LOG2W, BM, UP2, 128, M16, ROWS2, F16>;
  static size_t have = 0;
  set_lds(pk, have, lds);
  const int ids = cdiv(a.nt, 8) * 8 * a.mt, nb = band_persistent_blocks() > 0 ? band_persistent_blocks() : 256;
  launch_timed(pk, dim3(ids < nb ? ids : nb, splits), dim3(768), lds, st, a);
}
template <int LOG2W, int BM, bool UP2, bool F16>
static void launch_fwd_p3_wide_cfg(const ConvArgsP2& a, int splits, size_t lds, int rows2, hipStream_t st) {
  const bool m16 = F16 || band_m16();
  if (rows2) {
    if (m16) launch_fwd_p3_wide_k<LOG2W, BM, UP2, true, true, F16>(a, splits, lds, st);
    else if constexpr (!F16) launch_fwd_p3_wide_k<LOG2W, BM, UP2, false, true, false>(a, splits, lds, st);
    return;
  }
  if (m16) launch_fwd_p3_wide_k<LOG2W, BM, UP2, true, false, F16>(a, splits, lds, st);
  else if constexpr (!F16) launch_fwd_p3_wide_k<LOG2W, BM, UP2, false, false, false>(a, splits, lds, st);
}
template <int LOG2W, bool F16>
static void launch_fwd_p3_wide(const ConvArgsP2& a, int bm, int up2, int splits, size_t lds, int rows2, hipStream_t st) {
  if (bm == 64) {
    if (up2) launch_fwd_p3_wide_cfg<LOG2W, 64, true, F16>(a, splits, lds, rows2, st);
    else launch_fwd_p3_wide_cfg<LOG2W, 64, false, F16>(a, splits, lds, rows2, st);
  } else {
    if (up2) launch_fwd_p3_wide_cfg<LOG2W, 128, true, F16>(a, splits, lds, rows2, st);
    else launch_fwd_p3_wide_cfg<LOG2W, 128, false, F16>(a, splits, lds, rows2, st);
  }
}

bool band_is_persistent(const ConvArgsP2& a, const FwdPlanP2& p) {
  const int ids = cdiv(a.nt, 8) * 8 * a.mt;
  return p.bn == 128 || (!a.stats && band_persistent_blocks() > 0 && ids > band_persistent_blocks());
}

template <bool F16>
static void launch_fwd_p2_t(const ConvArgsP2& a, const FwdPlanP2& p, int W, int up2, hipStream_t st) {
  if (p.bn == 128) {
    if (log2_exact(W) == 7) launch_fwd_p3_wide<7, F16>(a, p.bm, up2, p.splits, p.lds, p.rows2, st);
    else launch_fwd_p3_wide<8, F16>(a, p.bm, up2, p.splits, p.lds, p.rows2, st);
    return;
  }
  switch (log2_exact(W)) {
    case 3: launch_fwd_p2_w<3, F16>(a, p.bm, up2, p.splits, p.lds, st); break;
    case 4: launch_fwd_p2_w<4, F16>(a, p.bm, up2, p.splits, p.lds, st); break;
    case 5: launch_fwd_p2_w<5, F16>(a, p.bm, up2, p.splits, p.lds, st); break;
    default: launch_fwd_p2_w<6, F16>(a, p.bm, up2, p.splits, p.lds, st); break;
  }
}
void launch_fwd_p2(const ConvArgsP2& a, const FwdPlanP2& p, int W, int up2, int f16, hipStream_t st) {
  if (f16) launch_fwd_p2_t<true>(a, p, W, up2, st);
  else launch_fwd_p2_t<false>(a, p, W, up2, st);
}

}  // namespace itcv
