// Declarations shared by the conv translation units (conv_igemm.hip, conv_band.hip): vector typedefs, the
// diagnostic-build switches, the raw-ISA LDS-DMA helpers and the band kernel's host interface.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace itcv {

// Diagnostic instrumentation (operand ablation, in-kernel cycle stamps) exists only in -DITCV_DIAG builds
// (`make diag`, used by tools/abl.sh): in the shipped library the tests below are the constant 0 and the branches are
// compiled out.  The shipped library reads NO environment variable: the two launch-shape choices that remain selectable
// are explicit, validated options (itcv_set_option, include/itcv_hip.h), each exercised by the test matrix.
#ifdef ITCV_DIAG
#define ITCV_ABL(args, bits) (((args).ablate & (bits)) != 0)
#define ITCV_DBG(args) ((args).debug != 0)
static int diag_ablate() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("ITCV_ABLATE");
    v = e ? atoi(e) : 0;
  }
  return v;
}
#else
#define ITCV_ABL(args, bits) (false)
#define ITCV_DBG(args) (false)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA of one 16-byte chunk per lane (global -> LDS at lds_base + lane*16), issued as raw ISA: the
// compiler's own tracking of the builtin puts an `s_waitcnt vmcnt(0)` in front of EVERY later ds_read
// (it cannot tell the ring slots apart), which serialises the prefetch with the tile being computed.
// The caller orders the reads by hand (s_waitcnt vmcnt(N) + barrier).
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is a reserved register: named so the compiler re-materialises it
__device__ __forceinline__ void lds_dma16(const void* gptr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(gptr)
               : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)p;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

static inline int log2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

struct ConvArgsP2 {
  const u32x4* xp;
  const u32x4* wp;
  const float* bias;
  float* y;
  int B, Ci, H, Co;
  int Mp, N;
  int mt, nt;
  int cpt, cpt_per_split;        // 32-channel groups, and how many of them one split-K slice takes
  int SR, NSEG, NP, NPC, PXB;    // band geometry: segment rows, segments, halo pixels, 64-chunk pieces, LDS row stride
  int h_shift;
  float* stats;                  // optional [2][Co][stat_T]: per-tile sum / sum of squares of the output (BatchNorm statistics)
  int stat_T;
  const ScaleRec* xscale;        // fp16 planes: the input's scale record (results are multiplied by xscale->inv * 2^-kWeightScaleLog2)
  size_t slab_stride, plane_stride;
#ifdef ITCV_DIAG
  int debug;   // diagnostic only (ITCV_ABLATE & 64): block 0 / 100 report main-loop shader cycles and 100 MHz ticks in y[0..3]
#endif
};

// taps (K-tiles) per barrier stage of the band kernels: tiles with little MFMA work per tap take two
__host__ __device__ constexpr int band_taps_per_stage(int bm, int bn, int /*lw*/) {
  return (bm == 64 || bn == 128) ? 2 : 1;
}

// MFMA shape of the band kernels' inner product.  The chip lowers its clock under dense bf16 MFMA loops and holds a
// higher one on v_mfma_f32_16x16x32_bf16 than on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back
// item 7); with 16x16x32 a lane's fragment is still one 16-byte plane chunk (8 channels of a row / pixel), the four
// lane groups take the four chunks of a 32-channel group, and one MFMA covers the whole group.
template <bool M16, bool F16 = false>
struct BandMfma;
template <bool F16>
struct BandMfma<false, F16> {
  typedef f32x16 acc_t;
  static constexpr int TS = 32, NR = 16, KSN = 2;        // tile side, accumulator registers, MFMAs (k-steps) per 32-channel group
  static __device__ __forceinline__ acc_t mma(bf16x8 a, bf16x8 b, acc_t c) { return mma32x32x16<F16>(a, b, c); }
  static __device__ __forceinline__ int row(int r, int kq) { return (r & 3) + 8 * (r >> 2) + 4 * kq; }
};
template <bool F16>
struct BandMfma<true, F16> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static constexpr int TS = 16, NR = 4, KSN = 1;
  static __device__ __forceinline__ acc_t mma(bf16x8 a, bf16x8 b, acc_t c) { return mma16x16x32<F16>(a, b, c); }
  static __device__ __forceinline__ int row(int r, int kq) { return 4 * kq + r; }
};

// Options (itcv_set_option): process-wide, read at launch time.
//   band_m16             1 (default): v_mfma_f32_16x16x32_bf16 in the band kernels and the 128-pixel planes kernel; 0: 32x32x16.
//                        Same-box A/B of the c2 step: 16.02 -> 15.75 ms, the 64 -> 64 @ 64x64 launch 104 -> 96 us.  Launches that
//                        produce BatchNorm tile statistics always use the 32x32x16 instantiation (its staged epilogue is
//                        written for that accumulator layout).
//   band_persist_blocks  256 (default): block count of the persistent band kernel, used when a launch has more tiles than
//                        that; 0: one tile per block always.  Any value gives bit-identical results (tests).
//   wgrad_m16            1 (default): v_mfma_f32_16x16x32 in the planes weight-gradient kernel; 0: 32x32x16.
//   planes_mfma_waves    8 (default): the 128 x 128 tile of the 128-pixel planes kernel (the 4x4 layers) runs eight MFMA waves
//                        of 64 x 32 (two per SIMD: one fills the other's barrier / LDS-latency bubbles) instead of four of
//                        64 x 64: bit-identical, 1.5-8 % faster on the 4x4 layers (tools/planes_waves_bench.py).
struct Options {
  int band_m16 = 1;
  int band_persist_blocks = 256;
  int wgrad_m16 = 1;
  int planes_mfma_waves = 8;   // MFMA waves of the 128 x 128 tile of the 128-pixel planes kernel (the 4x4 layers): 4 or 8
};
extern Options g_opt;
inline int band_m16() { return g_opt.band_m16; }

struct FwdPlanP2 {
  int ok, bm, bn, mt, nt, cpt, splits, cps, SR, NSEG, NP, NPC, PXB;
  int rows2;   // wide images: the 128-pixel tile is 2 rows x 64 columns
  size_t lds;
};

// conv_band.hip
FwdPlanP2 plan_fwd_p2(int B, int Ci, int H, int W, int Co, int KS, int ns);
void launch_fwd_p2(const ConvArgsP2& a, const FwdPlanP2& p, int W, int up2, int f16, hipStream_t st);
bool band_is_persistent(const ConvArgsP2& a, const FwdPlanP2& p);   // will launch_fwd_p2 use the persistent kernel?

}  // namespace itcv
