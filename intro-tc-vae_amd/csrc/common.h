// Shared helpers for libitcv_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/itcv_hip.h"

namespace itcv {

extern thread_local char g_err[512];

inline int fail(const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return 1;
}

#define ITCV_CHECK_LAUNCH(name)                                                        \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) return itcv::fail("%s: launch failed: %lld", name, (long long)e_); \
  } while (0)

#define ITCV_REQUIRE(cond, name)                                              \
  do {                                                                        \
    if (!(cond)) return itcv::fail("%s: requirement failed: " #cond, name);   \
  } while (0)

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t a, size_t b) { return cdivz(a, b) * b; }

constexpr int kWave = 64;

// ---- wave / block reductions (wave = 64 lanes) -------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T u = __shfl_xor(v, o, 64);
    v = u > v ? u : v;
  }
  return v;
}
// block-wide sum; result valid in every thread.  `scratch` holds >= blockDim/64 elements.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  T r = 0;
  for (int i = 0; i < nw; ++i) r += scratch[i];  // fixed order: deterministic
  return r;
}

// 128-bit buffer resource for raw buffer loads with hardware bounds checking: an offset
// >= num_bytes returns 0, which is how padding / tile tails are zero-filled without branches.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t num_bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, num_bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
constexpr uint32_t kOOB = 0xFFFFFFFFu;

// init + p[0] + p[stride] + ... (n terms, ascending: a fixed order, bitwise reproducible) with the loads of 8 terms in
// flight -- a plain `for (k) s += p[k*stride]` is one exposed memory round trip per term.
template <typename T>
__device__ __forceinline__ T fold_strided(T init, const T* __restrict__ p, size_t stride, int n) {
  T s = init;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k + 2 <= n; k += 2) {
    const T v0 = p[(size_t)k * stride], v1 = p[(size_t)(k + 1) * stride];
    s += v0;
    s += v1;
  }
  for (; k < n; ++k) s += p[(size_t)k * stride];
  return s;
}

// ---- split operands: x = sum_p plane_p(x); 8 values -> one 16-byte chunk per plane -------------------------------
// bf16 planes (formats 2 and 3): each plane is the bf16 rounding of the remaining residual.
// fp16 planes (format ITCV_PLANES_F16X2, "f16x3"): hi = fp16(S*x), lo = fp16(S*x - hi) with a power-of-two scale S per
// tensor: 22 significand bits in two planes, so the same THREE matrix-core products as bf16x3 (hi*lo + lo*hi + hi*hi)
// reach ~2^-21 per product -- fp32 class -- where two bf16 planes reach 2^-16.  fp16's narrow exponent is what the scale is
// for: S maps the tensor's magnitude bound just under 2^15, and the fp16 matrix cores keep subnormal inputs (measured on
// gfx950: tools/f16_probe.hip), so values far below the bound degrade gracefully (absolute error 2^-25 / S) instead of
// flushing.  The product of two such operands comes out scaled by Sa*Sb; the consumer's epilogue multiplies by the exact
// inverse.  Every scaled tensor carries its {S, 1/S} next to the data (ScaleRec, "trailer" of a planes buffer).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ScaleRec {   // 16 bytes: lives behind the last plane of an fp16 planes buffer
  float scale, inv;
  float pad[2];
};
constexpr int kWeightScaleLog2 = 8;   // packed fp16 weights are stored times 2^8 (|w| < 255 representable; typical |w| ~ 0.02 -> 5)

// power-of-two scale that maps `bound` (>= every |value| of the tensor) into [2^14, 2^15); 1 for 0 / inf / nan bounds
__host__ __device__ inline float scale_for_bound(float bound) {
  if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(bound, &e);          // bound = m * 2^e, m in [0.5, 1)  =>  bound < 2^e
  int k = 15 - e;                   // bound * 2^k < 2^15
  k = k > 126 ? 126 : (k < -126 ? -126 : k);
  return ldexpf(1.f, k);
}

template <int NS, bool F16 = false>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[NS], float scale = 1.f) {
  if constexpr (F16) {
    static_assert(NS == 2, "fp16 planes come in pairs");
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float r = v[j] * scale;
      const _Float16 h = (_Float16)r;
      hi[j] = h;
      lo[j] = (_Float16)(r - (float)h);
    }
    out[0] = __builtin_bit_cast(u32x4, hi), out[1] = __builtin_bit_cast(u32x4, lo);
  } else {
    bf16x8 pl[NS];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = v[j];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        const __bf16 b = (__bf16)r;
        pl[p][j] = b;
        r -= (float)b;
      }
    }
#pragma unroll
    for (int p = 0; p < NS; ++p) out[p] = __builtin_bit_cast(u32x4, pl[p]);
  }
}

// matrix-core products on 16-byte fragments (8 x 16-bit) of either element type
typedef float mm_f32x4 __attribute__((ext_vector_type(4)));
typedef float mm_f32x16 __attribute__((ext_vector_type(16)));
template <bool F16>
__device__ __forceinline__ mm_f32x4 mma16x16x32(bf16x8 a, bf16x8 b, mm_f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ mm_f32x16 mma32x32x16(bf16x8 a, bf16x8 b, mm_f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 1/S of a scale record as a wave-uniform value in a scalar register: loaded ONCE where this is called (left as a plain
// load the compiler re-issues it next to every use when it cannot prove that the epilogue's stores do not alias it)
__device__ __forceinline__ float inv_scale_of(const ScaleRec* r) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r->inv)));
}

// max of n non-negative floats at p, by one wave (all 64 lanes call; result in every lane)
__device__ __forceinline__ float wave_absmax_of(const float* __restrict__ p, int n) {
  float m = 0.f;
  for (int i = threadIdx.x & 63; i < n; i += 64) m = fmaxf(m, p[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m;
}
constexpr int kAbsmaxParts = 256;   // block maxima written by itcv_absmax (one float each)

// ---- optional per-launch timing of the GEMM-class kernels (bench.py's roofline leg) ------------
// When enabled, a HIP event pair is recorded on the launch stream immediately around the MAIN kernel
// of a conv call (not its split-K reduce), tagged with the kernel family / template parameters and the
// algorithmic FLOP of the launch.  Off by default; never used under graph capture.
struct ProfScope {
  hipStream_t st;
  int slot;
  ProfScope(hipStream_t stream, int kind, int ks, int bm, int up2, int ns, double flop);
  ~ProfScope();
};
// Event pair of the innermost live ProfScope (null when not profiling).  launch_timed() hands it to
// hipExtLaunchKernelGGL, which stamps the events at the kernel's own start and end: the record is the kernel's
// execution time (what rocprofv3 reports), free of the dispatch latency a hipEventRecord pair would include
// whenever the stream runs dry.
extern thread_local hipEvent_t g_prof_start, g_prof_stop;

template <typename K, typename... Args>
inline void launch_timed(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
  if (g_prof_start) {
    hipExtLaunchKernelGGL(kernel, grid, block, lds, st, g_prof_start, g_prof_stop, 0, args...);
    g_prof_start = nullptr;   // one kernel per scope: a second launch in the same scope is not part of the record
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
}

}  // namespace itcv
