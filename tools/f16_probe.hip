// Probe (diagnostic, not product): does v_mfma_f32_*_f16 keep fp16 subnormal inputs, and how fast is the f16 MFMA loop
// against the bf16 one on random data (DVFS)?   hipcc --offload-arch=gfx950 tools/f16_probe.hip -o /tmp/f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void denorm_probe(float* out, float av, float bv) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)av, b[i] = (_Float16)bv;
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  f16v d;
  for (int i = 0; i < 16; ++i) d[i] = 0;
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0], out[1] = d[0], out[2] = (float)a[0], out[3] = (float)b[0];
}

template <bool F16>
__global__ __launch_bounds__(256) void rate_probe(const unsigned* src, float* out, int iters) {
  // operands from memory (random bits with sane exponents), 8 independent 16x16 accumulators per wave
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = src[(threadIdx.x * 8 + i + blockIdx.x * 2048) & 0xFFFFF];
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 ra = {u[0], u[1], u[2], u[3]}, rb = {u[4], u[5], u[6], u[7]};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (F16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ra), __builtin_bit_cast(h8, rb), acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, ra), __builtin_bit_cast(b8, rb), acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 22);
  float h[4];
  const float cases[][2] = {{9.5367431640625e-07f, 1.0f}, {1.0f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 1024.f}, {3.0517578125e-05f, 3.0517578125e-05f}};
  for (auto& cs : cases) {
    denorm_probe<<<1, 64>>>(d, cs[0], cs[1]);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a=%g b=%g (as f16: %g %g): 16x16x32 -> %g (expect %g), 32x32x16 -> %g (expect %g)\n", cs[0], cs[1], h[2], h[3], h[0],
           32.0 * h[2] * h[3], h[1], 16.0 * h[2] * h[3]);
  }
  // random operands: fp16 values ~N(0,1)-ish: random mantissa, exponent in [2^-4, 2^3]
  std::vector<unsigned> src(1 << 20);
  srand(1);
  for (auto& v : src) {
    unsigned lo = ((rand() & 0x83FF) | ((11 + rand() % 8) << 10)), hi = ((rand() & 0x83FF) | ((11 + rand() % 8) << 10));
    v = lo | (hi << 16);
  }
  std::vector<unsigned> srcb(1 << 20);
  for (auto& v : srcb) {
    unsigned lo = ((rand() & 0x807F) | ((123 + rand() % 8) << 7)), hi = ((rand() & 0x807F) | ((123 + rand() % 8) << 7));
    v = lo | (hi << 16);
  }
  // fp16 operands with half of the elements SUBNORMAL (exponent field 0), as the lo plane of O(1) activations has
  std::vector<unsigned> srcd(1 << 20);
  for (auto& v : srcd) {
    unsigned lo = (rand() & 1) ? (rand() & 0x83FF) : ((rand() & 0x83FF) | ((11 + rand() % 8) << 10));
    unsigned hi = (rand() & 1) ? (rand() & 0x83FF) : ((rand() & 0x83FF) | ((11 + rand() % 8) << 10));
    v = lo | (hi << 16);
  }
  unsigned *ds, *dsb, *dsd;
  hipMalloc(&ds, 4 << 20);
  hipMalloc(&dsb, 4 << 20);
  hipMalloc(&dsd, 4 << 20);
  hipMemcpy(dsd, srcd.data(), 4 << 20, hipMemcpyHostToDevice);
  hipMemcpy(ds, src.data(), 4 << 20, hipMemcpyHostToDevice);
  hipMemcpy(dsb, srcb.data(), 4 << 20, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 4000, blocks = 1024;
  for (int rep = 0; rep < 3; ++rep)
    for (int f = 0; f < 3; ++f) {
      for (int w = 0; w < 20; ++w) {
        if (f == 2) rate_probe<true><<<blocks, 256>>>(dsd, d, iters);
        else if (f) rate_probe<true><<<blocks, 256>>>(ds, d, iters);
        else rate_probe<false><<<blocks, 256>>>(dsb, d, iters);
      }
      hipEventRecord(e0);
      for (int w = 0; w < 20; ++w) {
        if (f == 2) rate_probe<true><<<blocks, 256>>>(dsd, d, iters);
        else if (f) rate_probe<true><<<blocks, 256>>>(ds, d, iters);
        else rate_probe<false><<<blocks, 256>>>(dsb, d, iters);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flop = 20.0 * blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
      printf("%s 16x16x32 random data: %.1f TFLOP/s\n", f == 2 ? "f16 (half subnormal)" : (f ? "f16 " : "bf16"), flop / (ms * 1e-3) / 1e12);
    }
  return 0;
}
