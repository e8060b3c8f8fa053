// Microbenchmark: what feeds v_mfma_f32_32x32x16_f16 at which rate on gfx950?
// variants: chains (independent accumulators per wave), LDS read per MFMA or not, barrier per block,
// waves per SIMD.  Prints TFLOP/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int NK, bool LDS, bool BAR, int BOUND>
__global__ __launch_bounds__(256, BOUND) void k(const half8 *F, float *out, int iters) {
  extern __shared__ __align__(16) unsigned char smem[];
  half8 *sb = reinterpret_cast<half8 *>(smem);
  const int lane = threadIdx.x & 63;
  half8 th[NK];
  for (int i = 0; i < NK; ++i) th[i] = F[i * 64 + lane];
  for (int i = threadIdx.x; i < NK * 64; i += 256) sb[i] = F[(NK + i / 64) * 64 + (i & 63)];
  __syncthreads();
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  half8 areg[NK];
  if (!LDS) for (int i = 0; i < NK; ++i) areg[i] = sb[i * 64 + lane];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      half8 a = LDS ? sb[ks * 64 + lane] : areg[ks];
      acc[ks % CH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, th[ks], acc[ks % CH], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int CH, int NK, bool LDS, bool BAR, int BOUND>
void run(const char *name, const half8 *F, float *out, int wg_per_cu) {
  const int iters = 2000, grid = 256 * wg_per_cu;
  size_t lds = (size_t)NK * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<CH, NK, LDS, BAR, BOUND>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH, NK, LDS, BAR, BOUND><<<grid, 256, lds>>>(F, out, 10);
  hipEventRecord(e0);
  k<CH, NK, LDS, BAR, BOUND><<<grid, 256, lds>>>(F, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)grid * 4 * iters * NK * 32768.0;
  printf("%-44s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s\n", name, wg_per_cu, ms, flop / ms / 1e9);
}

int main() {
  half8 *F; float *out;
  hipMalloc(&F, 128 * 64 * 16); hipMalloc(&out, 4096);
  hipMemset(F, 0, 128 * 64 * 16);
#define R(CH, NK, L, B, BD, W) run<CH, NK, L, B, BD>("chains=" #CH " NK=" #NK " lds=" #L " barrier=" #B, F, out, W)
  R(1, 32, false, false, 2, 1); R(1, 32, false, false, 2, 2);
  R(2, 32, false, false, 2, 1); R(2, 32, false, false, 2, 2);
  R(4, 32, false, false, 2, 1); R(4, 32, false, false, 2, 2);
  R(1, 32, true, false, 2, 2); R(2, 32, true, false, 2, 2); R(4, 32, true, false, 2, 1); R(4, 32, true, false, 2, 2);
  R(4, 32, true, true, 2, 2); R(1, 32, true, true, 2, 2);
  R(2, 7, true, false, 3, 3); R(2, 7, true, true, 3, 3); R(4, 7, true, true, 3, 3); R(2, 7, false, false, 3, 3);
  R(2, 14, true, true, 3, 3); R(4, 14, true, true, 3, 3);
  return 0;
}
