// Microbenchmark 2: the screen's main loop skeleton -- every iteration a workgroup stages a fresh
// candidate group (CTG*NK KiB of fp16 fragments, out of a few-MB L2-resident chunk) into LDS and
// every wave multiplies it with its register-resident target fragments.  No epilogue.
// STAGE 0 = no staging (same LDS data every iteration), 1 = global_load + ds_write double buffer,
// 2 = LDS-DMA double buffer, 3 = LDS-DMA ring of 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NK, int CTG, int WPB, int STAGE, int BOUND>
__global__ __launch_bounds__(64 * WPB, BOUND) void k(const half8 *F, float *out, int iters, int n_groups) {
  constexpr int TILE = CTG * NK * 64, NTH = 64 * WPB, NPT = (TILE + NTH - 1) / NTH;
  constexpr int NSLOT = STAGE == 3 ? 3 : 2, NPW = (CTG * NK + WPB - 1) / WPB;
  extern __shared__ __align__(16) unsigned char smem[];
  half8 *sb = reinterpret_cast<half8 *>(smem);
  const int lane = threadIdx.x & 63, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  half8 th[NK];
  for (int i = 0; i < NK; ++i) th[i] = F[i * 64 + lane];
  f32x16 acc[CTG];
  for (int c = 0; c < CTG; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  auto dma = [&](int g, int slot) {
    const half8 *src = F + (size_t)g * TILE;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      int p = wave + i * WPB;
      if (p >= CTG * NK) p -= WPB;
      __builtin_amdgcn_global_load_lds(src + p * 64 + lane, (__attribute__((address_space(3))) void *)(sb + slot * TILE + p * 64), 16, 0, 0);
    }
  };
  half8 pre[STAGE == 1 ? NPT : 1];
  if (STAGE == 1 || STAGE == 0) { for (int i = tid; i < TILE; i += NTH) sb[i] = F[i]; }
  if (STAGE == 2) dma(0, 0);
  if (STAGE == 3) { dma(0, 0); dma(1, 1); }
  __syncthreads();
  int g = blockIdx.x % 7;   // workgroups of a launch stream the same chunk, slightly out of phase
  for (int it = 0; it < iters; ++it) {
    const int slot = STAGE == 3 ? it % 3 : (it & 1);
    g = g + 1 < n_groups ? g + 1 : 0;
    if (STAGE == 1) {
#pragma unroll
      for (int p = 0; p < NPT; ++p) if (p * NTH + tid < TILE) pre[p] = F[(size_t)g * TILE + p * NTH + tid];
    }
    if (STAGE == 2) dma(g, slot ^ 1);
    if (STAGE == 3) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      __builtin_amdgcn_s_barrier();
      dma(g, (it + 2) % 3);
    }
    const half8 *s = sb + (STAGE == 0 ? 0 : slot) * TILE;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
      for (int c = 0; c < CTG; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(s[(c * NK + ks) * 64 + lane], th[ks], acc[c], 0, 0, 0);
    if (STAGE == 1) {
#pragma unroll
      for (int p = 0; p < NPT; ++p) if (p * NTH + tid < TILE) sb[(slot ^ 1) * TILE + p * NTH + tid] = pre[p];
    }
    if (STAGE != 3) __syncthreads();
  }
  float sum = 0.f;
  for (int c = 0; c < CTG; ++c) for (int r = 0; r < 16; ++r) sum += acc[c][r];
  if (sum == 12345.678f) out[threadIdx.x] = sum;
}

template <int NK, int CTG, int WPB, int STAGE, int BOUND>
void run(const char *name, const half8 *F, float *out, int wg_per_cu, int n_groups) {
  const int iters = 2000;
  int grid = 256 * wg_per_cu;
  if (const char *e = getenv("UB_GRID")) grid = atoi(e);   // fewer workgroups than slots: power headroom?
  size_t lds = (size_t)(STAGE == 3 ? 3 : 2) * CTG * NK * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<NK, CTG, WPB, STAGE, BOUND>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NK, CTG, WPB, STAGE, BOUND><<<grid, 64 * WPB, lds>>>(F, out, 10, n_groups);
  hipEventRecord(e0);
  k<NK, CTG, WPB, STAGE, BOUND><<<grid, 64 * WPB, lds>>>(F, out, iters, n_groups);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)grid * WPB * iters * NK * CTG * 32768.0;
  double bytes = (double)grid * iters * CTG * NK * 1024.0;
  printf("%-40s WG/CU %d  %8.3f ms  %7.1f TFLOP/s  staged %6.2f TB/s\n", name, wg_per_cu, ms, flop / ms / 1e9,
         STAGE ? bytes / ms / 1e9 : 0.0);
}

int main() {
  half8 *F; float *out;
  const size_t chunk = 3 << 20;
  hipMalloc(&F, chunk + (1 << 20)); hipMalloc(&out, 4096);
  {
    std::vector<unsigned short> h((chunk + (1 << 20)) / 2);
    unsigned int x = 12345u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3800u + ((x >> 16) & 0x07ffu) + ((x >> 28) & 8u) * 0x1000u); }
    if (getenv("UB_ZERO")) for (auto &v : h) v = 0;
    hipMemcpy(F, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  }
#define R(NK, CTG, WPB, ST, BD, W) run<NK, CTG, WPB, ST, BD>("NK=" #NK " CTG=" #CTG " WPB=" #WPB " stage=" #ST, F, out, W, (int)(chunk / (CTG * NK * 1024)))
  if (getenv("UB_GRID")) { R(32, 1, 4, 2, 2, 2); R(7, 2, 4, 3, 3, 3); return 0; }
  R(32, 1, 4, 0, 2, 2); R(32, 1, 4, 1, 2, 2); R(32, 1, 4, 2, 2, 2);
  R(32, 1, 8, 0, 2, 1); R(32, 1, 8, 1, 2, 1); R(32, 1, 8, 2, 2, 1); R(32, 1, 8, 3, 2, 1);
  R(32, 2, 8, 2, 2, 1);
  R(7, 2, 4, 0, 3, 3); R(7, 2, 4, 1, 3, 3); R(7, 2, 4, 2, 3, 3); R(7, 2, 4, 3, 3, 3);
  R(7, 2, 8, 2, 3, 1); R(7, 2, 8, 3, 3, 1);
  return 0;
}
