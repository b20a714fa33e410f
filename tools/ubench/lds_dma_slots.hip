// Does an LDS-DMA (global_load_lds_dwordx4) into a per-wave LDS slot behave for all 8 waves of a workgroup?
// Each wave copies 4 KiB (4 x 1 KiB chunks, immediate offsets, one M0) from a pattern table into its slot at
// base + wave*4 KiB, waits vmcnt(0), reads the slot back and compares.  Mismatches are counted per wave index.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_slots lds_dma_slots.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

__global__ void __launch_bounds__(512, 2) k(const float* table, int n_tabs, int iters, int base_floats, int filler, unsigned* bad, unsigned* first_bad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* slot = lds + base_floats + wave * 1024;
  // touch the low LDS like the real kernel does (weights image), then barrier
  for (int i = threadIdx.x; i < base_floats; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  unsigned nbad = 0;
  float sink = 0.0f;
  for (int it = 0; it < iters; ++it) {
    const int tab = (it * 8 + wave + blockIdx.x) % n_tabs;
    const float* src = table + (size_t)tab * 1024;
    gvoid* g = (gvoid*)(src + lane * 4);
    lvoid* l = (lvoid*)slot;
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
    // filler work between issue and use (like the MFMA phase): LDS reads of the low image + VALU
    for (int f = 0; f < filler; ++f) sink += lds[(lane * 4 + f * 64 + wave * 7) % (base_floats > 0 ? base_floats : 1)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(slot + c * 256 + lane * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float want = (float)(tab * 1024 + c * 256 + lane * 4 + r);
        if (v[r] != want) {
          if (nbad == 0) atomicCAS(&first_bad[wave], 0u, (unsigned)(it + 1));
          ++nbad;
        }
      }
    }
    asm volatile("" ::: "memory");
  }
  if (nbad) atomicAdd(&bad[wave], nbad);
  if (sink == 12345.678f) bad[0] = 0xFFFFFFFFu;
}

int main() {
  const int n_tabs = 1024;
  std::vector<float> h((size_t)n_tabs * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float* table; unsigned *bad, *first;
  hipMalloc(&table, h.size() * 4); hipMalloc(&bad, 32); hipMalloc(&first, 32);
  hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int base_floats : {0, 24576}) {
    for (int filler : {0, 16, 256}) {
      const size_t ldsb = (size_t)(base_floats + 8 * 1024) * 4;
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
      hipMemset(bad, 0, 32); hipMemset(first, 0, 32);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), ldsb, 0, table, n_tabs, 2000, base_floats, filler, bad, first);
      hipError_t e = hipDeviceSynchronize();
      unsigned hb[8], hf[8];
      hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost);
      printf("base=%6d floats filler=%3d lds=%zu B (%s): bad per wave:", base_floats, filler, ldsb, hipGetErrorString(e));
      for (int w = 0; w < 8; ++w) printf(" %u", hb[w]);
      printf("  first-bad iter:");
      for (int w = 0; w < 8; ++w) printf(" %u", hf[w]);
      printf("\n");
    }
  }
  return 0;
}
