// Does a packed-fp32 op that consumes LDS-read data right after its s_waitcnt see stale data when two waves share
// a SIMD?  Each wave: writes a fresh pattern to its private LDS slot, reads it back with ds_read_b128 and feeds the
// four registers to v_pk_add_f32 (PK=1) or v_add_f32 (PK=0) immediately after the wait; the sums are compared with
// the expected values.  Mismatches are counted per (lane group g = lane/16, register r).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_pk_hazard lds_pk_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(512, 2) k(int iters, unsigned* bad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* slot = lds + wave * 2048;  // two 4 KiB halves per wave, alternated so a stale read returns the older pattern
  unsigned nbad[4] = {0, 0, 0, 0};
  f32x4 keep = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    float* s = slot + (it & 1) * 1024;
    const float base = (float)(it & 1023);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 w;
      for (int r = 0; r < 4; ++r) w[r] = base + (float)(c * 256 + lane * 4 + r);
      *reinterpret_cast<f32x4*>(s + c * 256 + lane * 4) = w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const f32x4*>(s + c * 256 + ((lane + 16 * c) & 63) * 4);  // rotated lanes: real LDS traffic
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float o[4];
      if (PK) {
        const f32x2 lo = f32x2{v[c][0], v[c][1]} + f32x2{1.0f, 1.0f};
        const f32x2 hi = f32x2{v[c][2], v[c][3]} + f32x2{1.0f, 1.0f};
        o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
      } else {
        for (int r = 0; r < 4; ++r) o[r] = v[c][r] + 1.0f;
      }
      for (int r = 0; r < 4; ++r) {
        const float want = base + (float)(c * 256 + ((lane + 16 * c) & 63) * 4 + r) + 1.0f;
        nbad[r] += (o[r] != want);
        keep[r] += o[r];
      }
    }
  }
  for (int r = 0; r < 4; ++r)
    if (nbad[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], nbad[r]);
  if (keep[0] + keep[1] + keep[2] + keep[3] == 1234.5f) bad[0] = 0xFFFFFFFFu;
}

template <int PK>
void run(unsigned* bad, int waves) {
  unsigned h[16];
  hipMemset(bad, 0, 64);
  hipLaunchKernelGGL((k<PK>), dim3(256), dim3(64 * waves), waves * 8192, 0, 100000, bad);
  hipDeviceSynchronize();
  hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
  unsigned tot = 0;
  for (int i = 0; i < 16; ++i) tot += h[i];
  printf("%s consumer, %d waves/WG (%d per SIMD): mismatches %u  by [g][r]:", PK ? "v_pk_add_f32" : "v_add_f32   ", waves, waves / 4, tot);
  for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
  printf("\n");
}

int main() {
  unsigned* bad; hipMalloc(&bad, 64);
  for (int waves : {4, 8}) { run<1>(bad, waves); run<0>(bad, waves); }
  return 0;
}
