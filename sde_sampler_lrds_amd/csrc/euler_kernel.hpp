// EulerIntegrator.integrate (eq/integrator.py:93-129) for SDEs WITHOUT a drift net, gfx950:
//   uncontrolled linear SDEs (OU.drift / OU.diff, eq/sdes.py:143-148: the inference processes of solver/oc.py:162-180)
//   and the classic Langevin SDE (LangevinSDE.drift / .diff, eq/sdes.py:63-76: solver/langevin.py:36-66).
//
//     x' = x + (c1 x + clip(c7 score_pi(x), clip_score)) c4 + c2 (z c5)
//
// c1 = drift_coeff_t(s), c2 = diff_coeff(s), c4 = t - s, c5 = sqrt(t - s) (1 when Brownian increments are injected),
// c7 = diff_coeff^2 / 2 (Langevin).  Same particle layout as the step loop of sim_kernel.hpp (a wave = 16 particles,
// state in registers for all N steps, Philox noise keyed by global particle index), without its LDS image and
// matrix work: a controlled SDE (ControlledSDE, eq/sdes.py:681-720) goes through k_simulate instead.
// xs_out (all N+1 states; the integrator interpolates them onto the caller's grid) makes it HBM-store-bound.
#pragma once
#include "sim_device.hpp"
#include "sim_kernel.hpp"

#define SD_EULER_THREADS 256

template <int NT, int SC>
__global__ void __launch_bounds__(SD_EULER_THREADS) k_euler(const SimArgs a) {
  constexpr int dpad = 16 * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  const bool full_d = a.d == dpad;
  float* trash = a.trash + tid * 4;
  constexpr int WAVES = SD_EULER_THREADS / 64;
  for (int tile = blockIdx.x * WAVES + wave; tile < a.ntiles; tile += gridDim.x * WAVES) {
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = row < static_cast<uint32_t>(a.B);
    const uint32_t pidx = static_cast<uint32_t>(a.particle0 + row);
    f32x4 x[NT];
    load_rows<NT>(a.x_in, row, a.d, live, g, x);
    if (a.xs_out) store_rows<NT>(a.xs_out, trash, row, a.d, live, g, x);
    for (int k = 0; k < a.N; ++k) {
      const float* cf = a.coef + static_cast<size_t>(k) * SDENG_NCOEF;
      const float c1 = cf[1], c2 = cf[2], c4 = cf[4], c5 = cf[5], c7 = cf[7];
      f32x4 ts[SC != SC_NONE ? NT : 1];
      if constexpr (SC == SC_GMM) {
        if (NT == 1 && a.target.kind == SDENG_DIST_RINGS) ts[0] = rings_score(x[0], a.target, g);
        else gmm_score<NT>(x, a.target.tab, a.target.consts, 4, a.target.k, a.target.p0, g, ts);
      }
      if constexpr (SC == SC_PHI4) phi4_score<NT>(x, a.target, a.d, g, lane, ts);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 z;
        if (a.noise_in) z = load_quad(a.noise_in + static_cast<size_t>(k) * a.B * a.d, row, a.d, live, t, g);
        else z = philox_normal4(pidx, static_cast<uint32_t>(k), static_cast<uint32_t>(4 * t + g), 0u, a.seed_lo, a.seed_hi);
        f32x4 sv = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (SC != SC_NONE) {  // clip(score * g^2/2): the bound rarely binds -- one test per tile (sim_device.hpp)
          sv = ts[t] * c7;
          if (a.clip_score > 0.0f) clamp_tile_rare(sv, a.clip_score);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool in_range = full_d || feat_lt(t, r, 4 * g, a.d);  // pad features stay exactly zero
          float drift = c1 * x[t][r];
          if constexpr (SC != SC_NONE) drift = drift + sv[r];
          const float xn = (x[t][r] + drift * c4) + c2 * (z[r] * c5);
          x[t][r] = in_range ? xn : 0.0f;
        }
      }
      if (a.xs_out) store_rows<NT>(a.xs_out + static_cast<size_t>(k + 1) * a.B * a.d, trash, row, a.d, live, g, x);
    }
    store_rows<NT>(a.x_out, trash, row, a.d, live, g, x);
    if (live && g == 0) a.rnd_out[row] = 0.0f;
  }
}

template <int NT, int SC>
static int launch_euler_one(const SimArgs& a, hipStream_t stream) {
  const int waves = SD_EULER_THREADS / 64;
  int grid = (a.ntiles + waves - 1) / waves;
  if (grid > 256 * 8) grid = 256 * 8;  // 8 resident workgroups per CU cover the chip; beyond that tiles are strided
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_euler<NT, SC>), dim3(grid), dim3(SD_EULER_THREADS), 0, stream, a);
  return static_cast<int>(hipGetLastError());
}
template <int NT>
static int launch_euler(const SimArgs& a, int sc, hipStream_t stream) {
  if (sc == SC_NONE) return launch_euler_one<NT, SC_NONE>(a, stream);
  if (sc == SC_GMM) return launch_euler_one<NT, SC_GMM>(a, stream);
  if (sc == SC_PHI4) return launch_euler_one<NT, SC_PHI4>(a, stream);
  return static_cast<int>(hipErrorInvalidValue);
}
#define SD_DEFINE_EULER(NT) \
  int sd_launch_euler_##NT(const SimArgs& a, int sc, hipStream_t s) { return launch_euler<NT>(a, sc, s); }
