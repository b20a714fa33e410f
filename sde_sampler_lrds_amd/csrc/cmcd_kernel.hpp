// CMCD step loop (ControlledLangevinSDELoss.simulate, losses/oc.py:666-755) for a Bayesian logistic-regression
// target (distr/logistic_regression.py) with a full-covariance or isotropic Gaussian prior, on gfx950.
//
// The reference evaluates, per step, the drift net twice and the annealed score twice, each target score being
// an autograd pass (4 per step with ScoreCtrl).  (t_k, y_k) of step k is (s_{k+1}, x_{k+1}) of step k+1 and the
// control and the drift use the same target score, so ONE evaluation per step suffices; it is carried in
// registers.  An evaluation is three FP32-MFMA chains: the drift net, logits = X w (+c) -> residual -> X^T r
// (design matrix once in LDS, read in both orientations), and the prior score -P (y - mu).
#pragma once
#include "sim_device.hpp"

struct CmcdArgs {
  SimArgs s;                // common fields (coef has N+1 rows: row k col 0 = ts[k])
  const float* x_image;     // global copy of the LDS design-matrix image [SD_LR_ROWS][SD_LR_STRIDE] (col d-1 = 1)
  const float* y_pad;       // [SD_LR_ROWS] labels (0 on pad rows)
  const float* prec_pack;   // packed prior precision (A operands, like a [d x d] layer) or nullptr (isotropic)
  const float* prior_loc;   // [16*NT] prior mean (0-padded)
  float w_scale2, c_mean, c_scale2, thr;   // weight_scale^2, intercept_mean, intercept_scale^2, threshold
  float iso_loc, iso_var;   // isotropic prior
  int n_tiles_rows;         // data-row tiles actually populated (ceil(n/16))
};

__host__ __device__ inline int cmcd_lds_floats(int NT, bool full_prior) {
  return sd_lds_weight_floats(NT) + SD_LR_ROWS * SD_LR_STRIDE + (full_prior ? NT * NT * 256 : 0);
}

// (u, b) at (time index ki, state x): u = ctrl(t, x) (reparam.py:112-117), b = annealed drift (eq/sdes.py:101-110)
template <int NT>
SD_INLINE void cmcd_eval(const CmcdArgs& a, const f32x4 (&x)[NT], int ki, float w_t, float w_1mt, const float* lds,
                         const float* bias, int lane, f32x4 (&u)[NT], f32x4 (&b)[NT]) {
  const int g = lane >> 4;
  const SimArgs& s = a.s;
  const float* xim = lds + sd_lds_weight_floats(NT);
  // ---- target score: prior part + X^T r ----
  f32x4 ts[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = feat(t, r, g);
      const float xv = x[t][r];
      float v = -xv / a.w_scale2;                                  // logistic_regression.py:72
      v = (f == s.d - 1) ? -(xv - a.c_mean) / a.c_scale2 : v;      // :74
      ts[t][r] = (f < s.d) ? v : 0.0f;
    }
  for (int nt = 0; nt < a.n_tiles_rows; nt += 2) {  // two 16-row tiles of the design matrix per pass
    f32x4 lg[2];
    lg[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    lg[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    dense_plain<NT, 2, 0>(x, lg, xim, lane, 0, nt);               // logits of 32 data rows (intercept via the 1-column)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const f32x4 yv = load_tile4(a.y_pad, nt + o, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) lg[o][r] = logreg_residual(lg[o][r], yv[r], a.thr);
    }
    dense_plain<2, NT, 1>(lg, ts, xim, lane, nt, 0);              // ts += X^T r (pad rows of X are zero)
  }
  // ---- prior score ----
  f32x4 ps[NT];
  if (a.prec_pack) {  // GaussFull: -P (x - mu)   distr/gauss.py:129-135
    f32x4 df[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      df[t] = x[t] - load_tile4(a.prior_loc, t, g);
      ps[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    dense_f32<NT, NT>(df, ps, xim + SD_LR_ROWS * SD_LR_STRIDE, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) ps[t] = -ps[t];
  } else {  // IsotropicGauss.score  distr/gauss.py:764-766
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) ps[t][r] = (feat(t, r, g) < s.d) ? (a.iso_loc - x[t][r]) / a.iso_var : 0.0f;
  }
  // ---- annealed drift: 0.5 g^2 clip(score_pi * t/T + score_prior * (1 - t/T)) ----
  const float hg2 = 0.5f * (s.cmcd_g * s.cmcd_g);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = ts[t][r] * w_t + ps[t][r] * w_1mt;
      v = v * hg2;
      if (s.cmcd_clip > 0.0f) v = clampf(v, s.cmcd_clip);
      b[t][r] = v;
    }
  // ---- control ----
  f32x4 hid[SD_HT];
  mlp_hidden<NT>(x, hid, lds, bias, s.temb + static_cast<size_t>(ki) * SD_H, lane);
  const float st = s.stheta ? s.stheta[ki] : 1.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4 o[1];
    mlp_out_tiles<NT, 1>(hid, lds, bias, t, lane, o);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float uv = o[0][r];
      if (s.clip_model > 0.0f) uv = clampf(uv, s.clip_model);
      if (s.ctrl_kind == SDENG_CTRL_SCORE) {
        float sv = ts[t][r];
        if (s.clip_score > 0.0f) sv = clampf(sv, s.clip_score);
        float v = s.scale_score * sv;
        v = v * st;
        uv = uv + v;
      }
      u[t][r] = uv;
    }
  }
}

template <int NT>
__global__ void __launch_bounds__(SD_THREADS, SD_WAVES / 4) k_simulate_cmcd(const CmcdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SimArgs& s = a.s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int nw = sd_lds_weight_floats(NT);
    for (int i = tid; i < nw / 4; i += SD_THREADS) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(s.wpack)[i];
    float* xim = lds + nw;
    for (int i = tid; i < SD_LR_ROWS * SD_LR_STRIDE; i += SD_THREADS) xim[i] = a.x_image[i];
    if (a.prec_pack) {
      float* pp = xim + SD_LR_ROWS * SD_LR_STRIDE;
      for (int i = tid; i < NT * NT * 256; i += SD_THREADS) pp[i] = a.prec_pack[i];
    }
  }
  __syncthreads();
  const float* bias = s.wpack + sd_off_bias(NT);
  const int p = lane & 15, g = lane >> 4;
  float* trash = s.trash + tid * 4;
  const float gg = s.cmcd_g;

  for (int tile = blockIdx.x + gridDim.x * wave; tile < s.ntiles; tile += gridDim.x * SD_WAVES) {  // CUs first
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = row < static_cast<uint32_t>(s.B);
    const uint32_t pidx = static_cast<uint32_t>(s.particle0 + row);
    f32x4 x[NT];
    load_rows<NT>(s.x_in, row, s.d, live, g, x);
    float rnd = 0.0f;
    if (s.rnd_init) rnd = (live ? s.rnd_init[row] : 0.0f);  // rnd0 = log p_prior(x0)  (losses/oc.py:695-699)
    if (s.xs_out) store_rows<NT>(s.xs_out, trash, row, s.d, live, g, x);
    f32x4 u_s[NT], b_s[NT];
    if (s.N > 0) cmcd_eval<NT>(a, x, 0, s.coef[4], s.coef[5], lds, bias, lane, u_s, b_s);

    for (int k = 0; k < s.N; ++k) {
      const float* cf = s.coef + static_cast<size_t>(k) * SDENG_NCOEF;
      const float dt = cf[2], sqdt = cf[3];
      // y = x + (b_s + u_s g) dt + g db ,  db = sqrt(dt) z      (losses/oc.py:722-724)
      f32x4 y[NT], db[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 z;
        if (s.noise_in) {
          z = load_quad(s.noise_in + static_cast<size_t>(k) * s.B * s.d, row, s.d, live, t, g);
        } else {
          z = philox_normal4(pidx, static_cast<uint32_t>(k), static_cast<uint32_t>(4 * t + g), 0u, s.seed_lo, s.seed_hi);
#pragma unroll
          for (int r = 0; r < 4; ++r) z[r] = feat_lt(t, r, 4 * g, s.d) ? z[r] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dbv = sqdt * z[r];
          db[t][r] = dbv;
          y[t][r] = x[t][r] + (b_s[t][r] + u_s[t][r] * gg) * dt + gg * dbv;
        }
      }
      f32x4 u_t[NT], b_t[NT];
      cmcd_eval<NT>(a, y, k + 1, cf[6], cf[7], lds, bias, lane, u_t, b_t);
      // cost = (b_s + b_t)/g + u_s - u_t ;  rnd += 0.5 |cost|^2 dt + <cost, db>   (losses/oc.py:737-742)
      float c2 = 0.0f, cdb = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float c = ((b_s[t][r] + b_t[t][r]) / gg + u_s[t][r]) - u_t[t][r];
          c2 = __builtin_fmaf(c, c, c2);
          cdb = __builtin_fmaf(c, db[t][r], cdb);
        }
      c2 = group_sum(c2);
      cdb = group_sum(cdb);
      rnd += (0.5f * c2) * dt;
      rnd += cdb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        x[t] = y[t];
        u_s[t] = u_t[t];
        b_s[t] = b_t[t];
      }
      if (s.xs_out) store_rows<NT>(s.xs_out + static_cast<size_t>(k + 1) * s.B * s.d, trash, row, s.d, live, g, x);
    }
    store_rows<NT>(s.x_out, trash, row, s.d, live, g, x);
    if (live && g == 0) s.rnd_out[row] = rnd;
  }
}

template <int NT>
static int launch_cmcd(const CmcdArgs& a, int grid, hipStream_t stream) {
  const size_t lds_bytes = static_cast<size_t>(cmcd_lds_floats(NT, a.prec_pack != nullptr)) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_simulate_cmcd<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((k_simulate_cmcd<NT>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  return static_cast<int>(hipGetLastError());
}
