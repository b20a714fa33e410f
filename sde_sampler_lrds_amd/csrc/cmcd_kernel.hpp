// CMCD step loop (ControlledLangevinSDELoss.simulate, losses/oc.py:666-755) for a Bayesian logistic-regression
// target (distr/logistic_regression.py) or a diagonal Gaussian mixture target, with a full-covariance, diagonal or
// isotropic Gaussian prior, on gfx950.
//
// The reference evaluates, per step, the drift net twice and the annealed score twice, each target score being
// an autograd pass (4 per step with ScoreCtrl).  (t_k, y_k) of step k is (s_{k+1}, x_{k+1}) of step k+1 and the
// control and the drift use the same target score, so ONE evaluation per step suffices; it is carried in
// registers.  An evaluation is four split-f16 MFMA chains that share one split of the state: the drift net,
// logits = Xa w -> residual y - sigmoid -> Xa^T r (the augmented design matrix sits in LDS in both orientations as
// packed A operands), and the prior score -P (w - mu) (precision matrix read through L2: LDS is full).
#pragma once
#include "sim_device.hpp"

struct CmcdArgs {
  SimArgs s;                // common fields (coef has N+1 rows: row k col 0 = ts[k]); s.lr = logistic-regression target
  const float* prec_pack;   // packed prior precision (A operands of a [d x d] layer, split f16) or nullptr (isotropic)
  const float* prior_loc;   // [16*NT] prior mean (0-padded)
  float iso_loc, inv_iso_var;   // isotropic prior
};

__host__ __device__ inline int cmcd_lds_floats(int NT, int n_rows) { return sd_lds_weight_floats(NT) + sd_lr_floats(NT, n_rows); }

// (u, b) at (time index ki, state x): u = ctrl(t, x) (reparam.py:112-117), b = annealed drift (eq/sdes.py:101-110)
enum { CT_LOGREG = 0, CT_GMM = 1, CT_PHI4 = 2 };  // target kind is a template parameter: one score body per kernel

// TWO = true also returns b2, the drift with a second pair of annealing weights (the noising loop of compute_eubo
// needs drift(t, y) for the cost and drift(s, y) for the next move: same scores, two mixes)
//
// The results leave one 16 x 16 tile at a time through sink(t, u_t, b_t, b2_t) (t is a compile-time unrolled index).  d <= 64: the drift is
// mixed and clipped ahead of the net, as the registers allow.  d > 64 (NT > 4): u[NT], b[NT] beside x, w_s, the normals and the target
// score do not fit 256 registers (452-844 B of scratch per lane in round 2), so the drift of a tile is mixed when the output layer reaches
// that tile and the caller folds the tile into its sums at once; only the target score stays live across the net.
template <int NT, int TGT, bool TWO = false, class Sink>
SD_INLINE void cmcd_eval(const CmcdArgs& a, const f32x4 (&x)[NT], int ki, float w_t, float w_1mt, const float* lds,
                         const float* bias, const NetScale& ns, int lane, float w2_t, float w2_1mt, Sink&& sink) {
  constexpr int KB = (NT + 1) / 2;
  const int g = lane >> 4;
  const SimArgs& s = a.s;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  // the LDS images never change inside the step loop; without a barrier the compiler hoists their reads out of
  // it and keeps the A operands in (spilled) registers
  asm volatile("" ::: "memory");
  f16x8 xh[KB], xl[KB];
  split_tiles<NT>(x, xh, xl);
  // From here to the end of the hidden layers the wave is in chains of dependent matrix instructions (score products, prior precision,
  // drift net): it wins issue arbitration against the other wave of its SIMD meanwhile (sim_kernel.hpp, same reason).  Measured on
  // cfg 4, same box: 6.81 ms without, 6.61 around the hidden layers only, 6.47 from here (level 2; output layer, control and cost at 1).
  __builtin_amdgcn_s_setprio(2);

  f32x4 ts[NT];
  if constexpr (TGT == CT_PHI4) {
    phi4_score<NT>(x, s.target, s.d, g, lane, ts);  // distr/phi_four.py:81-96 (pad features of x stay exactly 0 in this kernel)
  } else if constexpr (TGT == CT_GMM) {
    // diagonal Gaussian / mixture target (distr/gauss.py:97-107, 124-126): tables prepared by k_dist_tables
    gmm_score<NT>(x, s.target.tab, s.target.consts, 4, s.target.k, s.target.p0, g, ts);
  } else {
    static_assert(TGT != CT_LOGREG || NT <= 4, "logistic regression: d <= 64 (design matrix in LDS)");
    // (two copies of the body on purpose: a pointer selected between LDS and global memory would make every A-operand read a flat load)
    if (s.lr.in_lds) logreg_score<NT, true>(x, xh, xl, s.lr, s.d, lds + sd_lds_weight_floats(NT), lane, ts);
    else logreg_score<NT, true>(x, xh, xl, s.lr, s.d, s.lr.image, lane, ts);
  }

  const float hg2 = 0.5f * (s.cmcd_g * s.cmcd_g);
  if constexpr (NT <= 4) {
    // ---- annealed drift: 0.5 g^2 clip(score_pi * t/T + score_prior * (1 - t/T)) ----
    f32x4 b[NT], b2[TWO ? NT : 1];
    if (a.prec_pack) {  // GaussFull: -P (w - mu)   distr/gauss.py:129-135
      f32x4 df[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) df[t] = x[t] - load_tile4(a.prior_loc, t, g);
      f16x8 dh[KB], dl[KB];
      split_tiles<NT>(df, dh, dl);
#pragma unroll
      for (int t = 0; t < NT; ++t) {  // one output tile at a time: 2*KB A-operand vectors live instead of 2*KB*NT
        f32x4 ps[1] = {zero}, pm[1] = {zero};
        dense_pre_buf<KB, 1>(dh, dl, ps, pm, image_rsrc(a.prec_pack, NT * KB * 2 * 1024), static_cast<uint32_t>(t) * KB * 2 * 1024, lane);
        fold_lo<1>(ps, pm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b[t][r] = ts[t][r] * w_t + (-ps[0][r]) * w_1mt;
          if constexpr (TWO) b2[t][r] = ts[t][r] * w2_t + (-ps[0][r]) * w2_1mt;
        }
      }
    } else if (s.prior.kind == SDENG_DIST_GAUSS_DIAG) {  // Gauss.score (score_gauss, distr/gauss.py:124-126)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 pv = gauss_score_tile<NT>(x, s.prior.tab, g, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b[t][r] = ts[t][r] * w_t + pv[r] * w_1mt;
          if constexpr (TWO) b2[t][r] = ts[t][r] * w2_t + pv[r] * w2_1mt;
        }
      }
    } else {  // IsotropicGauss.score  distr/gauss.py:764-766
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = feat_live<NT>(t, r, 4 * g, s.d) ? (a.iso_loc - x[t][r]) * a.inv_iso_var : 0.0f;
          b[t][r] = ts[t][r] * w_t + pv * w_1mt;
          if constexpr (TWO) b2[t][r] = ts[t][r] * w2_t + pv * w2_1mt;
        }
    }
    // uniform run-time switches (clips on / off, control kind) are tested once per tile, never per element: a scalar branch per
    // element stalls issue (sim_kernel.hpp, add_ctrl_score_tile)
    const bool clip_b = s.cmcd_clip > 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      b[t] = b[t] * hg2;
      if (clip_b) clamp_tile_rare(b[t], s.cmcd_clip);
      if constexpr (TWO) {
        b2[t] = b2[t] * hg2;
        if (clip_b) clamp_tile_rare(b2[t], s.cmcd_clip);
      }
    }

    // ---- control ----
    f32x4 hid[SD_HT];
    mlp_hidden_pre<NT>(xh, xl, hid, lds, bias, s.temb + static_cast<size_t>(ki) * SD_H, lane, ns);
    __builtin_amdgcn_s_setprio(1);  // end of the matrix phase (raised at the top of this function); the rest of the step runs at 1
    const HidSplit hs = split_hidden(hid);
    const float st = s.stheta ? s.stheta[ki] : 1.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 o[1];
      mlp_out_tiles<NT, 1>(hs, lds, bias, t, lane, o, ns.inv_out);
      if (s.clip_model > 0.0f) clamp_tile_rare(o[0], s.clip_model);
      if (s.ctrl_kind == SDENG_CTRL_SCORE) {
        f32x4 sv = ts[t];
        if (s.clip_score > 0.0f) clamp_tile_rare(sv, s.clip_score);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s.scale_score * sv[r];
          v = v * st;
          o[0][r] = o[0][r] + v;
        }
      }
      sink(t, o[0], b[t], b2[TWO ? t : 0]);
    }
  } else {
    // ---- d > 64: control first, the drift of a tile beside its control ----
    const bool clip_b = s.cmcd_clip > 0.0f;
    const float st = s.stheta ? s.stheta[ki] : 1.0f;
    f32x4 hid[SD_HT];
    mlp_hidden_pre<NT>(xh, xl, hid, lds, bias, s.temb + static_cast<size_t>(ki) * SD_H, lane, ns);
    __builtin_amdgcn_s_setprio(1);
    const HidSplit hs = split_hidden(hid);
    f16x8 dh[KB], dl[KB];
    const bool full_prior = a.prec_pack != nullptr;
    if (full_prior) {  // split of (w - mu), the B operand of every row block of the precision image: AFTER the hidden layers, in the registers
      f32x4 df[NT];    // the split of the state occupied until then
#pragma unroll
      for (int t = 0; t < NT; ++t) df[t] = x[t] - load_tile4(a.prior_loc, t, g);
      split_tiles<NT>(df, dh, dl);
    }
    const bool diag_prior = s.prior.kind == SDENG_DIST_GAUSS_DIAG;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      asm volatile("" ::: "memory");  // one tile's A operands at a time
      __builtin_amdgcn_sched_barrier(0);  // ... and one tile's products: left free, the scheduler interleaves all NT independent tiles
      f32x4 o[1];
      mlp_out_tiles<NT, 1>(hs, lds, bias, t, lane, o, ns.inv_out);
      f32x4 pv;
      if (full_prior) {  // GaussFull: -P (w - mu)   distr/gauss.py:129-135
        f32x4 ps[1] = {zero}, pm[1] = {zero};
        dense_pre_buf<KB, 1>(dh, dl, ps, pm, image_rsrc(a.prec_pack, NT * KB * 2 * 1024), static_cast<uint32_t>(t) * KB * 2 * 1024, lane);
        fold_lo<1>(ps, pm);
        pv = f32x4{-ps[0][0], -ps[0][1], -ps[0][2], -ps[0][3]};
      } else if (diag_prior) {  // Gauss.score (score_gauss, distr/gauss.py:124-126)
        pv = gauss_score_tile<NT>(x, s.prior.tab, g, t);
      } else {  // IsotropicGauss.score  distr/gauss.py:764-766
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = feat_live<NT>(t, r, 4 * g, s.d) ? (a.iso_loc - x[t][r]) * a.inv_iso_var : 0.0f;
      }
      f32x4 bt, bt2 = zero;
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
        o[0][r] = uv;
        float bv = ts[t][r] * w_t + pv[r] * w_1mt;
        bv = bv * hg2;
        if (clip_b) bv = clampf(bv, s.cmcd_clip);
        bt[r] = bv;
        if constexpr (TWO) {
          float b2v = ts[t][r] * w2_t + pv[r] * w2_1mt;
          b2v = b2v * hg2;
          if (clip_b) b2v = clampf(b2v, s.cmcd_clip);
          bt2[r] = b2v;
        }
      }
      sink(t, o[0], bt, TWO ? bt2 : bt);
    }
  }
}

// PAR = 1: the twin that can replay injected noise and write the trajectory (return_traj, parity tests); the plain sampler (PAR = 0)
// carries neither path -- their masked loads and stores cost scalar registers in the step loop even when they never run.
template <int NT, int TGT, bool EUBO, int PAR>
__global__ void __launch_bounds__(SD_THREADS, SD_WAVES / 4) k_simulate_cmcd(const CmcdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SimArgs& s = a.s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int nw = sd_lds_weight_floats(NT);
    for (int i = tid; i < nw / 4; i += SD_THREADS) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(s.wpack)[i];
    const int ni = s.lr.in_lds ? sd_lr_floats(NT, s.lr.n_rows) : 0;
    for (int i = tid; i < ni / 4; i += SD_THREADS) reinterpret_cast<f32x4*>(lds + nw)[i] = reinterpret_cast<const f32x4*>(s.lr.image)[i];
  }
  __syncthreads();
  const float* bias = s.wpack + sd_off_bias(NT);
  const NetScale ns = load_net_scale(bias, NT);
  const int p = lane & 15, g = lane >> 4;
  float* trash = s.trash + tid * 4;
  const float gg = s.cmcd_g, inv_g = 1.0f / s.cmcd_g;

  for (int tile = blockIdx.x + gridDim.x * wave; tile < s.ntiles; tile += gridDim.x * SD_WAVES) {  // CUs first
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = row < static_cast<uint32_t>(s.B);
    const uint32_t pidx = static_cast<uint32_t>(s.particle0 + row);
    f32x4 x[NT];
    load_rows<NT>(s.x_in, row, s.d, live, g, x);  // (x0 drawn by the engine is materialised first: the initial log-density needs it, sdeng_api.hip)
    float rnd = 0.0f;
    if (s.rnd_init) rnd = (live ? s.rnd_init[row] : 0.0f);  // rnd0 = log p_prior(x0)  (losses/oc.py:695-699)
    if constexpr (PAR != 0) {
      if (s.xs_out) store_rows<NT>(s.xs_out, trash, row, s.d, live, g, x);
    }
    // carried between steps: w_s = b_s/g + u_s, the only combination of (u_s, b_s) the step needs:
    //   y = x + (b_s + u_s g) dt + g db = x + g w_s dt + g db ;   cost = (b_s + b_t)/g + u_s - u_t = w_s + (b_t/g - u_t)
    // (16 registers per 64 features less than carrying u_s and b_s; identical arithmetic for g = 1, the conf default).
    // Noising direction (compute_eubo, losses/oc.py:782-823), rows in iteration order: the move uses -u, so the carried
    // combination is v = drift(t,x)/g - u(t,x);  cost = v + (drift(t,y)/g + u(s,y));  next v = drift(s,y)/g - u(s,y).
    constexpr float sgn = EUBO ? -1.0f : 1.0f;
    f32x4 w_s[NT];
    if (s.N > 0) {
      cmcd_eval<NT, TGT>(a, x, 0, s.coef[4], s.coef[5], lds, bias, ns, lane, 0.0f, 0.0f, [&](int t, const f32x4& u0, const f32x4& b0, const f32x4&) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) w_s[t][r] = b0[r] * inv_g + sgn * u0[r];
      });
    }

    for (int k = 0; k < s.N; ++k) {
      const float* cf = s.coef + static_cast<size_t>(k) * SDENG_NCOEF;
      const float dt = cf[2], sqdt = cf[3];
      const float g_dt = gg * dt, g_sqdt = gg * sqdt;
      // the Philox key re-read through an opaque move every step: left alone, the compiler hoists the ten round keys (20 SGPRs) out of
      // the step loop, runs out of scalar registers and spills them to VGPR lanes -- one v_readlane + wait states per key and use
      uint32_t key_lo = s.seed_lo, key_hi = s.seed_hi;
      asm volatile("" : "+s"(key_lo), "+s"(key_hi));
      // d > 64, same reason for the per-lane part of every address (LDS images beyond the 64 KiB immediate range, bias rows, tables): left
      // loop-invariant, ~60 of them are computed once ahead of the step loop and parked in scratch; derived from a value the compiler
      // cannot see through, they are one add next to their use
      int lane_k = lane;
      if constexpr (NT > 4) asm volatile("" : "+v"(lane_k));
      // y = x + (b_s + u_s g) dt + g db ,  db = sqrt(dt) z      (losses/oc.py:722-724; :800-802 with -u for the noising loop)
      f32x4 db[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 z;
        if (PAR != 0 && s.noise_in) {
          z = load_quad(s.noise_in + static_cast<size_t>(k) * s.B * s.d, row, s.d, live, t, g);
        } else {
          z = philox_normal4(pidx, static_cast<uint32_t>(k), static_cast<uint32_t>(4 * t + g), 0u, key_lo, key_hi);
#pragma unroll
          for (int r = 0; r < 4; ++r) z[r] = feat_live<NT>(t, r, 4 * g, s.d) ? z[r] : 0.0f;
        }
        db[t] = z;  // the normals; db = sqrt(dt) z is applied through the per-step products g sqrt(dt) and (below) sqrt(dt) <c, z>
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] = __builtin_fmaf(g_sqdt, z[r], __builtin_fmaf(w_s[t][r], g_dt, x[t][r]));  // fused (sim_kernel.hpp, update)
      }
      // cost = (b_s + b_t)/g + u_s - u_t ;  rnd += 0.5 |cost|^2 dt + <cost, db>   (losses/oc.py:737-742; subtracted at :815-818)
      float c2 = 0.0f, cdb = 0.0f;
      auto fold_tile = [&](int t, const f32x4& u_t, const f32x4& b_t, const f32x4& b_n) __attribute__((always_inline)) {  // (t: compile-time index of an unrolled loop)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bg = b_t[r] * inv_g;
          const float c = w_s[t][r] + (bg - sgn * u_t[r]);
          c2 = __builtin_fmaf(c, c, c2);
          cdb = __builtin_fmaf(c, db[t][r], cdb);
          if constexpr (EUBO) w_s[t][r] = b_n[r] * inv_g - u_t[r];
          else w_s[t][r] = bg + u_t[r];
        }
      };
      // EUBO: weights of t (cols 6,7 of the row) for the cost, of s (cols 4,5 of the NEXT row) for the next move
      const float wn_t = EUBO ? cf[SDENG_NCOEF + 4] : 0.0f, wn_1mt = EUBO ? cf[SDENG_NCOEF + 5] : 0.0f;
      if constexpr (NT <= 4) {  // all tiles of (u_t, b_t) first, then the sums (the schedule the d <= 64 kernels were tuned with)
        f32x4 u_t[NT], b_t[NT], b_n[NT];
        cmcd_eval<NT, TGT, EUBO>(a, x, k + 1, cf[6], cf[7], lds, bias, ns, lane_k, wn_t, wn_1mt, [&](int t, const f32x4& u, const f32x4& b, const f32x4& bn) __attribute__((always_inline)) {
          u_t[t] = u;
          b_t[t] = b;
          b_n[t] = bn;
        });
#pragma unroll
        for (int t = 0; t < NT; ++t) fold_tile(t, u_t[t], b_t[t], b_n[t]);
      } else {  // d > 64: each tile is folded into the sums as the output layer delivers it (no u[NT], b[NT] arrays: they spilled)
        cmcd_eval<NT, TGT, EUBO>(a, x, k + 1, cf[6], cf[7], lds, bias, ns, lane_k, wn_t, wn_1mt, fold_tile);
      }
      __builtin_amdgcn_s_setprio(0);  // noise and move of the next step at the lowest level (three levels: 6.49 -> 6.39 ms against two)
      c2 = group_sum(c2);
      cdb = group_sum(cdb);
      rnd += sgn * ((0.5f * c2) * dt);
      rnd += sgn * (cdb * sqdt);
      if constexpr (PAR != 0) {
        if (s.xs_out) store_rows<NT>(s.xs_out + static_cast<size_t>(k + 1) * s.B * s.d, trash, row, s.d, live, g, x);
      }
    }
    store_rows<NT>(s.x_out, trash, row, s.d, live, g, x);
    if (live && g == 0) s.rnd_out[row] = rnd;
  }
}

template <int NT, int TGT, bool EUBO, int PAR>
static int launch_cmcd_p(const CmcdArgs& a, int grid, hipStream_t stream) {
  const size_t lds_bytes = static_cast<size_t>(a.s.lr.in_lds ? cmcd_lds_floats(NT, a.s.lr.n_rows) : sd_lds_weight_floats(NT)) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_simulate_cmcd<NT, TGT, EUBO, PAR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((k_simulate_cmcd<NT, TGT, EUBO, PAR>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  return static_cast<int>(hipGetLastError());
}
template <int NT, int TGT, bool EUBO>
static int launch_cmcd_t(const CmcdArgs& a, int grid, hipStream_t stream) {
  if (a.s.noise_in || a.s.xs_out) return launch_cmcd_p<NT, TGT, EUBO, 1>(a, grid, stream);
  return launch_cmcd_p<NT, TGT, EUBO, 0>(a, grid, stream);
}
template <int NT>
static int launch_cmcd(const CmcdArgs& a, int grid, hipStream_t stream) {
  if (a.s.form == SDENG_FORM_CMCD_EUBO) {  // noising loop: mixture / Gaussian targets (the targets that can be sampled from)
    if (a.s.target.kind == SDENG_DIST_GMM_DIAG || a.s.target.kind == SDENG_DIST_GAUSS_DIAG) return launch_cmcd_t<NT, CT_GMM, true>(a, grid, stream);
    return static_cast<int>(hipErrorInvalidValue);
  }
  if (a.s.target.kind == SDENG_DIST_PHI4) return launch_cmcd_t<NT, CT_PHI4, false>(a, grid, stream);
  if (a.s.target.kind != SDENG_DIST_LOGREG) return launch_cmcd_t<NT, CT_GMM, false>(a, grid, stream);
  if constexpr (NT <= 4) return launch_cmcd_t<NT, CT_LOGREG, false>(a, grid, stream);
  return static_cast<int>(hipErrorInvalidValue);
}
