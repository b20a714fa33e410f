// Fused forward + backward of the drift net over M = N * B (time, state) rows: the batched control pass of the training direction
// (SURVEY 8f-1).  Reference: the autograd graph of FourierMLP.forward (models/mlp.py:135-143) under ClippedCtrl (models/reparam.py:33-43)
// that losses/oc.py:83-103 (generative_and_sde_ctrl) builds once per step and loss.backward() walks.  Here the states are constants (the
// trajectory came from the step-loop kernel), so all N * B evaluations are independent rows:
//
//   forward   h0 = W_in x + b_in + e_t ; a0 = gelu(h0) ; h1 = W_1 a0 + b_1 ; a1 = gelu(h1) ; h2 = W_2 a1 + b_2 ; a2 = gelu(h2) ;
//             u = W_out a2 + b_out                                                  (u only decides the clip mask)
//   backward  d_out = cot * [|u| <= clip] ; d2 = (W_out^T d_out) gelu'(h2) ; d1 = (W_2^T d2) gelu'(h1) ; d0 = (W_1^T d1) gelu'(h0)
//             [gx = W_in^T d0 : the gradient w.r.t. the state, for the KL adjoint]
//
// and the parameter gradients are six skinny GEMMs over the rows -- dW_out = d_out^T a2, dW_2 = d2^T a1, dW_1 = d1^T a0, dW_in = d0^T x,
// bias gradients = column sums, time-embedding cotangent = sum_b d0 -- which the host forms from the per-row arrays this kernel writes.
// One wave = 16 rows, same register layout as the step loop: every product is a split-f16 MFMA chain; the transposed products read
// transposed weight images (k_pack_mlp with `transpose`: W_out^T is an input-layer-shaped matrix, W_in^T an output-layer-shaped one).
// Range: a cotangent row is scaled by a per-row power of two to ~2^9 before each transposed product and scaled back after (the backward
// pass is linear in the cotangent) -- gradients of 1e-6 magnitude keep fp32's relative accuracy through the f16 split.
#pragma once
#include "sim_device.hpp"

struct VjpArgs {
  int M, B, d, N;          // rows, rows per time (row r is evaluated at time r / B), dimension, number of times
  const float* x;          // [M, d]
  const float* cot;        // [M, d] cotangent of u; nullptr: forward only (u_out)
  const float* wpack;      // forward image (+ biases + layer scales behind it)
  const float* wpack_t;    // transposed image: [W_out^T | W_2^T | W_1^T | W_in^T] in the slots of [W_in | W_1 | W_2 | W_out]
  const float* temb;       // [N][64]
  float clip_model;
  float *a0, *a1, *a2;     // [M, 64] gelu(h0), gelu(h1), gelu(h2)
  float *d0, *d1, *d2;     // [M, 64] cotangents of h0, h1, h2
  float* dout;             // [M, d]  cotangent of the net's output (after the clip mask)
  float* gx;               // [M, d]  W_in^T d0, or nullptr
  float* u_out;            // [M, d]  the control itself (after the clip), or nullptr
  float* trash;
  int ntiles;
};

// gelu'(v) = Phi(v) + v phi(v), Phi from the same polynomial as gelu_fast (2^P(|v|) = Phi(-|v|))
SD_INLINE float gelu_grad(float v) {
  const float t = __builtin_fabsf(v);
  float r = -1.797168238e-06f;
  r = __builtin_fmaf(r, t, 2.659268830e-05f);
  r = __builtin_fmaf(r, t, -1.231626375e-04f);
  r = __builtin_fmaf(r, t, -2.968774061e-04f);
  r = __builtin_fmaf(r, t, 7.287443150e-03f);
  r = __builtin_fmaf(r, t, -5.266828835e-02f);
  r = __builtin_fmaf(r, t, -4.591407180e-01f);
  r = __builtin_fmaf(r, t, -1.151116490e+00f);
  r = __builtin_fmaf(r, t, -9.999995232e-01f);
  const float pt = __builtin_amdgcn_exp2f(r);                     // Phi(-|v|)
  const float Phi = v >= 0.0f ? 1.0f - pt : pt;
  const float phi = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * (v * v));  // exp(-v^2/2) / sqrt(2 pi)
  return __builtin_fmaf(v, phi, Phi);
}

// per-row power of two that brings the row's largest |entry| to [2^8, 2^9) -- up or down; 1 for an all-zero or non-finite row
template <int T>
SD_INLINE float row_normalise(const f32x4 (&v)[T]) {
  float m = 0.0f;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, __builtin_fabsf(v[t][r]));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (!(m > 0.0f) || !(m <= 3.0e38f)) return 1.0f;
  int ex;
  frexpf(m, &ex);
  int e = 9 - ex;
  e = e > 120 ? 120 : (e < -120 ? -120 : e);
  return ldexpf(1.0f, e);
}

SD_INLINE void store_h(float* dst, float* trash, uint32_t row, bool live, int g, const f32x4 (&v)[SD_HT]) {
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) *reinterpret_cast<f32x4*>(live ? dst + static_cast<size_t>(row) * SD_H + 16 * t + 4 * g : trash) = v[t];
}

// One 16-row tile: forward, cotangent, backward.  `cot_of(t, u_t)` gives the cotangent of output tile t (u_t = the clipped control of that
// tile when `need_u`, else unspecified): read from memory (k_ctrl_vjp) or formed from the adjoint state (k_kl_adjoint).  `have_cot` false:
// forward only.  The per-row arrays are written at `row`; with GX the state gradient W_in^T d0 is returned in `gx`.
template <int NT, bool GX, class CotFn>
SD_INLINE void vjp_tile(const VjpArgs& a, const float* lds, const float* lds_t, const float* bias, const NetScale& ns, float* trash, uint32_t row,
                        bool live, const float* te, int lane, const f32x4 (&x)[NT], bool have_cot, bool need_u, CotFn&& cot_of, f32x4 (&gx)[NT]) {
  const int g = lane >> 4;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 h0[SD_HT], h1[SD_HT], h2[SD_HT], act[SD_HT];
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) h0[t] = load_tile4(bias, t, g);
  dense<NT, SD_HT>(x, h0, lds + sd_off_win(NT), lane);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) {
    const f32x4 e = load_tile4(te, t, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h0[t][r] = __builtin_fmaf(h0[t][r], ns.inv_in, e[r]);
      act[t][r] = gelu_fast(h0[t][r]);
    }
    h1[t] = load_tile4(bias + 64, t, g);
  }
  if (a.a0) store_h(a.a0, trash, row, live, g, act);
  dense<SD_HT, SD_HT>(act, h1, lds + sd_off_wh1(NT), lane);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) {
    h1[t] = h1[t] * ns.inv_h1;
#pragma unroll
    for (int r = 0; r < 4; ++r) act[t][r] = gelu_fast(h1[t][r]);
    h2[t] = load_tile4(bias + 128, t, g);
  }
  if (a.a1) store_h(a.a1, trash, row, live, g, act);
  dense<SD_HT, SD_HT>(act, h2, lds + sd_off_wh2(NT), lane);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) {
    h2[t] = h2[t] * ns.inv_h2;
#pragma unroll
    for (int r = 0; r < 4; ++r) act[t][r] = gelu_fast(h2[t][r]);
  }
  if (a.a2) store_h(a.a2, trash, row, live, g, act);
  __builtin_amdgcn_sched_barrier(0);
  // ---- output layer: the clip mask of ClippedCtrl (torch.clip passes the gradient where min <= u <= max; NaN: nowhere), and u itself ----
  f32x4 dl[NT];
  {
    const HidSplit hs = split_hidden(act);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 u[1] = {zero};
      bool pass[4] = {true, true, true, true};
      if (a.clip_model > 0.0f || need_u) {
        mlp_out_tiles<NT, 1>(hs, a.wpack, bias, t, lane, u, ns.inv_out);  // A operands through L2
        if (a.clip_model > 0.0f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pass[r] = __builtin_fabsf(u[0][r]) <= a.clip_model;
            u[0][r] = clampf(u[0][r], a.clip_model);
          }
        }
        if (a.u_out) store_quad(a.u_out, trash, row, a.d, live, t, g, u[0]);
      }
      f32x4 c = have_cot ? cot_of(t, u[0]) : zero;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = pass[r] ? c[r] : 0.0f;
      dl[t] = c;
      if (a.dout) store_quad(a.dout, trash, row, a.d, live, t, g, c);
      __builtin_amdgcn_sched_barrier(0);  // one output tile at a time: hoisted, the A operands of all tiles (read through L2) spill
    }
  }
  if (!have_cot) return;  // forward only (wave-uniform)
  __builtin_amdgcn_sched_barrier(0);
  // ---- backward through the three hidden activations ----
  f32x4 dh[SD_HT], gacc[SD_HT];
  {
    const float sg = row_normalise<NT>(dl), back = ns.inv_out / sg;
#pragma unroll
    for (int t = 0; t < SD_HT; ++t) gacc[t] = zero;
    dense<NT, SD_HT, true>(dl, gacc, lds_t + sd_off_win(NT), lane, sg);  // (W_out^T 2^e) (d_out sigma)
#pragma unroll
    for (int t = 0; t < SD_HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[t][r] = (gacc[t][r] * back) * gelu_grad(h2[t][r]);
  }
  store_h(a.d2, trash, row, live, g, dh);
  {
    const float sg = row_normalise<SD_HT>(dh), back = ns.inv_h2 / sg;
#pragma unroll
    for (int t = 0; t < SD_HT; ++t) gacc[t] = zero;
    dense<SD_HT, SD_HT, true>(dh, gacc, lds_t + sd_off_wh1(NT), lane, sg);  // W_2^T
#pragma unroll
    for (int t = 0; t < SD_HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[t][r] = (gacc[t][r] * back) * gelu_grad(h1[t][r]);
  }
  store_h(a.d1, trash, row, live, g, dh);
  {
    const float sg = row_normalise<SD_HT>(dh), back = ns.inv_h1 / sg;
#pragma unroll
    for (int t = 0; t < SD_HT; ++t) gacc[t] = zero;
    dense<SD_HT, SD_HT, true>(dh, gacc, lds_t + sd_off_wh2(NT), lane, sg);  // W_1^T
#pragma unroll
    for (int t = 0; t < SD_HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[t][r] = (gacc[t][r] * back) * gelu_grad(h0[t][r]);
  }
  store_h(a.d0, trash, row, live, g, dh);
  if constexpr (GX) {  // gradient w.r.t. the state: W_in^T d0, an output-layer-shaped product (A operands through L2)
    const float sg = row_normalise<SD_HT>(dh), back = ns.inv_in / sg;
    f32x4 sc[SD_HT];
#pragma unroll
    for (int t = 0; t < SD_HT; ++t) sc[t] = dh[t] * sg;
    const HidSplit hs = split_hidden(sc);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 o[1] = {zero}, mx[1] = {zero};
      dense_pre<2, 1>(hs.h, hs.l, o, mx, reinterpret_cast<const f16x8*>(a.wpack_t + sd_off_wout(NT) + t * 2 * 512), lane);
      fold_lo<1>(o, mx);
      gx[t] = o[0] * back;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// LDS of both kernels below: forward W_in, W_1, W_2, then the transposed W_out^T, W_2^T, W_1^T (same three slots of the transposed image);
// the forward W_out (clip mask, u) and W_in^T (state gradient only) are read through L2
SD_INLINE void vjp_stage_weights(const VjpArgs& a, float* lds, int NF, int tid) {
  const f32x4* s0 = reinterpret_cast<const f32x4*>(a.wpack);
  const f32x4* s1 = reinterpret_cast<const f32x4*>(a.wpack_t);
  f32x4* dst = reinterpret_cast<f32x4*>(lds);
  for (int i = tid; i < NF / 4; i += SD_THREADS) {
    dst[i] = s0[i];
    dst[NF / 4 + i] = s1[i];
  }
  __syncthreads();
}

template <int NT, bool GX>
__global__ void __launch_bounds__(SD_THREADS, SD_WAVES / 4) k_ctrl_vjp(const VjpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NF = sd_off_wout(NT);
  vjp_stage_weights(a, lds, NF, tid);
  const float* lds_t = lds + NF;
  const float* bias = a.wpack + sd_off_bias(NT);
  const NetScale ns = load_net_scale(bias, NT);
  const int p = lane & 15, g = lane >> 4;
  float* trash = a.trash + tid * 4;
  for (int tile = blockIdx.x + gridDim.x * wave; tile < a.ntiles; tile += gridDim.x * SD_WAVES) {
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = row < static_cast<uint32_t>(a.M);
    const int kt = live ? static_cast<int>(row / static_cast<uint32_t>(a.B)) : 0;
    f32x4 x[NT], gx[NT];
    load_rows<NT>(a.x, row, a.d, live, g, x);
    vjp_tile<NT, GX>(a, lds, lds_t, bias, ns, trash, row, live, a.temb + static_cast<size_t>(kt) * SD_H, lane, x, a.cot != nullptr, a.u_out != nullptr,
                     [&](int t, const f32x4&) __attribute__((always_inline)) { return load_quad(a.cot, row, a.d, live, t, g); }, gx);
    if constexpr (GX) {
      if (a.cot) {
#pragma unroll
        for (int t = 0; t < NT; ++t) store_quad(a.gx, trash, row, a.d, live, t, g, gx[t]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// KL training: the discrete adjoint of the step loop in ONE launch (BaseOCLoss.compute_loss kl branch, losses/oc.py:105-131, back-
// propagated through simulate(): :258-284 EM, :478-502 EI, :1361-1390 DDS).  The states x_k came from the step-loop kernel and are
// constants; a wave owns 16 particles and walks k = N-1 .. 0 with the adjoint state lambda in registers:
//     cot_k  = alpha_k lambda + w_b (beta_k u_k + gamma_k z_k)           cotangent of the control of step k (u_k recomputed here)
//     lambda = A_k lambda + C_k H_ref(x_k) lambda + J_u(x_k)^T cot_k      H_ref = Jacobian of the reference score (symmetric), J_u^T via vjp_tile
// LIN (x' = c1 x + c2 (ref + u) + c3 z, rnd += c4 |u|^2 + c5 <u,z>): alpha = c2, beta = 2 c4, gamma = c5, A = c1, C = c2;
// EM  (x' = x + (c1 x + c3 ref + c2 u) c4 + c2 c5 z, rnd += 0.5 |u|^2 c4 + c5 <u,z>): alpha = c2 c4, beta = c4, gamma = c5, A = 1 + c4 c1, C = c4 c3.
// The per-row arrays of vjp_tile (row = k B + b) carry the parameter gradients exactly as in log-variance training.
// ------------------------------------------------------------------------------------------------------------------------------------
struct AdjArgs {
  VjpArgs v;              // x = the states x_0 .. x_{N-1} as [N * B, d] rows; cot / gx / u_out unused
  const float* coef;      // [N][SDENG_NCOEF]
  const float* noise;     // [N][B][d] normals of the trajectory, or nullptr (no Ito term)
  const float* w;         // [B] d loss / d rnd_b
  const float* lam_in;    // [B, d] lambda_N
  float* lam_out;         // [B, d] lambda_0, or nullptr
  const float* ref_tab;   // [N][K][2][dpad] (mean, 1/var) of the noised reference, or nullptr (no reference drift)
  const float* ref_consts;  // [N][K][2]
  int ref_k;
  float ref_c1;
  int lin;                // 1: FORM_LIN, 0: FORM_EM
  int ntiles_b;           // ceil(B / 16)
  // ScoreCtrl (models/reparam.py:63-117): u = clip(net) + scale clip(score_pi(x)) s_theta(t) on a diagonal mixture target
  int has_score;          // ADJ_NONE: ClippedCtrl; ADJ_GMM / ADJ_PHI4: ScoreCtrl on that target
  const float* stheta;    // [N] clipped s_theta(t_k), or nullptr (no score model: 1)
  DistDev target;         // SDENG_DIST_GMM_DIAG tables (k_dist_tables), or the PHI4 constants
  float scale_score, clip_score;
  int ctrl_kind;          // SDENG_CTRL_SCORE / _LERP / _CANCEL_DRIFT (per-step gains: coef cols 7, 8 as in the step loop, sim_kernel.hpp ctrl_score_term)
  float prior_loc, prior_scale;  // LerpCtrl: the IsotropicGauss prior whose score is interpolated with the target's
  int score_detached;     // detach_score: the target score is a constant of x (no Hessian term in the state gradient)
  float* dst;             // [N * B] <cot, scale clip(score)>: the cotangent of s_theta(t_k), per particle
  const float* score_ext; // ADJ_EXT: [N * B, d] target score of every row
};

// H(x) lam for a diagonal Gaussian mixture (K = 1: a Gaussian): with q_k = (x - m_k)/var_k, p = softmax of the component logits, qbar = sum p_k q_k,
//   H lam = sum_k p_k ( -lam/var_k + q_k <q_k, lam> ) - qbar <qbar, lam>          (the Hessian of log sum_k w_k N(x; m_k, var_k))
// one pass over the table with the online softmax of gmm_score_accum.
template <int NT>
SD_INLINE void gmm_hvp(const f32x4 (&x)[NT], const float* __restrict__ tab, const float* __restrict__ consts, int cstride, int K, float c1, int g,
                       const f32x4 (&lam)[NT], f32x4 (&out)[NT]) {
  constexpr int dpad = 16 * NT;
  f32x4 T[NT], Q[NT];
  float m_run = -INFINITY, l_run = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) T[t] = Q[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int k = 0; k < K; ++k) {
    const float* mp = tab + static_cast<size_t>(k) * 2 * dpad;
    f32x4 q[NT], il[NT];
    float part = 0.0f, s = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 m = load_tile4(mp, t, g);
      const f32x4 iv = load_tile4(mp + dpad, t, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dl = x[t][r] - m[r];
        q[t][r] = dl * iv[r];
        il[t][r] = iv[r] * lam[t][r];
        part = __builtin_fmaf(dl, q[t][r], part);
        s = __builtin_fmaf(q[t][r], lam[t][r], s);
      }
    }
    part = group_sum(part);
    s = group_sum(s);
    const float lp = consts[k * cstride + 1] + (((-0.5f * part) - c1) - consts[k * cstride + 0]);
    const float m_new = fmaxf(m_run, lp);
    const float so = exp_nonpos(m_run - m_new);
    const float pk = exp_nonpos(lp - m_new);
    l_run = l_run * so + pk;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T[t][r] = __builtin_fmaf(pk, __builtin_fmaf(q[t][r], s, -il[t][r]), T[t][r] * so);
        Q[t][r] = __builtin_fmaf(pk, q[t][r], Q[t][r] * so);
      }
  }
  const float inv = 1.0f / l_run;
  float qs = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Q[t][r] *= inv;
      qs = __builtin_fmaf(Q[t][r], lam[t][r], qs);
    }
  qs = group_sum(qs);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[t][r] = __builtin_fmaf(-Q[t][r], qs, T[t][r] * inv);
}

// H(x) v for the phi^4 lattice score s_i = A (x_i^3 - x_i) + K0 + C (2 x_i - x_{i+1} - x_{i-1}) (phi4_score, distr/phi_four.py:81-96):
// (A (3 x_i^2 - 1) + 2 C) v_i - C (v_{i+1} + v_{i-1}), Dirichlet-0 neighbours (v is 0 on pad features)
template <int NT>
SD_INLINE void phi4_hvp(const f32x4 (&x)[NT], const DistDev& ds, int d, int g, int lane, const f32x4 (&v)[NT], f32x4 (&out)[NT]) {
  const float coef = ds.p0 * static_cast<float>(d);
  const float A = -ds.p2 / coef, C = -ds.p2 * coef;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float le, re;
    phi4_edges<NT>(v, t, g, lane, le, re);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xv = x[t][r], vv = v[t][r];
      const float vl = (r == 0) ? le : v[t][r > 0 ? r - 1 : 0];
      const float vr = (r == 3) ? re : v[t][r < 3 ? r + 1 : 3];
      const float diag = __builtin_fmaf(A, __builtin_fmaf(3.0f * xv, xv, -1.0f), 2.0f * C);
      const float hv = __builtin_fmaf(diag, vv, -C * (vl + vr));
      out[t][r] = feat_live<NT>(t, r, 4 * g, d) ? hv : 0.0f;
    }
  }
}

// target of the score control the adjoint differentiates (0: ClippedCtrl).  ADJ_EXT: the target score of every row is an INPUT (one
// sdeng_dist_eval launch over all rows) and a constant of x -- targets whose score the reference makes with autograd and without a graph
// (Distribution.score, distr/base.py:146-154: logistic regression), so that backward() sees no Hessian term either
enum { ADJ_NONE = 0, ADJ_GMM = 1, ADJ_PHI4 = 2, ADJ_EXT = 3 };
template <int NT, int SCORE>
__global__ void __launch_bounds__(SD_THREADS, SD_WAVES / 4) k_kl_adjoint(const AdjArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const VjpArgs& v = a.v;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NF = sd_off_wout(NT);
  vjp_stage_weights(v, lds, NF, tid);
  const float* lds_t = lds + NF;
  const float* bias = v.wpack + sd_off_bias(NT);
  const NetScale ns = load_net_scale(bias, NT);
  const int p = lane & 15, g = lane >> 4;
  float* trash = v.trash + tid * 4;
  constexpr int dpad = 16 * NT;
  for (int tile = blockIdx.x + gridDim.x * wave; tile < a.ntiles_b; tile += gridDim.x * SD_WAVES) {
    const uint32_t b = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = b < static_cast<uint32_t>(v.B);
    const float wb = live ? a.w[b] : 0.0f;
    f32x4 lam[NT];
    load_rows<NT>(a.lam_in, b, v.d, live, g, lam);
    for (int k = v.N - 1; k >= 0; --k) {
      const float* cf = a.coef + static_cast<size_t>(k) * SDENG_NCOEF;
      const float c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5];
      const float alpha = a.lin ? c2 : c2 * c4, beta = wb * (a.lin ? 2.0f * c4 : c4), gamma = a.noise ? wb * c5 : 0.0f;
      const float A = a.lin ? c1 : __builtin_fmaf(c4, c1, 1.0f), C = a.lin ? c2 : c4 * c3;
      const float lerp_w = cf[8];  // LerpCtrl: t/T; CancelDriftCtrl: drift gain
      const float inv_pvar = SCORE != ADJ_NONE ? 1.0f / (a.prior_scale * a.prior_scale) : 0.0f;
      const uint32_t row = static_cast<uint32_t>(k) * static_cast<uint32_t>(v.B) + b;
      f32x4 x[NT], gx[NT], jl[NT];
      load_rows<NT>(v.x, row, v.d, live, g, x);
      if (a.ref_tab) {
        asm volatile("" ::: "memory");
        gmm_hvp<NT>(x, a.ref_tab + static_cast<size_t>(k) * a.ref_k * 2 * dpad, a.ref_consts + static_cast<size_t>(k) * a.ref_k * 2, 2, a.ref_k, a.ref_c1, g,
                    lam, jl);
      }
      f32x4 sr[SCORE != ADJ_NONE ? NT : 1], gfull[SCORE != ADJ_NONE ? NT : 1];  // ScoreCtrl: raw target score, and the cotangent of the whole control (before the net's clip mask)
      float gain_st = 0.0f;
      if constexpr (SCORE != ADJ_NONE) {
        asm volatile("" ::: "memory");
        if constexpr (SCORE == ADJ_GMM) gmm_score<NT>(x, a.target.tab, a.target.consts, 4, a.target.k, a.target.p0, g, sr);
        else if constexpr (SCORE == ADJ_PHI4) phi4_score<NT>(x, a.target, v.d, g, lane, sr);
        else {
#pragma unroll
          for (int t = 0; t < NT; ++t) sr[t] = load_quad(a.score_ext, row, v.d, live, t, g);
        }
        // LerpCtrl (models/reparam.py:166-199): the score that is clipped and scaled is lerp(score_prior, score_pi, t/T), times g(t);
        // CancelDriftCtrl (:120-145): + drift(t,x)/g(t) (linear in x, gain cf[8]) and the score part times g(t)/2 (cf[7])
        if (a.ctrl_kind == SDENG_CTRL_LERP) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float ps = feat_live<NT>(t, r, 4 * g, v.d) ? (a.prior_loc - x[t][r]) * inv_pvar : 0.0f;
              sr[t][r] = __builtin_fmaf(lerp_w, sr[t][r] - ps, ps);
            }
        }
        gain_st = a.scale_score * (a.stheta ? a.stheta[k] : 1.0f) * (a.ctrl_kind == SDENG_CTRL_SCORE ? 1.0f : cf[7]);
      }
      vjp_tile<NT, true>(v, lds, lds_t, bias, ns, trash, row, live, v.temb + static_cast<size_t>(k) * SD_H, lane, x, true, true,
                         [&](int t, const f32x4& u) __attribute__((always_inline)) {
                           f32x4 c;
                           const f32x4 z = a.noise ? load_quad(a.noise, row, v.d, live, t, g) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                           for (int r = 0; r < 4; ++r) {
                             float uv = u[r];
                             if constexpr (SCORE != ADJ_NONE) {
                               uv = __builtin_fmaf(gain_st, a.clip_score > 0.0f ? clampf(sr[t][r], a.clip_score) : sr[t][r], uv);
                               if (a.ctrl_kind == SDENG_CTRL_CANCEL_DRIFT && feat_live<NT>(t, r, 4 * g, v.d)) uv = __builtin_fmaf(lerp_w, x[t][r], uv);
                             }
                             c[r] = __builtin_fmaf(alpha, lam[t][r], __builtin_fmaf(beta, uv, gamma * z[r]));
                           }
                           if constexpr (SCORE != ADJ_NONE) gfull[t] = c;
                           return c;
                         },
                         gx);
      if constexpr (SCORE != ADJ_NONE) {
        // d u / d s_theta = scale clip(score): its cotangent, one number per (step, particle); d u / d x through the score: scale s_theta H_pi,
        // under the clip's mask (torch.clip passes the gradient where |score| <= clip_score)
        float ds = 0.0f;
        f32x4 gm[NT], hv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool pass = !(a.clip_score > 0.0f) || __builtin_fabsf(sr[t][r]) <= a.clip_score;
            ds = __builtin_fmaf(gfull[t][r], a.clip_score > 0.0f ? clampf(sr[t][r], a.clip_score) : sr[t][r], ds);
            gm[t][r] = pass ? gfull[t][r] : 0.0f;
          }
        ds = group_sum(ds) * (a.scale_score * (a.ctrl_kind == SDENG_CTRL_SCORE ? 1.0f : cf[7]));
        if (live && g == 0) a.dst[row] = ds;
        if (SCORE != ADJ_EXT && !a.score_detached) {
          asm volatile("" ::: "memory");
          if constexpr (SCORE == ADJ_GMM) gmm_hvp<NT>(x, a.target.tab, a.target.consts, 4, a.target.k, a.target.p0, g, gm, hv);
          else if constexpr (SCORE == ADJ_PHI4) phi4_hvp<NT>(x, a.target, v.d, g, lane, gm, hv);
          if (a.ctrl_kind == SDENG_CTRL_LERP) {  // d lerp / d x = (1 - w) (-1/var) + w H_pi
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) hv[t][r] = __builtin_fmaf(lerp_w, hv[t][r], (lerp_w - 1.0f) * inv_pvar * gm[t][r]);
          }
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) gx[t][r] = __builtin_fmaf(gain_st, hv[t][r], gx[t][r]);
        }
        if (a.ctrl_kind == SDENG_CTRL_CANCEL_DRIFT) {  // the drift/g term: lerp_w x on the live features
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) gx[t][r] = feat_live<NT>(t, r, 4 * g, v.d) ? __builtin_fmaf(lerp_w, gfull[t][r], gx[t][r]) : gx[t][r];
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float l = __builtin_fmaf(A, lam[t][r], gx[t][r]);
          if (a.ref_tab) l = __builtin_fmaf(C, jl[t][r], l);
          lam[t][r] = l;
        }
    }
    if (a.lam_out) store_rows<NT>(a.lam_out, trash, b, v.d, live, g, lam);
  }
}

template <int NT>
static int launch_ctrl_vjp(const VjpArgs& a, int grid, hipStream_t stream) {
  const size_t lds_bytes = static_cast<size_t>(2 * sd_off_wout(NT)) * sizeof(float);
  hipError_t e;
  if (a.gx) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctrl_vjp<NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL((k_ctrl_vjp<NT, true>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctrl_vjp<NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL((k_ctrl_vjp<NT, false>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  }
  return static_cast<int>(hipGetLastError());
}
template <int NT, int SCORE>
static int launch_kl_adjoint_s(const AdjArgs& a, int grid, hipStream_t stream) {
  const size_t lds_bytes = static_cast<size_t>(2 * sd_off_wout(NT)) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kl_adjoint<NT, SCORE>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((k_kl_adjoint<NT, SCORE>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  return static_cast<int>(hipGetLastError());
}
template <int NT>
static int launch_kl_adjoint(const AdjArgs& a, int grid, hipStream_t stream) {
  if (a.has_score == ADJ_GMM) return launch_kl_adjoint_s<NT, ADJ_GMM>(a, grid, stream);
  if (a.has_score == ADJ_PHI4) return launch_kl_adjoint_s<NT, ADJ_PHI4>(a, grid, stream);
  if (a.has_score == ADJ_EXT) return launch_kl_adjoint_s<NT, ADJ_EXT>(a, grid, stream);
  return launch_kl_adjoint_s<NT, ADJ_NONE>(a, grid, stream);
}
#define SD_DEFINE_VJP(NT) \
  int sd_launch_vjp_##NT(const VjpArgs& a, int grid, hipStream_t s) { return launch_ctrl_vjp<NT>(a, grid, s); } \
  int sd_launch_adjoint_##NT(const AdjArgs& a, int grid, hipStream_t s) { return launch_kl_adjoint<NT>(a, grid, s); }
