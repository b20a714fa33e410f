// The persistent whole-trajectory simulate kernel (FORM_LIN / FORM_EM / FORM_EUBO) for gfx950.
//
// One launch runs all N steps.  A wave keeps its 16 particles' state x[d] in registers for the whole
// trajectory; HBM sees x once in and once out ((2d+1)*4 bytes per particle per TRAJECTORY).  Per step a
// wave does: drift net (split-f16 MFMA chain with fp32 accumulation, weights in LDS), optional target score inside the control
// (ScoreCtrl/LerpCtrl), optional reference score (noised Gaussian / mixture; per-wave LDS table refilled by
// LDS-DMA), noise (Philox in-register or injected), the integrator update and the log-RND accumulation
// (two cross-lane adds per reduction).  No barrier inside the step loop: the two waves sharing a SIMD interleave
// freely (matrix and vector issue do not overlap on this part, so instruction count is what is optimised).
#pragma once
#include <type_traits>

#include "sim_device.hpp"

enum { RF_NONE = 0, RF_GAUSS = 1, RF_GMM = 2, RF_GMM_BIG = 3, RF_GMM_FULL = 4, RF_GMM_MM = 5 };  // GMM: K <= SD_KREG (responsibilities in registers); FULL: full covariances; MM: shared-variance mixture on the matrix pipe
enum { SC_NONE = 0, SC_GMM = 1, SC_PHI4 = 2, SC_LOGREG = 3, SC_REFSLOT = 4 };  // LOGREG: no reference table slots, d <= 64 (LDS)
// SC_REFSLOT: the target of the Score / Lerp / CancelDrift control is a FULL-covariance mixture (GMMFull / TwoModesFull, distr/gauss.py:310-520,
// score_mog_full :110-121).  Its precision images sit in the reference slot (REF = RF_GMM_FULL, tables of the static marginal s = 1,
// sigma^2 = 0 written N times by k_ref_full_tables), the mixture score the slot produces feeds the control, and the SDE has no reference
// drift (PIS / DDS / DIS with a target-informed control): every reference term of the tail compiles out.

// score part of the generative control (added to clip(net)):
//   ScoreCtrl (models/reparam.py:112-117):  scale*clip(score_pi(x)) * s_theta(t)
//   LerpCtrl  (models/reparam.py:166-199):  g(t) * (scale*clip(lerp(score_prior, score_pi, t/T)) * s_theta(t))
//   CancelDriftCtrl (models/reparam.py:120-145):  drift(t,x)/g(t) + g(t)/2 * (scale*clip(score_pi(x)) * s_theta(t))
// KIND and CLIP are compile-time here and chosen ONCE per tile by add_ctrl_score_tile: tested per element, these uniform
// run-time conditions cost a scalar branch each (the step loop of the PIS kernel carried 100+ of them per step).
template <int KIND, bool CLIP>
SD_INLINE float ctrl_score_term(const SimArgs& a, float sv, float xv, float st, float score_gain, float lerp_w, bool in_range,
                                float inv_prior_var) {
  constexpr bool lerp = KIND == SDENG_CTRL_LERP;
  if constexpr (lerp) {  // torch.lerp(prior_score, target_score, w); IsotropicGauss.score = (loc - x)/scale^2
    const float ps = (a.prior.p0 - xv) * inv_prior_var;  // reciprocal once per tile, not a division per element
    const float df = sv - ps;
    sv = (lerp_w < 0.5f) ? ps + lerp_w * df : sv - df * (1.0f - lerp_w);
  }
  if constexpr (CLIP) sv = clampf(sv, a.clip_score);
  float v = a.scale_score * sv;
  v = v * st;
  if constexpr (lerp) v = in_range ? score_gain * v : 0.0f;  // only the prior term can be non-zero on a pad feature
  // CancelDriftCtrl (models/reparam.py:131-145): + drift/g + (g/2) score, both gains per step at the net's time.  Pad features
  // carry Philox noise in their state (nothing else reads it), so the drift term must not leak it into the control
  if constexpr (KIND == SDENG_CTRL_CANCEL_DRIFT) v = in_range ? lerp_w * xv + score_gain * v : 0.0f;
  return v;
}
template <int KIND, bool CLIP>
SD_INLINE void add_ctrl_score_tile_k(const SimArgs& a, f32x4& u, const f32x4& sv, const f32x4& xv, float st, float score_gain, float lerp_w,
                                     int t, int g4, int d) {
  if constexpr (KIND == SDENG_CTRL_SCORE) {
    // ScoreCtrl (models/reparam.py:112-117): u += scale * clip(score) * s_theta.  The clip almost never binds (1e4): tile-level test,
    // 4-instruction clamp only when some lane is out of range; the two gains are one per-step scalar, the add is fused -- one
    // instruction per element instead of seven (one rounding instead of three: within an ulp of the reference's separate ops).
    f32x4 s = sv;
    if constexpr (CLIP) clamp_tile_rare(s, a.clip_score);
    const float gain = a.scale_score * st;
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = __builtin_fmaf(gain, s[r], u[r]);
    return;
  }
  const float inv_prior_var = KIND == SDENG_CTRL_LERP ? 1.0f / (a.prior.p1 * a.prior.p1) : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    u[r] = u[r] + ctrl_score_term<KIND, CLIP>(a, sv[r], xv[r], st, score_gain, lerp_w, feat_lt(t, r, g4, d), inv_prior_var);
}
// u += score part of the control for one feature tile (one uniform dispatch per tile)
SD_INLINE void add_ctrl_score_tile(const SimArgs& a, f32x4& u, const f32x4& sv, const f32x4& xv, float st, float score_gain, float lerp_w,
                                   int t, int g4, int d) {
  const bool clip = a.clip_score > 0.0f;
  if (a.ctrl_kind == SDENG_CTRL_LERP) {
    if (clip) add_ctrl_score_tile_k<SDENG_CTRL_LERP, true>(a, u, sv, xv, st, score_gain, lerp_w, t, g4, d);
    else add_ctrl_score_tile_k<SDENG_CTRL_LERP, false>(a, u, sv, xv, st, score_gain, lerp_w, t, g4, d);
  } else if (a.ctrl_kind == SDENG_CTRL_CANCEL_DRIFT) {
    if (clip) add_ctrl_score_tile_k<SDENG_CTRL_CANCEL_DRIFT, true>(a, u, sv, xv, st, score_gain, lerp_w, t, g4, d);
    else add_ctrl_score_tile_k<SDENG_CTRL_CANCEL_DRIFT, false>(a, u, sv, xv, st, score_gain, lerp_w, t, g4, d);
  } else {
    if (clip) add_ctrl_score_tile_k<SDENG_CTRL_SCORE, true>(a, u, sv, xv, st, score_gain, lerp_w, t, g4, d);
    else add_ctrl_score_tile_k<SDENG_CTRL_SCORE, false>(a, u, sv, xv, st, score_gain, lerp_w, t, g4, d);
  }
}

// Waves per workgroup of an instantiation (one persistent workgroup per CU).  Two per SIMD (256-register budget) by default; THREE per
// SIMD where the step loop fits the 168-register budget of three waves without scratch: the third wave fills issue slots the other
// two leave while they wait (PIS phi^4, 131 072 x 512, d = 100: 18.9 -> 16.1 ms before the matrix-phase priority below existed, 16.1 -> 15.6
// with it).  The list is the plain sampling kernels (PAR = 0, forward
// forms) whose register need the build reports at or below 168 (csrc/obj/kernel_resources.txt); kernels with a workgroup-shared
// table (RF_GMM_BIG / _FULL / _MM) stage it with SD_WAVES waves and stay there.
template <int NT, int REF, int SC, int FORM, int PAR>
constexpr int sd_waves_of() {
#if SD_WAVES != 8 || defined(SD_NO_W12)
  return SD_WAVES;  // occupancy experiments (build.py SDENG_WAVES; SD_NO_W12: two waves per SIMD everywhere)
#else
  if (PAR != 0 || FORM == SDENG_FORM_EUBO) return SD_WAVES;
  if (REF == RF_NONE && (SC == SC_NONE || SC == SC_PHI4)) return SD_WAVES_MAX;
  // (kernels with per-wave LDS reference tables stay at two waves per SIMD: with the matrix-phase priority in place, cfg 2's model
  // runs 3.26 ms at d = 64 with 8 waves against 3.63-3.77 with 12, 4.00 against 4.63 at d = 96, 4.8 against 5.8 at d = 128)
  return SD_WAVES;
#endif
}

// PAR = 1 adds the parity-mode paths (injected noise, trajectory dump); PAR = 0 keeps them out of the step loop.
template <int NT, int REF, int SC, int FORM, int PAR>
__global__ void __launch_bounds__((64 * sd_waves_of<NT, REF, SC, FORM, PAR>()), (sd_waves_of<NT, REF, SC, FORM, PAR>() / 4)) k_simulate(const SimArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int dpad = 16 * NT;
  constexpr int W = sd_waves_of<NT, REF, SC, FORM, PAR>(), THREADS = 64 * W;
  static_assert(W == SD_WAVES || !(REF == RF_GMM_BIG || REF == RF_GMM_FULL || REF == RF_GMM_MM), "shared-table staging is laid out for SD_WAVES waves");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {  // packed drift-net weights -> LDS, once per workgroup
    const f32x4* src = reinterpret_cast<const f32x4*>(a.wpack);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    const int n4 = sd_lds_weight_floats(NT) / 4;
    for (int i = tid; i < n4; i += THREADS) dst[i] = src[i];
    if constexpr (SC == SC_LOGREG) {  // the two design-matrix images behind the weights (when they fit: else they stay in L2)
      const int ni = a.lr.in_lds ? sd_lr_floats(NT, a.lr.n_rows) / 4 : 0;
      for (int i = tid; i < ni; i += THREADS) dst[n4 + i] = reinterpret_cast<const f32x4*>(a.lr.image)[i];
    }
  }
  __syncthreads();
#ifdef SD_STAGGER  // A/B builds only: the second wave of each SIMD starts SD_STAGGER x 64 x 127 cycles late (~half a step at 2)
  if constexpr (REF != RF_GMM_BIG && REF != RF_GMM_FULL && REF != RF_GMM_MM)  // (the shared-table kernels meet at barriers every step)
    if (wave >= W / 2)
      for (int i = 0; i < SD_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
#ifdef SD_DBG_PRIO_HI47  // DESIGN 4a experiments: which wave of a SIMD is the victim of the packed-fp32 corruption?
  if (wave >= 4) __builtin_amdgcn_s_setprio(3);
#endif
#ifdef SD_DBG_ONLY47
  if (wave < 4) return;
#endif
#ifdef SD_DBG_ONLY03
  if (wave >= 4) return;
#endif
  static_assert(SC != SC_LOGREG || (REF == RF_NONE && NT <= 4), "in-loop logistic-regression score: no reference, d <= 64");
  static_assert(SC != SC_REFSLOT || (REF == RF_GMM_FULL && FORM != SDENG_FORM_EUBO), "full-covariance target score: lives in the reference slot, forward forms");
  constexpr bool has_ref = REF != RF_NONE && SC != SC_REFSLOT;  // is there a reference DRIFT (the slot may hold the control's target instead)
  const float* bias = a.wpack + sd_off_bias(NT);
  const NetScale ns = load_net_scale(bias, NT);
#ifdef SD_NO_RANGE_GUARD  // A/B builds only (tools/variant_lib.sh): the step loop without the range guard of DESIGN 4b
  const bool scaled_net = false;
#else
  // Only scaled INPUT / HIDDEN layers send a net through the twin; a scaled OUTPUT layer -- the state of every freshly initialised
  // make_model, |w| <= 1.25e-7 -- is un-scaled where the plain output tiles are produced (one uniform test per tile group): before this
  // such a net ran plain pass + twin every step (cfg 2 4.8 -> 7.8 ms, cfg 3 15.6 -> 26.8 ms, profiles/r03_scaled_net.log).
  const bool scaled_net = ns.any_hidden();
#endif
  const float inv_out_plain = ns.inv_out;
  // this wave's private copy of the current step's reference table (Gaussian / small-mixture references)
#ifdef SD_DBG_NODMA
  constexpr bool ref_lds = false;
#else
  constexpr bool ref_lds = (REF == RF_GAUSS || REF == RF_GMM);
#endif
  float* my_tab = lds + sd_lds_weight_floats(NT) + wave * SD_REFTAB_FLOATS;
  const int tab_floats = a.ref_k * 2 * dpad;
  // larger mixtures: the workgroup shares one LDS copy of the step's table, double-buffered behind the weights
  // (one LDS-DMA copy per WORKGROUP instead of every wave streaming the table from L2).  Its 8 waves then
  // walk rounds and steps in lock-step -- one barrier per table piece, reached by every wave: a wave whose tile index is
  // past the end computes on zeros with all stores masked, so trip counts are uniform by construction.
  const int share = (REF == RF_GMM_BIG || REF == RF_GMM_FULL || REF == RF_GMM_MM) ? __builtin_amdgcn_readfirstlane(a.ref_share) : 0;
  float* sh_tab = lds + sd_lds_weight_floats(NT);
  const int sh_floats = sd_share_buf_floats(share);
  // full-covariance mixtures: a component's precision image (NT*KB*512 floats) is staged in FULL_PP pieces of FULL_TO output
  // tiles each (32 KiB at d = 128), piece after piece through the same two buffers
  constexpr int FULL_PP = NT > 4 ? 2 : 1, FULL_TO = NT / FULL_PP, FULL_PIECE = FULL_TO * ((NT + 1) / 2) * 512;
  static_assert(REF != RF_GMM_FULL || NT % FULL_PP == 0, "pieces hold whole output tiles");
  const int p = lane & 15, g = lane >> 4;
  constexpr bool lin = FORM == SDENG_FORM_LIN;
  constexpr bool eubo = FORM == SDENG_FORM_EUBO;  // noising direction (compute_eubo)
  static_assert(!eubo || (SC == SC_NONE) != (REF == RF_NONE), "EUBO kernels: reference-SDE losses with a ClippedCtrl, or DIS (no reference)");
  const bool full_d = a.d == dpad;
  float* trash = a.trash + tid * 4;
  bool same_var = false;
  if constexpr (REF == RF_GMM) same_var = a.N > 0 && a.ref_same_var[0] != 0.0f;  // written by k_same_var ahead of the table kernel
  const bool kfull = REF == RF_GMM && a.ref_k == SD_KREG;  // all SD_KREG components present: the unguarded instantiations
  const bool kfast = kfull && same_var;

  // CUs first; shared-table mode: the round's wave-0 tile decides, so all 8 waves run the same rounds
  for (int tile = blockIdx.x + gridDim.x * wave; (share ? tile - static_cast<int>(gridDim.x) * wave : tile) < a.ntiles;
       tile += gridDim.x * W) {
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = row < static_cast<uint32_t>(a.B);
    const uint32_t pidx = static_cast<uint32_t>(a.particle0 + row);
    f32x4 x[NT];
    load_rows<NT>(a.x_in, row, a.d, live, g, x);  // x0: the caller's, or the engine's own draw (k_sample_x0 ahead of this launch)
    // rnd0 = log p_prior(x0) when the loss asks for it (losses/oc.py:695-699, 935-939), from k_dist_eval
    float rnd = 0.0f;
    if (a.rnd_init) rnd = (live ? a.rnd_init[row] : 0.0f);
    if constexpr (PAR) {
      if (a.xs_out) store_rows<NT>(a.xs_out, trash, row, a.d, live, g, x);
    }
    if constexpr (ref_lds) {
      if (a.N > 0) dma_table_to_lds(a.ref_tab, my_tab, tab_floats, lane);
    }
    if constexpr (REF == RF_GMM_BIG) {
      if (share) {  // the previous round's last reads of buffer 0 are done once every wave is here
        __syncthreads();
        if (a.N > 0) dma_table_shared(a.ref_tab, sh_tab, min(a.ref_kc, a.ref_k) * 2 * dpad, share, wave, lane);
      }
    }
    if constexpr (REF == RF_GMM_MM) {  // always workgroup-shared: piece 0 = the logit image of step 0
      __syncthreads();
      if (a.N > 0) dma_table_shared(a.ref_tab, sh_tab, a.ref_kc * ((NT + 1) / 2) * 512, share, wave, lane);
    }
    if constexpr (REF == RF_GMM_FULL) {  // always workgroup-shared: piece 0 of the first precision image
      __syncthreads();
      if (a.N > 0) dma_table_shared(a.ref_tab, sh_tab, FULL_PIECE, share, wave, lane);
    }

    for (int k = 0; k < a.N; ++k) {  // EUBO: the host lays the rows out in iteration order (times T - s run backwards)
      const float* cf = a.coef + static_cast<size_t>(k) * SDENG_NCOEF;
      const float c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5], c6 = cf[6];
      const float score_gain = cf[7], lerp_w = cf[8];
      const float e1 = __builtin_fmaf(c4, c1, 1.0f), e2 = c4 * c2, e3 = c4 * c3, e4 = c2 * c5;  // EM update, see the tail
      // `d` re-read through an opaque move every step: keeps the pad masks (f < d) from being hoisted out of
      // the step loop, where they would occupy SGPR pairs and spill
      int d_dyn = a.d;
      asm volatile("" : "+s"(d_dyn));

      auto noise_tile = [&](int t) -> f32x4 {
        if (PAR && a.noise_in) return load_quad(a.noise_in + static_cast<size_t>(k) * a.B * a.d, row, d_dyn, live, t, g);
        f32x4 z = philox_normal4(pidx, static_cast<uint32_t>(k), static_cast<uint32_t>(4 * t + g), 0u, a.seed_lo, a.seed_hi);
        // pad features (f >= d) may carry noise: their weights, table entries and outputs are all zero, so
        // they never reach a live feature -- except through the phi^4 lattice's neighbour coupling
        if constexpr (SC == SC_PHI4) {
          if (!full_d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = feat_live<NT>(t, r, 4 * g, d_dyn) ? z[r] : 0.0f;
          }
        }
        return z;
      };
      if constexpr (eubo) {  // noise the samples first (losses/oc.py:335-337, 549-551): x *= mean_factor; x += std_factor * z
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 z = noise_tile(t);
#pragma unroll
          for (int r = 0; r < 4; ++r) x[t][r] = x[t][r] * c1 + c3 * z[r];
        }
      }

      // ---- drift net up to the last hidden activation (split-f16 MFMA chain) ----
      f32x4 hid[SD_HT];
      // The three hidden layers are one long chain of dependent matrix instructions: while a wave is in it, it wins the SIMD's issue
      // arbitration against the other wave(s), whose vector stream (noise, mixture score, tail) would otherwise be interleaved
      // between its MFMAs and stretch the chain.  Measured, same box: cfg 2 5.28 -> 4.70 ms, cfg 3 15.74 -> 14.84 ms; priority 1, 2
      // and 3 are equal; raising it around the output-layer MFMAs of the tail as well is slightly worse (4.73 / 15.09).  Three levels
      // -- hidden layers 2, the whole tail (output layer, noise, integrator) 1, scores and bookkeeping 0 -- are worth another 2.4 %
      // (4.74 -> 4.63, 15.22 -> 14.86): profiles/r02_issue_priority.log.
      __builtin_amdgcn_s_setprio(2);
      // (kernels whose reference / target score puts the state itself through split-f16 products -- matrix-pipe and full-covariance mixtures,
      // the in-loop logistic-regression score -- carry no range-safe twin: those products have none, the guard would be half a guard;
      // scaled weights are un-scaled layer by layer there)
      constexpr bool has_twin = REF != RF_GMM_MM && REF != RF_GMM_FULL && SC != SC_LOGREG;
      if constexpr (has_twin) mlp_hidden<NT>(x, hid, lds, bias, a.temb + static_cast<size_t>(k) * SD_H, lane);
      else mlp_hidden_scaled<NT>(x, hid, lds, bias, a.temb + static_cast<size_t>(k) * SD_H, lane, ns);
      __builtin_amdgcn_s_setprio(1);
      HidSplit hs = split_hidden(hid);
      // The range-safe twin of the net (sim_device.hpp mlp_hidden_safe) takes over -- at the first output tile of the step, below --
      // when an f16 operand overflowed or when a weight matrix is stored with a power-of-two scale (wave-uniform, fixed for the
      // launch: e.g. the reference's default initialisation; the plain pass above is then wasted work, for as long as the net sits
      // there).  A healthy net with in-range weights pays one compare per step.
      bool safe_net = false;
      float hid_rs = 1.0f;

      // ---- scores at the OLD state: target score inside the control, reference drift ----
      f32x4 ts[(SC != SC_NONE && SC != SC_REFSLOT) ? NT : 1];
      if constexpr (SC == SC_GMM) {
        if (NT == 1 && a.target.kind == SDENG_DIST_RINGS) ts[0] = rings_score(x[0], a.target, g);
        else gmm_score<NT>(x, a.target.tab, a.target.consts, 4, a.target.k, a.target.p0, g, ts);
      }
      if constexpr (SC == SC_PHI4) {
        // (eight feature tiles at three waves per SIMD: the neighbour-lane indices of the lattice score, left loop-invariant, are parked in
        // scratch and reloaded every step; derived from an opaque copy of the lane id they are two integer instructions next to their use)
        int lane_p = lane;
        if constexpr (NT == 8) asm volatile("" : "+v"(lane_p));
        phi4_score<NT>(x, a.target, d_dyn, g, lane_p, ts);
      }
      if constexpr (SC == SC_LOGREG) {
        asm volatile("" ::: "memory");
        f16x8 xh[(NT + 1) / 2], xl[(NT + 1) / 2];
        split_tiles<NT>(x, xh, xl);
        // two copies of the body on purpose: a pointer selected between LDS and global memory would turn every A-operand read into a flat load
        if (a.lr.in_lds) logreg_score<NT>(x, xh, xl, a.lr, d_dyn, lds + sd_lds_weight_floats(NT), lane, ts);
        else logreg_score<NT>(x, xh, xl, a.lr, d_dyn, a.lr.image, lane, ts);
      }
      // reference drift (eq/sdes.py:265-279, 329-345): small mixtures keep only the K responsibilities and
      // assemble the score tile by tile in the tail; larger ones use the online-softmax accumulator
      const float* rtab = ref_lds ? my_tab : a.ref_tab + static_cast<size_t>(k) * tab_floats;
      const float* rcs = a.ref_consts + static_cast<size_t>(k) * a.ref_k * 2;
      if constexpr (ref_lds) wait_dma();
      float resp[REF == RF_GMM ? SD_KREG : 1];
      f32x4 rs[(REF == RF_GMM_BIG || REF == RF_GMM_FULL || REF == RF_GMM_MM) ? NT : 1];
#ifdef SD_DBG_NOREF
      if constexpr (REF == RF_GMM) { resp[0] = 1.0f; resp[1] = resp[2] = resp[3] = 0.0f; }
#else
      if constexpr (REF == RF_GMM) {
        if (kfast) gmm_resp_centred<NT>(x, rtab, rcs, g, resp);  // k_ref_tables wrote the centred table under the same condition
        else if (kfull) gmm_resp<NT, SD_KREG>(x, rtab, rcs, 2, SD_KREG, a.ref_c1, g, resp);
        else gmm_resp<NT>(x, rtab, rcs, 2, a.ref_k, a.ref_c1, g, resp);
      }
#endif
      if constexpr (REF == RF_GMM_BIG) {
        if (share) {
          // The table of step k arrives in nch pieces of <= kc components (one piece when two copies of the whole
          // table fit); piece q = k*nch + c lives in buffer q & 1.  Per piece: my slice of it (issued one piece ago)
          // has landed, and after the barrier so has everyone's; every wave is also past its reads of piece q-1,
          // whose buffer the copy of piece q+1 now overwrites.  The accumulator runs over the components in the
          // same order as the one-piece form, so chunking does not change a bit of the result.
          const int kc = a.ref_kc, nch = (a.ref_k + kc - 1) / kc;
          float m_run, l_run;
          gmm_score_begin<NT>(rs, m_run, l_run);
          for (int c = 0; c < nch; ++c) {
            const int q = k * nch + c;
            wait_dma();
            __syncthreads();
            const int c2 = c + 1 < nch ? c + 1 : 0, k2 = c + 1 < nch ? k : k + 1;
            if (k2 < a.N)
              dma_table_shared(a.ref_tab + static_cast<size_t>(k2) * tab_floats + c2 * kc * 2 * dpad,
                               sh_tab + ((q + 1) & 1) * sh_floats, min(kc, a.ref_k - c2 * kc) * 2 * dpad, share, wave, lane);
            gmm_score_accum<NT>(x, sh_tab + (q & 1) * sh_floats, rcs + c * kc * 2, 2, min(kc, a.ref_k - c * kc), a.ref_c1, g, rs,
                                m_run, l_run);
          }
          gmm_score_end<NT>(rs, l_run);
        } else {
          gmm_score<NT>(x, rtab, rcs, 2, a.ref_k, a.ref_c1, g, rs);
        }
      }
      if constexpr (REF == RF_GMM_MM) {
        // Diagonal mixtures whose components share one variance vector (the reference's default initialisation; then true of every
        // noised marginal), K <= 64, on the MATRIX pipe -- score_mog (distr/gauss.py:97-107) is two GEMMs around a softmax:
        //   logit_k = log w_k - |x - mu_k|^2_iv / 2  =  <x - c, (mu_k - c) iv> + b_k + (terms common to all k, which the softmax drops)
        //   score   = (sum_k p_k mu_k - x) iv        =  (sum_k p_k (mu_k - c) - (x - c)) iv
        // with c = the centre of the noised means (keeps the products at the size of the mixture's SPREAD, so the expanded square
        // loses nothing to an offset; measured against fp64 the form is as accurate as the reference's own fp32 expression,
        // DESIGN 4).  Per step the workgroup stages two split-f16 A-operand images through the double-buffered LDS pieces of the
        // full-covariance path: [components x features] for the logits, [features x components] for the p-weighted mean; the logit
        // tile (rows = components, columns = particles) IS the B operand of the second product.  ~36-96 MFMAs + ~300 vector
        // instructions per tile-step instead of ~4.5 vector instructions per element and component.
        constexpr int KBX = (NT + 1) / 2, KTM = 4;
        const int kt = a.ref_kc, kbk = (kt + 1) / 2;       // live component tiles, K-blocks of the mean product
        const float* cs = a.ref_mean + static_cast<size_t>(k) * 2 * dpad;  // [centre][1/var] of this step
        const float* bk = a.ref_consts + static_cast<size_t>(k) * (KTM * 16);
        const int logit_floats = kt * KBX * 512, mean_floats = NT * kbk * 512;
        const float* img = a.ref_tab + static_cast<size_t>(k) * (logit_floats + mean_floats);
        f16x8 ch[KBX], cl[KBX];
        {
          f32x4 xc[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) xc[t] = x[t] - load_tile4(cs, t, g);
          split_tiles<NT>(xc, ch, cl);
        }
        // ---- piece 2k: logits ----
        wait_dma();
        __syncthreads();
        dma_table_shared(img + logit_floats, sh_tab + sh_floats, mean_floats, share, wave, lane);
        f32x4 lt[KTM];
        {
          const f16x8* im = reinterpret_cast<const f16x8*>(sh_tab);
#pragma unroll
          for (int j = 0; j < KTM; ++j) {
            lt[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (j < kt) {  // wave-uniform
              f32x4 acc = load_tile4(bk, j, g), mx = {0.0f, 0.0f, 0.0f, 0.0f};  // b_k (-inf on pad components)
#pragma unroll
              for (int kb = 0; kb < KBX; ++kb) {
                const f16x8 ah = im[((j * KBX + kb) * 2 + 0) * 64 + lane], al = im[((j * KBX + kb) * 2 + 1) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ch[kb], acc, 0, 0, 0);
                mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, cl[kb], mx, 0, 0, 0);
                mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ch[kb], mx, 0, 0, 0);
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) lt[j][r] = __builtin_fmaf(mx[r], SD_LO_INV, acc[r]);
            }
            __builtin_amdgcn_sched_barrier(0);  // one component tile at a time: hoisted, the 8 KB of A operands per tile spill
          }
        }
        {  // softmax over the components of each particle: registers and tiles of this lane, then the four row groups
          float mxv = -INFINITY;
#pragma unroll
          for (int j = 0; j < KTM; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) mxv = fmaxf(mxv, lt[j][r]);
          mxv = fmaxf(mxv, __shfl_xor(mxv, 16, 64));
          mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
          float den = 0.0f;
#pragma unroll
          for (int j = 0; j < KTM; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              lt[j][r] = (j < kt) ? exp_nonpos(lt[j][r] - mxv) : 0.0f;
              den += lt[j][r];
            }
          den = group_sum(den);
          const float inv = rcp_fast(den);
#pragma unroll
          for (int j = 0; j < KTM; ++j) lt[j] = lt[j] * inv;
        }
        f16x8 ph[2], pl[2];
        split8(lt[0], lt[1], ph[0], pl[0]);
        split8(lt[2], lt[3], ph[1], pl[1]);
        // ---- piece 2k + 1: p-weighted mean, score ----
        wait_dma();
        __syncthreads();
        if (k + 1 < a.N)
          dma_table_shared(img + logit_floats + mean_floats, sh_tab, logit_floats, share, wave, lane);
        {
          const f16x8* im = reinterpret_cast<const f16x8*>(sh_tab + sh_floats);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, mx = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              if (kb < kbk) {  // wave-uniform
                const f16x8 ah = im[((t * kbk + kb) * 2 + 0) * 64 + lane], al = im[((t * kbk + kb) * 2 + 1) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ph[kb], acc, 0, 0, 0);
                mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, pl[kb], mx, 0, 0, 0);
                mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ph[kb], mx, 0, 0, 0);
              }
            }
            const f32x4 c = load_tile4(cs, t, g), iv = load_tile4(cs + dpad, t, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) rs[t][r] = (__builtin_fmaf(mx[r], SD_LO_INV, acc[r]) - (x[t][r] - c[r])) * iv[r];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (REF == RF_GMM_FULL) {
        // score_mog_full (distr/gauss.py:110-121): per component v = P (m - x) on the matrix pipe (P staged in LDS for the whole
        // workgroup, split-f16 like the drift net), logit = log w - quad/2 - logdet/2 - c1 with quad = <m - x, v>, then the same
        // online softmax as the diagonal path; the score is sum_c p_c v_c.
        constexpr int KB = (NT + 1) / 2;
        float m_run, l_run;
        gmm_score_begin<NT>(rs, m_run, l_run);
        const int n_pieces = a.N * a.ref_k * FULL_PP;
        for (int c = 0; c < a.ref_k; ++c) {
          const float* mc = a.ref_mean + (static_cast<size_t>(k) * a.ref_k + c) * dpad;
          f16x8 dh[KB], dl[KB];
          {
            f32x4 dm[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) dm[t] = load_tile4(mc, t, g) - x[t];
            split_tiles<NT>(dm, dh, dl);
          }
          f32x4 v[NT];
          float quad = 0.0f;
#pragma unroll
          for (int h = 0; h < FULL_PP; ++h) {
            const int q = (k * a.ref_k + c) * FULL_PP + h;
            wait_dma();
            __syncthreads();
            if (q + 1 < n_pieces)
              dma_table_shared(a.ref_tab + static_cast<size_t>(q + 1) * FULL_PIECE, sh_tab + ((q + 1) & 1) * sh_floats, FULL_PIECE, share,
                               wave, lane);
            f32x4 out[FULL_TO], mx[FULL_TO];
#pragma unroll
            for (int o = 0; o < FULL_TO; ++o) out[o] = mx[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            dense_pre<KB, FULL_TO>(dh, dl, out, mx, reinterpret_cast<const f16x8*>(sh_tab + (q & 1) * sh_floats), lane);
            fold_lo<FULL_TO>(out, mx);
#pragma unroll
            for (int o = 0; o < FULL_TO; ++o) {
              const int t = h * FULL_TO + o;
              v[t] = out[o];
              const f32x4 dmt = load_tile4(mc, t, g) - x[t];
#pragma unroll
              for (int r = 0; r < 4; ++r) quad = __builtin_fmaf(dmt[r], out[o][r], quad);
            }
          }
          quad = group_sum(quad);
          const float lp = rcs[c * 2 + 1] + (((-0.5f * quad) - a.ref_c1) - rcs[c * 2 + 0]);
          const float m_new = fmaxf(m_run, lp);
          const float so = exp_nonpos(m_run - m_new);
          const float pk = exp_nonpos(lp - m_new);
          l_run = l_run * so + pk;
          m_run = m_new;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) rs[t][r] = __builtin_fmaf(pk, v[t][r], rs[t][r] * so);
        }
        gmm_score_end<NT>(rs, l_run);
      }
      float st = 1.0f;
      if constexpr (SC != SC_NONE) st = a.stheta ? a.stheta[k] : 1.0f;

      // ---- per pair of output tiles: out_layer (MFMA) -> clip -> cost -> noise -> integrator -> Ito term ----
      float su2 = 0.0f, suz = 0.0f, sux = 0.0f;
      // (NT odd: pairs, then the last tile alone -- `t0` is a constant after unrolling, the dead branch goes away)
      constexpr bool can_remove = SC != SC_NONE && has_ref && !eubo;
      auto ref_tile = [&](int t) __attribute__((always_inline)) -> f32x4 {  // reference score of feature tile t
        f32x4 rq = {0.0f, 0.0f, 0.0f, 0.0f};
#ifdef SD_DBG_NOREF
        if constexpr (REF == RF_GMM) rq = f32x4{resp[0], resp[0], resp[0], resp[0]};
#else
        if constexpr (REF == RF_GMM) {
          if (kfast) rq = gmm_score_tile_centred<NT>(x, rtab, g, resp, t);  // the reference's default reference: one test
          else if (kfull) rq = gmm_score_tile<NT, SD_KREG>(x, rtab, SD_KREG, g, resp, t);
          else rq = same_var ? gmm_score_tile_shared_var<NT>(x, rtab, a.ref_k, g, resp, t) : gmm_score_tile<NT>(x, rtab, a.ref_k, g, resp, t);
        }
#endif
        if constexpr (REF == RF_GMM_BIG || REF == RF_GMM_FULL || REF == RF_GMM_MM) rq = rs[t];
        if constexpr (REF == RF_GAUSS) rq = gauss_score_tile<NT>(x, rtab, g, t);
        return rq;
      };
      auto out_group = [&](auto otc, int t0) __attribute__((always_inline)) {
        constexpr int OT = decltype(otc)::value;
        f32x4 u[OT];
        // ONE copy of the output layer in the step loop: a step that went through the range-safe twin (below, first group) left the split
        // of hid * sigma in `hs`; the plain product then gives b 2^e + sigma 2^e W a, which out_tiles_unscale turns into (W a + b)
        if constexpr (has_twin) {
          mlp_out_tiles<NT, OT>(hs, lds, bias, t0, lane, u);
          if (__builtin_expect(safe_net, 0)) out_tiles_unscale<OT>(u, bias, t0, lane, hid_rs, ns.inv_out);
          else if (__builtin_expect(inv_out_plain != 1.0f, 0)) {
            asm volatile("" ::: "memory");  // a real branch: without it the compiler multiplies unconditionally (x * 1.0 = x: +32 instructions per tile-step)
#pragma unroll
            for (int o = 0; o < OT; ++o) u[o] = u[o] * inv_out_plain;
          }
        } else {
          mlp_out_tiles<NT, OT>(hs, lds, bias, t0, lane, u, ns.inv_out);
        }
#ifdef SD_NO_RANGE_GUARD
        constexpr bool range_guard = false;
#else
        constexpr bool range_guard = has_twin;
#endif
        if (has_twin && t0 == 0) {
          // A state or an activation beyond f16's range (65 504) turned into inf in a split operand: every output of that particle is
          // then inf or NaN, so ONE compare on the first output register finds it.  The net of this step is evaluated again through
          // the range-safe twin (per-particle power-of-two scaling of every layer's input) -- x has not been touched yet.  The
          // reference's fp32 GEMMs stay finite there, and so does this; inputs that are non-finite themselves stay non-finite.
          const bool bad = range_guard && !(__builtin_fabsf(u[0][0]) <= 3.0e38f);
          const bool trip = scaled_net | (__builtin_amdgcn_ballot_w64(bad) != 0);
          if (__builtin_expect(trip, 0)) {
            asm volatile("" ::: "memory");  // nothing of the cold path is to be prepared ahead of this test
            safe_net = true;
            mlp_hidden_safe<NT>(x, hid, lds, bias, a.temb + static_cast<size_t>(k) * SD_H, lane, ns);
            hid_rs = split_hidden_safe(hid, hs);
            mlp_out_tiles<NT, OT>(hs, lds, bias, t0, lane, u);
            out_tiles_unscale<OT>(u, bias, t0, lane, hid_rs, ns.inv_out);
            asm volatile("" ::: "memory");
          }
        }
        if (a.clip_model > 0.0f) {
          // ClippedCtrl's clip (reparam.py:42) almost never binds (clip_model = 1e4): one compare per element into a
          // wave-wide mask, and the 4-instruction NaN-preserving clamp only when some lane is out of range or NaN
          bool out_of_range = false;
#pragma unroll
          for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) out_of_range |= !(__builtin_fabsf(u[o][r]) <= a.clip_model);
          if (__builtin_amdgcn_ballot_w64(out_of_range) != 0) {
#pragma unroll
            for (int o = 0; o < OT; ++o)
#pragma unroll
              for (int r = 0; r < 4; ++r) u[o][r] = clampf(u[o][r], a.clip_model);
          }
        }
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          const int t = t0 + o;
          if constexpr (SC == SC_REFSLOT) add_ctrl_score_tile(a, u[o], rs[t], x[t], st, score_gain, lerp_w, t, 4 * g, d_dyn);
          else if constexpr (SC != SC_NONE) add_ctrl_score_tile(a, u[o], ts[t], x[t], st, score_gain, lerp_w, t, 4 * g, d_dyn);
          f32x4 rq = {0.0f, 0.0f, 0.0f, 0.0f};  // reference score of this tile
          if constexpr (can_remove) {
            // RemoveReferenceCtrl (models/reparam.py:46-64, use_rescaling = False): the control is ctrl - ref_score; the reference score of
            // the tile is needed for the drift anyway, it is only formed before the running cost instead of after the noise.  Only in the
            // instantiations with a score control AND a reference (upstream: "only used for Langevin init") -- the plain samplers carry no test.
            rq = ref_tile(t);
            if (a.flags & SDENG_FLAG_REMOVE_REF) u[o] = u[o] - rq;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float uv = u[o][r];
            if constexpr (eubo) uv = uv * c2;  // use_rescaling: generative_ctrl /= sde_diff (losses/oc.py:348-349); 1 for EI
            u[o][r] = uv;
            if constexpr (!eubo) su2 = __builtin_fmaf(uv, uv, su2);
          }
          const f32x4 z = noise_tile(t);  // EUBO: the same counters as in the noising phase above, regenerated
          if constexpr (!can_remove && has_ref) rq = ref_tile(t);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xv = x[t][r], uv = u[o][r];
            if constexpr (eubo) {  // losses/oc.py:352-358, 560-564: running_cost = u (ref + u/2); <u, x>; <u, z>
              su2 += uv * (rq[r] + 0.5f * uv);
              sux = __builtin_fmaf(uv, xv, sux);
              suz = __builtin_fmaf(uv, z[r], suz);
            } else if constexpr (lin) {  // eq/sdes.py:535-538: ret = c1*x + c2*(ref + u); ret += c3*z
              float sc = uv;
              if constexpr (has_ref) sc = rq[r] + uv;
              // fused multiply-adds (round 2): one rounding per term less than the reference's separate torch ops -- within an ulp
              // of them per step, like the drift-net products -- and two instructions fewer per element
              x[t][r] = __builtin_fmaf(c3, z[r], __builtin_fmaf(c2, sc, c1 * xv));
              suz = __builtin_fmaf(uv, z[r], suz);
            } else {  // losses/oc.py:277-284
              // x + (c1 x + c3 ref + c2 u) dt + c2 sqrt(dt) z with the per-step products formed once (e1 = 1 + c1 dt, e2 = c2 dt,
              // e3 = c3 dt, e4 = c2 sqrt(dt)): three fused instructions per element; <u, db> = sqrt(dt) <u, z> is scaled after the sum
              float acc = e1 * xv;
              if constexpr (has_ref) acc = __builtin_fmaf(e3, rq[r], acc);
              x[t][r] = __builtin_fmaf(e4, z[r], __builtin_fmaf(e2, uv, acc));
              suz = __builtin_fmaf(uv, z[r], suz);
            }
          }
        }
      };
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += 2) {
        if (t0 + 1 < NT) out_group(std::integral_constant<int, 2>{}, t0);
        else out_group(std::integral_constant<int, 1>{}, t0);
      }
      __builtin_amdgcn_s_setprio(0);  // end of the tail
      if constexpr (ref_lds) {  // next step's table: every read of this step's copy has been consumed above
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < a.N) dma_table_to_lds(a.ref_tab + static_cast<size_t>(k + 1) * tab_floats, my_tab, tab_floats, lane);
      }
      // running cost 0.5*omega*|u|^2 (losses/oc.py:493) or 0.5*|u|^2*dt (:274); per-step constant
      // (TimeReversalLoss: -int drift divergence, :1218-1219); stochastic integral (:284, :499)
      su2 = group_sum(su2);
      if constexpr (eubo) {  // rnd -= cost * c4;  rnd += <u,x> * c6 (EM only);  rnd -= <u,z> * c5
        rnd -= su2 * c4;
        if (c6 != 0.0f) rnd += group_sum(sux) * c6;
        rnd -= group_sum(suz) * c5;
      } else {
        rnd += lin ? c4 * su2 : (0.5f * su2) * c4;
        rnd += c6;
        if (a.flags & SDENG_FLAG_ITO) {
          suz = group_sum(suz);
          rnd += c5 * suz;  // LIN: sqrt(omega) <u, z> (:499); EM: <u, db>, db = sqrt(dt) z (:284)
        }
      }
      if constexpr (PAR) {
        if (a.xs_out) store_rows<NT>(a.xs_out + static_cast<size_t>(k + 1) * a.B * a.d, trash, row, d_dyn, live, g, x);
      }
    }

    // the terminal cost (losses/oc.py:290, :973) is added by k_terminal from x_out: it runs once per
    // trajectory, and keeping every distribution's log-density out of this kernel keeps its registers free
    store_rows<NT>(a.x_out, trash, row, a.d, live, g, x);
    if (live && g == 0) a.rnd_out[row] = rnd;
  }
}

// Generative control alone, u(t, x) for a whole batch at one time (unit parity tests of the drift net and
// the ctrl wrappers against the oracle; same device code as the step loop, k = 0 of one-row tables).
template <int NT, int SC>
__global__ void __launch_bounds__(SD_THREADS, SD_WAVES / 4) k_ctrl_forward(const SimArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.wpack);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    const int n4 = sd_lds_weight_floats(NT) / 4;
    for (int i = tid; i < n4; i += SD_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  const float* bias = a.wpack + sd_off_bias(NT);
  const NetScale ns = load_net_scale(bias, NT);
  const int p = lane & 15, g = lane >> 4;
  float* trash = a.trash + tid * 4;
  for (int tile = blockIdx.x + gridDim.x * wave; tile < a.ntiles; tile += gridDim.x * SD_WAVES) {  // CUs first
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = row < static_cast<uint32_t>(a.B);
    f32x4 x[NT];
    load_rows<NT>(a.x_in, row, a.d, live, g, x);
    const float score_gain = a.coef[7], lerp_w = a.coef[8];
    f32x4 hid[SD_HT];
    mlp_hidden<NT>(x, hid, lds, bias, a.temb, lane);
    HidSplit hs = split_hidden(hid);
    bool safe_net = false;  // scaled weights or an overflowed f16 operand: the range-safe twin, as in the step loop
    float hid_rs = 1.0f;
    f32x4 ts[SC != SC_NONE ? NT : 1];
    if constexpr (SC == SC_GMM) {
      if (NT == 1 && a.target.kind == SDENG_DIST_RINGS) ts[0] = rings_score(x[0], a.target, g);
      else gmm_score<NT>(x, a.target.tab, a.target.consts, 4, a.target.k, a.target.p0, g, ts);
    }
    if constexpr (SC == SC_PHI4) phi4_score<NT>(x, a.target, a.d, g, lane, ts);
    static_assert(SC != SC_LOGREG, "ctrl_forward is not instantiated for the in-loop logistic-regression score");
    float st = 1.0f;
    if constexpr (SC != SC_NONE) st = a.stheta ? a.stheta[0] : 1.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 u[1];
      if (safe_net) mlp_out_tiles_safe<NT, 1>(hs, hid_rs, lds, bias, t, lane, u, ns.inv_out);
      else mlp_out_tiles<NT, 1>(hs, lds, bias, t, lane, u);
      if (t == 0) {
        const bool bad = !(__builtin_fabsf(u[0][0]) <= 3.0e38f);
        if (ns.any() || __builtin_amdgcn_ballot_w64(bad) != 0) {
          safe_net = true;
          mlp_hidden_safe<NT>(x, hid, lds, bias, a.temb, lane, ns);
          hid_rs = split_hidden_safe(hid, hs);
          mlp_out_tiles_safe<NT, 1>(hs, hid_rs, lds, bias, t, lane, u, ns.inv_out);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (a.clip_model > 0.0f) u[0][r] = clampf(u[0][r], a.clip_model);
      if constexpr (SC != SC_NONE) add_ctrl_score_tile(a, u[0], ts[t], x[t], st, score_gain, lerp_w, t, 4 * g, a.d);
      store_quad(a.x_out, trash, row, a.d, live, t, g, u[0]);
    }
  }
}

template <int NT, int SC>
static int launch_ctrl_forward(const SimArgs& a, int grid, hipStream_t stream) {
  const size_t lds_bytes = static_cast<size_t>(sd_lds_total_bytes(NT, false));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctrl_forward<NT, SC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((k_ctrl_forward<NT, SC>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  return static_cast<int>(hipGetLastError());
}
#define SD_DEFINE_CTRL(NT, SC) \
  int sd_launch_ctrl_##NT##_##SC(const SimArgs& a, int grid, hipStream_t s) { return launch_ctrl_forward<NT, SC>(a, grid, s); }

// host-side launcher, one per instantiation (defined in gen/sim_*.hip)
template <int NT, int REF, int SC, int FORM, int PAR>
static int launch_simulate_par(const SimArgs& a, int grid, hipStream_t stream) {
  constexpr int W = sd_waves_of<NT, REF, SC, FORM, PAR>();
  const size_t lds_bytes = static_cast<size_t>(sd_lds_total_bytes(NT, REF == RF_GAUSS || REF == RF_GMM, W)) +
                           ((SC == SC_LOGREG && a.lr.in_lds) ? sizeof(float) * sd_lr_floats(NT, a.lr.n_rows) : 0) +
                           ((REF == RF_GMM_BIG || REF == RF_GMM_FULL || REF == RF_GMM_MM) ? sizeof(float) * 2 * sd_share_buf_floats(a.ref_share) : 0);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_simulate<NT, REF, SC, FORM, PAR>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((k_simulate<NT, REF, SC, FORM, PAR>), dim3(grid), dim3(64 * W), lds_bytes, stream, a);
  return static_cast<int>(hipGetLastError());
}
template <int NT, int REF, int SC, int FORM>
static int launch_simulate(const SimArgs& a, int grid, hipStream_t stream) {
  if (a.noise_in || a.xs_out) return launch_simulate_par<NT, REF, SC, FORM, 1>(a, grid, stream);
  return launch_simulate_par<NT, REF, SC, FORM, 0>(a, grid, stream);
}

#define SD_DEFINE_SIM(NT, REF, SC, FORM) \
  int sd_launch_sim_##NT##_##REF##_##SC##_##FORM(const SimArgs& a, int grid, hipStream_t s) { return launch_simulate<NT, REF, SC, FORM>(a, grid, s); }
