// Device building blocks of the simulate kernels (gfx950 only; see sim_common.hpp for the layout).
#pragma once
#include "sim_common.hpp"
#include "../../include/sdeng.h"

#define SD_INLINE __device__ __forceinline__

// ----------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------
// The two lanes p and p+32 hold the two feature-halves of one particle: one cross-half add finishes
// every per-particle reduction (log-weight increments, mixture logits).
SD_INLINE float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// torch.clip semantics: NaN stays NaN (fminf/fmaxf or v_med3 would swallow it).
SD_INLINE float clampf(float v, float m) {
  v = (v < -m) ? -m : v;
  v = (v > m) ? m : v;
  return v;
}

// erf, branch-free (both pieces evaluated, one select): the two minimax pieces are N. Juffa's
// single-precision erff (max error < 1 ulp, checked against scipy over [-6,6]); libm's erff branches
// per element, which costs a divergent branch per hidden unit in a 64-wide wave.
SD_INLINE float erf_bf(float a) {
  const float t = __builtin_fabsf(a);
  const float s = a * a;
  float r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = __builtin_fmaf(r, s, u);
  r = __builtin_fmaf(r, t, -1.06777877e-1f);
  r = __builtin_fmaf(r, t, -6.34846687e-1f);
  r = __builtin_fmaf(r, t, -1.28717512e-1f);
  r = __builtin_fmaf(r, t, -t);
  r = 1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f);
  const float big = __builtin_copysignf(r, a);
  float q = -5.96761703e-4f;
  q = __builtin_fmaf(q, s, 4.99119423e-3f);
  q = __builtin_fmaf(q, s, -2.67681349e-2f);
  q = __builtin_fmaf(q, s, 1.12819925e-1f);
  q = __builtin_fmaf(q, s, -3.76125336e-1f);
  q = __builtin_fmaf(q, s, 1.28379166e-1f);
  const float small = __builtin_fmaf(q, a, a);
  return (t > 0.927734375f) ? big : small;
}

// exact-erf GELU, torch's CPU formula (x*0.5)*(1+erf(x/sqrt2))  -- models/mlp.py activation nn.GELU()
SD_INLINE float gelu(float v) { return (v * 0.5f) * (1.0f + erf_bf(v * 0.70710678118654752440f)); }

SD_INLINE int feat(int t, int r, int h) { return 32 * t + 8 * (r >> 2) + 4 * h + (r & 3); }
// feat(t, r, h) < d with the lane-dependent part (4h) on one side only: the compare takes a scalar
// operand, so no per-element index register is ever materialised.
SD_INLINE bool feat_lt(int t, int r, int h4, int d) { return h4 < d - (32 * t + 8 * (r >> 2) + (r & 3)); }

// 16 registers of one feature tile for this lane from a dense [..] vector (tile base = 32*t)
SD_INLINE f32x16 load_tile16(const float* base, int h) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 b = *reinterpret_cast<const f32x4*>(base + 8 * q + 4 * h);
    v[4 * q + 0] = b.x;
    v[4 * q + 1] = b.y;
    v[4 * q + 2] = b.z;
    v[4 * q + 3] = b.w;
  }
  return v;
}

// ----------------------------------------------------------------------------------------------
// FP32 MFMA dense layer:  out[to] += W[to-tile][ti-tile] * in[ti]   (Y^T = W X^T form)
// `w` is the LDS image written by k_pack_mlp: float4 index ((to*TI+ti)*4+r4)*64+lane holds the A
// operands of k-steps r = 4*r4 .. 4*r4+3, i.e. W[32 to + (lane&31)][feat(ti, r, lane>>5)].
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (1e-5 parity rules out bf16/xf32 paths).
// ----------------------------------------------------------------------------------------------
template <int TI, int TO>
SD_INLINE void dense(const f32x16 (&in)[TI], f32x16 (&out)[TO], const float* w, int lane) {
  const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
#pragma unroll
  for (int to = 0; to < TO; ++to) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 a = w4[((to * TI + ti) * 4 + r4) * 64 + lane];
        out[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, in[ti][4 * r4 + 0], out[to], 0, 0, 0);
        out[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, in[ti][4 * r4 + 1], out[to], 0, 0, 0);
        out[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, in[ti][4 * r4 + 2], out[to], 0, 0, 0);
        out[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, in[ti][4 * r4 + 3], out[to], 0, 0, 0);
      }
    }
  }
}

template <int T>
SD_INLINE void gelu_tiles(f32x16 (&v)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[t][r] = gelu(v[t][r]);
}

// FourierMLP.forward (models/mlp.py:135-143) for a 32-particle tile, split in two so that the d-wide
// output never has to be live at once: mlp_hidden() runs input_embed + time embedding + the two hidden
// layers and returns gelu(h) (64 channels = 2 register tiles); mlp_out_tile() produces ONE 32-feature
// tile of out_layer, which the caller consumes (clip, cost, integrator) before asking for the next.
// `temb` = this step's time embedding [64] (hoisted: the reference recomputes the identical row for
// every particle, :136-137).
template <int DT>
SD_INLINE void mlp_hidden(const f32x16 (&x)[DT], f32x16 (&a)[2], const float* lds, const float* bias, const float* temb,
                           int lane) {
  const int h = lane >> 5;
  f32x16 b[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) a[t] = load_tile16(bias + 32 * t, h);  // b_in
  dense<DT, 2>(x, a, lds + sd_off_win(DT), lane);
#pragma unroll
  for (int t = 0; t < 2; ++t) {  // embed = embed_x + embed_t
    const f32x16 e = load_tile16(temb + 32 * t, h);
    a[t] = a[t] + e;
  }
  gelu_tiles<2>(a);
#pragma unroll
  for (int t = 0; t < 2; ++t) b[t] = load_tile16(bias + 64 + 32 * t, h);  // b_h1
  dense<2, 2>(a, b, lds + sd_off_wh1(DT), lane);
  gelu_tiles<2>(b);
#pragma unroll
  for (int t = 0; t < 2; ++t) a[t] = load_tile16(bias + 128 + 32 * t, h);  // b_h2
  dense<2, 2>(b, a, lds + sd_off_wh2(DT), lane);
  gelu_tiles<2>(a);
}

template <int DT>
SD_INLINE f32x16 mlp_out_tile(const f32x16 (&a)[2], const float* lds, const float* bias, int to, int lane) {
  const int h = lane >> 5;
  f32x16 u[1];
  u[0] = load_tile16(bias + 192 + 32 * to, h);  // b_out
  dense<2, 1>(a, u, lds + sd_off_wout(DT) + to * 2048, lane);
  return u[0];
}

// ----------------------------------------------------------------------------------------------
// per-wave reference table in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no VGPR staging).  The copy for
// step k+1 is issued once the wave has read step k's table for the last time and lands under the next step's
// MFMA phase; the reader waits with s_waitcnt vmcnt(0) (nothing else orders a ds_read behind an LDS-DMA).
// ----------------------------------------------------------------------------------------------
SD_INLINE void dma_table_to_lds(const float* __restrict__ gsrc, float* lds_dst, int n_floats, int lane) {
  for (int c = 0; c * 256 < n_floats; ++c) {  // 1 KiB per wave-instruction
    typedef __attribute__((address_space(1))) void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    __builtin_amdgcn_global_load_lds((gvoid*)(gsrc + c * 256 + lane * 4), (lvoid*)(lds_dst + c * 256), 16, 0, 0);
  }
}
SD_INLINE void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ----------------------------------------------------------------------------------------------
// counter-based noise: Philox4x32-10 + Box-Muller (definition shared with oracle.philox_normal)
// ----------------------------------------------------------------------------------------------
SD_INLINE void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t (&o)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0;
  o[1] = c1;
  o[2] = c2;
  o[3] = c3;
}

SD_INLINE float u01(uint32_t bits) { return (static_cast<float>(bits >> 9) + 0.5f) * 1.1920928955078125e-07f; }

// normals of features 4*jb .. 4*jb+3 of global particle `pidx` at step `step`.  Counter order
// (pidx, jb, step, stream): the first round multiplies c0 and c2 and XORs c1 into the c2 product, so the step
// product is scalar work, the particle product is the only loop-invariant (two registers), and no per-quad
// partial round can be hoisted out of the step loop (with the step in c0 the compiler hoisted 16 of them
// per tile and spilled them).
SD_INLINE f32x4 philox_normal4(uint32_t pidx, uint32_t step, uint32_t jb, uint32_t stream, uint32_t k0, uint32_t k1) {
  uint32_t r[4];
  philox4x32_10(pidx, jb, step, stream, k0, k1, r);
  f32x4 z;
  // rad = sqrt(-2 ln u) = sqrt(-2 ln2 * log2 u); v_sin/v_cos take revolutions: sin(2 pi u) directly
  const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r[0])));
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r[2])));
  const float a0 = u01(r[1]), a1 = u01(r[3]);
  z.x = r0 * __builtin_amdgcn_cosf(a0);
  z.y = r0 * __builtin_amdgcn_sinf(a0);
  z.z = r1 * __builtin_amdgcn_cosf(a1);
  z.w = r1 * __builtin_amdgcn_sinf(a1);
  return z;
}

// ----------------------------------------------------------------------------------------------
// Gaussian-mixture score (distr/gauss.py:97-107 score_mog) with an online softmax over components.
// tab: [K][2][dpad] (mean, 1/var); consts: [K][cstride] with [0] = 0.5*sum log var, [1] = log w_k.
// ----------------------------------------------------------------------------------------------
template <int DT>
SD_INLINE void gmm_score(const f32x16 (&x)[DT], const float* __restrict__ tab, const float* __restrict__ consts,
                         int cstride, int K, float c1, int h, f32x16 (&acc)[DT]) {
  constexpr int dpad = 32 * DT;
  float m_run = -INFINITY, l_run = 0.0f;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  for (int k = 0; k < K; ++k) {
    const float* mp = tab + static_cast<size_t>(k) * 2 * dpad + 4 * h;
    const float* vp = mp + dpad;
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mp + 32 * t + 8 * q);
        const f32x4 iv = *reinterpret_cast<const f32x4*>(vp + 32 * t + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = x[t][4 * q + e] - m[e];
          part = __builtin_fmaf(dl * dl, iv[e], part);
        }
      }
    part = half_sum(part);
    float lp = ((-0.5f * part) - c1) - consts[k * cstride + 0];  // distr/gauss.py:70-72
    lp = consts[k * cstride + 1] + lp;                           // torch.log(weights) + log_prob
    const float m_new = fmaxf(m_run, lp);
    const float so = expf(m_run - m_new);
    const float pk = expf(lp - m_new);
    l_run = l_run * so + pk;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mp + 32 * t + 8 * q);
        const f32x4 iv = *reinterpret_cast<const f32x4*>(vp + 32 * t + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = (m[e] - x[t][4 * q + e]) * iv[e];  // -(x - mean)/var
          acc[t][4 * q + e] = __builtin_fmaf(pk, g, acc[t][4 * q + e] * so);
        }
      }
  }
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] *= inv;
}

// Small-mixture variant (K <= SD_KREG; the reference's default n_modes is 4, conf/target/many_modes.yaml): component responsibilities p_k = softmax_k(log w_k + log N_k(x)) are
// computed once (pass 1) and kept in registers; the score of a feature is then assembled where it is
// consumed, sum_k p_k (m_k - x) / var_k, so no d-wide score array stays live across the output layer.
#define SD_KREG 4
template <int DT>
SD_INLINE void gmm_resp(const f32x16 (&x)[DT], const float* __restrict__ tab, const float* __restrict__ consts,
                        int cstride, int K, float c1, int h, float (&p)[SD_KREG]) {
  constexpr int dpad = 32 * DT;
  float lp[SD_KREG];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    lp[k] = -INFINITY;
    if (k < K) {
      const float* mp = tab + static_cast<size_t>(k) * 2 * dpad + 4 * h;
      const float* vp = mp + dpad;
      float part = 0.0f;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(mp + 32 * t + 8 * q);
          const f32x4 iv = *reinterpret_cast<const f32x4*>(vp + 32 * t + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dl = x[t][4 * q + e] - m[e];
            part = __builtin_fmaf(dl * dl, iv[e], part);
          }
        }
      part = half_sum(part);
      float v = ((-0.5f * part) - c1) - consts[k * cstride + 0];  // distr/gauss.py:70-72
      v = consts[k * cstride + 1] + v;                           // torch.log(weights) + log_prob
      lp[k] = v;
      mx = fmaxf(mx, v);
    }
  }
  float den = 0.0f;
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    p[k] = (k < K) ? expf(lp[k] - mx) : 0.0f;
    den += p[k];
  }
  const float inv = 1.0f / den;
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) p[k] *= inv;
}

// score of the four features of quad (t, q) from the responsibilities: -sum_k p_k (x - m_k)/var_k
template <int DT>
SD_INLINE f32x4 gmm_score_quad(const f32x16 (&x)[DT], const float* __restrict__ tab, int K, int h, const float (&p)[SD_KREG],
                               int t, int q) {
  constexpr int dpad = 32 * DT;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    if (k < K) {
      const float* mp = tab + static_cast<size_t>(k) * 2 * dpad + 4 * h + 32 * t + 8 * q;
      const f32x4 m = *reinterpret_cast<const f32x4*>(mp);
      const f32x4 iv = *reinterpret_cast<const f32x4*>(mp + dpad);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(p[k], (m[e] - x[t][4 * q + e]) * iv[e], acc[e]);
    }
  }
  return acc;
}

// Gaussian (one component) score: -(x - mean)/var  (distr/gauss.py:124-126)
template <int DT>
SD_INLINE void gauss_score(const f32x16 (&x)[DT], const float* __restrict__ tab, int h, f32x16 (&acc)[DT]) {
  constexpr int dpad = 32 * DT;
  const float* mp = tab + 4 * h;
  const float* vp = mp + dpad;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 m = *reinterpret_cast<const f32x4*>(mp + 32 * t + 8 * q);
      const f32x4 iv = *reinterpret_cast<const f32x4*>(vp + 32 * t + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = -((x[t][4 * q + e] - m[e]) * iv[e]);
    }
}

// phi^4 lattice neighbours of one group of four features (g = 4 t + q; features 8 g + 4 h + 0..3).
// The partner lane (p, 1-h) holds the sites adjacent to the group's two ends; 0 outside the lattice.
template <int DT>
SD_INLINE void phi4_group_edges(const f32x16 (&x)[DT], int g, int h, float& l_edge, float& r_edge) {
  const int t = g >> 2, q = g & 3;
  const float p3 = __shfl_xor(x[t][4 * q + 3], 32, 64);  // partner's last site of group g
  const float p0 = __shfl_xor(x[t][4 * q + 0], 32, 64);  // partner's first site of group g
  float p3prev = 0.0f, p0next = 0.0f;
  if (g > 0) p3prev = __shfl_xor(x[(g - 1) >> 2][4 * ((g - 1) & 3) + 3], 32, 64);
  if (g < 4 * DT - 1) p0next = __shfl_xor(x[(g + 1) >> 2][4 * ((g + 1) & 3) + 0], 32, 64);
  l_edge = h ? p3 : p3prev;  // h==0: site 8g-1 is (g-1, e=3, h=1); h==1: site 8g+3 is (g, e=3, h=0)
  r_edge = h ? p0next : p0;  // h==0: site 8g+4 is (g, e=0, h=1); h==1: site 8g+8 is (g+1, e=0, h=0)
}

// PhiFour.score = -beta * grad_U  (distr/phi_four.py:81-96); p0=a, p1=b, p2=beta
template <int DT>
SD_INLINE void phi4_score(const f32x16 (&x)[DT], const DistDev& ds, int d, int h, f32x16 (&acc)[DT]) {
  const float coef = ds.p0 * static_cast<float>(d);
#pragma unroll
  for (int g = 0; g < 4 * DT; ++g) {
    const int t = g >> 2, q = g & 3;
    float le, re;
    phi4_group_edges<DT>(x, g, h, le, re);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xv = x[t][4 * q + e];
      const float xl = (e == 0) ? le : x[t][4 * q + (e > 0 ? e - 1 : 0)];
      const float xr = (e == 3) ? re : x[t][4 * q + (e < 3 ? e + 1 : 3)];
      float gr = (ds.p1 - xv * (1.0f - xv * xv)) / coef;
      gr = gr + coef * ((2.0f * xv - xr) - xl);
      acc[t][4 * q + e] = feat_lt(t, 4 * q + e, 4 * h, d) ? (-ds.p2) * gr : 0.0f;
    }
  }
}

// Tile-row I/O without control flow: masked lanes (dead rows, pad features) read element 0 of the array
// and discard it, or write to a per-lane dump slot (`trash`, 4 floats per lane) instead of the array.
// Only trajectory start/end, injected noise (parity mode) and trajectory dumps come through here.
template <int DT>
SD_INLINE void load_rows(const float* __restrict__ src, uint32_t row, int d, bool live, int h, f32x16 (&v)[DT]) {
  const size_t base = static_cast<size_t>(row) * d;
  if ((d & 3) == 0) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = live && feat_lt(t, 4 * q, 4 * h, d);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + (ok ? base + (32 * t + 8 * q + 4 * h) : 0));
        v[t][4 * q + 0] = ok ? b.x : 0.0f;
        v[t][4 * q + 1] = ok ? b.y : 0.0f;
        v[t][4 * q + 2] = ok ? b.z : 0.0f;
        v[t][4 * q + 3] = ok ? b.w : 0.0f;
      }
  } else {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = live && feat_lt(t, r, 4 * h, d);
        const float b = src[ok ? base + feat(t, r, h) : 0];
        v[t][r] = ok ? b : 0.0f;
      }
  }
}

template <int DT>
SD_INLINE void store_rows(float* __restrict__ dst, float* __restrict__ trash, uint32_t row, int d, bool live, int h,
                          const f32x16 (&v)[DT]) {
  const size_t base = static_cast<size_t>(row) * d;
  if ((d & 3) == 0) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = live && feat_lt(t, 4 * q, 4 * h, d);
        f32x4 b;
        b.x = v[t][4 * q + 0];
        b.y = v[t][4 * q + 1];
        b.z = v[t][4 * q + 2];
        b.w = v[t][4 * q + 3];
        float* pdst = ok ? dst + base + (32 * t + 8 * q + 4 * h) : trash;
        *reinterpret_cast<f32x4*>(pdst) = b;
      }
  } else {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = live && feat_lt(t, r, 4 * h, d);
        float* pdst = ok ? dst + base + feat(t, r, h) : trash;
        *pdst = v[t][r];
      }
  }
}

// four consecutive features of one row (injected noise: parity mode only), tile t, quad q
SD_INLINE f32x4 load_quad(const float* __restrict__ src, uint32_t row, int d, bool live, int t, int q, int h) {
  const size_t base = static_cast<size_t>(row) * d;
  f32x4 z;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool ok = live && feat_lt(t, 4 * q + e, 4 * h, d);
    const float b = src[ok ? base + (32 * t + 8 * q + 4 * h + e) : 0];
    z[e] = ok ? b : 0.0f;
  }
  return z;
}

// ----------------------------------------------------------------------------------------------
// CMCD building blocks (eq/sdes.py:101-110, distr/logistic_regression.py, distr/gauss.py:129-135)
// ----------------------------------------------------------------------------------------------
#define SD_LR_ROWS 192      // data rows padded to 6 tiles of 32 (sonar: 166)
#define SD_LR_STRIDE 65     // LDS row stride of the design-matrix image (odd: conflict-free in both products)

// Dense product whose A operands come from ONE plain LDS image M[rows][SD_LR_STRIDE] (ds_read_b32 per k-step):
//   TRANS = 0: out[to] += M[32 to + i][feat(ti, r, h)]      (rows = outputs)      logits = X w
//   TRANS = 1: out[to] += M[feat(ti, r, h)][32 to + i]      (rows = summation)    grad   = X^T r
// Both address patterns are bank-conflict-free with the odd stride, so the design matrix is stored once.
template <int TI, int TO, int TRANS>
SD_INLINE void dense_plain(const f32x16 (&in)[TI], f32x16 (&out)[TO], const float* m, int lane, int in_tile0, int out_tile0) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int to = 0; to < TO; ++to)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kf = 32 * (ti + in_tile0) + 8 * (r >> 2) + (r & 3);  // + 4h below
        const int o = 32 * (to + out_tile0) + i;
        const float a = TRANS ? m[(kf + 4 * h) * SD_LR_STRIDE + o] : m[o * SD_LR_STRIDE + kf + 4 * h];
        out[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[ti][r], out[to], 0, 0, 0);
      }
}

// Per-datum factor of the logistic-regression score the reference obtains by autograd through
// sigmoid -> clip(thr) -> probs_to_logits(clamp eps) -> BCE-with-logits (distr/logistic_regression.py:41-61,
// distr/base.py:146-154); closed form checked against autograd incl. saturated rows.
SD_INLINE float logreg_residual(float logit, float y, float thr) {
  const float eps = 1.1920928955078125e-07f;
  const float p = 1.0f / (1.0f + expf(-logit));
  const float pc = fminf(fmaxf(p, thr), 1.0f - thr);
  const float pcc = fminf(fmaxf(pc, eps), 1.0f - eps);
  const bool pass = (p >= thr) && (p <= 1.0f - thr) && (pc >= eps) && (pc <= 1.0f - eps);
  const float l2 = logf(pcc) - log1pf(-pcc);
  const float sg = 1.0f / (1.0f + expf(-l2));
  const float r = (y - sg) * (1.0f / pcc + 1.0f / (1.0f - pcc)) * (p * (1.0f - p));
  return pass ? r : 0.0f;
}
