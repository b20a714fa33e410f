// Low-latency variant of the step loop for SMALL batches (SDENG_FLAG_SPLIT_TILES): one 16-particle tile is worked on by FOUR waves,
// one per SIMD of a CU, each owning a quarter of the features.
//
// Why: the persistent kernel of sim_kernel.hpp gives a tile to ONE wave; alone on its SIMD that wave needs ~9.8 us per step whatever the
// batch (its own instruction issue and latencies), so the reference's default evaluation batch (6 000 particles,
// conf/solver/basic_oc_base.yaml:28-30) takes 2.5 ms with three quarters of the chip idle (profiles/r01_small_batch_latency.log).  Here
// wave w of a tile's four owns the input K-block w = feature tiles 2w, 2w+1 and the hidden tile w:
//
//   A  own features: split x, partial input layer (4 hidden tiles x own K-block), partial mixture quadratic forms     -> LDS
//   -- barrier 1 --
//   B  hidden tile w: sum the four partials + bias + time embedding, GELU                                             -> LDS
//      mixture responsibilities from the summed quadratic forms (every wave, identical arithmetic)
//   -- barrier 2 --
//   C  hidden layer 1, output tile w (all four activated input tiles from LDS), GELU                                   -> LDS
//   -- barrier 3 --
//   D  hidden layer 2, output tile w, GELU                                                                             -> LDS
//   -- barrier 4 --
//   E  output layer for the own two feature tiles, clip, cost, noise, reference score, integrator update, Ito term
//
// A workgroup = 8 waves = two tiles side by side (the LDS weight image allows one workgroup per CU), so every SIMD interleaves two
// waves.  Per wave and step: ~36 MFMAs and ~800 vector instructions instead of 144 and 2 660.  Noise counters are those of the
// standard kernel (same normals); the hidden-layer sums and the per-particle reductions are formed in a different order, so results
// agree with the standard kernel to fp32 round-off, not bit for bit -- which is why the path is a flag of the ABI (sdeng.h); the Python
// solvers set it by default for evaluation / training batches of at most 8 192 particles (solver cfg 'split_tiles').
// Scope: ClippedCtrl, forward forms (LIN / EM), no / Gaussian / small-mixture (K <= 4) reference, d > 64, B <= 8 192, no injected noise;
// the trajectory (xs_out) is written.  No range-safe twin of the drift net here (sim_device.hpp mlp_hidden_safe): states / activations
// beyond 65 504 give NaN on this path.
#pragma once
#include "sim_kernel.hpp"

#define SD_SPLIT_W 4                                  // waves per tile
#define SD_SPLIT_SLOT_FLOATS (4096 + 1024 + 1024 + 256)  // P (4 hidden tiles x 4 partials), A0, A2, L (+ A1 overlays P)

#define SD_SPLIT_TAB_FLOATS 256                        // per wave: its two feature tiles of one step's reference table, [K <= 4][mean, 1/var][2 tiles][16]
__host__ __device__ inline int sd_split_lds_bytes(int NT) { return (sd_lds_weight_floats(NT) + 2 * SD_SPLIT_SLOT_FLOATS + SD_WAVES * SD_SPLIT_TAB_FLOATS) * 4; }

template <int NT, int REF, int FORM>
__global__ void __launch_bounds__(SD_THREADS, SD_WAVES / 4) k_simulate_split(const SimArgs a) {
#ifndef SD_EXPERIMENT_WAVES  // occupancy experiments build the other kernels with another wave count; this one is then not launched
  static_assert(SD_WAVES == 8, "two tiles of four waves per workgroup");
#endif
  static_assert(REF == RF_NONE || REF == RF_GAUSS || REF == RF_GMM, "split kernel: no / Gaussian / small-mixture reference");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int dpad = 16 * NT, KBX = (NT + 1) / 2;
  constexpr bool lin = FORM == SDENG_FORM_LIN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave >> 2, w = wave & 3;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.wpack);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    const int n4 = sd_lds_weight_floats(NT) / 4;
    for (int i = tid; i < n4; i += SD_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  const float* bias = a.wpack + sd_off_bias(NT);
  const NetScale ns = load_net_scale(bias, NT);  // per-layer 2^-e of the packed weights (all 1 unless a matrix is out of the f16 split's range)
  float* xch = lds + sd_lds_weight_floats(NT) + slot * SD_SPLIT_SLOT_FLOATS;
  f32x4* P = reinterpret_cast<f32x4*>(xch);            // [hidden tile][partial of wave v][lane]; A1 overlays its first 1024 floats
  f32x4* A0 = reinterpret_cast<f32x4*>(xch + 4096);    // [hidden tile][lane]
  f32x4* A2 = reinterpret_cast<f32x4*>(xch + 5120);
  f32x4* A1 = P;
  float* Lq = xch + 6144;                              // [wave v][particle][4 components] partial quadratic forms; final log-weights
  // this wave's copy of its part of the current step's reference table, filled by ONE LDS-DMA instruction per step (lane l fetches
  // the 16 bytes of component l >> 4, row (l >> 3) & 1 (mean | 1/var), own tile (l >> 2) & 1, feature group l & 3), issued as soon
  // as the previous step's last read is done: the table's L2 latency is off the step's critical path
  float* mytab = lds + sd_lds_weight_floats(NT) + 2 * SD_SPLIT_SLOT_FLOATS + wave * SD_SPLIT_TAB_FLOATS;
  const int p = lane & 15, g = lane >> 4;
  const int t0 = 2 * w, t1 = 2 * w + 1;               // own feature tiles (wave-uniform)
  const bool has0 = t0 < NT, has1 = t1 < NT;
  float* trash = a.trash + tid * 4;
  const f16x8* w8 = reinterpret_cast<const f16x8*>(lds);
  const int tab_floats = a.ref_k * 2 * dpad;
  const int ntiles2 = (a.ntiles + 1) & ~1;
  // loop-invariant bias tiles in registers (a global load per phase would sit on the critical path of every step)
  const f32x4 b_in = load_tile4(bias, w, g), b_h1 = load_tile4(bias + 64, w, g), b_h2 = load_tile4(bias + 128, w, g);
  const f32x4 b_o0 = load_tile4(bias + 192, t0 < NT ? t0 : 0, g), b_o1 = load_tile4(bias + 192, t1 < NT ? t1 : 0, g);
  // source of this lane's 16 bytes of a step's table (clamped to a valid address for absent components / tiles)
  int tab_src = 0;
  if constexpr (REF != RF_NONE) {
    const int kc = lane >> 4, which = (lane >> 3) & 1, j = (lane >> 2) & 1, gq = lane & 3;
    const int kk = kc < a.ref_k ? kc : 0, tt = (t0 + j) < NT ? (t0 + j) : 0;
    tab_src = kk * 2 * dpad + which * dpad + 16 * tt + 4 * gq;
  }
  auto fetch_table = [&](int k) __attribute__((always_inline)) {
    if constexpr (REF != RF_NONE) {
      typedef __attribute__((address_space(1))) void gvoid;
      typedef __attribute__((address_space(3))) void lvoid;
      __builtin_amdgcn_global_load_lds((gvoid*)(a.ref_tab + static_cast<size_t>(k) * tab_floats + tab_src), (lvoid*)mytab, 16, 0, 0);
    }
  };
  // table entry (component kc, row which, own tile j) of this lane's feature group
  auto tab4 = [&](int kc, int which, int j) __attribute__((always_inline)) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(mytab + ((kc * 2 + which) * 2 + j) * 16 + 4 * g);
  };

  for (int base = 2 * blockIdx.x; base < ntiles2; base += 2 * gridDim.x) {  // uniform trip count for all 8 waves
    const int tile = base + slot;
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = tile < a.ntiles && row < static_cast<uint32_t>(a.B);
    const uint32_t pidx = static_cast<uint32_t>(a.particle0 + row);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 x[2] = {zero, zero};
    if (has0) x[0] = load_quad(a.x_in, row, a.d, live, t0, g);
    if (has1) x[1] = load_quad(a.x_in, row, a.d, live, t1, g);
    if (a.xs_out) {
      if (has0) store_quad(a.xs_out, trash, row, a.d, live, t0, g, x[0]);
      if (has1) store_quad(a.xs_out, trash, row, a.d, live, t1, g, x[1]);
    }
    float rnd = 0.0f;  // this wave's share of the log-weight (its features' cost terms; wave 0 also carries rnd0 and the per-step constants)
    if (w == 0 && a.rnd_init) rnd = live ? a.rnd_init[row] : 0.0f;
    if (a.N > 0) fetch_table(0);
    f32x4 te_next = a.N > 0 ? load_tile4(a.temb, w, g) : zero;  // time embedding of step 0, hidden tile w

    for (int k = 0; k < a.N; ++k) {
      const float* cf = a.coef + static_cast<size_t>(k) * SDENG_NCOEF;
      const float c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5], c6 = cf[6];
      const float e1 = __builtin_fmaf(c4, c1, 1.0f), e2 = c4 * c2, e3 = c4 * c3, e4 = c2 * c5;
      const float* rcs = a.ref_consts + static_cast<size_t>(k) * a.ref_k * 2;
      const f32x4 te = te_next;
      float rc0[SD_KREG] = {0.0f, 0.0f, 0.0f, 0.0f}, rc1[SD_KREG] = {0.0f, 0.0f, 0.0f, 0.0f};  // mixture constants, read ahead of the barriers
      if constexpr (REF == RF_GMM) {
#pragma unroll
        for (int kc = 0; kc < SD_KREG; ++kc)
          if (kc < a.ref_k) { rc0[kc] = rcs[kc * 2 + 0]; rc1[kc] = rcs[kc * 2 + 1]; }
      }

      __builtin_amdgcn_s_setprio(1);  // matrix phases A-D: ahead of the other tile's wave on this SIMD (sim_kernel.hpp; 1.28 -> 1.26 ms at 6 000 x 256)
      // ---- A: own K-block of the input layer (partial sums of all four hidden tiles), partial mixture quadratic forms ----
      {
        f16x8 xh, xl;
        if (has1) split8(x[0], x[1], xh, xl);
        else split8_half(x[0], xh, xl);
#pragma unroll
        for (int to = 0; to < SD_HT; ++to) {
          f32x4 acc = zero, mx = zero;
          if (has0) {
            const f16x8 ah = w8[((to * KBX + w) * 2 + 0) * 64 + lane], al = w8[((to * KBX + w) * 2 + 1) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh, acc, 0, 0, 0);
            mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl, mx, 0, 0, 0);
            mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh, mx, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(mx[r], SD_LO_INV, acc[r]);
          }
          P[(to * SD_SPLIT_W + w) * 64 + lane] = acc;
        }
        if constexpr (REF != RF_NONE) wait_dma();  // this step's table has landed
        if (k + 1 < a.N) te_next = load_tile4(a.temb + static_cast<size_t>(k + 1) * SD_H, w, g);  // prefetch (after the wait: it counts loads in order)
        if constexpr (REF == RF_GMM) {
          f32x4 qk = zero;  // component k's partial quadratic form of this particle, own features
#pragma unroll
          for (int kc = 0; kc < SD_KREG; ++kc) {
            float part = 0.0f;
            if (kc < a.ref_k) {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int t = t0 + j;
                if (t < NT) {
                  const f32x4 m = tab4(kc, 0, j), iv = tab4(kc, 1, j);
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const float dl = x[j][r] - m[r];
                    part = __builtin_fmaf(dl * dl, iv[r], part);
                  }
                }
              }
            }
            qk[kc] = group_sum(part);
          }
          if (g == 0) *reinterpret_cast<f32x4*>(Lq + (w * 16 + p) * 4) = qk;
        }
      }
      __syncthreads();  // 1

      // ---- B: hidden tile w of the input layer: sum of the four partials + bias + time embedding, GELU ----
      {
        f32x4 h = b_in;
#pragma unroll
        for (int v = 0; v < SD_SPLIT_W; ++v) h = h + P[(w * SD_SPLIT_W + v) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = __builtin_fmaf(h[r], ns.inv_in, te[r]);  // (inv = 1: h + te bit for bit)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = gelu_fast(h[r]);
        A0[w * 64 + lane] = h;
      }
      float resp[SD_KREG] = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (REF == RF_GMM) {  // responsibilities from the summed quadratic forms (distr/gauss.py:97-107): every wave, same arithmetic
        f32x4 q = zero;
#pragma unroll
        for (int v = 0; v < SD_SPLIT_W; ++v) q = q + *reinterpret_cast<const f32x4*>(Lq + (v * 16 + p) * 4);
        float lp[SD_KREG], mxl = -INFINITY;
#pragma unroll
        for (int kc = 0; kc < SD_KREG; ++kc) {
          lp[kc] = -INFINITY;
          if (kc < a.ref_k) {
            lp[kc] = rc1[kc] + (((-0.5f * q[kc]) - a.ref_c1) - rc0[kc]);
            mxl = fmaxf(mxl, lp[kc]);
          }
        }
        float den = 0.0f;
#pragma unroll
        for (int kc = 0; kc < SD_KREG; ++kc) {
          resp[kc] = (kc < a.ref_k) ? expf(lp[kc] - mxl) : 0.0f;
          den += resp[kc];
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int kc = 0; kc < SD_KREG; ++kc) resp[kc] *= inv;
      }
      __syncthreads();  // 2

      // ---- C, D: the two hidden layers, output tile w each ----
#pragma unroll
      for (int layer = 0; layer < 2; ++layer) {
        const f32x4* Ain = layer == 0 ? A0 : A1;
        f32x4* Aout = layer == 0 ? A1 : A2;
        const int woff = layer == 0 ? sd_off_wh1(NT) : sd_off_wh2(NT);
        f32x4 in[SD_HT];
#pragma unroll
        for (int t = 0; t < SD_HT; ++t) in[t] = Ain[t * 64 + lane];
        f16x8 hh[2], hl[2];
        split_tiles<SD_HT>(in, hh, hl);
        const f16x8* wl = reinterpret_cast<const f16x8*>(lds + woff);
        f32x4 acc = layer == 0 ? b_h1 : b_h2, mx = zero;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const f16x8 ah = wl[((w * 2 + kb) * 2 + 0) * 64 + lane], al = wl[((w * 2 + kb) * 2 + 1) * 64 + lane];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hh[kb], acc, 0, 0, 0);
          mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hl[kb], mx, 0, 0, 0);
          mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, hh[kb], mx, 0, 0, 0);
        }
        const float inv_l = layer == 0 ? ns.inv_h1 : ns.inv_h2;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(mx[r], SD_LO_INV, acc[r]);
        if (inv_l != 1.0f) acc = acc * inv_l;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = gelu_fast(acc[r]);
        Aout[w * 64 + lane] = acc;
        __syncthreads();  // 3, 4
      }

      __builtin_amdgcn_s_setprio(0);
      // ---- E: output layer of the own feature tiles and everything element-wise ----
      float su2 = 0.0f, suz = 0.0f;
      if (has0) {
        f32x4 hid[SD_HT];
#pragma unroll
        for (int t = 0; t < SD_HT; ++t) hid[t] = A2[t * 64 + lane];
        const HidSplit hs = split_hidden(hid);
        const f16x8* wo = reinterpret_cast<const f16x8*>(lds + sd_off_wout(NT));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int t = t0 + j;
          if (t < NT) {  // wave-uniform
            f32x4 u = j == 0 ? b_o0 : b_o1, mx = zero;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              const f16x8 ah = wo[((t * 2 + kb) * 2 + 0) * 64 + lane], al = wo[((t * 2 + kb) * 2 + 1) * 64 + lane];
              u = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hs.h[kb], u, 0, 0, 0);
              mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hs.l[kb], mx, 0, 0, 0);
              mx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, hs.h[kb], mx, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = __builtin_fmaf(mx[r], SD_LO_INV, u[r]);
            if (ns.inv_out != 1.0f) u = u * ns.inv_out;
            if (a.clip_model > 0.0f) clamp_tile_rare(u, a.clip_model);  // ClippedCtrl (reparam.py:42)
            const f32x4 z = philox_normal4(pidx, static_cast<uint32_t>(k), static_cast<uint32_t>(4 * t + g), 0u, a.seed_lo, a.seed_hi);
            f32x4 rq = zero;  // reference score of this tile (eq/sdes.py:265-279, 329-345)
            if constexpr (REF == RF_GAUSS) {
              const f32x4 m = tab4(0, 0, j), iv = tab4(0, 1, j);
#pragma unroll
              for (int r = 0; r < 4; ++r) rq[r] = -((x[j][r] - m[r]) * iv[r]);
            }
            if constexpr (REF == RF_GMM) {
#pragma unroll
              for (int kc = 0; kc < SD_KREG; ++kc) {
                if (kc < a.ref_k) {
                  const f32x4 m = tab4(kc, 0, j), iv = tab4(kc, 1, j);
#pragma unroll
                  for (int r = 0; r < 4; ++r) rq[r] = __builtin_fmaf(resp[kc], (m[r] - x[j][r]) * iv[r], rq[r]);
                }
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float xv = x[j][r], uv = u[r];
              su2 = __builtin_fmaf(uv, uv, su2);
              if constexpr (lin) {  // eq/sdes.py:535-538
                float sc = uv;
                if constexpr (REF != RF_NONE) sc = rq[r] + uv;
                x[j][r] = __builtin_fmaf(c3, z[r], __builtin_fmaf(c2, sc, c1 * xv));
                suz = __builtin_fmaf(uv, z[r], suz);
              } else {              // losses/oc.py:277-284
                float acc = e1 * xv;  // (sim_kernel.hpp, EM update)
                if constexpr (REF != RF_NONE) acc = __builtin_fmaf(e3, rq[r], acc);
                x[j][r] = __builtin_fmaf(e4, z[r], __builtin_fmaf(e2, uv, acc));
                suz = __builtin_fmaf(uv, z[r], suz);
              }
            }
          }
        }
      }
      if constexpr (REF != RF_NONE) {  // every read of this step's table is done: fetch the next one
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < a.N) fetch_table(k + 1);
      }
      if (a.xs_out) {  // trajectory (return_traj, log-variance training): each wave stores its own feature tiles of state k + 1
        float* xs = a.xs_out + static_cast<size_t>(k + 1) * a.B * a.d;
        if (has0) store_quad(xs, trash, row, a.d, live, t0, g, x[0]);
        if (has1) store_quad(xs, trash, row, a.d, live, t1, g, x[1]);
      }
      // running cost and stochastic integral of the own features (the sums over features are linear: the four shares add up)
      su2 = group_sum(su2);
      rnd += lin ? c4 * su2 : (0.5f * su2) * c4;
      if (w == 0) rnd += c6;
      if (a.flags & SDENG_FLAG_ITO) {
        suz = group_sum(suz);
        rnd += c5 * suz;
      }
    }

    if (has0) store_quad(a.x_out, trash, row, a.d, live, t0, g, x[0]);
    if (has1) store_quad(a.x_out, trash, row, a.d, live, t1, g, x[1]);
    // the four shares of the log-weight, summed in a fixed order by wave 0 of the tile
    __syncthreads();
    if (g == 0) Lq[w * 16 + p] = rnd;
    __syncthreads();
    if (w == 0 && g == 0 && live) a.rnd_out[row] = ((Lq[p] + Lq[16 + p]) + Lq[32 + p]) + Lq[48 + p];
    __syncthreads();
  }
}

template <int NT, int REF, int FORM>
static int launch_split(const SimArgs& a, hipStream_t stream) {
  const size_t lds_bytes = static_cast<size_t>(sd_split_lds_bytes(NT));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_simulate_split<NT, REF, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  int grid = (a.ntiles + 1) / 2;
  grid = grid > 256 ? 256 : (grid < 1 ? 1 : grid);
  hipLaunchKernelGGL((k_simulate_split<NT, REF, FORM>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  return static_cast<int>(hipGetLastError());
}
// one launcher per feature-tile count: reference kind and form are picked at run time
template <int NT>
static int launch_split_nt(const SimArgs& a, int rf, hipStream_t stream) {
  const bool lin = a.form == SDENG_FORM_LIN;
  if (rf == RF_NONE) return lin ? launch_split<NT, RF_NONE, SDENG_FORM_LIN>(a, stream) : launch_split<NT, RF_NONE, SDENG_FORM_EM>(a, stream);
  if (rf == RF_GAUSS) return lin ? launch_split<NT, RF_GAUSS, SDENG_FORM_LIN>(a, stream) : launch_split<NT, RF_GAUSS, SDENG_FORM_EM>(a, stream);
  if (rf == RF_GMM) return lin ? launch_split<NT, RF_GMM, SDENG_FORM_LIN>(a, stream) : launch_split<NT, RF_GMM, SDENG_FORM_EM>(a, stream);
  return static_cast<int>(hipErrorInvalidValue);
}
#define SD_DEFINE_SPLIT(NT) \
  int sd_launch_split_##NT(const SimArgs& a, int rf, hipStream_t s) { return launch_split_nt<NT>(a, rf, s); }
