// Device building blocks of the simulate kernels (gfx950 only; see sim_common.hpp for the layout).
#pragma once
#include "sim_common.hpp"
#include "../../include/sdeng.h"

#define SD_INLINE __device__ __forceinline__

// ----------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------
// The four lanes p, p+16, p+32, p+48 hold the four feature groups of one particle: two cross-lane adds finish
// every per-particle reduction (log-weight increments, mixture logits).
SD_INLINE float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// exp of a non-positive argument / reciprocal of a sum of such terms, for softmax weights: the hardware exp2 and reciprocal
// (1 ulp each).  libm's expf spends ~15 instructions on overflow / denormal handling that cannot occur here, the IEEE division ~10.
SD_INLINE float exp_nonpos(float v) { return __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }
SD_INLINE float rcp_fast(float v) { return __builtin_amdgcn_rcpf(v); }

// torch.clip semantics: NaN stays NaN (fminf/fmaxf or v_med3 would swallow it).
SD_INLINE float clampf(float v, float m) {
  v = (v < -m) ? -m : v;
  v = (v > m) ? m : v;
  return v;
}

// torch.clip of one feature tile with a bound that almost never binds (1e4 / 1e5 in every config): one compare per element into
// a wave-wide mask, and the 4-instruction NaN-preserving clamp only when some lane is out of range or NaN
SD_INLINE void clamp_tile_rare(f32x4& v, float m) {
  bool out_of_range = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) out_of_range |= !(__builtin_fabsf(v[r]) <= m);
  if (__builtin_amdgcn_ballot_w64(out_of_range) != 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = clampf(v[r], m);
  }
}

SD_INLINE int feat(int t, int r, int g) { return 16 * t + 4 * g + r; }
// feat(t, r, g) < d with the lane-dependent part (4g) on one side only: the compare takes a scalar operand,
// so no per-element index register is ever materialised.
SD_INLINE bool feat_lt(int t, int r, int g4, int d) { return g4 < d - (16 * t + r); }
// The same for a kernel with NT feature tiles: NT = ceil(d / 16) exactly (sdeng_api.hip tiles_of), so every tile but the last
// holds live features only -- no mask instructions there (t is a compile-time constant after unrolling).  (The full-covariance
// reference kernels may run with one tile more than that; they never use this helper.)
template <int NT>
SD_INLINE bool feat_live(int t, int r, int g4, int d) { return (t < NT - 1) ? true : feat_lt(t, r, g4, d); }

// the 4 registers of feature tile t for this lane from a dense vector
SD_INLINE f32x4 load_tile4(const float* base, int t, int g) { return *reinterpret_cast<const f32x4*>(base + 16 * t + 4 * g); }

// ----------------------------------------------------------------------------------------------
// Split-f16 dense layer on the f16 matrix pipe (v_mfma_f32_16x16x32_f16, fp32 accumulate).
//
// Why: on gfx950 an MFMA and vector instructions of the waves sharing a SIMD do not overlap (microbenchmark
// tools/ubench/pipe_share.hip, profiles/r01_ubench_mfma_valu_serialize.log: times add), so the step costs
// T_mfma + T_valu and the FP32 MFMA (64 FLOP/clk/SIMD) was 46 % of it.  Writing each operand as
// v = hi + lo * 2^-11 with hi = f16(v), lo = f16((v - hi) * 2^11) (v - hi is exact in fp32) and summing the
// three products hi*hi + 2^-11 (hi*lo + lo*hi) in fp32 reproduces the fp32 GEMM to fp32 round-off (22+ bits
// per operand; measured rms error 1.00-1.06x that of an fp32 GEMM against fp64) at 3/16 of its matrix time.
//
// K-block kb covers feature tiles 2kb, 2kb+1: lane (p,g) contributes its 8 registers of those two tiles as the
// 8 k-values of its group, k-slot (g, j) <-> feature 16 (2kb + j/4) + 4 g + j%4; the weights are stored in the
// same permutation (k_pack_mlp).  `w` image: 16-byte vectors, index ((to*KB + kb)*2 + part)*64 + lane.
// ----------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define SD_LO_SCALE 2048.0f
#define SD_LO_INV 4.8828125e-04f

// One pair of values -> packed (hi, lo) halves in 6 instructions: v_cvt_pk_f16_f32 (round to nearest even), two
// v_fma_mix_f32 that subtract the f16 halves from the fp32 values exactly (v*1.0 - hi with the f16 operand read in
// place: no separate f16->f32 conversion), two scalings, one v_cvt_pk_f16_f32.
//
// NO INLINE ASM HERE.  Round 1 wrote these four instructions as asm statements.  LLVM's hazard recognizer
// runs its MFMA rules only for instructions it classifies as VALU; an INLINEASM node is not one, so the rule "a VALU
// write of a VGPR that an XDL MFMA issued < 3 wait states earlier reads as SrcC" (GCNHazardRecognizer
// checkMAIVALUHazards, 4-pass case) was never applied to them: wherever the allocator handed a just-dead accumulator
// register to one of these statements, the v_cvt_pk / v_fma_mix overwrote it 0-2 wait states after the MFMA -- while
// the MFMA's last pass (rows 12-15 = lanes 48-63) had not read it yet (tools/isa_hazard_scan.py finds the sites;
// for every compiler-visible VALU writer hipcc emits the `s_nop 2`).  That was the "lanes 48-63, one register"
// corruption at two waves per SIMD.  Written with builtins the same six instructions come out (the opaque SGPR 1.0
// keeps fma(a, 1, -hi) from being folded to a subtraction, which would need two separate f16->f32 conversions), and
// every one of them is visible to the hazard recognizer (tests/test_build_cpu.py keeps it that way).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
SD_INLINE float opaque_one() {
  float one = 1.0f;
  asm("" : "+s"(one));  // no instruction: only hides the constant from the optimizer (not volatile: CSE'd, one SGPR per kernel)
  return one;
}
SD_INLINE void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  const float one = opaque_one();
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  float d0 = __builtin_fmaf(a, one, -static_cast<float>(h[0]));
  float d1 = __builtin_fmaf(b, one, -static_cast<float>(h[1]));
  d0 *= SD_LO_SCALE;
  d1 *= SD_LO_SCALE;
  const f16x2 l = __builtin_convertvector(f32x2{d0, d1}, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
SD_INLINE void split8(const f32x4& t0, const f32x4& t1, f16x8& hi, f16x8& lo) {
  uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
  split_pair(t0[0], t0[1], h0, l0);
  split_pair(t0[2], t0[3], h1, l1);
  split_pair(t1[0], t1[1], h2, l2);
  split_pair(t1[2], t1[3], h3, l3);
  hi = __builtin_bit_cast(f16x8, u32x4{h0, h1, h2, h3});
  lo = __builtin_bit_cast(f16x8, u32x4{l0, l1, l2, l3});
}

// odd tile counts: the last K-block holds one live tile, its upper four k-slots are zero
SD_INLINE void split8_half(const f32x4& t0, f16x8& hi, f16x8& lo) {
  uint32_t h0, h1, l0, l1;
  split_pair(t0[0], t0[1], h0, l0);
  split_pair(t0[2], t0[3], h1, l1);
  hi = __builtin_bit_cast(f16x8, u32x4{h0, h1, 0u, 0u});
  lo = __builtin_bit_cast(f16x8, u32x4{l0, l1, 0u, 0u});
}

// one layer: out[to] = bias[to] (preloaded in `out`) + W in.  The activations are split K-block by K-block right
// before use, so only 8 packed registers of hi/lo pieces are live at a time.
// SCALED: the B operand is in * sg (sg per lane, i.e. per particle: the range-safe twin below), formed K-block by K-block so that no
// scaled copy of the whole input is ever live
template <int NTI, int TO, bool SCALED = false>
SD_INLINE void dense(const f32x4 (&in)[NTI], f32x4 (&out)[TO], const float* w, int lane, float sg = 1.0f) {
  constexpr int KB = (NTI + 1) / 2;
  const f16x8* w8 = reinterpret_cast<const f16x8*>(w);
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 mx[TO];
#pragma unroll
  for (int to = 0; to < TO; ++to) mx[to] = zero;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    f16x8 xh, xl;
    if constexpr (SCALED) {
      if (2 * kb + 1 < NTI) split8(in[2 * kb] * sg, in[2 * kb + 1 < NTI ? 2 * kb + 1 : 0] * sg, xh, xl);
      else split8_half(in[2 * kb] * sg, xh, xl);
    } else {
      if (2 * kb + 1 < NTI) split8(in[2 * kb], in[2 * kb + 1 < NTI ? 2 * kb + 1 : 0], xh, xl);
      else split8_half(in[2 * kb], xh, xl);
    }
    f16x8 ah[TO], al[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      ah[to] = w8[((to * KB + kb) * 2 + 0) * 64 + lane];
      al[to] = w8[((to * KB + kb) * 2 + 1) * 64 + lane];
    }
    // acc_hh += Whi Xhi ; acc_mx += Whi Xlo + Wlo Xhi ; consecutive MFMAs go to different accumulators
#pragma unroll
    for (int to = 0; to < TO; ++to) out[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[to], xh, out[to], 0, 0, 0);
#pragma unroll
    for (int to = 0; to < TO; ++to) mx[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[to], xl, mx[to], 0, 0, 0);
#pragma unroll
    for (int to = 0; to < TO; ++to) mx[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[to], xh, mx[to], 0, 0, 0);
  }
#pragma unroll
  for (int to = 0; to < TO; ++to)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[to][r] = __builtin_fmaf(mx[to][r], SD_LO_INV, out[to][r]);
}

// The same product with the activations already split (all K-blocks at once): used where one split feeds several
// products (CMCD: drift net, data logits and prior score all read the state).  `w8` may point to LDS or global.
template <int KB, int TO>
SD_INLINE void dense_pre(const f16x8 (&xh)[KB], const f16x8 (&xl)[KB], f32x4 (&out)[TO], f32x4 (&mx)[TO], const f16x8* w8, int lane) {
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    f16x8 ah[TO], al[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      ah[to] = w8[((to * KB + kb) * 2 + 0) * 64 + lane];
      al[to] = w8[((to * KB + kb) * 2 + 1) * 64 + lane];
    }
#pragma unroll
    for (int to = 0; to < TO; ++to) out[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[to], xh[kb], out[to], 0, 0, 0);
#pragma unroll
    for (int to = 0; to < TO; ++to) mx[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[to], xl[kb], mx[to], 0, 0, 0);
#pragma unroll
    for (int to = 0; to < TO; ++to) mx[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[to], xh[kb], mx[to], 0, 0, 0);
  }
}
// The same products with the A operands read from a GLOBAL-memory image (prior precision of the CMCD kernels, ...) through a buffer
// descriptor: a buffer load's address is descriptor base (SGPRs) + scalar offset + ONE 32-bit per-lane offset, so there is no per-lane
// 64-bit address per (tile, k-block) for the compiler to hoist out of the step loop -- it did that for all 2 * KB * NT of them with plain
// pointers (2 VGPRs each, every one spilled: 450-840 B of scratch per lane in the d = 128 CMCD kernels).  Out-of-range reads return 0.
SD_INLINE __amdgpu_buffer_rsrc_t image_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, static_cast<int>(bytes), 0x00020000);
}
template <int KB, int TO>
SD_INLINE void dense_pre_buf(const f16x8 (&xh)[KB], const f16x8 (&xl)[KB], f32x4 (&out)[TO], f32x4 (&mx)[TO], __amdgpu_buffer_rsrc_t img,
                             uint32_t tile_bytes, int lane) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int voff = lane * 16;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    f16x8 ah[TO], al[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) {
      const u32x4 vh = __builtin_amdgcn_raw_buffer_load_b128(img, voff, static_cast<int>(tile_bytes) + ((to * KB + kb) * 2 + 0) * 1024, 0);
      const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(img, voff, static_cast<int>(tile_bytes) + ((to * KB + kb) * 2 + 1) * 1024, 0);
      ah[to] = __builtin_bit_cast(f16x8, vh);
      al[to] = __builtin_bit_cast(f16x8, vl);
    }
#pragma unroll
    for (int to = 0; to < TO; ++to) out[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[to], xh[kb], out[to], 0, 0, 0);
#pragma unroll
    for (int to = 0; to < TO; ++to) mx[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[to], xl[kb], mx[to], 0, 0, 0);
#pragma unroll
    for (int to = 0; to < TO; ++to) mx[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[to], xh[kb], mx[to], 0, 0, 0);
  }
}
template <int TO>
SD_INLINE void fold_lo(f32x4 (&out)[TO], const f32x4 (&mx)[TO]) {
#pragma unroll
  for (int to = 0; to < TO; ++to)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[to][r] = __builtin_fmaf(mx[to][r], SD_LO_INV, out[to][r]);
}
template <int NT>
SD_INLINE void split_tiles(const f32x4 (&x)[NT], f16x8 (&xh)[(NT + 1) / 2], f16x8 (&xl)[(NT + 1) / 2]) {
#pragma unroll
  for (int kb = 0; kb < (NT + 1) / 2; ++kb) {
    if (2 * kb + 1 < NT) split8(x[2 * kb], x[2 * kb + 1 < NT ? 2 * kb + 1 : 0], xh[kb], xl[kb]);
    else split8_half(x[2 * kb], xh[kb], xl[kb]);
  }
}

// GELU for the step loop: one branch-free piece, 8 fma + exp2 + max + fma (the erf form above costs 2 polynomial
// pieces, a select and a sign transfer per element; the hidden layers apply 192 GELUs per particle-step and
// vector issue slots, not the matrix pipe, bound the kernel).  With P(t) = log2( Phi(-t) ) = log2(erfc(t/sqrt2)/2),
//     gelu(v) = v Phi(v) = max(v,0) - |v| * 2^P(|v|)
// P is a degree-8 weighted minimax fit on [0,6] (weight |v| Phi(-|v|), i.e. the absolute error of the result);
// beyond 6 the polynomial keeps decreasing, so 2^P underflows smoothly.  Max abs error against fp64
// x Phi(x): 3.5e-8 for |v|<0.5, 8e-8 for |v|<1.5, <= 0.6 ulp(v) above: at or below the rounding error of
// torch's own (0.5 v)(1 + erf(v/sqrt2)) in fp32 (tools/fit_gelu.py).
SD_INLINE float gelu_fast(float v) {
#ifdef SD_DBG_NOGELU
  return v * 0.5f;
#endif
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
  const float pt = __builtin_amdgcn_exp2f(r);
  return __builtin_fmaf(-t, pt, __builtin_fmaxf(v, 0.0f));
}

template <int T>
SD_INLINE void gelu_tiles(f32x4 (&v)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[t][r] = gelu_fast(v[t][r]);
}

// Layer scales of the packed drift net (sim_common.hpp SD_N_SCALES): inv_* = 2^-e of each layer, s_in = 2^e of the input layer.
// All 1.0 for a net whose weights sit in the f16 split's good range -- then nothing below changes a bit of the result.
struct NetScale {
  float inv_in, inv_h1, inv_h2, inv_out, s_in;
  SD_INLINE bool any() const { return (inv_in != 1.0f) | (inv_h1 != 1.0f) | (inv_h2 != 1.0f) | (inv_out != 1.0f); }
  SD_INLINE bool any_hidden() const { return (inv_in != 1.0f) | (inv_h1 != 1.0f) | (inv_h2 != 1.0f); }  // (the output layer is linear: un-scaled where it is produced)
};
SD_INLINE NetScale load_net_scale(const float* bias, int NT) {
  const float* sc = bias + 3 * 64 + 16 * NT;  // uniform address: scalar loads, loop invariant
  return NetScale{sc[4], sc[5], sc[6], sc[7], sc[0]};
}
// layer output *= 2^-e, behind ONE uniform test per layer (a multiply per element would cost the common case ~2 % for nothing)
template <int T>
SD_INLINE void unscale_tiles(f32x4 (&v)[T], float inv) {
  if (__builtin_expect(inv != 1.0f, 0)) {
    asm volatile("" ::: "memory");  // keeps the test a branch: x * 1.0 = x, so the compiler otherwise drops it and multiplies every time
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = v[t] * inv;
  }
}

// FourierMLP.forward (models/mlp.py:135-143) for a 16-particle tile, split in two so that the d-wide output
// never has to be live at once: mlp_hidden() runs input_embed + time embedding + the two hidden layers and
// returns gelu(h) (64 channels = 4 register tiles); mlp_out_tiles() produces OT 16-feature tiles of out_layer,
// which the caller consumes (clip, cost, integrator) before asking for the next.  `temb` = this step's time
// embedding [64] (hoisted: the reference recomputes the identical row for every particle, :136-137).
// (The step loop calls this plain form only for a net whose four layer scales are all 1 -- NetScale::any() false; a scaled net runs
// the range-safe twin below, which applies the scales: the common case carries no test per layer.)
template <int NT>
SD_INLINE void mlp_hidden(const f32x4 (&x)[NT], f32x4 (&a)[SD_HT], const float* lds, const float* bias, const float* temb,
                          int lane) {
  const int g = lane >> 4;
  f32x4 b[SD_HT];
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) a[t] = load_tile4(bias, t, g);  // b_in
  dense<NT, SD_HT>(x, a, lds + sd_off_win(NT), lane);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) a[t] = a[t] + load_tile4(temb, t, g);  // embed = embed_x + embed_t
  gelu_tiles<SD_HT>(a);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) b[t] = load_tile4(bias + 64, t, g);  // b_h1
  dense<SD_HT, SD_HT>(a, b, lds + sd_off_wh1(NT), lane);
  gelu_tiles<SD_HT>(b);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) a[t] = load_tile4(bias + 128, t, g);  // b_h2
  dense<SD_HT, SD_HT>(b, a, lds + sd_off_wh2(NT), lane);
  gelu_tiles<SD_HT>(a);
}

// The same with the layer scales applied (NetScale: weight matrices stored times 2^e) -- for the step loops that carry no
// range-safe twin (matrix-pipe / full-covariance mixture kernels, in-loop logistic regression: their own state products have none, and
// the twin cost them 4.5 % in spills).  Exact power-of-two scalings, one uniform test per layer: a net whose scales are all 1 gives the
// bits of mlp_hidden.
template <int NT>
SD_INLINE void mlp_hidden_scaled(const f32x4 (&x)[NT], f32x4 (&a)[SD_HT], const float* lds, const float* bias, const float* temb,
                                 int lane, const NetScale& ns) {
  const int g = lane >> 4;
  f32x4 b[SD_HT];
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) a[t] = load_tile4(bias, t, g);  // b_in 2^e
  dense<NT, SD_HT>(x, a, lds + sd_off_win(NT), lane);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) {
    const f32x4 te = load_tile4(temb, t, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) a[t][r] = __builtin_fmaf(te[r], ns.s_in, a[t][r]);  // + embed_t 2^e  (s_in = 1: a + te bit for bit)
  }
  unscale_tiles<SD_HT>(a, ns.inv_in);
  gelu_tiles<SD_HT>(a);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) b[t] = load_tile4(bias + 64, t, g);  // b_h1
  dense<SD_HT, SD_HT>(a, b, lds + sd_off_wh1(NT), lane);
  unscale_tiles<SD_HT>(b, ns.inv_h1);
  gelu_tiles<SD_HT>(b);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) a[t] = load_tile4(bias + 128, t, g);  // b_h2
  dense<SD_HT, SD_HT>(b, a, lds + sd_off_wh2(NT), lane);
  unscale_tiles<SD_HT>(a, ns.inv_h2);
  gelu_tiles<SD_HT>(a);
}

// ----------------------------------------------------------------------------------------------
// Range-safe twin of the drift net.  The split operands are f16: a state or an activation beyond 65 504 becomes inf there and the
// step would return NaN where the reference (fp32 GEMMs) stays finite.  The step loop detects that after the fact -- one compare per
// tile-step on an output every input of the particle feeds, sim_kernel.hpp -- and re-evaluates the net of that step here: every
// layer's B operand is the input times a per-PARTICLE power of two sigma that brings the row's largest entry below 2^10, the
// accumulator starts at zero and the output is (acc / sigma + bias) 2^-e.  Exact scalings; never taken by a healthy sampler.
// ----------------------------------------------------------------------------------------------
template <int T>
SD_INLINE float row_downscale(const f32x4 (&v)[T]) {
  float m = 0.0f;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, __builtin_fabsf(v[t][r]));  // (fmaxf drops NaNs: they poison the row by themselves)
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (!(m >= 1024.0f) || !(m <= 3.0e38f)) return 1.0f;
  int ex;
  frexpf(m, &ex);  // m = f 2^ex, f in [0.5, 1)
  return ldexpf(1.0f, 10 - ex);
}
template <int NTI, int TO>
SD_INLINE void dense_safe(const f32x4 (&in)[NTI], f32x4 (&out)[TO], const float* w, const float* bias_scaled, float inv, int lane) {
  const int g = lane >> 4;
  constexpr int KB = (NTI + 1) / 2;
  const float sg = row_downscale<NTI>(in), rs = 1.0f / sg;  // powers of two: exact
  // one output tile at a time (the input is split again for each): this path is cold, what matters is that it adds no register pressure
  // to the step loop it sits in (the three-waves-per-SIMD instantiations have 168 registers)
#pragma unroll
  for (int to = 0; to < TO; ++to) {
    f32x4 o1[1] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
    dense<NTI, 1, true>(in, o1, w + to * KB * 512, lane, sg);
    const f32x4 b = load_tile4(bias_scaled, to, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[to][r] = __builtin_fmaf(o1[0][r], rs, b[r]) * inv;
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int NT>
SD_INLINE void mlp_hidden_safe(const f32x4 (&x)[NT], f32x4 (&a)[SD_HT], const float* lds, const float* bias, const float* temb,
                               int lane, const NetScale& ns) {
  const int g = lane >> 4;
  f32x4 b[SD_HT];
  dense_safe<NT, SD_HT>(x, a, lds + sd_off_win(NT), bias, ns.inv_in, lane);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) a[t] = a[t] + load_tile4(temb, t, g);
  gelu_tiles<SD_HT>(a);
  dense_safe<SD_HT, SD_HT>(a, b, lds + sd_off_wh1(NT), bias + 64, ns.inv_h1, lane);
  gelu_tiles<SD_HT>(b);
  dense_safe<SD_HT, SD_HT>(b, a, lds + sd_off_wh2(NT), bias + 128, ns.inv_h2, lane);
  gelu_tiles<SD_HT>(a);
}

// mlp_hidden with the input layer fed from a pre-split state
template <int NT>
SD_INLINE void mlp_hidden_pre(const f16x8 (&xh)[(NT + 1) / 2], const f16x8 (&xl)[(NT + 1) / 2], f32x4 (&a)[SD_HT], const float* lds,
                              const float* bias, const float* temb, int lane, const NetScale& ns) {
  const int g = lane >> 4;
  f32x4 b[SD_HT], mx[SD_HT];
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) {
    const f32x4 bi = load_tile4(bias, t, g), te = load_tile4(temb, t, g);  // b_in 2^e + embed_t 2^e (fma(t, 1, b) = t + b bit for bit)
#pragma unroll
    for (int r = 0; r < 4; ++r) a[t][r] = __builtin_fmaf(te[r], ns.s_in, bi[r]);
    mx[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  dense_pre<(NT + 1) / 2, SD_HT>(xh, xl, a, mx, reinterpret_cast<const f16x8*>(lds + sd_off_win(NT)), lane);
  fold_lo<SD_HT>(a, mx);
  unscale_tiles<SD_HT>(a, ns.inv_in);
  gelu_tiles<SD_HT>(a);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) b[t] = load_tile4(bias + 64, t, g);  // b_h1
  dense<SD_HT, SD_HT>(a, b, lds + sd_off_wh1(NT), lane);
  unscale_tiles<SD_HT>(b, ns.inv_h1);
  gelu_tiles<SD_HT>(b);
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) a[t] = load_tile4(bias + 128, t, g);  // b_h2
  dense<SD_HT, SD_HT>(b, a, lds + sd_off_wh2(NT), lane);
  unscale_tiles<SD_HT>(a, ns.inv_h2);
  gelu_tiles<SD_HT>(a);
}

// the last hidden activation, split once per step for all output tiles
struct HidSplit {
  f16x8 h[2], l[2];
};
SD_INLINE HidSplit split_hidden(const f32x4 (&a)[SD_HT]) {
  HidSplit s;
  split_tiles<SD_HT>(a, s.h, s.l);
  return s;
}
template <int NT, int OT>
SD_INLINE void mlp_out_tiles(const HidSplit& hs, const float* lds, const float* bias, int t0, int lane, f32x4 (&u)[OT], float inv_out = 1.0f) {
  const int g = lane >> 4;
  f32x4 mx[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    u[o] = load_tile4(bias + 192, t0 + o, g);  // b_out (times the layer's scale)
    mx[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  dense_pre<2, OT>(hs.h, hs.l, u, mx, reinterpret_cast<const f16x8*>(lds + sd_off_wout(NT) + t0 * 2 * 512), lane);
  fold_lo<OT>(u, mx);
  unscale_tiles<OT>(u, inv_out);
}
// range-safe twin (mlp_hidden_safe): `hs` is the split of hid * sigma, `rs` = 1 / sigma of this lane's particle
template <int NT, int OT>
SD_INLINE void mlp_out_tiles_safe(const HidSplit& hs, float rs, const float* lds, const float* bias, int t0, int lane, f32x4 (&u)[OT],
                                  float inv_out) {
  const int g = lane >> 4;
  f32x4 mx[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) u[o] = mx[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  dense_pre<2, OT>(hs.h, hs.l, u, mx, reinterpret_cast<const f16x8*>(lds + sd_off_wout(NT) + t0 * 2 * 512), lane);
  fold_lo<OT>(u, mx);
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const f32x4 b = load_tile4(bias + 192, t0 + o, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) u[o][r] = __builtin_fmaf(u[o][r], rs, b[r]) * inv_out;
  }
}
// The plain output layer applied to the split of hid * sigma (range-safe step) returns acc = b 2^e + sigma 2^e (W a), b 2^e being the
// stored bias it starts from: (acc - b 2^e) / sigma + b 2^e, times 2^-e, is W a + b.  (acc - b 2^e is formed in fp32 from an fp32 sum that
// began with b 2^e: the difference is exact to an ulp of the larger term, like any fp32 evaluation of W a + b.)
template <int OT>
SD_INLINE void out_tiles_unscale(f32x4 (&u)[OT], const float* bias, int t0, int lane, float rs, float inv_out) {
  const int g = lane >> 4;
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const f32x4 b = load_tile4(bias + 192, t0 + o, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) u[o][r] = __builtin_fmaf(u[o][r] - b[r], rs, b[r]) * inv_out;
  }
}
// the last hidden activation of the safe path, split with its per-particle factor; returns 1 / sigma
SD_INLINE float split_hidden_safe(const f32x4 (&a)[SD_HT], HidSplit& s) {
  const float sg = row_downscale<SD_HT>(a);
  f32x4 sc[SD_HT];
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) sc[t] = a[t] * sg;
  split_tiles<SD_HT>(sc, s.h, s.l);
  return 1.0f / sg;
}

// ----------------------------------------------------------------------------------------------
// per-wave reference table in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no VGPR staging).  The copy for
// step k+1 is issued once the wave has read step k's table for the last time and lands under the next step's
// MFMA phase; the reader waits with s_waitcnt vmcnt(0) (nothing else orders a ds_read behind an LDS-DMA).
// ----------------------------------------------------------------------------------------------
// Every chunk uses the SAME M0 (the wave's slot base) and advances through the instruction's immediate offset,
// which applies to the global and the LDS address alike.  Rewriting M0 between the chunks (one s_add m0 per
// 1 KiB) is not safe on gfx950: under load the queued LDS-DMA of chunk i picked up the M0 written for chunk
// i+1 and landed 1 KiB off -- silently wrong tables in ~20 % of the tiles at full occupancy, none in a small
// launch (tools/probe_determinism.py, tests/test_gpu_fullsize.py).
SD_INLINE void dma_table_to_lds(const float* __restrict__ gsrc, float* lds_dst, int n_floats, int lane) {
  typedef __attribute__((address_space(1))) void gvoid;
  typedef __attribute__((address_space(3))) void lvoid;
  gvoid* g = (gvoid*)(gsrc + lane * 4);
  lvoid* l = (lvoid*)lds_dst;
  __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);  // 1 KiB per wave-instruction
  if (n_floats > 256) __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
  if (n_floats > 512) __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
  if (n_floats > 768) __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
  static_assert(SD_REFTAB_FLOATS <= 1024, "four 1 KiB chunks");
}
// One step's table of a larger mixture, copied once per WORKGROUP: wave w moves chunks [w*share, (w+1)*share).
SD_INLINE void dma_table_shared(const float* __restrict__ gsrc, float* lds_buf, int n_floats, int share, int wave, int lane) {
  typedef __attribute__((address_space(1))) void gvoid;
  typedef __attribute__((address_space(3))) void lvoid;
  const int f0 = wave * share * 256;  // first float of this wave's slice (wave-uniform)
  gvoid* g = (gvoid*)(gsrc + f0 + lane * 4);
  lvoid* l = (lvoid*)(lds_buf + f0);
  if (f0 < n_floats) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
  if (share > 1 && f0 + 256 < n_floats) __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
  if (share > 2 && f0 + 512 < n_floats) __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
  if (share > 3 && f0 + 768 < n_floats) __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
  static_assert(SD_SHARE_MAX == 4, "four 1 KiB chunks per wave");
}
#ifdef SD_DBG_WAITMORE
SD_INLINE void wait_dma() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_sleep 8\n\ts_nop 7" ::: "memory"); }
#else
SD_INLINE void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// ----------------------------------------------------------------------------------------------
// counter-based noise: Philox4x32-10 + Box-Muller (definition shared with oracle.philox_normal)
// ----------------------------------------------------------------------------------------------
SD_INLINE void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t (&o)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32x32->64 product per multiplier (v_mad_u64_u32) instead of a mul_hi/mul_lo pair: integer multiplies
    // are quarter-rate and were the single largest vector cost of the step
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
    const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
    // three-input xor in one instruction (gfx950 v_bitop3_b32, truth table 0x96); the compiler emits two v_xor_b32 for a ^ b ^ c
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
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

// (n + 0.5) * 2^-23 with n = bits >> 9 < 2^23: every step is exact in fp32, so the single fma below is the same number
SD_INLINE float u01(uint32_t bits) { return __builtin_fmaf(static_cast<float>(bits >> 9), 1.1920928955078125e-07f, 5.9604644775390625e-08f); }

// normals of features 4*jb .. 4*jb+3 of global particle `pidx` at step `step`.  Counter order
// (pidx, jb, step, stream): the first round multiplies c0 and c2 and XORs c1 into the c2 product, so the step
// product is scalar work, the particle product is the only loop-invariant (two registers), and no per-quad
// partial round can be hoisted out of the step loop.
SD_INLINE f32x4 philox_normal4(uint32_t pidx, uint32_t step, uint32_t jb, uint32_t stream, uint32_t k0, uint32_t k1) {
  uint32_t r[4];
#ifdef SD_DBG_NOPHILOX
  r[0] = pidx * 2654435761u + step; r[1] = jb * 40503u + step; r[2] = r[0] ^ 0x9E3779B9u; r[3] = r[1] + k0;
#else
  philox4x32_10(pidx, jb, step, stream, k0, k1, r);
#endif
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
// Initial particles: prior.sample((B,)) as a pure function of (seed, global particle, feature) -- the Philox
// normals of stream 1 at step 0.  IsotropicGauss.sample (distr/gauss.py:777): loc + scale * randn; Gauss.sample: loc + scale * z per
// feature; Delta.sample (distr/delta.py:31): loc.  One quad = features 4 jb .. 4 jb + 3 of one particle; pads are 0.
// Used by the sampler kernel k_sample_x0 (prep_kernels.hip), which runs ahead of the step loop when the caller passes no x0.
// ----------------------------------------------------------------------------------------------
#define SD_X0_STREAM 1u
SD_INLINE f32x4 x0_quad(const X0Dev& q, uint32_t pidx, int jb, int d, uint32_t k0, uint32_t k1) {
  f32x4 x = {0.0f, 0.0f, 0.0f, 0.0f};
  if (q.kind == SDENG_DIST_GAUSS_DIAG && q.scale == nullptr) {  // Delta: no draw
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = (4 * jb + r < d) ? q.loc[4 * jb + r] : 0.0f;
    return x;
  }
  const f32x4 z = philox_normal4(pidx, 0u, static_cast<uint32_t>(jb), SD_X0_STREAM, k0, k1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 4 * jb + r;
    float v;
    if (q.kind == SDENG_DIST_ISO_GAUSS) v = q.p0 + q.p1 * z[r];
    else v = (f < d) ? q.loc[f] + q.scale[f] * z[r] : 0.0f;
    x[r] = (f < d) ? v : 0.0f;
  }
  return x;
}
// ----------------------------------------------------------------------------------------------
// Gaussian-mixture score (distr/gauss.py:97-107 score_mog).
// tab: [K][2][dpad] (mean, 1/var); consts: [K][cstride] with [0] = 0.5*sum log var, [1] = log w_k.
// ----------------------------------------------------------------------------------------------
// component logit: log w_k + log N(x; m_k, v_k)   (distr/gauss.py:70-72, :103)
template <int NT>
SD_INLINE float gmm_logit(const f32x4 (&x)[NT], const float* __restrict__ tab, const float* __restrict__ consts, int cstride,
                          int k, float c1, int g) {
  constexpr int dpad = 16 * NT;
  const float* mp = tab + static_cast<size_t>(k) * 2 * dpad;
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 m = load_tile4(mp, t, g);
    const f32x4 iv = load_tile4(mp + dpad, t, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dl = x[t][r] - m[r];
      part = __builtin_fmaf(dl * dl, iv[r], part);
    }
  }
  part = group_sum(part);
  const float v = ((-0.5f * part) - c1) - consts[k * cstride + 0];
  return consts[k * cstride + 1] + v;
}

// any K: online softmax over components, score accumulated in a d-wide register array.  One pass over the
// table per component: q = (x - m)/var is kept from the logit's quadratic form (sum (x - m) q) and reused for
// the score term -p_k q, and the accumulator is rescaled only when some particle of the wave found a new
// maximum (exp(0) = 1 otherwise, so skipping the multiply is exact).
template <int NT>
SD_INLINE void gmm_score_accum(const f32x4 (&x)[NT], const float* __restrict__ tab, const float* __restrict__ consts,
                               int cstride, int K, float c1, int g, f32x4 (&acc)[NT], float& m_run, float& l_run) {
  constexpr int dpad = 16 * NT;
  for (int k = 0; k < K; ++k) {
    const float* mp = tab + static_cast<size_t>(k) * 2 * dpad;
    f32x4 q[NT];
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 m = load_tile4(mp, t, g);
      const f32x4 iv = load_tile4(mp + dpad, t, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dl = x[t][r] - m[r];
        q[t][r] = dl * iv[r];
        part = __builtin_fmaf(dl, q[t][r], part);
      }
    }
    part = group_sum(part);
    const float v = ((-0.5f * part) - c1) - consts[k * cstride + 0];
    const float lp = consts[k * cstride + 1] + v;
    const float m_new = fmaxf(m_run, lp);
    const float so = exp_nonpos(m_run - m_new);
    const float pk = exp_nonpos(lp - m_new);
    l_run = l_run * so + pk;
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(pk, -q[t][r], acc[t][r] * so);  // -(x-mean)/var
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(pk, -q[t][r], acc[t][r]);
    }
    m_run = m_new;
  }
}
// the online-softmax accumulator over a table in one piece (global memory, or one LDS chunk holding all K components);
// a table staged in several chunks calls begin / accum per chunk / end -- the same operations in the same order
template <int NT>
SD_INLINE void gmm_score_begin(f32x4 (&acc)[NT], float& m_run, float& l_run) {
  m_run = -INFINITY;
  l_run = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
template <int NT>
SD_INLINE void gmm_score_end(f32x4 (&acc)[NT], float l_run) {
  const float inv = rcp_fast(l_run);
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = acc[t] * inv;
}
template <int NT>
SD_INLINE void gmm_score(const f32x4 (&x)[NT], const float* __restrict__ tab, const float* __restrict__ consts,
                         int cstride, int K, float c1, int g, f32x4 (&acc)[NT]) {
  float m_run, l_run;
  gmm_score_begin<NT>(acc, m_run, l_run);
  gmm_score_accum<NT>(x, tab, consts, cstride, K, c1, g, acc, m_run, l_run);
  gmm_score_end<NT>(acc, l_run);
}

// Small-mixture variant (K <= SD_KREG; the reference's default n_modes is 4, conf/target/many_modes.yaml):
// component responsibilities p_k = softmax_k(logit_k) are computed once and kept in registers; the score of a
// feature tile is then assembled where it is consumed, sum_k p_k (m_k - x) / var_k, so no d-wide score array
// stays live across the output layer.
#define SD_KREG 4
// KC = SD_KREG: the component count is known to be SD_KREG (the reference's default, 4 modes): no per-component guards
template <int NT, int KC = 0>
SD_INLINE void gmm_resp(const f32x4 (&x)[NT], const float* __restrict__ tab, const float* __restrict__ consts,
                        int cstride, int K, float c1, int g, float (&p)[SD_KREG]) {
  if constexpr (KC == SD_KREG) K = SD_KREG;
  float lp[SD_KREG];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    lp[k] = -INFINITY;
    if (k < K) {
      lp[k] = gmm_logit<NT>(x, tab, consts, cstride, k, c1, g);
      mx = fmaxf(mx, lp[k]);
    }
  }
  float den = 0.0f;
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    p[k] = (k < K) ? exp_nonpos(lp[k] - mx) : 0.0f;
    den += p[k];
  }
  const float inv = rcp_fast(den);
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) p[k] *= inv;
}

template <int NT, int KC = 0>
SD_INLINE f32x4 gmm_score_tile(const f32x4 (&x)[NT], const float* __restrict__ tab, int K, int g, const float (&p)[SD_KREG], int t) {
  constexpr int dpad = 16 * NT;
  if constexpr (KC == SD_KREG) K = SD_KREG;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    if (k < K) {
      const float* mp = tab + static_cast<size_t>(k) * 2 * dpad;
      const f32x4 m = load_tile4(mp, t, g);
      const f32x4 iv = load_tile4(mp + dpad, t, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(p[k], (m[r] - x[t][r]) * iv[r], acc[r]);
    }
  }
  return acc;
}

// The same when every component has the same variance vector (the reference's default initialisation, and then
// true of every noised marginal): sum_k p_k (m_k - x)/var = (sum_k p_k m_k - x)/var, one fma per component and element.
template <int NT, int KC = 0>
SD_INLINE f32x4 gmm_score_tile_shared_var(const f32x4 (&x)[NT], const float* __restrict__ tab, int K, int g, const float (&p)[SD_KREG], int t) {
  constexpr int dpad = 16 * NT;
  if constexpr (KC == SD_KREG) K = SD_KREG;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    if (k < K) {
      const f32x4 m = load_tile4(tab + static_cast<size_t>(k) * 2 * dpad, t, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(p[k], m[r], acc[r]);
    }
  }
  const f32x4 iv = load_tile4(tab + dpad, t, g);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (acc[r] - x[t][r]) * iv[r];
  return acc;
}

// Exactly SD_KREG components with one variance vector, centred table (k_ref_tables, `centred`): with c the centre of the noised means,
// y = x - c and a_k = (m_k - c)/var,
//     logit_k = b_k + <y, a_k>   (the quadratic term -0.5 sum y^2/var is common to all components and drops out of the softmax;
//                                  b_k = log w_k - 0.5 sum (m_k - c)^2/var)
//     score   = sum_k p_k (m_k - x)/var = sum_k p_k a_k - y/var                      (sum_k p_k = 1)
// one fma per element and component for the logits and one for the score: 352 instead of 576 vector instructions per tile-step at
// d = 128.  Centring keeps the magnitudes of the dot products at those of the mode separations.
// Table rows: [k][0] = a_k, [0][1] = 1/var, [1][1] = c; consts[k][1] = b_k.
template <int NT>
SD_INLINE void gmm_resp_centred(const f32x4 (&x)[NT], const float* __restrict__ tab, const float* __restrict__ consts, int g,
                                float (&p)[SD_KREG]) {
  constexpr int dpad = 16 * NT;
  float lp[SD_KREG];
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) lp[k] = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 y = x[t] - load_tile4(tab + 3 * dpad, t, g);
#pragma unroll
    for (int k = 0; k < SD_KREG; ++k) {
      const f32x4 ak = load_tile4(tab + static_cast<size_t>(k) * 2 * dpad, t, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) lp[k] = __builtin_fmaf(y[r], ak[r], lp[k]);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    lp[k] = consts[2 * k + 1] + group_sum(lp[k]);
    mx = fmaxf(mx, lp[k]);
  }
  float den = 0.0f;
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    p[k] = exp_nonpos(lp[k] - mx);
    den += p[k];
  }
  const float inv = rcp_fast(den);
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) p[k] *= inv;
}
template <int NT>
SD_INLINE f32x4 gmm_score_tile_centred(const f32x4 (&x)[NT], const float* __restrict__ tab, int g, const float (&p)[SD_KREG], int t) {
  constexpr int dpad = 16 * NT;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < SD_KREG; ++k) {
    const f32x4 ak = load_tile4(tab + static_cast<size_t>(k) * 2 * dpad, t, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(p[k], ak[r], acc[r]);
  }
  const f32x4 iv = load_tile4(tab + dpad, t, g), c = load_tile4(tab + 3 * dpad, t, g);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(c[r] - x[t][r], iv[r], acc[r]);
  return acc;
}

// Gaussian (one component) score of one tile: -(x - mean)/var  (distr/gauss.py:124-126)
template <int NT>
SD_INLINE f32x4 gauss_score_tile(const f32x4 (&x)[NT], const float* __restrict__ tab, int g, int t) {
  constexpr int dpad = 16 * NT;
  const f32x4 m = load_tile4(tab, t, g);
  const f32x4 iv = load_tile4(tab + dpad, t, g);
  f32x4 out;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r] = -((x[t][r] - m[r]) * iv[r]);
  return out;
}

// Rings target (distr/rings.py:100-109), d = 2: both coordinates sit in registers 0,1 of the g = 0 lanes.
//   n = |x| + eps;  score = x * (score_radius(n)/n - 1/n^2),  score_radius = score_mog over the radii (gauss.py:97-107)
// Lanes of the other feature groups (g != 0) hold pad features -- which carry noise in Philox mode -- and return 0.
SD_INLINE f32x4 rings_score(const f32x4& x, const DistDev& ds, int g) {
  const float n = __builtin_sqrtf(x[0] * x[0] + x[1] * x[1]) + 1e-7f;
  const float var = ds.p0 * ds.p0;
  float wsum = 0.0f;
  for (int k = 0; k < ds.k; ++k) wsum += ds.aux1[k];
  const float cn = 0.5f * 1.8378770664093453f + 0.5f * logf(var);  // 0.5*1*log(2 pi) + 0.5*log(var)
  float lp[8], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    lp[k] = -INFINITY;
    if (k < ds.k) {
      const float dl = n - ds.aux0[k];
      lp[k] = logf(ds.aux1[k] / wsum) + ((-0.5f * ((dl * dl) / var)) - cn);
      mx = fmaxf(mx, lp[k]);
    }
  }
  float den = 0.0f, acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < ds.k) {
      const float e = expf(lp[k] - mx);
      den += e;
      acc += e * ((n - ds.aux0[k]) / var);
    }
  }
  const float sr = -(acc / den);
  const float f = (sr / n) - (1.0f / (n * n));
  return g == 0 ? f32x4{x[0] * f, x[1] * f, 0.0f, 0.0f} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// phi^4 lattice neighbours of tile t (sites 16 t + 4 g + 0..3).  Lane (p, g-1) holds the site left of the
// group, lane (p, g+1) the site right of it; across a tile border they come from the neighbouring tile.
template <int NT>
SD_INLINE void phi4_edges(const f32x4 (&x)[NT], int t, int g, int lane, float& l_edge, float& r_edge) {
  const int up = (lane + 48) & 63, dn = (lane + 16) & 63;  // lanes (p, g-1) and (p, g+1), wrapping
  const float l_same = __shfl(x[t][3], up, 64);
  const float r_same = __shfl(x[t][0], dn, 64);
  float l_prev = 0.0f, r_next = 0.0f;
  if (t > 0) l_prev = __shfl(x[t > 0 ? t - 1 : 0][3], up, 64);            // g == 0: site 16t-1 = (t-1, r=3, g=3)
  if (t < NT - 1) r_next = __shfl(x[t < NT - 1 ? t + 1 : t][0], dn, 64);  // g == 3: site 16t+16 = (t+1, r=0, g=0)
  l_edge = (g == 0) ? l_prev : l_same;
  r_edge = (g == 3) ? r_next : r_same;
}

// PhiFour.score = -beta * grad_U  (distr/phi_four.py:81-96); p0=a, p1=b, p2=beta
template <int NT>
SD_INLINE void phi4_score(const f32x4 (&x)[NT], const DistDev& ds, int d, int g, int lane, f32x4 (&acc)[NT]) {
  const float coef = ds.p0 * static_cast<float>(d);
  // -beta * [ (b - x (1 - x^2)) / coef + coef (2 x - x_r - x_l) ]  =  A (x^3 - x) + K0 + C (2 x - x_r - x_l)
  // with A = -beta / coef, K0 = A b, C = -beta coef formed once per call: 6 fused instructions per element instead of 11 plain ones
  // (the reference divides by coef and multiplies by beta element by element; each fused term is rounded once where the reference
  // rounds two or three times, so the two agree to a few ulp of the larger term -- within the parity tolerance, tests/test_gpu_units.py)
  const float A = -ds.p2 / coef, K0 = A * ds.p1, C = -ds.p2 * coef;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float le, re;
    phi4_edges<NT>(x, t, g, lane, le, re);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xv = x[t][r];
      const float xl = (r == 0) ? le : x[t][r > 0 ? r - 1 : 0];
      const float xr = (r == 3) ? re : x[t][r < 3 ? r + 1 : 3];
      const float cub = __builtin_fmaf(xv * xv, xv, -xv);                      // x^3 - x
      const float lap = __builtin_fmaf(2.0f, xv, -xr) - xl;                     // 2 x - x_r - x_l
      const float sc = __builtin_fmaf(C, lap, __builtin_fmaf(A, cub, K0));
      acc[t][r] = feat_live<NT>(t, r, 4 * g, d) ? sc : 0.0f;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Tile-row I/O without control flow: masked lanes (dead rows, pad features) read element 0 of the array
// and discard it, or write to a per-lane dump slot (`trash`, 4 floats per lane) instead of the array.
// Only trajectory start/end, injected noise (parity mode) and trajectory dumps come through here.
// ----------------------------------------------------------------------------------------------
SD_INLINE f32x4 load_quad(const float* __restrict__ src, uint32_t row, int d, bool live, int t, int g) {
  const size_t base = static_cast<size_t>(row) * d + 16 * t + 4 * g;
  f32x4 z;
  if ((d & 3) == 0) {
    const bool ok = live && feat_lt(t, 0, 4 * g, d);
    z = *reinterpret_cast<const f32x4*>(src + (ok ? base : 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = ok ? z[r] : 0.0f;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = live && feat_lt(t, r, 4 * g, d);
      const float b = src[ok ? base + r : 0];
      z[r] = ok ? b : 0.0f;
    }
  }
  return z;
}

SD_INLINE void store_quad(float* __restrict__ dst, float* __restrict__ trash, uint32_t row, int d, bool live, int t, int g, f32x4 v) {
  const size_t base = static_cast<size_t>(row) * d + 16 * t + 4 * g;
  if ((d & 3) == 0) {
    const bool ok = live && feat_lt(t, 0, 4 * g, d);
    *reinterpret_cast<f32x4*>(ok ? dst + base : trash) = v;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = live && feat_lt(t, r, 4 * g, d);
      *(ok ? dst + base + r : trash) = v[r];
    }
  }
}

template <int NT>
SD_INLINE void load_rows(const float* __restrict__ src, uint32_t row, int d, bool live, int g, f32x4 (&v)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) v[t] = load_quad(src, row, d, live, t, g);
}
template <int NT>
SD_INLINE void store_rows(float* __restrict__ dst, float* __restrict__ trash, uint32_t row, int d, bool live, int g,
                          const f32x4 (&v)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) store_quad(dst, trash, row, d, live, t, g, v[t]);
}
// ----------------------------------------------------------------------------------------------
// CMCD building blocks (eq/sdes.py:101-110, distr/logistic_regression.py, distr/gauss.py:129-135)
// ----------------------------------------------------------------------------------------------
// Logistic-regression data for the CMCD kernel live in LDS as two packed split-f16 A-operand images of the
// augmented design matrix Xa = [X | 1] (the ones column carries the intercept, logistic_regression.py:52-55):
//   logits image: blocks [row tile][feature K-block]   (out = data rows, sum over features)      logits = Xa w
//   grad   image: blocks [feature tile][row K-block]    (out = features,  sum over data rows)     grad   = Xa^T r
// both in the (to, kb, part, lane, 8 halves) layout of k_pack_mlp; data rows are padded with zero rows.
__host__ __device__ inline int sd_lr_row_tiles(int n) { return (n + 15) / 16; }
__host__ __device__ inline int sd_lr_row_kb(int n) { return (n + 31) / 32; }
__host__ __device__ inline int sd_lr_logit_floats(int NT, int n) { return sd_lr_row_tiles(n) * sd_kb(NT) * 512; }
__host__ __device__ inline int sd_lr_grad_floats(int NT, int n) { return NT * sd_lr_row_kb(n) * 512; }
__host__ __device__ inline int sd_lr_floats(int NT, int n) { return sd_lr_logit_floats(NT, n) + sd_lr_grad_floats(NT, n); }

// Per-datum factor of the logistic-regression score.  The reference differentiates
// sigmoid -> clip(thr) -> probs_to_logits(clamp eps) -> BCE-with-logits by autograd
// (distr/logistic_regression.py:41-61, distr/base.py:146-154).  With p = sigmoid(logit) inside both clamps the
// chain collapses: logits2 = logit(p), sigmoid(logits2) = p and d logits2/dp * dp/dlogit = 1, so the factor is
// y - p; outside a clamp the gradient is zero.  (k_dist_eval keeps the long form; the two agree to fp32
// round-off, tests/test_gpu_units.py.)
SD_INLINE float logreg_residual(float logit, float y, float p_lo, float p_hi) {
  const float e = __builtin_amdgcn_exp2f(logit * -1.4426950408889634f);
  const float p = __builtin_amdgcn_rcpf(1.0f + e);
  return (p >= p_lo && p <= p_hi) ? (y - p) : 0.0f;
}

// score of the logistic-regression posterior for a 16-particle tile: prior part + Xa^T (y - sigmoid(Xa w)), both
// products on the split-f16 matrix path with the A operands read from the two LDS images at `images`.
// PADS_ZERO: the caller keeps the pad features of x at exactly 0 (the CMCD kernel); otherwise they may hold noise and are masked here.
template <int NT, bool PADS_ZERO = false>
SD_INLINE void logreg_score(const f32x4 (&x)[NT], const f16x8 (&xh)[(NT + 1) / 2], const f16x8 (&xl)[(NT + 1) / 2], const LogregDev& lr,
                            int d, const float* images, int lane, f32x4 (&ts)[NT]) {
  constexpr int KB = (NT + 1) / 2;
  const int g = lane >> 4;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  const f16x8* im_logit = reinterpret_cast<const f16x8*>(images);
  const f16x8* im_grad = reinterpret_cast<const f16x8*>(images + sd_lr_logit_floats(NT, lr.n_rows));
  const int row_kb = sd_lr_row_kb(lr.n_rows), row_tiles = sd_lr_row_tiles(lr.n_rows);
  f32x4 tm[NT];
  // Prior part, with as few per-feature masks as possible: only the last tile can hold pad features (NT = ceil(d / 16)) -- and in the
  // CMCD kernel the pads of x are exactly 0 (x0 pads, masked noise, zero weight / image rows), so -x / s^2 is 0 there by itself; the
  // intercept (feature d - 1) is patched in the one tile that holds it.  The 2 x 16 lane masks of a fully masked form are loop
  // invariants: the compiler kept them in 64 SGPRs across the step loop, ran out of scalar registers and paid a v_readlane pair +
  // wait states per use (110 in the cfg-4 step loop).
  const int fi = d - 1, ti = fi >> 4, ri = fi & 15;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ts[t][r] = -x[t][r] * lr.inv_w_scale2;   // logistic_regression.py:72
      if constexpr (!PADS_ZERO) ts[t][r] = feat_live<NT>(t, r, 4 * g, d) ? ts[t][r] : 0.0f;
    }
    if (t == ti) {  // wave-uniform
#pragma unroll
      for (int r = 0; r < 4; ++r) ts[t][r] = (4 * g + r == ri) ? -(x[t][r] - lr.c_mean) * lr.inv_c_scale2 : ts[t][r];  // :74
    }
    tm[t] = zero;
  }
  for (int pr = 0; pr < row_kb; ++pr) {  // 32 data rows per pass = one K-block of the Xa^T r product
    asm volatile("" ::: "memory");       // keeps the image reads inside the step loop (see cmcd_kernel.hpp)
    f32x4 lg[2], lm[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      lg[o] = zero;
      lm[o] = zero;
      const int tile = 2 * pr + o;
      if (tile < row_tiles) {  // wave-uniform
        f32x4 acc[1] = {zero}, mx[1] = {zero};
        dense_pre<KB, 1>(xh, xl, acc, mx, im_logit + static_cast<size_t>(tile) * KB * 2 * 64, lane);
        lg[o] = acc[0];
        lm[o] = mx[0];
      }
      const f32x4 yv = load_tile4(lr.y_pad, tile, g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        lg[o][r] = logreg_residual(__builtin_fmaf(lm[o][r], SD_LO_INV, lg[o][r]), yv[r], lr.p_lo, lr.p_hi);  // pad rows: Xa row = 0
    }
    f16x8 rh, rl;
    split8(lg[0], lg[1], rh, rl);
#pragma unroll
    for (int to = 0; to < NT; ++to) {
      const f16x8 ah = im_grad[((to * row_kb + pr) * 2 + 0) * 64 + lane];
      const f16x8 al = im_grad[((to * row_kb + pr) * 2 + 1) * 64 + lane];
      ts[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, rh, ts[to], 0, 0, 0);
      tm[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, rl, tm[to], 0, 0, 0);
      tm[to] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, rh, tm[to], 0, 0, 0);
    }
  }
  fold_lo<NT>(ts, tm);
}

