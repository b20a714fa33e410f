// Shared device/host definitions of the MI355X (gfx950) simulate engine.
//
// Wave-tile layout used by every kernel in this directory
// --------------------------------------------------------
// One wavefront (64 lanes) owns a tile of 16 particles.  Lane l holds particle p = l & 15 and the feature
// group g = l >> 4.  A [16 particles x 16 features] block of state lives in ONE f32x4 register group per lane:
// register r of lane (p,g) is feature
//
//        feat(t, r, g) = 16 t + 4 g + r                 t = feature tile, r = 0..3
//
// which is exactly the C/D layout of the 16x16 MFMAs (v_mfma_f32_16x16x32_f16 here) when the product is computed as Y^T = W * X^T
// (rows = output features, columns = particles).  Because an MFMA's summation index may be permuted as long
// as both operands agree, that same register group is directly the B operand of the next layer (k-step r of
// tile t <-> feature feat(t,r,g)), so the whole [d -> 64 -> 64 -> 64 -> d] drift net runs out of registers with
// no cross-lane traffic; the A operands (weights) are pre-permuted once into an LDS image by k_pack_mlp.
//
// Why 16-particle tiles (an earlier build used 32x32x2 MFMA, 32 particles per wave): the state of a tile is
// half as many registers per lane (d=128: 32 instead of 64), so the whole step loop at d=128 fits the 256-VGPR
// budget of 2 waves per SIMD with NO scratch traffic (208 VGPRs, 0 spills).  Measured on MI355X, cfg 2:
// 4 waves/SIMD (128 VGPRs, spills) 10.2 ms, 2 waves/SIMD 7.6 ms, 1 wave/SIMD 9.6 ms.  On gfx950 MFMA and
// VALU issue of the waves of one SIMD do not overlap (tools/ubench/pipe_share.hip), so extra occupancy buys
// nothing once memory latency is covered; instruction count is what matters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SD_H 64            // hidden channels (models/mlp.py: channels=64)
#define SD_HT 4            // hidden channel tiles (64 / 16)
#ifndef SD_WAVES
#define SD_WAVES 8         // waves per workgroup (2 per SIMD, 256-VGPR budget)
#endif
#define SD_THREADS (SD_WAVES * 64)
#define SD_WAVES_MAX 12    // step-loop instantiations that fit the 168-register budget run 3 waves per SIMD (sim_kernel.hpp sd_waves_of)

// Packed drift-net image.  The GEMMs run on the f16 matrix pipe as a two-piece split (see sim_device.hpp,
// dense_f16x2): every weight is stored as hi = f16(w) and lo = f16((w - hi) * 2^11).  One block = one
// (16-output tile, 32-input K-block) = 64 lanes x 8 halves for hi, then the same for lo = 2 KiB = 512 floats.
//   W_in [4][KB(NT)] blocks, W_h1 [4][2], W_h2 [4][2], W_out [NT][2]; then b_in, b_h1, b_h2 (64 each), b_out (16*NT)
__host__ __device__ inline int sd_kb(int NT) { return (NT + 1) / 2; }  // 32-feature K-blocks covering NT tiles
__host__ __device__ inline int sd_off_win(int NT) { return 0; }
__host__ __device__ inline int sd_off_wh1(int NT) { return 4 * sd_kb(NT) * 512; }
__host__ __device__ inline int sd_off_wh2(int NT) { return sd_off_wh1(NT) + 4 * 2 * 512; }
__host__ __device__ inline int sd_off_wout(int NT) { return sd_off_wh2(NT) + 4 * 2 * 512; }
__host__ __device__ inline int sd_lds_weight_floats(int NT) { return sd_off_wout(NT) + NT * 2 * 512; }
__host__ __device__ inline int sd_off_bias(int NT) { return sd_lds_weight_floats(NT); }
// Power-of-two prescale of the four weight matrices (k_weight_scales, prep_kernels.hip): the f16 split v = hi + lo 2^-11 carries fp32's
// 22+ bits only while |v| sits in f16's NORMAL range, so a matrix whose largest entry is below 2^-10 (the reference initialises the last
// layer at ~1e-7, models/utils.py:7-22: f16-subnormal) or at / above 2^14 is stored as W * 2^e with max |W| 2^e in [1, 2), its bias as
// b * 2^e, and the layer's output is multiplied by 2^-e (exact).  Matrices already in range keep e = 0 and the old bits.
// Behind the biases: scale[4] = 2^e per layer (in, h1, h2, out), then inv[4] = 2^-e.
#define SD_N_SCALES 8
__host__ __device__ inline int sd_off_scales(int NT) { return sd_off_bias(NT) + 3 * 64 + 16 * NT; }
__host__ __device__ inline int sd_pack_floats(int NT) { return sd_off_scales(NT) + SD_N_SCALES; }
// per-wave LDS copy of one step's reference table [K][2][dpad]; used while K*2*dpad <= SD_REFTAB_FLOATS
#define SD_REFTAB_FLOATS 1024
// workgroup-shared copy of a larger mixture's table (RF_GMM_BIG): `share` chunks of 1 KiB per wave, two buffers
#define SD_SHARE_MAX 4
__host__ __device__ inline int sd_share_buf_floats(int share) { return share * SD_WAVES * 256; }
__host__ __device__ inline int sd_lds_total_bytes(int NT, bool with_ref, int waves = SD_WAVES) {
  return (sd_lds_weight_floats(NT) + (with_ref ? waves * SD_REFTAB_FLOATS : 0)) * 4;
}

// device-side view of a distribution (tables prepared by k_dist_tables)
struct DistDev {
  int kind;
  int k;
  const float* tab;     // GMM/GAUSS_DIAG: [k][2][dpad]  (mean, 1/var); GAUSS_FULL: precision [d,d]
  const float* consts;  // GMM/GAUSS_DIAG: [k][4]
  const float* aux0;    // LOGREG: X; GAUSS_FULL: loc
  const float* aux1;    // LOGREG: y; GAUSS_FULL: L^-1
  float p0, p1, p2, p3;
  float clip;
};

// logistic-regression target, in-kernel form (images built by k_logreg_images; see sim_device.hpp)
struct LogregDev {
  const float* image;       // global copy of the two LDS images: logits image, then grad image
  const float* y_pad;       // [32 * row K-blocks] labels (0 on pad rows)
  float inv_w_scale2, c_mean, inv_c_scale2;   // 1/weight_scale^2, intercept_mean, 1/intercept_scale^2
  float p_lo, p_hi;         // sigmoid range with non-zero gradient (clip threshold and eps clamp)
  int n_rows;               // data rows n (0: no logistic-regression target)
  int in_lds;               // 1: the two images fit behind the drift-net weights in LDS; 0: the kernel reads them through L2
};

// initial particles drawn by the engine (k_sample_x0; sdeng_desc.x_in == NULL): x0 = loc + scale * z, z = Philox stream 1 at step 0
struct X0Dev {
  int kind;            // ISO_GAUSS / GAUSS_DIAG
  const float* loc;    // GAUSS_DIAG [d]
  const float* scale;  // GAUSS_DIAG [d], or nullptr: x0 = loc (Delta)
  float p0, p1;        // ISO_GAUSS: loc, scale
};

struct SimArgs {
  int form;
  unsigned flags;
  int B, d, N;
  long long particle0;
  unsigned seed_lo, seed_hi;
  const float* coef;      // [N][16]
  const float* x_in;
  float* x_out;
  float* rnd_out;
  float* xs_out;
  const float* noise_in;
  const float* wpack;     // packed drift-net image (global copy)
  const float* temb;      // [N][64] time embedding of the drift net, per step
  const float* stheta;    // [N] clipped score_model(t) or nullptr
  int ctrl_kind;
  float clip_model, clip_score, scale_score;
  int ref_k;
  int ref_share;            // RF_GMM_BIG: > 0 = 1 KiB chunks per wave of the workgroup-shared, double-buffered table copy
  int ref_kc;               //   components per staged piece of the table (>= ref_k: the whole table in one piece)
  const float* ref_tab;     // [N][K][2][dpad]
  const float* ref_consts;  // [N][K][2]  (0.5*sum log var, log w)
  const float* ref_mean;      // RF_GMM_FULL: [N][K][dpad] noised means (ref_tab then holds the precision images [N][K][NT*KB*512])
  const float* ref_same_var;  // [1] 1.0: all components share one variance vector (written by k_ref_tables)
  float ref_c1;             // 0.5*d*log(2*pi)
  DistDev target;         // in-loop target score (ScoreCtrl / LerpCtrl / CMCD)
  DistDev prior;          // LerpCtrl / CMCD prior score
  const float* rnd_init;  // [B] initial log-weight (log p_prior(x0)) or nullptr
  float* trash;           // [SD_THREADS*4] dump slots for masked stores
  float cmcd_g, cmcd_clip;
  int ntiles;             // ceil(B / 16)
  LogregDev lr;           // in-loop logistic-regression score (ScoreCtrl on a LOGREG target, CMCD)
};
