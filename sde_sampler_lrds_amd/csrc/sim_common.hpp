// Shared device/host definitions of the MI355X (gfx950) simulate engine.
//
// Wave-tile layout used by every kernel in this directory
// --------------------------------------------------------
// One wavefront (64 lanes) owns a tile of 32 particles.  Lane l holds particle p = l & 31 and the
// "half" h = l >> 5.  A [32 particles x 32 features] block of state lives in ONE f32x16 register
// group per lane: register r of lane (p,h) is feature
//
//        feat(t, r, h) = 32 t + 8 (r >> 2) + 4 h + (r & 3)           t = feature tile, r = 0..15
//
// which is exactly the C/D layout of v_mfma_f32_32x32x2_f32 when the product is computed as
// Y^T = W * X^T (rows = output features, columns = particles).  Because an MFMA's summation index is
// free to be permuted as long as both operands agree, that same register group is directly the
// B operand of the next layer (k-step r of tile t <-> feature feat(t,r,h)), so the whole
// [d -> 64 -> 64 -> 64 -> d] drift net runs out of registers with no cross-lane traffic; the A operands
// (weights) are pre-permuted once into an LDS image by k_pack_mlp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SD_H 64            // hidden channels (models/mlp.py: channels=64)
#ifndef SD_WAVES
#define SD_WAVES 8         // waves per workgroup (2 per SIMD)
#endif
#define SD_THREADS (SD_WAVES * 64)

__host__ __device__ inline int sd_lds_floats(int DT) {
  // packed image: W_in (2*DT tile pairs) + W_h1 + W_h2 (4 each) + W_out (DT*2), 1024 floats per (out-tile,in-tile)
  // pair, then b_in, b_h1, b_h2 (64 each) and b_out (32*DT).  The weights live in LDS; the biases are read from
  // the global copy (a few hundred bytes, L1-resident) so that LDS keeps room for per-wave reference tables.
  return (2 * DT + 4 + 4 + 2 * DT) * 1024 + 3 * 64 + 32 * DT;
}
// per-wave LDS copy of one step's reference table [K][2][dpad]; available while K*2*dpad <= SD_REFTAB_FLOATS
#define SD_REFTAB_FLOATS 1024
__host__ __device__ inline int sd_lds_weight_floats(int DT) { return (4 * DT + 8) * 1024; }
__host__ __device__ inline int sd_lds_total_bytes(int DT, bool with_ref) {
  return (sd_lds_weight_floats(DT) + (with_ref ? SD_WAVES * SD_REFTAB_FLOATS : 0)) * 4;
}
__host__ __device__ inline int sd_off_win(int DT) { return 0; }
__host__ __device__ inline int sd_off_wh1(int DT) { return 2 * DT * 1024; }
__host__ __device__ inline int sd_off_wh2(int DT) { return (2 * DT + 4) * 1024; }
__host__ __device__ inline int sd_off_wout(int DT) { return (2 * DT + 8) * 1024; }
__host__ __device__ inline int sd_off_bias(int DT) { return (4 * DT + 8) * 1024; }

// device-side view of a distribution (tables prepared by k_dist_tables)
struct DistDev {
  int kind;
  int k;
  const float* tab;     // GMM/GAUSS_DIAG: [k][2][dpad]  (mean, 1/var)
  const float* consts;  // GMM/GAUSS_DIAG: [k][2]  (sum log sigma + d*log sqrt(2pi), log mixture prob)
  const float* aux0;    // LOGREG: packed X images; GAUSS_FULL: packed precision / L^-1 images
  const float* aux1;
  float p0, p1, p2, p3;
  float clip;
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
  const float* wpack;     // packed MLP LDS image (global copy)
  const float* temb;      // [N][64] time embedding of the drift net, per step
  const float* stheta;    // [N] clipped score_model(t) or nullptr
  int ctrl_kind;
  float clip_model, clip_score, scale_score;
  int ref_k;
  const float* ref_tab;     // [N][K][2][dpad]
  const float* ref_consts;  // [N][K][2]  (0.5*sum log var, log w)
  float ref_c1;             // 0.5*d*log(2*pi)
  DistDev target;         // in-loop target score (ScoreCtrl / LerpCtrl / CMCD)
  DistDev prior;          // LerpCtrl / CMCD prior score
  const float* rnd_init;  // [B] initial log-weight (log p_prior(x0)) or nullptr
  float* trash;           // [SD_THREADS*4] dump slots for masked stores
  float cmcd_g, cmcd_clip;
  int ntiles;
  int stagger;            // start delay of waves 4..7, in units of s_sleep(127) (~8k cycles): see k_simulate
};
