// Argument blocks and host launchers of the small kernels in prep_kernels.hip.
#pragma once
#include "sim_common.hpp"
#include "../../include/sdeng.h"

#define SD_LOGZ_MAX_BLOCKS 1024

struct PackArgs {
  int NT, d;
  const float *w_in, *b_in, *w_h1, *b_h1, *w_h2, *b_h2, *w_out, *b_out;
  float* out;
  // transpose = 1: the image of the TRANSPOSED net for the backward products (grad_kernel.hpp): slot of W_in <- W_out^T, W_1 <- W_2^T,
  // W_2 <- W_1^T, W_out <- W_in^T, each with the power-of-two scale of its own matrix, read from `scales` (the forward image's
  // scale block); no biases.  transpose = 0: `scales` is ignored (the forward pack computes and stores them).
  int transpose;
  const float* scales;
};

struct TimeEmbedArgs {
  sdeng_time_embed te;
  const float* coef;  // [N][16]; time = coef[k][col]
  int col;
  int t_direct;       // 1: use t_value for every block (single-time evaluation)
  float t_value;
  float clip;         // <= 0: none
  float* out;         // [N][dim_out]
};

struct RefTabArgs {
  int K, d, dpad;
  const float* coef;
  const float *means, *vars, *weights;
  float* tab;     // [N][K][2][dpad]
  float* consts;  // [N][K][2]
  float* same_var;  // [1]: 1.0 when every component has the same variance vector (then so has every noised marginal)
  int centred;      // 1: K = SD_KREG shared-variance mixtures get the centred table of gmm_resp_centred (the standard step loop asks for it)
};

// full-covariance mixture reference (SDENG_REF_GMM_FULL): per step and component the noised precision as an MFMA image
struct RefFullArgs {
  int K, d, dpad, NT;
  const float* coef;  // per-step marginal (cols 9..11); nullptr = the mixture itself at every step
  const float *means, *eigvals, *eigvecs, *weights;
  float* images;     // [N][K][NT*KB*512]  split-f16 A-operand image of P = U diag(1/(VA + S2 lambda)) U^T
  float* means_out;  // [N][K][dpad]       S * mean (0 on pad features)
  float* consts;     // [N][K][2]          (0.5 log det, log w)
};

// shared-variance mixture reference on the matrix pipe (sim_kernel.hpp RF_GMM_MM): per step the logit image [components x features],
// the mean image [features x components] (split-f16 A operands), the centre / inverse variance vectors and the logit constants
struct RefMMArgs {
  int K, d, dpad, NT, kt;     // kt = ceil(K / 16) component tiles (<= 4)
  const float* coef;
  const float *means, *vars, *weights;  // vars: [K][d], all rows equal (row 0 is read)
  float* images;    // [N][kt*KB(NT)*512 + NT*KB(kt)*512]
  float* centre;    // [N][2][dpad]   centre of the noised means, 1/var (0 on pad features)
  float* consts;    // [N][64]        b_k = log w_k - 0.5 sum_f (mu_kf - c_f)^2 / var_f ; -inf on pad components
  float* same_var;  // [1] scratch: k_same_var's verdict on `vars` -- a caller whose shared_var promise is false gets NaN, not a wrong score
};

struct DistTabArgs {
  int K, d, dpad;
  const float *loc, *scale, *weights;  // weights nullptr: single Gaussian
  float* tab;     // [K][2][dpad]
  float* consts;  // [K][4]
};

struct DistEvalArgs {
  DistDev ds;
  int B, d, dpad;
  const float* x;
  float* logp_out;
  float* score_out;
};

// K Langevin moves (MALA or ULA) of B chains at fixed tempering weights, state in LDS: additions/mcmc.py:77-135 (mala_step), :189-221
// (ula_step) with the step-size heuristic of :55-74, on the geometric path log pi_t = (1 - t) log p_prior + t log pi~ (ebm_mle.py's
// tempered densities; prior.ds.kind == NONE: the target alone, e.g. the MALA chains of experiments/benchmark_utils.py:268-333)
struct MovesArgs {
  DistEvalArgs prior, target;   // .ds / .d / .dpad of the two ends of the path
  int B, d, K, keep_from, ula;  // chains, dimension, moves, first stored move, 1 = unadjusted
  float target_acc;             // <= 0: step sizes stay fixed
  const float* t;               // [B] tempering weight of each chain, or nullptr (1)
  float *x, *lp, *grad, *step;  // [B,d], [B], [B,d], [B]: chain state, carried in and out
  const float *z, *u;           // injected normals [K,B,d] and uniforms [K,B] (MALA), or nullptr: counter-based Philox draws
  unsigned seed_lo, seed_hi;
  long long chain0;             // global index of chain 0 (Philox counter)
  float* samples;               // [K - keep_from, B, d] states after each stored move, or nullptr
  float* acc_sum;               // [B] sum over the stored moves of min(1, acceptance ratio), or nullptr
  float* acc_last;              // [B] min(1, acceptance ratio) of the last move, or nullptr
};
int sd_launch_moves(const MovesArgs& a, hipStream_t s);

struct TerminalArgs {
  DistDev ref, target;
  int use_ref, use_target;
  int B, d, dpad;
  const float* x;
  float* rnd;
};

int sd_launch_pack(const PackArgs& a, hipStream_t s);
int sd_launch_time_embed(const TimeEmbedArgs& a, int N, hipStream_t s);
int sd_launch_ref_tables(const RefTabArgs& a, int N, hipStream_t s);
int sd_launch_ref_full_tables(const RefFullArgs& a, int N, hipStream_t s);
int sd_launch_dist_tables(const DistTabArgs& a, hipStream_t s);
int sd_launch_dist_eval(const DistEvalArgs& a, hipStream_t s);
int sd_launch_terminal(const TerminalArgs& a, hipStream_t s);
int sd_launch_logz(const float* rnd, long long B, float* stats, float* weights, float* scratch, hipStream_t s);
int sd_launch_philox(unsigned lo, unsigned hi, int step, int n_steps, long long p0, int B, int d, unsigned stream_id, float* out, hipStream_t s);
int sd_launch_sample_x0(const sdeng_dist& ds, unsigned lo, unsigned hi, long long p0, int B, int d, float* out, hipStream_t s);
int sd_launch_ref_mm_tables(const RefMMArgs& a, int N, hipStream_t s);
