/*
 * sdeng.h -- C ABI of the MI355X-native SDE sampling engine (libsdeng.so).
 *
 * The reference (vanilladucky/sde_sampler_lrds) has no FFI: its seam for this path is the
 * Python callable protocol of sde_sampler/losses/oc.py (`loss.simulate(ts, x, ...)`,
 * SURVEY.md section 8b).  This header is the binding a maintainer would add below that seam:
 * one call per `simulate()` loop, plain pointers and sizes, no torch types.  Every entry point
 * names the reference code it replaces (paths relative to /root/reference/sde_sampler).
 *
 * Conventions
 *   - all arrays are fp32, row-major, contiguous, in DEVICE memory unless noted;
 *   - the callee borrows every pointer for the duration of the stream-ordered work and never
 *     frees or allocates: outputs and the workspace are caller-allocated;
 *   - every function enqueues on `stream` (a hipStream_t) and returns without synchronising;
 *   - return value 0 = ok, negative = error (see SDENG_E_*); the message for the calling
 *     thread's last error is available from sdeng_last_error(); nothing ever throws;
 *   - re-entrant per stream; the only global state is the thread-local error string.
 */
#ifndef SDENG_H
#define SDENG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDENG_ABI_VERSION 3

/* error codes */
#define SDENG_OK 0
#define SDENG_E_INVALID (-1)     /* malformed descriptor (null pointer, bad size)          */
#define SDENG_E_UNSUPPORTED (-2) /* valid, but no kernel is built for this combination      */
#define SDENG_E_WORKSPACE (-3)   /* workspace too small (see sdeng_workspace_bytes)          */
#define SDENG_E_HIP (-4)         /* a HIP runtime call failed                                */

/* ---- which simulate() loop -------------------------------------------------------------
 * SDENG_FORM_LIN   x' = (c1*x + c2*(u [+ ref])) + c3*z ;  rnd += c4*<u,u> + c5*<u,z> + c6
 *                  EIReferenceSDELoss.simulate        losses/oc.py:444-510  (+ eq/sdes.py:532-539, 658-666)
 *                  DDPMLikeReferenceSDELoss.simulate  losses/oc.py:584-651  (+ eq/sdes.py:541-555, 668-678)
 *                  DiscreteTimeReversalLossEI.simulate losses/oc.py:906-978
 *                  ExponentialIntegratorSDELoss.simulate losses/oc.py:1319-1397 (DDS)
 * SDENG_FORM_EM    x' = x + ((c1*x [+ c3*ref]) + c2*u)*c4 + c2*(c5*z) ; rnd += 0.5*<u,u>*c4 + <u,c5*z> + c6
 *                  EMReferenceSDELoss.simulate        losses/oc.py:218-296  (PIS: no reference)
 *                  TimeReversalLoss.simulate          losses/oc.py:1133-1238 (no inference control)
 * SDENG_FORM_CMCD  ControlledLangevinSDELoss.simulate losses/oc.py:666-755 (+ eq/sdes.py:101-110)
 * SDENG_FORM_EUBO  the noising loops (x_in = samples of the target; coef rows in ITERATION order, i.e. times T - s run backwards):
 *                  x' = x*c1 + c3*z ; u = c2*ctrl(t_net, x') ; rnd -= c4*<u, ref(x') + u/2> ; rnd += c6*<u, x'> ; rnd -= c5*<u, z>
 *                  EIReferenceSDELoss.compute_eubo    losses/oc.py:512-568  (c2 = 1, c6 = 0)
 *                  EMReferenceSDELoss.compute_eubo    losses/oc.py:298-362  (also DDPMLikeReferenceSDELoss)
 *                  FLAG_TERM_REF / FLAG_TERM_TARGET here mean the INITIAL cost rnd0 = log p_ref(x_in) - log pi~(x_in) (:322, :536);
 *                  DiscreteTimeReversalLossEI.compute_eubo losses/oc.py:980-1036 (no reference, Score/LerpCtrl, c2 = 1, c6 = 0;
 *                  FLAG_INIT_LOGP here adds log p_prior of the NOISED samples at the end, :1032)
 *                  needs either a reference with a ClippedCtrl or no reference with a Score/LerpCtrl.
 */
#define SDENG_FORM_LIN 0
#define SDENG_FORM_EM 1
#define SDENG_FORM_CMCD 2
#define SDENG_FORM_EUBO 3
#define SDENG_FORM_CMCD_EUBO 4 /* ControlledLangevinSDELoss.compute_eubo losses/oc.py:757-828: the CMCD loop run from target samples
                                  with -u; coef rows in iteration order ([0] = ts[N-k], [2],[3] = dt, sqrt(dt) of iteration k,
                                  [4..7] = annealing weights of that time); rnd0 = -log pi~(x_in), + log p_prior(x_out) at the end */

/* Per-step coefficient table: coef[N][SDENG_NCOEF], filled by the host with the reference's own
 * fp32 scalar formulas (eq/sdes.py:456-555, 609-678; losses/oc.py:1369-1371).  Column meaning:
 *            LIN              EM                 CMCD
 *   [0]  t_net            t_net              s            time fed to the drift net
 *   [1]  c1 (x gain)      c1 (+-drift coef)  t            (CMCD: time of the second evaluation)
 *   [2]  c2 (ctrl gain)   c2 = g             dt
 *   [3]  c3 (noise gain)  c3 = g^2           sqrt(dt)
 *   [4]  c4 (0.5*omega)   c4 = dt            s/T
 *   [5]  c5 (sqrt omega)  c5 = sqrt(dt)      1 - s/T
 *   [6]  c6 (rnd const)   c6 (rnd const)     t/T
 *   [7]  score gain       score gain         1 - t/T      LerpCtrl: g(t); ScoreCtrl: 1
 *   [8]  lerp weight t/T  lerp weight        (unused)     LerpCtrl only
 *   [9]  s(tau)   [10] s(tau)^2*sigma_sq(tau)   [11] s(tau)^2   reference marginal (eq/sdes.py:228-229,247)
 *   [12..15] reserved
 * SDENG_FORM_EUBO (rows in iteration order): [0] t_net = T - s, [1] mean factor, [2] control gain (1, or 1/g with
 *   use_rescaling), [3] std factor, [4] running-cost weight (omega | dt g^2), [5] Ito weight (sqrt omega | std/mean),
 *   [6] <u,x> weight (0 | 1/mean - 1 + drift_coeff dt), [9..11] reference marginal at t_net.
 */
#define SDENG_NCOEF 16

/* flags */
#define SDENG_FLAG_ITO 1u          /* accumulate the stochastic integral <u,z> (compute_ito_int)     */
#define SDENG_FLAG_INIT_LOGP 2u    /* rnd0 = log p_prior(x0)  (losses/oc.py:695-699, 935-939, 1164-1168) */
#define SDENG_FLAG_TERM_REF 4u     /* terminal: rnd += log p_ref(x_N)  (losses/oc.py:290, 505, 645, 1390) */
#define SDENG_FLAG_TERM_TARGET 8u  /* terminal: rnd -= log pi~(x_N)                                 */
#define SDENG_FLAG_SPLIT_TILES 16u /* hint for small batches (B <= 8 192 = two tiles per CU; the reference's default evaluation batch is 6 000,
                                      conf/solver/basic_oc_base.yaml:28-30): work on every 16-particle tile with four waves (one per SIMD,
                                      a quarter of the features each) instead of one -- ~2x lower latency when the batch cannot fill the
                                      chip.  Same noise counters; sums are formed in another order, so results equal the default path's
                                      to fp32 round-off, not bit for bit.  Honoured for ClippedCtrl, LIN / EM forms, no / Gaussian /
                                      small-mixture reference, d > 64, no noise_in (xs_out is written); ignored otherwise. */

#define SDENG_FLAG_REMOVE_REF 32u  /* RemoveReferenceCtrl(score, ref_score, use_rescaling=False) (models/reparam.py:46-64): the control the loss sees is
                                      ctrl(t, x) - ref_score(t, x), ref_score = the reference drift of `ref`.  Forward forms (LIN / EM) with a
                                      Score / Lerp / CancelDrift control and a diagonal reference; SDENG_E_UNSUPPORTED otherwise. */

#define SDENG_FLAG_REUSE_PACK 64u  /* sdeng_ctrl_vjp only: the packed weight images of an earlier call with the same net are still in the
                                      workspace -- skip re-packing (a caller that walks the times one by one, e.g. an adjoint recursion) */

/* ---- distributions (the distr package): log-density and score by hand-coded formulas ------------ */
#define SDENG_DIST_NONE 0
#define SDENG_DIST_GMM_DIAG 1   /* distr/gauss.py:138-244  GMM / TwoModes / ManyModes (MixtureSameFamily)   */
#define SDENG_DIST_GAUSS_DIAG 2 /* distr/gauss.py:597-629  Gauss, Delta (one component, no mixture weights) */
#define SDENG_DIST_ISO_GAUSS 3  /* distr/gauss.py:720-787  IsotropicGauss                                   */
#define SDENG_DIST_PHI4 4       /* distr/phi_four.py:8-96  1-D lattice, Dirichlet-0 boundary                */
#define SDENG_DIST_LOGREG 5     /* distr/logistic_regression.py:11-92 + autograd score distr/base.py:146-154 */
#define SDENG_DIST_GAUSS_FULL 6 /* distr/gauss.py:632-717  GaussFull (MultivariateNormal)                   */
#define SDENG_DIST_RINGS 7      /* distr/rings.py:38-109   2-D rings: radial Gaussian mixture x uniform angle     */
#define SDENG_DIST_GMM_FULL 8   /* distr/gauss.py:310-520  GMMFull / TwoModesFull (full covariances), as the TARGET of a Score / Lerp /
                                   CancelDrift control only (score_mog_full :110-121 inside the step loop): loc [k,d], scale = eigenvalues
                                   [k,d] and aux = eigenvectors [k,d,d] (row-major, columns = vectors) of each covariance, w [k].  Forward
                                   forms without a reference drift (PIS / DDS / DIS); its log-density is not evaluated here (no
                                   FLAG_TERM_TARGET with it): SDENG_E_UNSUPPORTED otherwise. */

typedef struct sdeng_dist {
  int32_t kind;      /* SDENG_DIST_*                                                              */
  int32_t k;         /* GMM: number of components; LOGREG: number of data rows                    */
  const float* loc;  /* GMM [k,d]; GAUSS_DIAG/GAUSS_FULL [d]; LOGREG: X [k,d-1]; RINGS: radii [k <= 8]  */
  const float* scale;/* GMM [k,d]; GAUSS_DIAG [d] (std-dev); GAUSS_FULL: precision [d,d]; LOGREG: y [k] */
  const float* w;    /* GMM, RINGS: unnormalised mixture weights [k]; GAUSS_FULL: inverse Cholesky factor L^-1 [d,d] */
  float p0, p1, p2, p3; /* ISO_GAUSS: loc, scale, norm_const = -0.5*d*log(2*pi*scale^2), scale^2 (both as the
                           reference computes them in fp32, distr/gauss.py:759-760); PHI4: a, b, beta;
                           LOGREG: weight_scale, intercept_mean, intercept_scale, threshold;
                           GAUSS_FULL: p0 = sum(log diag L); RINGS: p0 = radial std-dev (d must be 2) */
  float clip;        /* clip applied to the log-density (solver/oc.py:80-87 clip_target); <=0: none */
  const float* aux;  /* GAUSS_FULL as x0_dist: Cholesky factor L [d,d] (row-major, lower) of the covariance; else NULL */
} sdeng_dist;

/* ---- drift net: models/mlp.py:99-143 FourierMLP(num_layers=4, channels=64, GELU) wrapped by
 *      models/reparam.py:18-43 ClippedCtrl / :63-117 ScoreCtrl / :148-199 LerpCtrl ------------ */
#define SDENG_CTRL_CLIPPED 0
#define SDENG_CTRL_SCORE 1
#define SDENG_CTRL_LERP 2
/* no drift net: EulerIntegrator.integrate (eq/integrator.py:93-129) of an uncontrolled linear SDE (OU.drift/diff,
 * eq/sdes.py:143-148; the inference process of solver/oc.py:162-180) or of the classic Langevin SDE (LangevinSDE,
 * eq/sdes.py:46-76; solver/langevin.py:36-66).  form = SDENG_FORM_EM, no reference; per step
 *   x' = x + (coef[1] x + clip(coef[7] score_target(x), net.clip_score)) coef[4] + coef[2] (z coef[5])
 * (target.kind NONE: no score term).  rnd_out is zero-filled; xs_out carries the N+1 states. */
#define SDENG_CTRL_NONE 3
/* CancelDriftCtrl (models/reparam.py:120-145, conf/model/langevin_init.yaml): a ScoreCtrl that also cancels the SDE's drift,
 *   u = clip(net) + coef[8] x + coef[7] (scale clip(score_target(x)) s_theta(t))
 * with coef[8] = drift_coeff(t)/g(t), coef[7] = g(t)/2 at the net's time (use_rescaling; else drift_coeff/g^2 and 1/2). */
#define SDENG_CTRL_CANCEL_DRIFT 4
#define SDENG_HIDDEN 64

typedef struct sdeng_time_embed { /* models/mlp.py:57-96 TimeEmbed */
  const float* coeff;             /* timestep_coeff [64] (linspace(0.1,100,64))          */
  const float* phase;             /* timestep_phase [64]                                  */
  const float* w[4];              /* hidden_layer[i].weight: [64,128] for i=0, [64,64] after */
  const float* b[4];              /* hidden_layer[i].bias [64]                            */
  int32_t n_hidden;               /* num_layers - 1 (1 for the MLP's embed, 3 for score_model) */
  int32_t dim_out;                /* 64 (MLP embed) or 1 (score_model)                    */
  const float* w_out;             /* out_layer.weight [dim_out,64]                        */
  const float* b_out;             /* out_layer.bias [dim_out]                             */
} sdeng_time_embed;

typedef struct sdeng_net {
  int32_t ctrl_kind;              /* SDENG_CTRL_*                                         */
  int32_t reserved;
  const float *w_in, *b_in;       /* input_embed  [64,d], [64]                            */
  const float *w_h1, *b_h1;       /* hidden_layer.0 [64,64], [64]                         */
  const float *w_h2, *b_h2;       /* hidden_layer.1 [64,64], [64]                         */
  const float *w_out, *b_out;     /* out_layer [d,64], [d]                                */
  sdeng_time_embed t_embed;       /* base_model.timestep_embed (dim_out 64)               */
  sdeng_time_embed score_model;   /* ScoreCtrl/LerpCtrl score_model (dim_out 1); n_hidden=0: absent */
  float clip_model;               /* reparam.py:33-40; <=0: no clipping                   */
  float clip_score;               /* reparam.py:91-100; <=0: none                         */
  float scale_score;              /* reparam.py:114                                       */
  float reserved_f;
} sdeng_net;

/* ---- reference drift (solver/oc.py:535-576 change_reference_type -> eq/sdes.py:265-345) ---- */
#define SDENG_REF_NONE 0
#define SDENG_REF_GAUSS_DIAG 1 /* marginal_score, diagonal var_init            eq/sdes.py:265-279 */
#define SDENG_REF_GMM_DIAG 2   /* marginal_gmm_score, diagonal variances_init  eq/sdes.py:329-345 */

/* marginal_gmm_score with full covariance matrices, score_mog_full (distr/gauss.py:110-121, eq/sdes.py:329-345), in the
 * eigen form the reference itself accepts (eq/sdes.py:228-238: variances_init = (D, P), covariance_c = P_c diag(D_c) P_c^T):
 * vars_init holds the eigenvalues D [k,d], eigvecs the eigenvectors P [k,d,d] row-major (P[c][i][j] = i-th coordinate of
 * the j-th eigenvector).  A caller holding covariance matrices passes their symmetric eigendecomposition. */
#define SDENG_REF_GMM_FULL 3

typedef struct sdeng_ref {
  int32_t kind;
  int32_t k;               /* GMM components (1 for GAUSS_DIAG)                      */
  const float* means_init; /* [k,d]                                                   */
  const float* vars_init;  /* [k,d]  (GMM_FULL: eigenvalues of the covariances)      */
  const float* weights;    /* [k] unnormalised (normalised like distr/gauss.py:100)   */
  const float* eigvecs;    /* [k,d,d] GMM_FULL only, else NULL                        */
  int32_t shared_var;      /* GMM_DIAG: 1 = the caller guarantees that all k rows of vars_init are equal (the reference's default
                              initialisation, variances_init = v * ones): mixtures with 4 < k <= 64 then run on the matrix pipe.
                              The promise is checked on the device: if the rows differ, x_out and rnd_out come back NaN. */
  int32_t reserved;
} sdeng_ref;

/* ---- noise ---------------------------------------------------------------------------------
 * noise_in != NULL : injected, z[k] = noise_in[k*B*d ...] -- replays the reference's one
 *                    randn_like(x) per step (losses/oc.py:277, eq/sdes.py:537, losses/oc.py:722, 1372, 1222);
 * noise_in == NULL : counter-based Philox4x32-10, counter (particle0+p, feature/4, step, stream), key = seed,
 *                    two Box-Muller pairs on u = ((bits>>9)+0.5)*2^-23; independent of sharding.
 */
/* ---- initial particles -------------------------------------------------------------------------
 * x_in != NULL : the caller's x0 [B,d] (e.g. prior.sample((B,)) as solver/oc.py:132 does).
 * x_in == NULL : x0 is drawn by the engine from `x0_dist`, z = the Philox normals of stream 1 at step 0 (same counter layout as
 *                the step noise, so x0 of global particle p does not depend on the sharding):
 *     ISO_GAUSS   x0 = p0 + p1 * z                IsotropicGauss.sample   distr/gauss.py:772-787 (no truncation)
 *     GAUSS_DIAG  x0 = loc + scale * z            Gauss.sample distr/gauss.py:235-239;  scale == NULL: x0 = loc  (Delta.sample, distr/delta.py:27-31)
 *     GAUSS_FULL  x0 = loc + L z  (L = aux)       GaussFull.sample        distr/gauss.py:709-713 (MultivariateNormal)
 *   A sampler kernel writes x0 into the workspace -- or into x0_out ([B,d], optional) -- ahead of the step loop: one launch,
 *   0.2 % of a cfg-2 pass.  (A step-loop variant that drew x0 in registers was measured 1-4 % slower and dropped)
 */
typedef struct sdeng_desc {
  int32_t abi_version;   /* SDENG_ABI_VERSION                                              */
  int32_t form;          /* SDENG_FORM_*                                                   */
  uint32_t flags;        /* SDENG_FLAG_*                                                   */
  int32_t B, d, N;       /* particles (this shard), dimension (<=128), steps               */
  int64_t particle0;     /* global index of this shard's first particle (Philox counter)   */
  uint64_t seed;         /* Philox key                                                     */
  const float* coef;     /* [N][SDENG_NCOEF], device                                        */
  const float* x_in;     /* [B,d], or NULL: draw x0 from x0_dist                             */
  float* x_out;          /* [B,d]                                                           */
  float* rnd_out;        /* [B]  (the reference's rnd[B,1])                                 */
  float* xs_out;         /* optional trajectory [N+1,B,d] (return_traj=True), or NULL       */
  const float* noise_in; /* optional [N,B,d], or NULL                                       */
  sdeng_net net;
  sdeng_ref ref;
  sdeng_dist target;     /* terminal cost + (ScoreCtrl/LerpCtrl/CMCD) target score          */
  sdeng_dist ref_dist;   /* terminal reference log-density (FLAG_TERM_REF)                  */
  sdeng_dist prior;      /* initial log-density (FLAG_INIT_LOGP); LerpCtrl/CMCD prior score  */
  float cmcd_g;          /* ControlledLangevinSDE.diff_coeff          eq/sdes.py:93-95     */
  float cmcd_clip;       /* ControlledLangevinSDE.clip_score (<=0: none) eq/sdes.py:105-110 */
  void* workspace;       /* device scratch, >= sdeng_workspace_bytes(desc)                  */
  size_t workspace_bytes;
  void* ev_start;        /* optional hipEvent_t recorded on `stream` right before the step-loop kernel */
  void* ev_stop;         /* optional hipEvent_t recorded right after it (roofline timing)            */
  sdeng_dist x0_dist;    /* distribution of x0 when x_in == NULL (see above)                 */
  float* x0_out;         /* optional [B,d]: the x0 that was drawn                             */
} sdeng_desc;

/* Version of this ABI compiled into the library. */
int sdeng_abi_version(void);

/* Message of the calling thread's most recent error ("" if none). */
const char* sdeng_last_error(void);

/* Device scratch needed by sdeng_simulate for this descriptor (packed MFMA weight image,
 * per-step embedding and reference tables).  Pure host arithmetic; 0 on a malformed descriptor. */
size_t sdeng_workspace_bytes(const sdeng_desc* desc);

/* Run one whole simulate() loop (all N steps, initial and terminal cost) on `stream`.
 * Replaces the Python step loops listed under SDENG_FORM_*. */
int sdeng_simulate(const sdeng_desc* desc, void* stream);

/* Estimators of BaseOCLoss.compute_results (losses/oc.py:150-161) and the normalised ESS of
 * eval/metrics.py:135-140 from rnd[B]:
 *   stats[0] = mean(-rnd) (elbo)      stats[1] = logsumexp(-rnd) - log(B) (log_norm_const_is)
 *   stats[2] = var(rnd) (unbiased)    stats[3] = (sum w)^2 / sum w^2 / B  with w = softmax(-rnd)
 *   stats[4] = max(-rnd)              stats[5] = sum exp(-rnd - max)      stats[6] = sum exp(2(-rnd - max))
 *   stats[7] = sum(-rnd)              (4..7: partials for the multi-GPU combine)
 * weights_out (optional, [B]) receives softmax(-rnd, 0).  stats is device memory, 8 floats. */
int sdeng_logz(const float* rnd, int64_t B, float* stats, float* weights_out, void* workspace, size_t workspace_bytes,
               void* stream);
size_t sdeng_logz_workspace_bytes(void);

/* Standalone pieces, exported for unit parity tests against the oracle:
 * FourierMLP (+ctrl wrapper) forward at one time, u[B,d] (models/mlp.py:135-143, reparam.py:42,112-117,186-199) */
int sdeng_ctrl_forward(const sdeng_desc* desc, float t_net, float score_gain, float lerp_w, const float* x, float* u_out,
                       void* stream);
/* distribution log-density [B] and score [B,d] (either output may be NULL) */
int sdeng_dist_eval(const sdeng_dist* dist, int32_t B, int32_t d, const float* x, float* logp_out, float* score_out,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t sdeng_dist_workspace_bytes(const sdeng_dist* dist, int32_t d);
/* Counter-based normals exactly as the step loop draws them: out[B,d] for one step. */
int sdeng_philox_normal(uint64_t seed, int32_t step, int64_t particle0, int32_t B, int32_t d, uint32_t stream_id,
                        float* out, void* stream);
/* The same for n_steps consecutive steps in one launch: out[n_steps,B,d] (the noise a training call keeps, losses/oc.py:277, :537). */
int sdeng_philox_normal_steps(uint64_t seed, int32_t step0, int32_t n_steps, int64_t particle0, int32_t B, int32_t d,
                              uint32_t stream_id, float* out, void* stream);

/* Training direction: fused forward + backward of the drift net -- FourierMLP under ClippedCtrl -- over
 * M = n_times * rows_per_time rows; row r = k * rows_per_time + b is the state x[r] at time desc->coef[k][0].  Replaces the autograd
 * graph that losses/oc.py:83-103 (generative_and_sde_ctrl) builds per step and loss.backward() walks (models/mlp.py:135-143,
 * models/reparam.py:33-43).  Reads desc->{abi_version, d, coef, net, workspace}.  `cot` [M,d] is the cotangent of the control u.
 * Per-row outputs, all [M,64] unless noted:  a0, a1, a2 = gelu of the three hidden pre-activations;  d0, d1, d2 = cotangents of those
 * pre-activations;  dout [M,d] = cot under ClippedCtrl's clip mask;  gx [M,d] (optional) = gradient w.r.t. the state;  u_out [M,d]
 * (optional) = the control itself.  cot == NULL: forward only (u_out required, the other outputs may be NULL).  The
 * parameter gradients are the caller's six products over the rows: dW_out = dout^T a2, dW_2 = d2^T a1, dW_1 = d1^T a0,
 * dW_in = d0^T x, bias gradients = column sums, cotangent of the time embedding of time k = sum over its rows of d0. */
int sdeng_ctrl_vjp(const sdeng_desc* desc, int32_t n_times, int32_t rows_per_time, const float* x, const float* cot, float* a0, float* a1,
                   float* a2, float* d0, float* d1, float* d2, float* dout, float* gx, float* u_out, void* stream);
size_t sdeng_ctrl_vjp_workspace_bytes(int32_t d, int32_t n_times);

/* KL-method training (BaseOCLoss.compute_loss kl branch, losses/oc.py:105-131): the gradient of the mean log-weight through the
 * trajectory -- what loss.backward() computes by walking the autograd graph of simulate() (losses/oc.py:258-284 EM, :478-502 EI) -- as the
 * discrete adjoint of the recursion sdeng_simulate integrates, in ONE launch.  The states xs[k] (k = 0 .. N-1, from sdeng_simulate's xs_out)
 * are constants; per 16 particles the kernel walks k = N-1 .. 0 with the adjoint state lambda in registers:
 *     cot_k  = alpha_k lambda + w_b (beta_k u_k + gamma_k z_k)        u_k = the control at (coef[k][0], xs[k]), recomputed
 *     lambda = A_k lambda + C_k H_ref(xs[k]) lambda + J_u^T cot_k     H_ref: Jacobian of the noised reference's score (eq/sdes.py:265-279, 329-345)
 * (FORM_LIN: alpha = c2, beta = 2 c4, gamma = c5, A = c1, C = c2;  FORM_EM: alpha = c2 c4, beta = c4, gamma = c5, A = 1 + c4 c1, C = c4 c3;
 * gamma = 0 without SDENG_FLAG_ITO) and writes the per-row arrays of sdeng_ctrl_vjp at row k * B + b: the parameter gradients are the same
 * six products.  Reads desc->{abi_version, B, d, N, form, flags & ITO, coef, net, ref (NONE / GAUSS_DIAG / GMM_DIAG), target, workspace}.
 * Controls: ClippedCtrl, and ScoreCtrl / LerpCtrl / CancelDriftCtrl (models/reparam.py:63-199; the latter two with the per-step gains of
 * coef[7], coef[8] and, for LerpCtrl, desc->prior = the IsotropicGauss whose score is interpolated: d lerp/dx = (1 - t/T)(-1/var) + (t/T) H_pi)
 * -- written out for ScoreCtrl (BASELINE configs 1 and 3: DDS, PIS) -- on a diagonal mixture or phi^4 target
 * (desc->target of kind GMM_DIAG / PHI4): u = clip(net) + scale clip(score_pi(x)) s_theta(t); the state gradient gains
 * scale s_theta H_pi(x) (mask cot) (closed-form Hessian-vector product: mixture, or the lattice's tridiagonal Hessian; skipped with detach_score), and `dst` receives the cotangent of s_theta(t_k) per particle -- the caller sums
 * it over the particles and back-propagates the N values through the small score model. */
typedef struct sdeng_adjoint {
  const float* xs;      /* [N][B][d] states x_0 .. x_{N-1}                                                   */
  const float* noise;   /* [N][B][d] the normals of the trajectory (sdeng_philox_normal_steps of its seed); required with FLAG_ITO */
  const float* w;       /* [B] d loss / d rnd_b (1/B on the particles that pass the loss's filter, else 0)   */
  const float* lam_in;  /* [B][d] lambda_N = d (sum_b w_b terminal(x_N,b)) / d x_N                           */
  float* lam_out;       /* [B][d] lambda_0, or NULL                                                          */
  float *a0, *a1, *a2, *d0, *d1, *d2; /* [N*B][64] each, as sdeng_ctrl_vjp                                    */
  float* dout;          /* [N*B][d]                                                                          */
  float* dst;           /* [N*B] ScoreCtrl: <cot, scale clip(score_pi)>, the cotangent of s_theta(t_k) per particle; else NULL */
  int32_t detach_score; /* ScoreCtrl(detach_score=True): the target score is treated as a constant of x     */
  const float* score;   /* [N*B][d] or NULL.  Given: the target score of every row, evaluated by the caller (sdeng_dist_eval) and treated as
                           a constant of x -- for targets whose score the reference makes by autograd without a graph (distr/base.py:146-154:
                           LogisticRegression), desc->target is then ignored */
} sdeng_adjoint;
int sdeng_kl_adjoint(const sdeng_desc* desc, const sdeng_adjoint* adj, void* stream);
size_t sdeng_kl_adjoint_workspace_bytes(const sdeng_desc* desc);

/* Annealed samplers: n_moves Langevin moves of B chains in ONE launch -- mala_step / ula_step of additions/mcmc.py:77-135,
 * 189-221 with the per-chain step-size heuristic of :55-74 (target_acceptance > 0), as smc_sampler / re_sampler / mcmc_sample apply
 * them move after move (additions/ebm_mle.py:120-160, 340-380; experiments/benchmark_utils.py:300-330).  Density: the geometric path
 * log pi_t = (1 - t[b]) log p_prior + t[b] log pi~ (prior == NULL or kind NONE: the target alone; t == NULL: 1).  State in / out:
 * x [B,d], lp [B] = log pi_t(x), grad [B,d], step [B].  Noise: injected normals z [n_moves,B,d] and uniforms u [n_moves,B] (u unused
 * by the unadjusted move) -- e.g. drawn from the host framework's generator in the reference's order -- or NULL: Philox streams 2 / 3
 * keyed by (seed, chain0 + chain, move).  samples (optional) [n_moves - keep_from, B, d]: the states after moves keep_from..;
 * acc_sum (optional) [B]: sum over those moves of min(1, acceptance ratio); acc_last (optional) [B]: that of the last move. */
int sdeng_langevin_moves(const sdeng_dist* prior, const sdeng_dist* target, int32_t B, int32_t d, int32_t n_moves, int32_t keep_from,
                         int32_t unadjusted, float target_acceptance, const float* t, float* x, float* lp, float* grad, float* step,
                         const float* z, const float* u, uint64_t seed, int64_t chain0, float* samples, float* acc_sum, float* acc_last,
                         void* workspace, size_t workspace_bytes, void* stream);
size_t sdeng_langevin_moves_workspace_bytes(const sdeng_dist* prior, const sdeng_dist* target, int32_t d);

/* prior.sample((B,)) on the device, out[B,d]: exactly the x0 that sdeng_simulate draws for x_in == NULL with the same
 * (dist, seed, particle0).  Replaces IsotropicGauss.sample / Gauss.sample / Delta.sample / GaussFull.sample (distr/gauss.py:772-787,
 * 235-239, 709-713; distr/delta.py:27-31). */
int sdeng_sample_x0(const sdeng_dist* dist, uint64_t seed, int64_t particle0, int32_t B, int32_t d, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDENG_H */
