// extern "C" entry points of libsdeng.so (see include/sdeng.h): descriptor validation, workspace
// carving, preparation kernels, dispatch to the simulate-kernel instantiations.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "../../include/sdeng.h"
#include "prep_kernels.hpp"
#include "sim_common.hpp"
#include "cmcd_kernel.hpp"
#include "grad_kernel.hpp"

#define SD_TILES(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8)      // feature tiles of 16: one instantiation per count (d <= 128)
#define SD_TILES_LOGREG(M) M(1) M(2) M(3) M(4)                 // in-loop logistic-regression score: d <= 64 (design matrix in LDS)
#define SD_TILES_FULL(M) M(1) M(2) M(3) M(4) M(6) M(8)          // full-covariance reference: 5 and 7 tiles run on 6 and 8 (piece staging)
#define SD_DECLARE_CMCD(DT) int sd_launch_cmcd_##DT(const CmcdArgs& a, int grid, hipStream_t s);
SD_TILES(SD_DECLARE_CMCD)
int sd_launch_logreg_images(const float* X, const float* y, int n, int dw, int NT, float* image, float* y_pad, hipStream_t s);
int sd_launch_pack_square(const float* P, const float* loc, int d, int NT, float* out, float* loc_pad, hipStream_t s);

enum { RF_NONE = 0, RF_GAUSS = 1, RF_GMM = 2, RF_GMM_BIG = 3, RF_GMM_FULL = 4, RF_GMM_MM = 5 };
enum { SC_NONE = 0, SC_GMM = 1, SC_PHI4 = 2, SC_LOGREG = 3, SC_REFSLOT = 4 };

typedef int (*sim_launch_fn)(const SimArgs&, int grid, hipStream_t);
#define SD_DECLARE_SIM(DT, REF, SC, FORM) int sd_launch_sim_##DT##_##REF##_##SC##_##FORM(const SimArgs& a, int grid, hipStream_t s);
#define SD_DECLARE_CTRL(DT, SC) int sd_launch_ctrl_##DT##_##SC(const SimArgs& a, int grid, hipStream_t s);
#define SD_ENTRY(DT, REF, SC, FORM) sd_launch_sim_##DT##_##REF##_##SC##_##FORM,
#define SD_CENTRY(DT, SC) sd_launch_ctrl_##DT##_##SC,
// forward forms (LIN, EM): [tiles - 1][reference kind 0..3][score kind 0..2][form]
#define SD_FOR_FORM(M, DT, REF, SC) M(DT, REF, SC, 0) M(DT, REF, SC, 1)
#define SD_FOR_SC(M, DT, REF) SD_FOR_FORM(M, DT, REF, 0) SD_FOR_FORM(M, DT, REF, 1) SD_FOR_FORM(M, DT, REF, 2)
#define SD_FOR_REF(M, DT) SD_FOR_SC(M, DT, 0) SD_FOR_SC(M, DT, 1) SD_FOR_SC(M, DT, 2) SD_FOR_SC(M, DT, 3)
#define SD_DECL_ALL(DT) SD_FOR_REF(SD_DECLARE_SIM, DT)
SD_TILES(SD_DECL_ALL)
#define SD_TAB_FORM(DT, REF, SC) {SD_ENTRY(DT, REF, SC, 0) SD_ENTRY(DT, REF, SC, 1)},
#define SD_TAB_SC(DT, REF) {SD_TAB_FORM(DT, REF, 0) SD_TAB_FORM(DT, REF, 1) SD_TAB_FORM(DT, REF, 2)},
#define SD_TAB_REF(DT) {SD_TAB_SC(DT, 0) SD_TAB_SC(DT, 1) SD_TAB_SC(DT, 2) SD_TAB_SC(DT, 3)},
static const sim_launch_fn kSimTable[8][4][3][2] = {SD_TILES(SD_TAB_REF)};
// compute_eubo kernels: [tiles - 1][reference kind - 1] for the reference-SDE losses (ClippedCtrl), [tiles - 1][2 + score kind]
// for DIS (no reference, ScoreCtrl)
#define SD_EUBO_ROW(M, DT) M(DT, 1, 0, 3) M(DT, 2, 0, 3) M(DT, 3, 0, 3) M(DT, 0, 1, 3) M(DT, 0, 2, 3)
#define SD_DECL_EUBO(DT) SD_EUBO_ROW(SD_DECLARE_SIM, DT)
SD_TILES(SD_DECL_EUBO)
#define SD_TAB_EUBO(DT) {SD_EUBO_ROW(SD_ENTRY, DT)},
static const sim_launch_fn kEuboTable[8][5] = {SD_TILES(SD_TAB_EUBO)};
// in-loop logistic-regression score (ScoreCtrl on a LOGREG target, no reference): [tiles - 1][form LIN, EM]
#define SD_LOGREG_ROW(M, DT) M(DT, 0, 3, 0) M(DT, 0, 3, 1)
#define SD_DECL_LOGREG(DT) SD_LOGREG_ROW(SD_DECLARE_SIM, DT)
SD_TILES_LOGREG(SD_DECL_LOGREG)
#define SD_TAB_LOGREG(DT) {SD_LOGREG_ROW(SD_ENTRY, DT)},
static const sim_launch_fn kLogregTable[4][2] = {SD_TILES_LOGREG(SD_TAB_LOGREG)};
// full-covariance mixture reference: [tiles - 1][LIN, EM, EUBO]; 5 and 7 tiles have no instantiation (tiles_full rounds them up)
#define SD_FULL_ROW(X, DT) X(DT, 4, 0, 0) X(DT, 4, 0, 1) X(DT, 4, 0, 3)
#define SD_DECL_FULL(DT) SD_FULL_ROW(SD_DECLARE_SIM, DT)
SD_TILES_FULL(SD_DECL_FULL)
// shared-variance mixture reference on the matrix pipe (REF = 5): [tiles - 1][LIN, EM]
#define SD_MM_ROW(X, DT) X(DT, 5, 0, 0) X(DT, 5, 0, 1)
#define SD_DECL_MM(DT) SD_MM_ROW(SD_DECLARE_SIM, DT)
SD_TILES(SD_DECL_MM)
#define SD_TAB_MM(DT) {SD_MM_ROW(SD_ENTRY, DT)},
static const sim_launch_fn kMMTable[8][2] = {SD_TILES(SD_TAB_MM)};
// full-covariance mixture TARGET of a Score / Lerp / CancelDrift control, held in the reference slot (SC = 4): [tiles - 1][LIN, EM]
#define SD_FULLSC_ROW(X, DT) X(DT, 4, 4, 0) X(DT, 4, 4, 1)
#define SD_DECL_FULLSC(DT) SD_FULLSC_ROW(SD_DECLARE_SIM, DT)
SD_TILES_FULL(SD_DECL_FULLSC)
static const sim_launch_fn kFullScoreTable[8][2] = {{SD_FULLSC_ROW(SD_ENTRY, 1)}, {SD_FULLSC_ROW(SD_ENTRY, 2)}, {SD_FULLSC_ROW(SD_ENTRY, 3)}, {SD_FULLSC_ROW(SD_ENTRY, 4)},
                                                    {nullptr, nullptr}, {SD_FULLSC_ROW(SD_ENTRY, 6)}, {nullptr, nullptr}, {SD_FULLSC_ROW(SD_ENTRY, 8)}};
static const sim_launch_fn kFullTable[8][3] = {{SD_FULL_ROW(SD_ENTRY, 1)}, {SD_FULL_ROW(SD_ENTRY, 2)}, {SD_FULL_ROW(SD_ENTRY, 3)}, {SD_FULL_ROW(SD_ENTRY, 4)},
                                               {nullptr, nullptr, nullptr}, {SD_FULL_ROW(SD_ENTRY, 6)}, {nullptr, nullptr, nullptr}, {SD_FULL_ROW(SD_ENTRY, 8)}};
#define SD_CTRL_ROW(M, DT) M(DT, 0) M(DT, 1) M(DT, 2)
#define SD_DECL_CTRL(DT) SD_CTRL_ROW(SD_DECLARE_CTRL, DT)
SD_TILES(SD_DECL_CTRL)
#define SD_TAB_CTRL(DT) {SD_CTRL_ROW(SD_CENTRY, DT)},
static const sim_launch_fn kCtrlTable[8][3] = {SD_TILES(SD_TAB_CTRL)};
// low-latency small-batch kernels (split_kernel.hpp): [tiles - 5] for 5..8 feature tiles
typedef int (*split_launch_fn)(const SimArgs&, int rf, hipStream_t);
int sd_launch_split_5(const SimArgs& a, int rf, hipStream_t s);
int sd_launch_split_6(const SimArgs& a, int rf, hipStream_t s);
int sd_launch_split_7(const SimArgs& a, int rf, hipStream_t s);
int sd_launch_split_8(const SimArgs& a, int rf, hipStream_t s);
static const split_launch_fn kSplitTable[4] = {sd_launch_split_5, sd_launch_split_6, sd_launch_split_7, sd_launch_split_8};
typedef int (*cmcd_launch_fn)(const CmcdArgs&, int grid, hipStream_t);
#define SD_TAB_CMCD(DT) sd_launch_cmcd_##DT,
static const cmcd_launch_fn kCmcdTable[8] = {SD_TILES(SD_TAB_CMCD)};

// ---- error string ------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define SD_HIP(expr)                                                                  \
  do {                                                                                \
    int e_ = (expr);                                                                  \
    if (e_ != 0) return fail(SDENG_E_HIP, "%s failed: %s", #expr, hipGetErrorString(static_cast<hipError_t>(e_))); \
  } while (0)

extern "C" int sdeng_abi_version(void) { return SDENG_ABI_VERSION; }
extern "C" const char* sdeng_last_error(void) { return g_err; }

// ---- workspace layout ----------------------------------------------------------------------------
static inline size_t align64(size_t n_floats) { return (n_floats + 63) & ~static_cast<size_t>(63); }
// feature tiles of 16: one instantiation per tile count, NT = ceil(d / 16) exactly (d = 100 runs 7 tiles, not 8: no vector or
// matrix work on whole tiles of pad features).  The full-covariance reference kernels stage each precision image in two
// pieces of NT/2 output tiles when NT > 4, so 5 and 7 tiles run on the 6- and 8-tile instantiations.
static int tiles_exact(int d) { return (d + 15) / 16; }
static int tiles_of(const sdeng_desc* d) {
  const int nt = tiles_exact(d->d);
  return (d->ref.kind == SDENG_REF_GMM_FULL && (nt == 5 || nt == 7)) ? nt + 1 : nt;
}
static int dt_index(int NT) { return NT - 1; }

static size_t dist_floats(const sdeng_dist& ds, int dpad) {
  if (ds.kind == SDENG_DIST_GMM_DIAG) return align64(static_cast<size_t>(ds.k) * 2 * dpad) + align64(static_cast<size_t>(ds.k) * 4);
  if (ds.kind == SDENG_DIST_GAUSS_DIAG) return align64(2 * dpad) + align64(4);
  return 0;
}

static bool use_mm(const sdeng_desc* d, int DT);
// SDENG_FLAG_SPLIT_TILES: is the low-latency kernel built for this call?  (The reference kind is checked where it is known.)
static bool split_eligible(const sdeng_desc* d, int DT) {
  return (d->flags & SDENG_FLAG_SPLIT_TILES) && d->B <= 8192 && DT >= 5 && d->net.ctrl_kind == SDENG_CTRL_CLIPPED &&
         (d->form == SDENG_FORM_LIN || d->form == SDENG_FORM_EM) && !d->noise_in &&
         (d->ref.kind == SDENG_REF_NONE || d->ref.kind == SDENG_REF_GAUSS_DIAG || (d->ref.kind == SDENG_REF_GMM_DIAG && d->ref.k <= 4));
}
static int check_x0_dist(const sdeng_desc* d);

// Diagonal mixture references with 4 < K <= 64 components that share one variance vector run on the matrix pipe (RF_GMM_MM):
// forward forms with a ClippedCtrl.  SDENG_REF_MM=0 keeps the vector path (A/B measurements).
static bool use_mm(const sdeng_desc* d, int DT) {
  static const bool allow = [] { const char* e = getenv("SDENG_REF_MM"); return !(e && e[0] == '0'); }();
  if (!allow || d->ref.kind != SDENG_REF_GMM_DIAG || !d->ref.shared_var) return false;
  const int K = d->ref.k;
  if (K <= 4 && K * 2 * 16 * DT <= SD_REFTAB_FLOATS) return false;  // small mixtures: responsibilities in registers (RF_GMM)
  if (K > 64) return false;
  if (d->net.ctrl_kind != SDENG_CTRL_CLIPPED || (d->form != SDENG_FORM_LIN && d->form != SDENG_FORM_EM)) return false;
  const int kt = (K + 15) / 16;
  const int piece = std::max(kt * sd_kb(DT), DT * ((kt + 1) / 2)) * 512;  // floats of the larger image
  const int room = (160 * 1024 - static_cast<int>(sizeof(float)) * sd_lds_weight_floats(DT)) / 2 / static_cast<int>(sizeof(float));
  return piece <= room && piece <= sd_share_buf_floats(SD_SHARE_MAX);
}

struct Layout {
  size_t wpack, temb, stheta, ref_tab, ref_mean, ref_consts, target, ref_dist, prior, rnd_init, trash, logz, cmcd, x0, total;
};

static bool make_layout(const sdeng_desc* d, Layout& L) {
  if (!d || d->d < 1 || d->d > 128 || d->N < 0 || d->B < 0) return false;
  const int DT = tiles_of(d), dpad = 16 * DT;
  size_t o = 0;
  L.wpack = o; o += align64(sd_pack_floats(DT));
  L.temb = o; o += align64(static_cast<size_t>(d->N + 1) * SD_H);  // CMCD evaluates the net at N+1 times
  L.stheta = o; o += align64(d->N + 1);
  const int K = d->ref.kind == SDENG_REF_NONE ? 0 : (d->ref.kind == SDENG_REF_GAUSS_DIAG ? 1 : d->ref.k);
  if (d->ref.kind == SDENG_REF_GMM_FULL) {  // precision images + noised means
    L.ref_tab = o; o += align64(static_cast<size_t>(d->N) * K * DT * sd_kb(DT) * 512 + 256);
    L.ref_mean = o; o += align64(static_cast<size_t>(d->N) * K * dpad);
  } else if (use_mm(d, DT)) {  // logit + mean images, centre / 1/var vectors (+ 64 logit constants per step under ref_consts)
    const int kt = (K + 15) / 16;
    L.ref_tab = o; o += align64(static_cast<size_t>(d->N) * (kt * sd_kb(DT) + DT * ((kt + 1) / 2)) * 512 + 1024);
    L.ref_mean = o; o += align64(static_cast<size_t>(d->N) * 2 * dpad);
  } else {
    L.ref_tab = o; o += align64(static_cast<size_t>(d->N) * K * 2 * dpad);
    L.ref_mean = o;
  }
  L.ref_consts = o; o += align64(static_cast<size_t>(d->N) * (K * 2 > 64 ? K * 2 : 64) + 1);  // + the shared-variance flag (MM: 64 per step)
  L.target = o; o += dist_floats(d->target, dpad);
  L.ref_dist = o; o += dist_floats(d->ref_dist, dpad);
  L.prior = o; o += dist_floats(d->prior, dpad);
  L.rnd_init = o; o += align64(d->B);
  L.trash = o; o += align64((SD_WAVES_MAX > SD_WAVES ? SD_WAVES_MAX : SD_WAVES) * 64 * 4);
  L.logz = o; o += align64(5 * SD_LOGZ_MAX_BLOCKS);
  L.cmcd = o;
  {  // logistic-regression images (CMCD, or the in-loop score of a Score/LerpCtrl) and the CMCD prior's packed precision
    const bool lr_used = d->target.kind == SDENG_DIST_LOGREG && d->target.k > 0 &&
                         (d->form == SDENG_FORM_CMCD || d->form == SDENG_FORM_CMCD_EUBO || d->net.ctrl_kind != SDENG_CTRL_CLIPPED);
    const int n = lr_used ? d->target.k : 0;
    o += align64(sd_lr_floats(DT, n)) + align64(32 * sd_lr_row_kb(n));
    if (d->form == SDENG_FORM_CMCD || d->form == SDENG_FORM_CMCD_EUBO) o += align64(DT * sd_kb(DT) * 512) + align64(16 * DT);
  }
  L.x0 = o;
  if (!d->x_in && !d->x0_out) o += align64(static_cast<size_t>(d->B) * d->d);  // the engine's own x0 draw
  L.total = o;
  return true;
}

// A full-covariance mixture as the target of a score control (SDENG_DIST_GMM_FULL) is evaluated by the machinery of the full-covariance
// REFERENCE: the descriptor is rewritten so that the mixture occupies the (empty) reference slot -- layout, table kernel and the
// RF_GMM_FULL step loop then need no second copy -- and the caller remembers that the slot feeds the control, not the drift.
static bool slot_target(const sdeng_desc* d, sdeng_desc& out) {
  if (!d || d->target.kind != SDENG_DIST_GMM_FULL) return false;
  out = *d;
  out.ref.kind = SDENG_REF_GMM_FULL;
  out.ref.k = d->target.k;
  out.ref.means_init = d->target.loc;
  out.ref.vars_init = d->target.scale;
  out.ref.eigvecs = d->target.aux;
  out.ref.weights = d->target.w;
  out.ref.shared_var = 0;
  memset(&out.target, 0, sizeof(out.target));
  return true;
}
static int check_slot_target(const sdeng_desc* d) {  // d: the caller's descriptor
  if (d->ref.kind != SDENG_REF_NONE)
    return fail(SDENG_E_UNSUPPORTED, "full-covariance mixture target of a score control together with a reference drift (ref.kind %d)", d->ref.kind);
  if (d->net.ctrl_kind != SDENG_CTRL_SCORE && d->net.ctrl_kind != SDENG_CTRL_LERP && d->net.ctrl_kind != SDENG_CTRL_CANCEL_DRIFT)
    return fail(SDENG_E_UNSUPPORTED, "SDENG_DIST_GMM_FULL is the target of a Score / Lerp / CancelDrift control only (ctrl_kind %d)", d->net.ctrl_kind);
  if (d->form != SDENG_FORM_LIN && d->form != SDENG_FORM_EM)
    return fail(SDENG_E_UNSUPPORTED, "full-covariance mixture target of a score control: forward forms only (form %d)", d->form);
  if (d->flags & (SDENG_FLAG_TERM_TARGET | SDENG_FLAG_REMOVE_REF))
    return fail(SDENG_E_UNSUPPORTED, "SDENG_DIST_GMM_FULL: no log-density kernel (FLAG_TERM_TARGET) and no RemoveReferenceCtrl");
  if (d->target.k < 1 || !d->target.loc || !d->target.scale || !d->target.aux)
    return fail(SDENG_E_INVALID, "SDENG_DIST_GMM_FULL: null means / eigenvalues / eigenvectors or k < 1");
  return 0;
}

extern "C" size_t sdeng_workspace_bytes(const sdeng_desc* desc) {
  Layout L;
  sdeng_desc ds;
  if (slot_target(desc, ds)) desc = &ds;
  if (!make_layout(desc, L)) return 0;
  return L.total * sizeof(float);
}

// ---- distribution descriptor -> device view (+ table kernels) -------------------------------------
static int build_dist(const sdeng_dist& in, int d, int dpad, float* ws, DistDev& out, hipStream_t s) {
  memset(&out, 0, sizeof(out));
  out.kind = in.kind;
  out.k = in.k;
  out.p0 = in.p0; out.p1 = in.p1; out.p2 = in.p2; out.p3 = in.p3;
  out.clip = in.clip;
  switch (in.kind) {
    case SDENG_DIST_NONE:
      return 0;
    case SDENG_DIST_GMM_DIAG:
    case SDENG_DIST_GAUSS_DIAG: {
      const int K = in.kind == SDENG_DIST_GMM_DIAG ? in.k : 1;
      if (K < 1 || !in.loc || !in.scale || (in.kind == SDENG_DIST_GMM_DIAG && !in.w))
        return fail(SDENG_E_INVALID, "diagonal Gaussian/mixture needs loc, scale%s", in.kind == SDENG_DIST_GMM_DIAG ? ", w and k >= 1" : "");
      DistTabArgs t;
      t.K = K; t.d = d; t.dpad = dpad;
      t.loc = in.loc; t.scale = in.scale; t.weights = in.kind == SDENG_DIST_GMM_DIAG ? in.w : nullptr;
      t.tab = ws; t.consts = ws + align64(static_cast<size_t>(K) * 2 * dpad);
      SD_HIP(sd_launch_dist_tables(t, s));
      out.k = K;
      out.tab = t.tab; out.consts = t.consts;
      out.p0 = static_cast<float>(0.5 * d * std::log(2.0 * M_PI));  // distr/gauss.py:71
      return 0;
    }
    case SDENG_DIST_ISO_GAUSS:
    case SDENG_DIST_PHI4:
      return 0;
    case SDENG_DIST_GAUSS_FULL:
      if (!in.loc || !in.scale || !in.w) return fail(SDENG_E_INVALID, "GAUSS_FULL needs loc, precision, inverse Cholesky factor");
      out.aux0 = in.loc; out.tab = in.scale; out.aux1 = in.w;
      return 0;
    case SDENG_DIST_LOGREG:
      if (!in.loc || !in.scale || in.k < 1) return fail(SDENG_E_INVALID, "LOGREG needs X, y and k >= 1 rows");
      out.aux0 = in.loc; out.aux1 = in.scale;
      return 0;
    case SDENG_DIST_RINGS:
      if (d != 2) return fail(SDENG_E_INVALID, "RINGS is two-dimensional (d = %d)", d);
      if (!in.loc || !in.w || in.k < 1 || in.k > 8 || !(in.p0 > 0.0f)) return fail(SDENG_E_INVALID, "RINGS needs radii, weights, 1 <= k <= 8, scale > 0");
      out.aux0 = in.loc; out.aux1 = in.w;
      return 0;
    default:
      return fail(SDENG_E_UNSUPPORTED, "unknown distribution kind %d", in.kind);
  }
}

static int check_net(const sdeng_net& n) {
  if (!n.w_in || !n.b_in || !n.w_h1 || !n.b_h1 || !n.w_h2 || !n.b_h2 || !n.w_out || !n.b_out)
    return fail(SDENG_E_INVALID, "drift net: null weight pointer");
  const sdeng_time_embed& te = n.t_embed;
  if (!te.coeff || !te.phase || te.n_hidden != 1 || te.dim_out != SD_H || !te.w[0] || !te.b[0] || !te.w_out || !te.b_out)
    return fail(SDENG_E_UNSUPPORTED, "drift net time embedding must be TimeEmbed(num_layers=2, channels=64)");
  if (n.ctrl_kind != SDENG_CTRL_CLIPPED) {
    const sdeng_time_embed& sm = n.score_model;
    if (sm.n_hidden > 0) {
      if (sm.n_hidden > 4 || sm.dim_out != 1 || !sm.coeff || !sm.phase || !sm.w_out || !sm.b_out)
        return fail(SDENG_E_UNSUPPORTED, "score_model must be TimeEmbed(dim_out=1, num_layers<=5)");
      for (int i = 0; i < sm.n_hidden; ++i)
        if (!sm.w[i] || !sm.b[i]) return fail(SDENG_E_INVALID, "score_model: null weight pointer");
    }
  }
  return 0;
}

// common preparation for simulate / ctrl_forward: pack weights, time embeddings, target tables.
static int prepare_net(const sdeng_desc* d, const Layout& L, float* ws, int DT, SimArgs& a, hipStream_t s, int n_times,
                       bool t_direct, float t_value) {
  int rc = check_net(d->net);
  if (rc) return rc;
  PackArgs pk;
  pk.NT = DT; pk.d = d->d;
  pk.w_in = d->net.w_in; pk.b_in = d->net.b_in; pk.w_h1 = d->net.w_h1; pk.b_h1 = d->net.b_h1;
  pk.w_h2 = d->net.w_h2; pk.b_h2 = d->net.b_h2; pk.w_out = d->net.w_out; pk.b_out = d->net.b_out;
  pk.out = ws + L.wpack;
  pk.transpose = 0; pk.scales = nullptr;
  SD_HIP(sd_launch_pack(pk, s));
  a.wpack = pk.out;
  if (n_times > 0) {
    TimeEmbedArgs te;
    te.te = d->net.t_embed; te.coef = d->coef; te.col = 0; te.t_direct = t_direct; te.t_value = t_value; te.clip = 0.0f;
    te.out = ws + L.temb;
    SD_HIP(sd_launch_time_embed(te, n_times, s));
    a.stheta = nullptr;
    if (d->net.ctrl_kind != SDENG_CTRL_CLIPPED && d->net.score_model.n_hidden > 0) {
      TimeEmbedArgs sm;
      sm.te = d->net.score_model; sm.coef = d->coef; sm.col = 0; sm.t_direct = t_direct; sm.t_value = t_value;
      sm.clip = d->net.clip_model;  // reparam.py:102-110 clips the score model with clip_model
      sm.out = ws + L.stheta;
      SD_HIP(sd_launch_time_embed(sm, n_times, s));
      a.stheta = sm.out;
    }
  }
  a.temb = ws + L.temb;
  a.ctrl_kind = d->net.ctrl_kind;
  a.clip_model = d->net.clip_model;
  a.clip_score = d->net.clip_score;
  a.scale_score = d->net.scale_score;
  return 0;
}

static int score_kind(const sdeng_desc* d, int& sc) {
  sc = SC_NONE;
  if (d->net.ctrl_kind == SDENG_CTRL_CLIPPED) return 0;
  if (d->net.ctrl_kind != SDENG_CTRL_SCORE && d->net.ctrl_kind != SDENG_CTRL_LERP && d->net.ctrl_kind != SDENG_CTRL_CANCEL_DRIFT)
    return fail(SDENG_E_UNSUPPORTED, "unknown ctrl_kind %d", d->net.ctrl_kind);
  if (d->target.kind == SDENG_DIST_GMM_DIAG || d->target.kind == SDENG_DIST_RINGS) sc = SC_GMM;  // rings: runtime branch of the d <= 16 kernel
  else if (d->target.kind == SDENG_DIST_PHI4) sc = SC_PHI4;
  else if (d->target.kind == SDENG_DIST_LOGREG) sc = SC_LOGREG;
  else return fail(SDENG_E_UNSUPPORTED, "ScoreCtrl/LerpCtrl: no in-loop score kernel for target kind %d", d->target.kind);
  if (d->net.ctrl_kind == SDENG_CTRL_LERP && d->prior.kind != SDENG_DIST_ISO_GAUSS)
    return fail(SDENG_E_UNSUPPORTED, "LerpCtrl needs an IsotropicGauss prior (kind %d given)", d->prior.kind);
  return 0;
}

static int grid_for(int ntiles) {
  // one persistent workgroup per CU (LDS image + 2 waves/SIMD fill a CU); tiles are dealt to workgroups
  // first, to the waves of a workgroup second, so a small batch spreads over the whole chip
  int g = ntiles > 256 ? 256 : ntiles;
  return g < 1 ? 1 : g;
}

static int grid_for(int ntiles);
// logistic-regression target -> the two LDS images + the in-kernel constants (a.lr); returns the first float after them
static int prepare_logreg(const sdeng_desc* d, const Layout& L, float* ws, int DT, SimArgs& a, hipStream_t s, float** next) {
  const int n = d->target.k;
  float* image = ws + L.cmcd;
  float* y_pad = image + align64(sd_lr_floats(DT, n));
  *next = y_pad + align64(32 * sd_lr_row_kb(n));
  if (n < 1 || !d->target.loc || !d->target.scale) return fail(SDENG_E_INVALID, "LOGREG needs X, y and k >= 1 rows");
  SD_HIP(sd_launch_logreg_images(d->target.loc, d->target.scale, n, d->d - 1, DT, image, y_pad, s));
  a.lr.image = image; a.lr.y_pad = y_pad; a.lr.n_rows = n;
  // sonar (166 x 61) sits in LDS next to the drift net; larger design matrices (credit: 800 rows) are read through L2 instead
  a.lr.in_lds = static_cast<size_t>(cmcd_lds_floats(DT, n)) * sizeof(float) <= 160 * 1024 ? 1 : 0;
  a.lr.inv_w_scale2 = 1.0f / (d->target.p0 * d->target.p0); a.lr.c_mean = d->target.p1; a.lr.inv_c_scale2 = 1.0f / (d->target.p2 * d->target.p2);
  // sigmoid range with a gradient: inside clip(thr, 1 - thr) and inside the eps clamp of probs_to_logits
  const float thr = d->target.p3, eps = 1.1920928955078125e-07f;
  a.lr.p_lo = thr > eps ? thr : eps;
  a.lr.p_hi = (1.0f - thr) < (1.0f - eps) ? (1.0f - thr) : (1.0f - eps);
  return 0;
}

// ControlledLangevinSDELoss.simulate (losses/oc.py:666-755): logistic-regression target, Gaussian prior
static int simulate_cmcd(const sdeng_desc* d, const Layout& L, float* ws, int DT, SimArgs& a, hipStream_t s) {
  const int dpad = 16 * DT;
  const bool logreg = d->target.kind == SDENG_DIST_LOGREG;
  if (!logreg && d->target.kind != SDENG_DIST_GMM_DIAG && d->target.kind != SDENG_DIST_GAUSS_DIAG && d->target.kind != SDENG_DIST_PHI4)
    return fail(SDENG_E_UNSUPPORTED, "CMCD kernel: target must be LOGREG, GMM_DIAG, GAUSS_DIAG or PHI4 (kind %d)", d->target.kind);
  if (logreg && DT > 4) return fail(SDENG_E_UNSUPPORTED, "CMCD kernel: logistic regression with d <= 64 (got %d)", d->d);
  const int n = logreg ? d->target.k : 0;  // data rows held in LDS
  if (logreg && n < 1) return fail(SDENG_E_INVALID, "CMCD kernel: logistic regression without data rows");
  if (d->prior.kind != SDENG_DIST_GAUSS_FULL && d->prior.kind != SDENG_DIST_ISO_GAUSS && d->prior.kind != SDENG_DIST_GAUSS_DIAG)
    return fail(SDENG_E_UNSUPPORTED, "CMCD kernel: prior must be GAUSS_FULL, GAUSS_DIAG or ISO_GAUSS (kind %d)", d->prior.kind);
  if (d->net.ctrl_kind != SDENG_CTRL_CLIPPED && d->net.ctrl_kind != SDENG_CTRL_SCORE)
    return fail(SDENG_E_UNSUPPORTED, "CMCD kernel: ClippedCtrl or ScoreCtrl");
  if (!(d->flags & SDENG_FLAG_INIT_LOGP) || !(d->flags & SDENG_FLAG_TERM_TARGET))
    return fail(SDENG_E_UNSUPPORTED, "CMCD kernel implements the eval path (rnd0 = log p_prior, terminal -log pi)");
  int rc = prepare_net(d, L, ws, DT, a, s, d->N + 1, false, 0.0f);
  if (rc) return rc;
  DistDev target, prior;
  rc = build_dist(d->target, d->d, dpad, ws + L.target, target, s);
  if (rc) return rc;
  rc = build_dist(d->prior, d->d, dpad, ws + L.prior, prior, s);
  if (rc) return rc;
  CmcdArgs c;
  memset(&c, 0, sizeof(c));
  float* prec = ws + L.cmcd;
  if (logreg) {
    rc = prepare_logreg(d, L, ws, DT, a, s, &prec);
    if (rc) return rc;
  }
  float* locp = prec + align64(DT * sd_kb(DT) * 512);
  if (d->prior.kind == SDENG_DIST_GAUSS_FULL) {
    SD_HIP(sd_launch_pack_square(d->prior.scale, d->prior.loc, d->d, DT, prec, locp, s));
    c.prec_pack = prec; c.prior_loc = locp;
  } else if (d->prior.kind == SDENG_DIST_ISO_GAUSS) {
    c.iso_loc = d->prior.p0; c.inv_iso_var = 1.0f / d->prior.p3;
  }
  const bool eubo = d->form == SDENG_FORM_CMCD_EUBO;
  TerminalArgs t;
  t.B = d->B; t.d = d->d; t.dpad = dpad;
  if (eubo) {  // rnd0 = -log pi~(x_in)   (losses/oc.py:779)
    if (d->target.kind != SDENG_DIST_GMM_DIAG && d->target.kind != SDENG_DIST_GAUSS_DIAG)
      return fail(SDENG_E_UNSUPPORTED, "CMCD compute_eubo kernels: diagonal Gaussian / mixture targets (kind %d)", d->target.kind);
    SD_HIP(hipMemsetAsync(ws + L.rnd_init, 0, sizeof(float) * d->B, s));
    t.ref = target; t.target = target; t.use_ref = 0; t.use_target = 1; t.x = d->x_in; t.rnd = ws + L.rnd_init;
    SD_HIP(sd_launch_terminal(t, s));
  } else {     // rnd0 = log p_prior(x0)
    DistEvalArgs e;
    e.ds = prior; e.B = d->B; e.d = d->d; e.dpad = dpad; e.x = d->x_in; e.logp_out = ws + L.rnd_init; e.score_out = nullptr;
    SD_HIP(sd_launch_dist_eval(e, s));
  }
  a.rnd_init = ws + L.rnd_init;
  a.cmcd_g = d->cmcd_g; a.cmcd_clip = d->cmcd_clip;
  a.target = target; a.prior = prior;
  c.s = a;
  if (d->ev_start) SD_HIP(hipEventRecord(static_cast<hipEvent_t>(d->ev_start), s));
  SD_HIP(kCmcdTable[dt_index(DT)](c, grid_for(a.ntiles), s));
  if (d->ev_stop) SD_HIP(hipEventRecord(static_cast<hipEvent_t>(d->ev_stop), s));
  // terminal: -log pi~(x_N)  (:752), or for the noising loop + log p_prior(x_noised)  (:825)
  if (eubo) { t.ref = prior; t.target = prior; t.use_ref = 1; t.use_target = 0; }
  else { t.ref = target; t.target = target; t.use_ref = 0; t.use_target = 1; }
  t.x = d->x_out; t.rnd = d->rnd_out;
  SD_HIP(sd_launch_terminal(t, s));
  return 0;
}

typedef int (*euler_launch_fn)(const SimArgs&, int sc, hipStream_t);
#define SD_DECLARE_EULER(DT) int sd_launch_euler_##DT(const SimArgs& a, int sc, hipStream_t s);
SD_TILES(SD_DECLARE_EULER)
#define SD_TAB_EULER(DT) sd_launch_euler_##DT,
static const euler_launch_fn kEulerTable[8] = {SD_TILES(SD_TAB_EULER)};

// SDENG_CTRL_NONE: Euler-Maruyama of an SDE without a drift net (euler_kernel.hpp)
static int simulate_euler(const sdeng_desc* d, const Layout& L, float* ws, int DT, SimArgs& a, hipStream_t s) {
  if (d->form != SDENG_FORM_EM) return fail(SDENG_E_UNSUPPORTED, "CTRL_NONE (no drift net) runs the Euler-Maruyama form only (form %d given)", d->form);
  if (d->ref.kind != SDENG_REF_NONE) return fail(SDENG_E_UNSUPPORTED, "CTRL_NONE with a reference drift");
  if (d->flags & (SDENG_FLAG_TERM_REF | SDENG_FLAG_TERM_TARGET | SDENG_FLAG_INIT_LOGP))
    return fail(SDENG_E_UNSUPPORTED, "CTRL_NONE carries no log-weight: terminal / initial cost flags are not accepted");
  const int dpad = 16 * DT;
  int sc = SC_NONE;
  DistDev target;
  int rc = build_dist(d->target, d->d, dpad, ws + L.target, target, s);
  if (rc) return rc;
  if (d->target.kind == SDENG_DIST_GMM_DIAG || d->target.kind == SDENG_DIST_RINGS) sc = SC_GMM;
  else if (d->target.kind == SDENG_DIST_PHI4) sc = SC_PHI4;
  else if (d->target.kind != SDENG_DIST_NONE)
    return fail(SDENG_E_UNSUPPORTED, "Langevin drift: no in-loop score kernel for target kind %d", d->target.kind);
  a.target = target;
  a.clip_score = d->net.clip_score;
  if (d->ev_start) SD_HIP(hipEventRecord(static_cast<hipEvent_t>(d->ev_start), s));
  rc = kEulerTable[dt_index(DT)](a, sc, s);
  SD_HIP(rc);
  if (d->ev_stop) SD_HIP(hipEventRecord(static_cast<hipEvent_t>(d->ev_stop), s));
  return 0;
}

extern "C" int sdeng_simulate(const sdeng_desc* d, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!d) return fail(SDENG_E_INVALID, "null descriptor");
  if (d->abi_version != SDENG_ABI_VERSION) return fail(SDENG_E_INVALID, "ABI version %d, library has %d", d->abi_version, SDENG_ABI_VERSION);
  sdeng_desc dslot;
  const bool target_in_slot = d->target.kind == SDENG_DIST_GMM_FULL;
  if (target_in_slot) {
    int rcs = check_slot_target(d);
    if (rcs) return rcs;
    slot_target(d, dslot);
    d = &dslot;
  }
  Layout L;
  if (!make_layout(d, L)) return fail(SDENG_E_INVALID, "bad sizes: B=%d d=%d N=%d (need 1 <= d <= 128)", d->B, d->d, d->N);
  if (d->B == 0) return 0;
  if ((!d->coef && d->N > 0) || !d->x_out || !d->rnd_out) return fail(SDENG_E_INVALID, "null coef/x_out/rnd_out");
  if (!d->x_in) {
    int rcx = check_x0_dist(d);
    if (rcx) return rcx;
  }
  if (!d->workspace || d->workspace_bytes < L.total * sizeof(float))
    return fail(SDENG_E_WORKSPACE, "workspace %zu bytes, need %zu", d->workspace_bytes, L.total * sizeof(float));
  if (static_cast<long long>(d->B) * d->d >= (1ll << 31)) return fail(SDENG_E_UNSUPPORTED, "B*d >= 2^31");
  if (d->form != SDENG_FORM_LIN && d->form != SDENG_FORM_EM && d->form != SDENG_FORM_CMCD && d->form != SDENG_FORM_EUBO &&
      d->form != SDENG_FORM_CMCD_EUBO)
    return fail(SDENG_E_INVALID, "unknown form %d", d->form);
  const int DT = tiles_of(d), dpad = 16 * DT;
  float* ws = static_cast<float*>(d->workspace);

  SimArgs a;
  memset(&a, 0, sizeof(a));
  a.form = d->form; a.flags = d->flags;
  a.B = d->B; a.d = d->d; a.N = d->N;
  a.particle0 = d->particle0;
  a.seed_lo = static_cast<unsigned>(d->seed & 0xFFFFFFFFull);
  a.seed_hi = static_cast<unsigned>(d->seed >> 32);
  a.coef = d->coef; a.x_in = d->x_in; a.x_out = d->x_out; a.rnd_out = d->rnd_out;
  a.xs_out = d->xs_out; a.noise_in = d->noise_in;
  a.trash = ws + L.trash;
  a.ntiles = (d->B + 15) / 16;
  // x0 drawn by the engine: k_sample_x0 writes it (workspace, or the caller's x0_out), then everything runs as if the caller had passed
  // it.  A twin of the step-loop kernel that drew x0 in registers was built and dropped: same instruction count, but its loop came out
  // of the register allocator 4 % slower on cfg 2 and 1.3 % on cfg 3 (profiles/r02_x0_draw_ab.log); the sampler kernel costs 0.2 %.
  sdeng_desc dm;
  if (!d->x_in) {
    float* x0 = d->x0_out ? d->x0_out : ws + L.x0;
    SD_HIP(sd_launch_sample_x0(d->x0_dist, a.seed_lo, a.seed_hi, d->particle0, d->B, d->d, x0, s));
    dm = *d;
    dm.x_in = x0;
    d = &dm;
    a.x_in = x0;
  }

  if (d->form == SDENG_FORM_CMCD || d->form == SDENG_FORM_CMCD_EUBO) return simulate_cmcd(d, L, ws, DT, a, s);
  if (d->net.ctrl_kind == SDENG_CTRL_NONE) return simulate_euler(d, L, ws, DT, a, s);
  int rc = prepare_net(d, L, ws, DT, a, s, d->N, false, 0.0f);
  if (rc) return rc;
  int sc;
  if (target_in_slot) {
    sc = SC_REFSLOT;
    if (d->net.ctrl_kind == SDENG_CTRL_LERP && d->prior.kind != SDENG_DIST_ISO_GAUSS)
      return fail(SDENG_E_UNSUPPORTED, "LerpCtrl needs an IsotropicGauss prior (kind %d given)", d->prior.kind);
  } else {
    rc = score_kind(d, sc);
    if (rc) return rc;
  }

  // reference drift tables
  int rf = RF_NONE;
  if (d->ref.kind == SDENG_REF_GAUSS_DIAG || d->ref.kind == SDENG_REF_GMM_DIAG) {
    const int K = d->ref.kind == SDENG_REF_GAUSS_DIAG ? 1 : d->ref.k;
    rf = d->ref.kind == SDENG_REF_GAUSS_DIAG ? RF_GAUSS : ((K <= 4 && K * 2 * dpad <= SD_REFTAB_FLOATS) ? RF_GMM : RF_GMM_BIG);
    if (K < 1 || !d->ref.means_init || !d->ref.vars_init) return fail(SDENG_E_INVALID, "reference: null means/vars or k < 1");
    if (use_mm(d, DT)) {
      rf = RF_GMM_MM;
      const int kt = (K + 15) / 16;
      if (d->N > 0) {
        RefMMArgs r;
        r.K = K; r.d = d->d; r.dpad = dpad; r.NT = DT; r.kt = kt; r.coef = d->coef;
        r.means = d->ref.means_init; r.vars = d->ref.vars_init; r.weights = d->ref.weights;
        r.images = ws + L.ref_tab; r.centre = ws + L.ref_mean; r.consts = ws + L.ref_consts;
        r.same_var = ws + L.ref_consts + static_cast<size_t>(d->N) * 64;  // the slot behind the per-step constants (make_layout: N * 64 + 1)
        SD_HIP(sd_launch_ref_mm_tables(r, d->N, s));
      }
      a.ref_k = K; a.ref_kc = kt;
      a.ref_tab = ws + L.ref_tab; a.ref_mean = ws + L.ref_mean; a.ref_consts = ws + L.ref_consts;
      const int piece = std::max(kt * sd_kb(DT), DT * ((kt + 1) / 2)) * 512;
      a.ref_share = (piece / 256 + SD_WAVES - 1) / SD_WAVES;
    } else if (d->N > 0) {
      RefTabArgs r;
      r.K = K; r.d = d->d; r.dpad = dpad; r.coef = d->coef;
      r.means = d->ref.means_init; r.vars = d->ref.vars_init; r.weights = rf != RF_GAUSS ? d->ref.weights : nullptr;
      r.tab = ws + L.ref_tab; r.consts = ws + L.ref_consts;
      r.same_var = ws + L.ref_consts + static_cast<size_t>(d->N) * K * 2;
      r.centred = (rf == RF_GMM && !split_eligible(d, DT)) ? 1 : 0;  // the split-tile kernel reads the plain (mean, 1/var) table
      SD_HIP(sd_launch_ref_tables(r, d->N, s));
    }
    if (rf != RF_GMM_MM) a.ref_k = K;
    if (rf == RF_GMM_BIG) {
      // workgroup-shared table copy: two LDS buffers of whole 1 KiB chunks behind the drift-net weights (160 KiB of
      // LDS per workgroup), each holding a piece of kc components -- the whole table when it fits, otherwise the
      // fewest equal pieces that do.  SDENG_REF_SHARE=0 keeps the streamed-from-L2 path (A/B measurements).
      static const bool allow = [] { const char* e = getenv("SDENG_REF_SHARE"); return !(e && e[0] == '0'); }();
      const int room = (160 * 1024 - static_cast<int>(sizeof(float)) * sd_lds_weight_floats(DT)) / 2 / static_cast<int>(sizeof(float));
      const int cap = std::min(room, sd_share_buf_floats(SD_SHARE_MAX)) / 256 * 256;  // floats per buffer
      const int kc_max = cap / (2 * dpad);
      if (allow && kc_max >= 1) {
        const int nch = (K + kc_max - 1) / kc_max;
        const int kc = (K + nch - 1) / nch;
        const int chunks = (kc * 2 * dpad + 255) / 256;
        a.ref_share = (chunks + SD_WAVES - 1) / SD_WAVES;
        a.ref_kc = kc;
      }
    }
    if (rf != RF_GMM_MM) {
      a.ref_tab = ws + L.ref_tab; a.ref_consts = ws + L.ref_consts;
      a.ref_same_var = ws + L.ref_consts + static_cast<size_t>(d->N) * K * 2;
    }
    a.ref_c1 = static_cast<float>(0.5 * d->d * std::log(2.0 * M_PI));
  } else if (d->ref.kind == SDENG_REF_GMM_FULL) {
    const int K = d->ref.k;
    rf = RF_GMM_FULL;
    if (K < 1 || !d->ref.means_init || !d->ref.vars_init || !d->ref.eigvecs)
      return fail(SDENG_E_INVALID, "full-covariance reference: null means / eigenvalues / eigenvectors or k < 1");
    if (d->N > 0) {
      RefFullArgs r;
      r.K = K; r.d = d->d; r.dpad = dpad; r.NT = DT; r.coef = target_in_slot ? nullptr : d->coef;  // a target does not diffuse
      r.means = d->ref.means_init; r.eigvals = d->ref.vars_init; r.eigvecs = d->ref.eigvecs; r.weights = d->ref.weights;
      r.images = ws + L.ref_tab; r.means_out = ws + L.ref_mean; r.consts = ws + L.ref_consts;
      SD_HIP(sd_launch_ref_full_tables(r, d->N, s));
    }
    a.ref_k = K;
    a.ref_tab = ws + L.ref_tab; a.ref_mean = ws + L.ref_mean; a.ref_consts = ws + L.ref_consts;
    a.ref_c1 = static_cast<float>(0.5 * d->d * std::log(2.0 * M_PI));
    const int piece = (DT > 4 ? DT / 2 : DT) * sd_kb(DT) * 512;  // floats per staged piece of an image (sim_kernel.hpp FULL_PIECE)
    a.ref_share = (piece / 256 + SD_WAVES - 1) / SD_WAVES;
  } else if (d->ref.kind != SDENG_REF_NONE) {
    return fail(SDENG_E_UNSUPPORTED, "reference kind %d", d->ref.kind);
  }

  DistDev target, ref_dist, prior;
  rc = build_dist(d->target, d->d, dpad, ws + L.target, target, s);
  if (rc) return rc;
  rc = build_dist(d->ref_dist, d->d, dpad, ws + L.ref_dist, ref_dist, s);
  if (rc) return rc;
  rc = build_dist(d->prior, d->d, dpad, ws + L.prior, prior, s);
  if (rc) return rc;
  a.target = target;
  a.prior = prior;

  // initial cost (EUBO: the prior log-density is added at the END, on the noised samples: losses/oc.py:1032)
  if ((d->flags & SDENG_FLAG_INIT_LOGP) && d->form != SDENG_FORM_EUBO) {
    if (prior.kind == SDENG_DIST_NONE) return fail(SDENG_E_INVALID, "FLAG_INIT_LOGP without a prior");
    DistEvalArgs e;
    e.ds = prior; e.B = d->B; e.d = d->d; e.dpad = dpad; e.x = d->x_in; e.logp_out = ws + L.rnd_init; e.score_out = nullptr;
    SD_HIP(sd_launch_dist_eval(e, s));
    a.rnd_init = ws + L.rnd_init;
  }

  const bool tr = d->flags & SDENG_FLAG_TERM_REF, tt = d->flags & SDENG_FLAG_TERM_TARGET;
  if (tr && ref_dist.kind == SDENG_DIST_NONE) return fail(SDENG_E_INVALID, "FLAG_TERM_REF without ref_dist");
  if (tt && target.kind == SDENG_DIST_NONE) return fail(SDENG_E_INVALID, "FLAG_TERM_TARGET without target");
  TerminalArgs t;
  t.ref = ref_dist; t.target = target; t.use_ref = tr; t.use_target = tt;
  t.B = d->B; t.d = d->d; t.dpad = dpad;

  if ((d->flags & SDENG_FLAG_REMOVE_REF) && (sc == SC_NONE || sc == SC_LOGREG || rf == RF_NONE || rf == RF_GMM_FULL || rf == RF_GMM_MM || d->form == SDENG_FORM_EUBO))
    return fail(SDENG_E_UNSUPPORTED, "FLAG_REMOVE_REF (RemoveReferenceCtrl): forward forms with a Score / Lerp / CancelDrift control on a mixture or "
                                     "phi^4 target and a diagonal Gaussian / mixture reference (ctrl_kind %d, ref.kind %d, form %d)",
                d->net.ctrl_kind, d->ref.kind, d->form);
  sim_launch_fn fn;
  if (d->form == SDENG_FORM_EUBO) {
    if ((rf == RF_NONE) == (sc == SC_NONE))
      return fail(SDENG_E_UNSUPPORTED, "compute_eubo kernels: a reference drift with a ClippedCtrl, or no reference with a Score/LerpCtrl "
                                       "(ref.kind %d, ctrl_kind %d)", d->ref.kind, d->net.ctrl_kind);
    if (sc == SC_LOGREG) return fail(SDENG_E_UNSUPPORTED, "compute_eubo kernels: no logistic-regression control score");
    fn = rf == RF_GMM_FULL ? kFullTable[dt_index(DT)][2] : kEuboTable[dt_index(DT)][rf != RF_NONE ? rf - 1 : 2 + sc];
    if (tr || tt) {  // cost at the data distribution: rnd0 = [log p_ref(x_in)] - log pi~(x_in)   (losses/oc.py:322, :536, :1003)
      SD_HIP(hipMemsetAsync(ws + L.rnd_init, 0, sizeof(float) * d->B, s));
      t.x = d->x_in; t.rnd = ws + L.rnd_init;
      SD_HIP(sd_launch_terminal(t, s));
      a.rnd_init = ws + L.rnd_init;
    }
  } else if (sc == SC_LOGREG) {
    // ScoreCtrl / LerpCtrl on a logistic-regression target (PIS, DDS, DIS on the Bayesian benchmarks): design matrix in LDS
    if (rf != RF_NONE) return fail(SDENG_E_UNSUPPORTED, "in-loop logistic-regression score together with a reference drift");
    if (DT > 4) return fail(SDENG_E_UNSUPPORTED, "in-loop logistic-regression score: d <= 64 (got %d)", d->d);
    float* unused;
    rc = prepare_logreg(d, L, ws, DT, a, s, &unused);
    if (rc) return rc;
    fn = kLogregTable[dt_index(DT)][d->form];
  } else if (rf == RF_GMM_MM) {
    fn = kMMTable[dt_index(DT)][d->form];
  } else if (rf == RF_GMM_FULL && sc == SC_REFSLOT) {
    fn = kFullScoreTable[dt_index(DT)][d->form];
  } else if (rf == RF_GMM_FULL) {
    if (sc != SC_NONE) return fail(SDENG_E_UNSUPPORTED, "full-covariance reference together with a Score/LerpCtrl");
    fn = kFullTable[dt_index(DT)][d->form];
  } else {
    fn = kSimTable[dt_index(DT)][rf][sc][d->form];
  }
  if (d->ev_start) SD_HIP(hipEventRecord(static_cast<hipEvent_t>(d->ev_start), s));
  if (split_eligible(d, DT) && (rf == RF_NONE || rf == RF_GAUSS || rf == RF_GMM)) SD_HIP(kSplitTable[DT - 5](a, rf, s));
  else SD_HIP(fn(a, grid_for(a.ntiles), s));
  if (d->ev_stop) SD_HIP(hipEventRecord(static_cast<hipEvent_t>(d->ev_stop), s));

  // terminal cost
  if (d->form != SDENG_FORM_EUBO && (tr || tt)) {
    t.x = d->x_out; t.rnd = d->rnd_out;
    SD_HIP(sd_launch_terminal(t, s));
  }
  if (d->form == SDENG_FORM_EUBO && (d->flags & SDENG_FLAG_INIT_LOGP)) {  // rnd += log p_prior(x_noised)
    if (prior.kind == SDENG_DIST_NONE) return fail(SDENG_E_INVALID, "FLAG_INIT_LOGP without a prior");
    TerminalArgs tp;
    tp.ref = prior; tp.target = prior; tp.use_ref = 1; tp.use_target = 0;
    tp.B = d->B; tp.d = d->d; tp.dpad = dpad; tp.x = d->x_out; tp.rnd = d->rnd_out;
    SD_HIP(sd_launch_terminal(tp, s));
  }
  return 0;
}

static int check_x0_dist(const sdeng_desc* d) {
  const sdeng_dist& q = d->x0_dist;
  if (d->form == SDENG_FORM_EUBO || d->form == SDENG_FORM_CMCD_EUBO)
    return fail(SDENG_E_INVALID, "the noising loops start from samples of the TARGET: x_in is required");
  if (q.kind == SDENG_DIST_ISO_GAUSS) return 0;
  if (q.kind == SDENG_DIST_GAUSS_DIAG) return q.loc ? 0 : fail(SDENG_E_INVALID, "x0_dist GAUSS_DIAG needs loc (scale may be NULL: x0 = loc)");
  if (q.kind == SDENG_DIST_GAUSS_FULL) return (q.loc && q.aux) ? 0 : fail(SDENG_E_INVALID, "x0_dist GAUSS_FULL needs loc and the Cholesky factor (aux)");
  return fail(SDENG_E_UNSUPPORTED, "x_in == NULL: no sampler for x0_dist kind %d (ISO_GAUSS, GAUSS_DIAG, GAUSS_FULL)", q.kind);
}

extern "C" int sdeng_sample_x0(const sdeng_dist* dist, uint64_t seed, int64_t particle0, int32_t B, int32_t d, float* out, void* stream) {
  if (!dist || !out || B < 0 || d < 1 || d > 128) return fail(SDENG_E_INVALID, "bad argument");
  if (B == 0) return 0;
  sdeng_desc tmp;
  memset(&tmp, 0, sizeof(tmp));
  tmp.x0_dist = *dist;
  int rc = check_x0_dist(&tmp);
  if (rc) return rc;
  SD_HIP(sd_launch_sample_x0(*dist, static_cast<unsigned>(seed & 0xFFFFFFFFull), static_cast<unsigned>(seed >> 32), particle0, B, d, out,
                             static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int sdeng_ctrl_forward(const sdeng_desc* d, float t_net, float score_gain, float lerp_w, const float* x, float* u_out,
                                  void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!d || !x || !u_out) return fail(SDENG_E_INVALID, "null argument");
  Layout L;
  if (!make_layout(d, L)) return fail(SDENG_E_INVALID, "bad sizes");
  if (!d->workspace || d->workspace_bytes < L.total * sizeof(float)) return fail(SDENG_E_WORKSPACE, "workspace too small");
  if (d->B == 0) return 0;
  const int DT = tiles_of(d), dpad = 16 * DT;
  float* ws = static_cast<float*>(d->workspace);
  SimArgs a;
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.d = d->d; a.N = 1;
  a.x_in = x; a.x_out = u_out;
  a.trash = ws + L.trash;
  a.ntiles = (d->B + 15) / 16;
  int rc = prepare_net(d, L, ws, DT, a, s, 1, true, t_net);
  if (rc) return rc;
  int sc;
  rc = score_kind(d, sc);
  if (rc) return rc;
  DistDev target, prior;
  rc = build_dist(d->target, d->d, dpad, ws + L.target, target, s);
  if (rc) return rc;
  rc = build_dist(d->prior, d->d, dpad, ws + L.prior, prior, s);
  if (rc) return rc;
  a.target = target; a.prior = prior;
  // one-row coefficient table for (score_gain, lerp_w), kept in the (unused) reference-consts slot
  float host_coef[SDENG_NCOEF] = {0};
  host_coef[0] = t_net; host_coef[7] = score_gain; host_coef[8] = lerp_w;
  float* dev_coef = ws + L.logz;
  SD_HIP(hipMemcpyAsync(dev_coef, host_coef, sizeof(host_coef), hipMemcpyHostToDevice, s));
  a.coef = dev_coef;
  SD_HIP(kCtrlTable[dt_index(DT)][sc](a, grid_for(a.ntiles), s));
  return 0;
}

// ---- fused forward + backward of the drift net over N * B rows (training direction) ---------------------------------------------
#define SD_DECLARE_VJP(DT) int sd_launch_vjp_##DT(const VjpArgs& a, int grid, hipStream_t s);
SD_TILES(SD_DECLARE_VJP)
typedef int (*vjp_launch_fn)(const VjpArgs&, int grid, hipStream_t);
#define SD_DECLARE_ADJ(DT) int sd_launch_adjoint_##DT(const AdjArgs& a, int grid, hipStream_t s);
SD_TILES(SD_DECLARE_ADJ)
typedef int (*adj_launch_fn)(const AdjArgs&, int grid, hipStream_t);
#define SD_TAB_ADJ(DT) sd_launch_adjoint_##DT,
static const adj_launch_fn kAdjTable[8] = {SD_TILES(SD_TAB_ADJ)};
#define SD_TAB_VJP(DT) sd_launch_vjp_##DT,
static const vjp_launch_fn kVjpTable[8] = {SD_TILES(SD_TAB_VJP)};

static size_t vjp_floats(int DT, int n_times, size_t* o_wt, size_t* o_temb, size_t* o_trash) {
  size_t o = align64(sd_pack_floats(DT));
  *o_wt = o; o += align64(sd_lds_weight_floats(DT));
  *o_temb = o; o += align64(static_cast<size_t>(n_times) * SD_H);
  *o_trash = o; o += align64(SD_WAVES_MAX * 64 * 4);
  return o;
}
extern "C" size_t sdeng_ctrl_vjp_workspace_bytes(int32_t d, int32_t n_times) {
  if (d < 1 || d > 128 || n_times < 1) return 0;
  size_t a, b, c;
  return vjp_floats(tiles_exact(d), n_times, &a, &b, &c) * sizeof(float);
}
extern "C" int sdeng_ctrl_vjp(const sdeng_desc* d, int32_t n_times, int32_t rows_per_time, const float* x, const float* cot, float* a0,
                              float* a1, float* a2, float* d0, float* d1, float* d2, float* dout, float* gx, float* u_out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!d || !x) return fail(SDENG_E_INVALID, "null argument");
  if (cot && (!a0 || !a1 || !a2 || !d0 || !d1 || !d2 || !dout)) return fail(SDENG_E_INVALID, "backward pass: every per-row output is required");
  if (!cot && !u_out) return fail(SDENG_E_INVALID, "nothing to compute: no cotangent and no u_out");
  if (d->abi_version != SDENG_ABI_VERSION) return fail(SDENG_E_INVALID, "ABI version %d, library has %d", d->abi_version, SDENG_ABI_VERSION);
  if (d->d < 1 || d->d > 128 || n_times < 1 || rows_per_time < 1 || !d->coef) return fail(SDENG_E_INVALID, "bad sizes (1 <= d <= 128, n_times, rows_per_time >= 1) or null coef");
  if (d->net.ctrl_kind != SDENG_CTRL_CLIPPED) return fail(SDENG_E_UNSUPPORTED, "ctrl_vjp: ClippedCtrl around the FourierMLP (ctrl_kind %d given)", d->net.ctrl_kind);
  const long long M = static_cast<long long>(n_times) * rows_per_time;
  if (M * d->d >= (1ll << 31)) return fail(SDENG_E_UNSUPPORTED, "rows * d >= 2^31");
  int rc = check_net(d->net);
  if (rc) return rc;
  const int DT = tiles_exact(d->d);
  size_t o_wt, o_temb, o_trash;
  const size_t need = vjp_floats(DT, n_times, &o_wt, &o_temb, &o_trash) * sizeof(float);
  if (!d->workspace || d->workspace_bytes < need) return fail(SDENG_E_WORKSPACE, "workspace %zu bytes, need %zu", d->workspace_bytes, need);
  float* ws = static_cast<float*>(d->workspace);
  if (!(d->flags & SDENG_FLAG_REUSE_PACK)) {  // (a caller stepping through the times one by one packs once: the images stay valid in the workspace)
    PackArgs pk;
    pk.NT = DT; pk.d = d->d;
    pk.w_in = d->net.w_in; pk.b_in = d->net.b_in; pk.w_h1 = d->net.w_h1; pk.b_h1 = d->net.b_h1;
    pk.w_h2 = d->net.w_h2; pk.b_h2 = d->net.b_h2; pk.w_out = d->net.w_out; pk.b_out = d->net.b_out;
    pk.out = ws; pk.transpose = 0; pk.scales = nullptr;
    SD_HIP(sd_launch_pack(pk, s));
    PackArgs pt = pk;
    pt.out = ws + o_wt; pt.transpose = 1; pt.scales = ws + sd_off_scales(DT);
    SD_HIP(sd_launch_pack(pt, s));
  }
  TimeEmbedArgs te;
  te.te = d->net.t_embed; te.coef = d->coef; te.col = 0; te.t_direct = 0; te.t_value = 0.0f; te.clip = 0.0f;
  te.out = ws + o_temb;
  SD_HIP(sd_launch_time_embed(te, n_times, s));
  VjpArgs a;
  memset(&a, 0, sizeof(a));
  a.M = static_cast<int>(M); a.B = rows_per_time; a.d = d->d; a.N = n_times;
  a.x = x; a.cot = cot; a.wpack = ws; a.wpack_t = ws + o_wt; a.temb = ws + o_temb;
  a.clip_model = d->net.clip_model;
  a.a0 = a0; a.a1 = a1; a.a2 = a2; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.dout = dout; a.gx = gx; a.u_out = u_out;
  a.trash = ws + o_trash;
  a.ntiles = static_cast<int>((M + 15) / 16);
  SD_HIP(kVjpTable[dt_index(DT)](a, grid_for(a.ntiles), s));
  return 0;
}

// ---- KL training: the adjoint of the step loop (grad_kernel.hpp k_kl_adjoint) ------------------------------------------------------
static size_t adjoint_floats(const sdeng_desc* d, int DT, size_t* o_wt, size_t* o_temb, size_t* o_trash, size_t* o_tab, size_t* o_consts,
                             size_t* o_stheta, size_t* o_target) {
  size_t o = vjp_floats(DT, d->N, o_wt, o_temb, o_trash);
  const int K = d->ref.kind == SDENG_REF_NONE ? 0 : (d->ref.kind == SDENG_REF_GAUSS_DIAG ? 1 : d->ref.k);
  *o_tab = o; o += align64(static_cast<size_t>(d->N) * K * 2 * 16 * DT);
  *o_consts = o; o += align64(static_cast<size_t>(d->N) * K * 2 + 1);
  *o_stheta = o; o += align64(static_cast<size_t>(d->N));
  *o_target = o; o += dist_floats(d->target, 16 * DT);
  return o;
}
static int check_adjoint(const sdeng_desc* d, bool ext_score) {
  if (!d) return fail(SDENG_E_INVALID, "null descriptor");
  if (d->abi_version != SDENG_ABI_VERSION) return fail(SDENG_E_INVALID, "ABI version %d, library has %d", d->abi_version, SDENG_ABI_VERSION);
  if (d->d < 1 || d->d > 128 || d->N < 1 || d->B < 1 || !d->coef) return fail(SDENG_E_INVALID, "bad sizes (1 <= d <= 128, N, B >= 1) or null coef");
  if (d->form != SDENG_FORM_LIN && d->form != SDENG_FORM_EM) return fail(SDENG_E_UNSUPPORTED, "kl_adjoint: forward forms LIN / EM (form %d)", d->form);
  const bool score_like = d->net.ctrl_kind == SDENG_CTRL_SCORE || d->net.ctrl_kind == SDENG_CTRL_LERP || d->net.ctrl_kind == SDENG_CTRL_CANCEL_DRIFT;
  if (d->net.ctrl_kind != SDENG_CTRL_CLIPPED && !(score_like && (ext_score || d->target.kind == SDENG_DIST_GMM_DIAG || d->target.kind == SDENG_DIST_PHI4)))
    return fail(SDENG_E_UNSUPPORTED, "kl_adjoint: ClippedCtrl, or Score / Lerp / CancelDrift control on a diagonal mixture / phi^4 target (ctrl_kind %d, "
                                     "target kind %d)", d->net.ctrl_kind, d->target.kind);
  if (d->net.ctrl_kind == SDENG_CTRL_LERP && d->prior.kind != SDENG_DIST_ISO_GAUSS)
    return fail(SDENG_E_UNSUPPORTED, "LerpCtrl needs an IsotropicGauss prior (kind %d given)", d->prior.kind);
  if (d->ref.kind != SDENG_REF_NONE && d->ref.kind != SDENG_REF_GAUSS_DIAG && d->ref.kind != SDENG_REF_GMM_DIAG)
    return fail(SDENG_E_UNSUPPORTED, "kl_adjoint: no reference, or a diagonal Gaussian / mixture reference (ref.kind %d)", d->ref.kind);
  if (d->ref.kind != SDENG_REF_NONE && ((d->ref.kind == SDENG_REF_GMM_DIAG && d->ref.k < 1) || !d->ref.means_init || !d->ref.vars_init))
    return fail(SDENG_E_INVALID, "reference: null means/vars or k < 1");
  if (static_cast<long long>(d->N) * d->B * d->d >= (1ll << 31)) return fail(SDENG_E_UNSUPPORTED, "N * B * d >= 2^31");
  return check_net(d->net);
}
extern "C" size_t sdeng_kl_adjoint_workspace_bytes(const sdeng_desc* d) {
  if (!d || d->d < 1 || d->d > 128 || d->N < 1) return 0;
  size_t a, b, c, e, f, g, h;
  return adjoint_floats(d, tiles_exact(d->d), &a, &b, &c, &e, &f, &g, &h) * sizeof(float);
}
extern "C" int sdeng_kl_adjoint(const sdeng_desc* d, const sdeng_adjoint* adj, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = check_adjoint(d, adj && adj->score);
  if (rc) return rc;
  if (!adj || !adj->xs || !adj->w || !adj->lam_in || !adj->a0 || !adj->a1 || !adj->a2 || !adj->d0 || !adj->d1 || !adj->d2 || !adj->dout)
    return fail(SDENG_E_INVALID, "kl_adjoint: null states / weights / lambda_N / per-row outputs");
  const bool ito = d->flags & SDENG_FLAG_ITO;
  if (ito && !adj->noise) return fail(SDENG_E_INVALID, "kl_adjoint: FLAG_ITO needs the normals of the trajectory");
  const int DT = tiles_exact(d->d), dpad = 16 * DT;
  size_t o_wt, o_temb, o_trash, o_tab, o_consts, o_stheta, o_target;
  const size_t need = adjoint_floats(d, DT, &o_wt, &o_temb, &o_trash, &o_tab, &o_consts, &o_stheta, &o_target) * sizeof(float);
  const bool score = d->net.ctrl_kind != SDENG_CTRL_CLIPPED;
  if (score && !adj->dst) return fail(SDENG_E_INVALID, "kl_adjoint: a score control needs the dst output");
  if (!d->workspace || d->workspace_bytes < need) return fail(SDENG_E_WORKSPACE, "workspace %zu bytes, need %zu", d->workspace_bytes, need);
  float* ws = static_cast<float*>(d->workspace);
  PackArgs pk;
  pk.NT = DT; pk.d = d->d;
  pk.w_in = d->net.w_in; pk.b_in = d->net.b_in; pk.w_h1 = d->net.w_h1; pk.b_h1 = d->net.b_h1;
  pk.w_h2 = d->net.w_h2; pk.b_h2 = d->net.b_h2; pk.w_out = d->net.w_out; pk.b_out = d->net.b_out;
  pk.out = ws; pk.transpose = 0; pk.scales = nullptr;
  SD_HIP(sd_launch_pack(pk, s));
  PackArgs pt = pk;
  pt.out = ws + o_wt; pt.transpose = 1; pt.scales = ws + sd_off_scales(DT);
  SD_HIP(sd_launch_pack(pt, s));
  TimeEmbedArgs te;
  te.te = d->net.t_embed; te.coef = d->coef; te.col = 0; te.t_direct = 0; te.t_value = 0.0f; te.clip = 0.0f;
  te.out = ws + o_temb;
  SD_HIP(sd_launch_time_embed(te, d->N, s));
  AdjArgs a;
  memset(&a, 0, sizeof(a));
  if (d->ref.kind != SDENG_REF_NONE) {  // the noised reference of every step: (mean, 1/var) tables + logit constants, as the step loop reads them
    const int K = d->ref.kind == SDENG_REF_GAUSS_DIAG ? 1 : d->ref.k;
    RefTabArgs r;
    r.K = K; r.d = d->d; r.dpad = dpad; r.coef = d->coef;
    r.means = d->ref.means_init; r.vars = d->ref.vars_init; r.weights = d->ref.kind == SDENG_REF_GMM_DIAG ? d->ref.weights : nullptr;
    r.tab = ws + o_tab; r.consts = ws + o_consts;
    r.same_var = ws + o_consts + static_cast<size_t>(d->N) * K * 2;
    r.centred = 0;
    SD_HIP(sd_launch_ref_tables(r, d->N, s));
    a.ref_tab = ws + o_tab; a.ref_consts = ws + o_consts; a.ref_k = K;
    a.ref_c1 = static_cast<float>(0.5 * d->d * std::log(2.0 * M_PI));
  }
  if (score) {
    a.has_score = adj->score ? 3 : (d->target.kind == SDENG_DIST_PHI4 ? 2 : 1);  // grad_kernel.hpp ADJ_EXT / ADJ_PHI4 / ADJ_GMM
    a.score_ext = adj->score;
    if (d->net.score_model.n_hidden > 0) {
      TimeEmbedArgs sm;
      sm.te = d->net.score_model; sm.coef = d->coef; sm.col = 0; sm.t_direct = 0; sm.t_value = 0.0f;
      sm.clip = d->net.clip_model;  // reparam.py:102-110 clips the score model with clip_model
      sm.out = ws + o_stheta;
      SD_HIP(sd_launch_time_embed(sm, d->N, s));
      a.stheta = sm.out;
    }
    if (!adj->score) {
      rc = build_dist(d->target, d->d, dpad, ws + o_target, a.target, s);
      if (rc) return rc;
    }
    a.scale_score = d->net.scale_score; a.clip_score = d->net.clip_score;
    a.ctrl_kind = d->net.ctrl_kind;
    a.prior_loc = d->prior.p0; a.prior_scale = d->net.ctrl_kind == SDENG_CTRL_LERP ? d->prior.p1 : 1.0f;
    a.score_detached = adj->detach_score ? 1 : 0;
    a.dst = adj->dst;
  }
  VjpArgs& v = a.v;
  v.M = d->N * d->B; v.B = d->B; v.d = d->d; v.N = d->N;
  v.x = adj->xs; v.cot = nullptr; v.wpack = ws; v.wpack_t = ws + o_wt; v.temb = ws + o_temb;
  v.clip_model = d->net.clip_model;
  v.a0 = adj->a0; v.a1 = adj->a1; v.a2 = adj->a2; v.d0 = adj->d0; v.d1 = adj->d1; v.d2 = adj->d2; v.dout = adj->dout;
  v.trash = ws + o_trash;
  a.coef = d->coef; a.noise = ito ? adj->noise : nullptr; a.w = adj->w; a.lam_in = adj->lam_in; a.lam_out = adj->lam_out;
  a.lin = d->form == SDENG_FORM_LIN ? 1 : 0;
  a.ntiles_b = (d->B + 15) / 16;
  SD_HIP(kAdjTable[dt_index(DT)](a, grid_for(a.ntiles_b), s));
  return 0;
}

extern "C" int sdeng_dist_eval(const sdeng_dist* dist, int32_t B, int32_t d, const float* x, float* logp_out, float* score_out,
                               void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!dist || !x || d < 1 || B < 0) return fail(SDENG_E_INVALID, "bad argument");
  if (d > 255) return fail(SDENG_E_UNSUPPORTED, "dist_eval: d <= 255 (rows are staged through LDS), got %d", d);
  if (B == 0) return 0;
  const int dpad = 16 * ((d + 15) / 16);
  const size_t need = dist_floats(*dist, dpad) * sizeof(float);
  if (need > 0 && (!workspace || workspace_bytes < need)) return fail(SDENG_E_WORKSPACE, "workspace %zu bytes, need %zu", workspace_bytes, need);
  DistDev dd;
  int rc = build_dist(*dist, d, dpad, static_cast<float*>(workspace), dd, s);
  if (rc) return rc;
  DistEvalArgs e;
  e.ds = dd; e.B = B; e.d = d; e.dpad = dpad; e.x = x; e.logp_out = logp_out; e.score_out = score_out;
  SD_HIP(sd_launch_dist_eval(e, s));
  return 0;
}

// ---- Langevin moves of the annealed samplers ------------------------------------------------------------------------
extern "C" size_t sdeng_langevin_moves_workspace_bytes(const sdeng_dist* prior, const sdeng_dist* target, int32_t d) {
  if (!target || d < 1) return 0;
  const int dpad = 16 * ((d + 15) / 16);
  return (dist_floats(*target, dpad) + (prior ? dist_floats(*prior, dpad) : 0) + 64) * sizeof(float);
}
extern "C" int sdeng_langevin_moves(const sdeng_dist* prior, const sdeng_dist* target, int32_t B, int32_t d, int32_t n_moves, int32_t keep_from,
                                    int32_t unadjusted, float target_acceptance, const float* t, float* x, float* lp, float* grad, float* step,
                                    const float* z, const float* u, uint64_t seed, int64_t chain0, float* samples, float* acc_sum,
                                    float* acc_last, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!target || !x || !lp || !grad || !step || B < 0 || d < 1 || n_moves < 0 || keep_from < 0) return fail(SDENG_E_INVALID, "bad argument");
  if (d > 255) return fail(SDENG_E_UNSUPPORTED, "langevin_moves: d <= 255 (chain rows live in LDS), got %d", d);
  if ((z == nullptr) != (u == nullptr) && !unadjusted) return fail(SDENG_E_INVALID, "MALA: inject both the normals and the uniforms, or neither");
  if (B == 0 || n_moves == 0) return 0;
  const int dpad = 16 * ((d + 15) / 16);
  const size_t need = sdeng_langevin_moves_workspace_bytes(prior, target, d);
  if (!workspace || workspace_bytes < need) return fail(SDENG_E_WORKSPACE, "workspace %zu bytes, need %zu", workspace_bytes, need);
  float* ws = static_cast<float*>(workspace);
  MovesArgs a;
  memset(&a, 0, sizeof(a));
  int rc = build_dist(*target, d, dpad, ws, a.target.ds, s);
  if (rc) return rc;
  a.target.d = d; a.target.dpad = dpad; a.target.B = B;
  a.prior.ds.kind = SDENG_DIST_NONE;
  if (prior && prior->kind != SDENG_DIST_NONE) {
    rc = build_dist(*prior, d, dpad, ws + dist_floats(*target, dpad), a.prior.ds, s);
    if (rc) return rc;
  }
  a.prior.d = d; a.prior.dpad = dpad; a.prior.B = B;
  a.B = B; a.d = d; a.K = n_moves; a.keep_from = keep_from; a.ula = unadjusted ? 1 : 0; a.target_acc = target_acceptance;
  a.t = t; a.x = x; a.lp = lp; a.grad = grad; a.step = step; a.z = z; a.u = u;
  a.seed_lo = static_cast<unsigned>(seed & 0xFFFFFFFFull); a.seed_hi = static_cast<unsigned>(seed >> 32); a.chain0 = chain0;
  a.samples = samples; a.acc_sum = acc_sum; a.acc_last = acc_last;
  SD_HIP(sd_launch_moves(a, s));
  return 0;
}

extern "C" size_t sdeng_dist_workspace_bytes(const sdeng_dist* dist, int32_t d) {
  if (!dist || d < 1) return 0;
  return dist_floats(*dist, 16 * ((d + 15) / 16)) * sizeof(float);
}

extern "C" size_t sdeng_logz_workspace_bytes(void) { return 5 * SD_LOGZ_MAX_BLOCKS * sizeof(float); }

extern "C" int sdeng_logz(const float* rnd, int64_t B, float* stats, float* weights_out, void* workspace, size_t workspace_bytes,
                          void* stream) {
  if (!rnd || !stats || B < 1) return fail(SDENG_E_INVALID, "bad argument");
  if (!workspace || workspace_bytes < sdeng_logz_workspace_bytes()) return fail(SDENG_E_WORKSPACE, "logz workspace too small");
  SD_HIP(sd_launch_logz(rnd, B, stats, weights_out, static_cast<float*>(workspace), static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int sdeng_philox_normal(uint64_t seed, int32_t step, int64_t particle0, int32_t B, int32_t d, uint32_t stream_id, float* out,
                                   void* stream) {
  return sdeng_philox_normal_steps(seed, step, 1, particle0, B, d, stream_id, out, stream);
}
extern "C" int sdeng_philox_normal_steps(uint64_t seed, int32_t step0, int32_t n_steps, int64_t particle0, int32_t B, int32_t d,
                                         uint32_t stream_id, float* out, void* stream) {
  if (!out || B < 0 || d < 1 || n_steps < 0) return fail(SDENG_E_INVALID, "bad argument");
  if (B == 0 || n_steps == 0) return 0;
  SD_HIP(sd_launch_philox(static_cast<unsigned>(seed & 0xFFFFFFFFull), static_cast<unsigned>(seed >> 32), step0, n_steps, particle0, B, d,
                          stream_id, out, static_cast<hipStream_t>(stream)));
  return 0;
}
