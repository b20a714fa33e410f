// Small per-call kernels around the persistent simulate kernel (gfx950):
//   k_pack_mlp      nn.Linear weights -> LDS image of MFMA A operands          (models/mlp.py:99-143 parameters)
//   k_time_embed    TimeEmbed.forward for all N step times at once              (models/mlp.py:85-96)
//   k_ref_tables    noised reference marginals per step                          (eq/sdes.py:208-248, 281-307)
//   k_dist_tables   static tables of a diagonal Gaussian / mixture               (distr/gauss.py:67-73, 217-221)
//   k_dist_eval     log-density / score of any supported distribution            (distr/*.py)
//   k_terminal      rnd += log p_ref(x_N) - log pi~(x_N)                         (losses/oc.py:290, 505, 973)
//   k_logz_*        elbo / logsumexp / variance / ESS / softmax weights          (losses/oc.py:150-161)
//   k_philox        the step loop's counter-based normals, standalone            (tests)
#include "sim_device.hpp"
#include "prep_kernels.hpp"

// ------------------------------------------------------------------------------------------------
// weights -> packed LDS image.  One thread per packed float.
// ------------------------------------------------------------------------------------------------
// One thread per 16-bit half of the image.  Block (to, kb), part (0 = hi, 1 = lo), lane, j:
//   w = W[16 to + (lane & 15)][16 (2 kb + j/4) + 4 (lane >> 4) + j%4];  hi = f16(w);  lo = f16((w - hi) * 2^11)
// Per-matrix power-of-two scale (sim_common.hpp SD_N_SCALES): four blocks, one matrix each.
__global__ void __launch_bounds__(256) k_weight_scales(PackArgs a) {
  __shared__ float red[256];
  float* sc = a.out + sd_off_scales(a.NT);
  {
    const int l = blockIdx.x;
    const float* W = l == 0 ? a.w_in : (l == 1 ? a.w_h1 : (l == 2 ? a.w_h2 : a.w_out));
    const int n = (l == 0 || l == 3) ? SD_H * a.d : SD_H * SD_H;
    float m = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float v = __builtin_fabsf(W[i]);
      m = (v <= 3.0e38f) ? fmaxf(m, v) : m;  // non-finite weights do not decide the scale (they poison the result either way)
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float mx = red[0];
      int e = 0;
      if (mx > 0.0f && (mx < 9.765625e-4f || mx >= 16384.0f)) {  // outside [2^-10, 2^14): normalise the largest entry to [1, 2)
        int ex;
        frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)
        e = 1 - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
      }
      sc[l] = ldexpf(1.0f, e);
      sc[4 + l] = ldexpf(1.0f, -e);
    }
    __syncthreads();
  }
}

__global__ void k_pack_mlp(PackArgs a) {
  const int NT = a.NT;
  const float* scales = a.transpose ? a.scales : a.out + sd_off_scales(NT);  // written by k_weight_scales, launched ahead of this kernel
  const int n_half = sd_lds_weight_floats(NT) * 2;
  const int n_bias = a.transpose ? 0 : 3 * 64 + 16 * NT;
  _Float16* img = reinterpret_cast<_Float16*>(a.out);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_half + n_bias; idx += gridDim.x * blockDim.x) {
    if (idx < n_half) {
      const int fl = idx >> 1;  // float-equivalent offset, to find the layer
      const float* W;
      int n_out, n_in, KB, base, layer;
      if (fl < sd_off_wh1(NT)) {
        W = a.w_in; n_out = SD_H; n_in = a.d; KB = sd_kb(NT); base = sd_off_win(NT); layer = 0;
      } else if (fl < sd_off_wh2(NT)) {
        W = a.w_h1; n_out = SD_H; n_in = SD_H; KB = 2; base = sd_off_wh1(NT); layer = 1;
      } else if (fl < sd_off_wout(NT)) {
        W = a.w_h2; n_out = SD_H; n_in = SD_H; KB = 2; base = sd_off_wh2(NT); layer = 2;
      } else {
        W = a.w_out; n_out = a.d; n_in = SD_H; KB = 2; base = sd_off_wout(NT); layer = 3;
      }
      const int local = idx - 2 * base;          // half index inside the layer
      const int j = local & 7;
      const int lane = (local >> 3) & 63;
      const int part = (local >> 9) & 1;
      const int blk = local >> 10;
      const int kb = blk % KB, to = blk / KB;
      const int o = 16 * to + (lane & 15);
      const int i = 16 * (2 * kb + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
      float w = 0.0f;
      if (o < n_out && i < n_in) {
        if (!a.transpose) {
          w = W[static_cast<size_t>(o) * n_in + i] * scales[layer];  // power of two: exact
        } else {  // slot `layer` holds the transpose of matrix 3 - layer, stored [n_in x n_out] row-major
          const float* Wt = layer == 0 ? a.w_out : (layer == 1 ? a.w_h2 : (layer == 2 ? a.w_h1 : a.w_in));
          w = Wt[static_cast<size_t>(i) * n_out + o] * scales[3 - layer];
        }
      }
      const _Float16 hi = static_cast<_Float16>(w);
      img[idx] = part == 0 ? hi : static_cast<_Float16>((w - static_cast<float>(hi)) * 2048.0f);
    } else {
      const int b = idx - n_half;
      float v;
      // biases carry their layer's scale too: they are the initial value of the layer's accumulator
      if (b < 64) v = a.b_in[b] * scales[0];
      else if (b < 128) v = a.b_h1[b - 64] * scales[1];
      else if (b < 192) v = a.b_h2[b - 128] * scales[2];
      else v = (b - 192 < a.d) ? a.b_out[b - 192] * scales[3] : 0.0f;
      a.out[sd_off_bias(NT) + b] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TimeEmbed.forward (models/mlp.py:85-96) for every step time: grid = N blocks of 64 threads.
// t comes from coef[k][col]; out[k][dim_out]; optional clip (score_model: reparam.py:102-110).
// ------------------------------------------------------------------------------------------------
__device__ inline float gelu_exact(float v) { return (v * 0.5f) * (1.0f + erff(v * 0.70710678118654752440f)); }

__global__ void __launch_bounds__(64) k_time_embed(TimeEmbedArgs a) {
  __shared__ float e[128];
  __shared__ float hbuf[2][64];
  const int k = blockIdx.x, j = threadIdx.x;
  const float t = a.t_direct ? a.t_value : a.coef[static_cast<size_t>(k) * SDENG_NCOEF + a.col];
  const float arg = (a.te.coeff[j] * t) + a.te.phase[j];
  e[j] = sinf(arg);
  e[64 + j] = cosf(arg);
  __syncthreads();
  const float* in = e;
  int n_in = 128, cur = 0;
  for (int l = 0; l < a.te.n_hidden; ++l) {
    const float* W = a.te.w[l] + static_cast<size_t>(j) * n_in;
    float acc = 0.0f;
    for (int i = 0; i < n_in; ++i) acc = __builtin_fmaf(W[i], in[i], acc);
    acc += a.te.b[l][j];
    hbuf[cur][j] = gelu_exact(acc);
    __syncthreads();
    in = hbuf[cur];
    n_in = 64;
    cur ^= 1;
  }
  if (j < a.te.dim_out) {
    const float* W = a.te.w_out + static_cast<size_t>(j) * 64;
    float acc = 0.0f;
    for (int i = 0; i < 64; ++i) acc = __builtin_fmaf(W[i], in[i], acc);
    acc += a.te.b_out[j];
    if (a.clip > 0.0f) acc = clampf(acc, a.clip);
    a.out[static_cast<size_t>(k) * a.te.dim_out + j] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// reference marginals per step (eq/sdes.py:228-229, 247): mean = s*m, var = s^2 sigma^2 + s^2 v.
// tab[k][c][0][f] = mean, tab[k][c][1][f] = 1/var (0 on pad features); consts[k][c] = (0.5*sum log var, log w_c).
// `centred` (the step loop's K = SD_KREG shared-variance path, sim_device.hpp gmm_resp_centred) and same_var[0] != 0: instead
// tab[k][c][0] = (mean_c - cen)/var, tab[k][0][1] = 1/var, tab[k][1][1] = cen = centre of the noised means, the other rows 0;
// consts[k][c][1] = log w_c - 0.5 sum (mean_c - cen)^2/var.
// grid = N*K blocks of 128 threads.  same_var[0] comes from k_same_var, launched ahead of this kernel.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) k_same_var(const float* vars, int K, int d, float* same_var) {
  // do all components share one variance vector?  (bitwise: the shortcuts it enables are exact algebra)
  int differ = 0;
  for (int i = threadIdx.x; i < K * d; i += 128) differ |= vars[i] != vars[i % d];
  differ = __syncthreads_or(differ);
  if (threadIdx.x == 0) same_var[0] = differ ? 0.0f : 1.0f;
}

__global__ void __launch_bounds__(128) k_ref_tables(RefTabArgs a) {
  __shared__ float red[128];
  const int k = blockIdx.x / a.K, c = blockIdx.x % a.K;
  const float S = a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 9];
  const float VA = a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 10];
  const float S2 = a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 11];
  const bool centred = a.centred && a.K == SD_KREG && a.same_var[0] != 0.0f;
  float* tm = a.tab + (static_cast<size_t>(k) * a.K + c) * 2 * a.dpad;
  float* tv = tm + a.dpad;
  float ls = 0.0f, quad = 0.0f;
  for (int f = threadIdx.x; f < a.dpad; f += 128) {
    float mean = 0.0f, iv = 0.0f, cen = 0.0f;
    if (f < a.d) {
      mean = S * a.means[static_cast<size_t>(c) * a.d + f];
      const float var = VA + S2 * a.vars[static_cast<size_t>(c) * a.d + f];
      iv = 1.0f / var;
      ls += logf(var);
      if (centred) {
        float sum = 0.0f;
        for (int j = 0; j < SD_KREG; ++j) sum += a.means[static_cast<size_t>(j) * a.d + f];
        cen = S * (sum * (1.0f / SD_KREG));
      }
    }
    if (centred) {
      const float dm = mean - cen;
      quad = __builtin_fmaf(dm * dm, iv, quad);
      tm[f] = dm * iv;
      tv[f] = c == 0 ? iv : (c == 1 ? cen : 0.0f);
    } else {
      tm[f] = mean;
      tv[f] = iv;
    }
  }
  red[threadIdx.x] = ls;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float ls_sum = red[0];
  __syncthreads();
  red[threadIdx.x] = quad;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float wsum = 0.0f;
    for (int i = 0; i < a.K; ++i) wsum += a.weights ? a.weights[i] : 1.0f;
    const float w = (a.weights ? a.weights[c] : 1.0f) / wsum;  // distr/gauss.py:100 (normalised in place upstream)
    float* cs = a.consts + (static_cast<size_t>(k) * a.K + c) * 2;
    cs[0] = 0.5f * ls_sum;
    cs[1] = centred ? logf(w) - 0.5f * red[0] : logf(w);
  }
}

// ------------------------------------------------------------------------------------------------
// Full-covariance mixture reference per step (eq/sdes.py:228-238 in the reference's own eigen form: covariance_c =
// U_c diag(lambda_c) U_c^T  =>  noised precision P = U diag(1/(VA + S2 lambda)) U^T, log det = sum log(VA + S2 lambda)).
// The precision is written directly as the split-f16 MFMA A-operand image the step loop multiplies with (layout of
// k_pack_mlp: half index -> (to, kb, part, lane, j), element P[16 to + lane%16][16 (2 kb + j/4) + 4 (lane/16) + j%4]).
// grid = N*K blocks of 256 threads.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ref_full_tables(RefFullArgs a) {
  __shared__ float invd[128];
  __shared__ float red[256];
  const int k = blockIdx.x / a.K, c = blockIdx.x % a.K;
  // coef == nullptr: the mixture itself at every step (a full-covariance TARGET held in the reference slot, sim_kernel.hpp SC_REFSLOT)
  const float S = a.coef ? a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 9] : 1.0f;
  const float VA = a.coef ? a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 10] : 0.0f;
  const float S2 = a.coef ? a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 11] : 1.0f;
  const int KB = sd_kb(a.NT);
  const int img_floats = a.NT * KB * 512;
  float ls = 0.0f;
  for (int j = threadIdx.x; j < 128; j += 256) {
    float iv = 0.0f;
    if (j < a.d) {
      const float var = VA + S2 * a.eigvals[static_cast<size_t>(c) * a.d + j];
      iv = 1.0f / var;
      ls += logf(var);
    }
    invd[j] = iv;
  }
  red[threadIdx.x] = ls;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const size_t blk = static_cast<size_t>(k) * a.K + c;
  if (threadIdx.x == 0) {
    float wsum = 0.0f;
    for (int i = 0; i < a.K; ++i) wsum += a.weights ? a.weights[i] : 1.0f;
    a.consts[blk * 2 + 0] = 0.5f * red[0];
    a.consts[blk * 2 + 1] = logf((a.weights ? a.weights[c] : 1.0f) / wsum);
  }
  for (int f = threadIdx.x; f < a.dpad; f += 256) a.means_out[blk * a.dpad + f] = f < a.d ? S * a.means[static_cast<size_t>(c) * a.d + f] : 0.0f;
  const float* U = a.eigvecs + static_cast<size_t>(c) * a.d * a.d;
  _Float16* img = reinterpret_cast<_Float16*>(a.images + blk * img_floats);
  for (int idx = threadIdx.x; idx < img_floats * 2; idx += 256) {
    const int j8 = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1, tb = idx >> 10;
    const int kb = tb % KB, to = tb / KB;
    const int o = 16 * to + (lane & 15);
    const int i = 16 * (2 * kb + (j8 >> 2)) + 4 * (lane >> 4) + (j8 & 3);
    float p = 0.0f;
    if (o < a.d && i < a.d) {
      const float* uo = U + static_cast<size_t>(o) * a.d;
      const float* ui = U + static_cast<size_t>(i) * a.d;
      for (int j = 0; j < a.d; ++j) p = __builtin_fmaf(uo[j] * invd[j], ui[j], p);
    }
    const _Float16 hi = static_cast<_Float16>(p);
    img[idx] = part == 0 ? hi : static_cast<_Float16>((p - static_cast<float>(hi)) * 2048.0f);
  }
}

// ------------------------------------------------------------------------------------------------
// Shared-variance mixture reference on the matrix pipe (RF_GMM_MM): noised marginal mu_k = S m_k, var = VA + S2 v (eq/sdes.py:228-229,
// 247), centre c = mean_k mu_k.  Per step: centre / 1/var vectors, logit constants, and the two split-f16 A-operand images in the
// (to, kb, part, lane, 8 halves) layout of k_pack_mlp:  logit image [kt component tiles][KB(NT)]: (mu_k - c) / var ;
// mean image [NT feature tiles][KB(kt)]: mu_k - c.   grid = N blocks of 256 threads.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ref_mm_tables(RefMMArgs a) {
  __shared__ float cbar[128], ivs[128];
  const int k = blockIdx.x;
  const float S = a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 9];
  const float VA = a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 10];
  const float S2 = a.coef[static_cast<size_t>(k) * SDENG_NCOEF + 11];
  const int KBX = sd_kb(a.NT), KBK = (a.kt + 1) / 2;
  for (int f = threadIdx.x; f < 128; f += 256) {
    float c = 0.0f, iv = 0.0f;
    if (f < a.d) {
      for (int j = 0; j < a.K; ++j) c += a.means[static_cast<size_t>(j) * a.d + f];
      c = S * (c / static_cast<float>(a.K));
      iv = 1.0f / (VA + S2 * a.vars[f]);
      // sdeng_ref.shared_var is the caller's promise; k_same_var checked it on the device (no host round trip): a false promise
      // poisons the table, so every score -- and the whole result -- is NaN instead of silently that of another mixture
      if (a.same_var[0] == 0.0f) iv = __builtin_nanf("");
    }
    cbar[f] = c;
    ivs[f] = iv;
    if (f < a.dpad) {
      a.centre[(static_cast<size_t>(k) * 2 + 0) * a.dpad + f] = c;
      a.centre[(static_cast<size_t>(k) * 2 + 1) * a.dpad + f] = iv;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // logit constants (fp64 accumulation: once per step and component)
    const int c = threadIdx.x;
    float b = -INFINITY;
    if (c < a.K) {
      float wsum = 0.0f;
      for (int i = 0; i < a.K; ++i) wsum += a.weights ? a.weights[i] : 1.0f;
      double q = 0.0;
      for (int f = 0; f < a.d; ++f) {
        const float mu = S * a.means[static_cast<size_t>(c) * a.d + f] - cbar[f];
        q += static_cast<double>(mu * mu) * ivs[f];
      }
      b = logf((a.weights ? a.weights[c] : 1.0f) / wsum) - 0.5f * static_cast<float>(q);
    }
    a.consts[static_cast<size_t>(k) * 64 + c] = b;
  }
  const int n_logit = a.kt * KBX * 1024, n_mean = a.NT * KBK * 1024;  // halves
  _Float16* img = reinterpret_cast<_Float16*>(a.images + static_cast<size_t>(k) * ((n_logit + n_mean) / 2));
  for (int idx = threadIdx.x; idx < n_logit + n_mean; idx += 256) {
    const bool logit = idx < n_logit;
    const int local = logit ? idx : idx - n_logit, KB = logit ? KBX : KBK;
    const int j8 = local & 7, lane = (local >> 3) & 63, part = (local >> 9) & 1, blk = local >> 10;
    const int kb = blk % KB, to = blk / KB;
    const int o = 16 * to + (lane & 15);
    const int i = 16 * (2 * kb + (j8 >> 2)) + 4 * (lane >> 4) + (j8 & 3);
    const int comp = logit ? o : i, f = logit ? i : o;
    float v = 0.0f;
    if (comp < a.K && f < a.d) {
      const float mu = S * a.means[static_cast<size_t>(comp) * a.d + f] - cbar[f];
      v = logit ? mu * ivs[f] : mu;
    }
    const _Float16 hi = static_cast<_Float16>(v);
    img[idx] = part == 0 ? hi : static_cast<_Float16>((v - static_cast<float>(hi)) * 2048.0f);
  }
}

// static tables of a diagonal Gaussian / mixture: tab[c][0][f] = loc, tab[c][1][f] = 1/scale^2;
// consts[c] = (0.5*sum log var, log w_norm, sum log scale + d*log sqrt(2pi), log-mixture-prob)
__global__ void __launch_bounds__(128) k_dist_tables(DistTabArgs a) {
  __shared__ float red[2][128];
  const int c = blockIdx.x;
  float* tm = a.tab + static_cast<size_t>(c) * 2 * a.dpad;
  float* tv = tm + a.dpad;
  float lv = 0.0f, lsig = 0.0f;
  for (int f = threadIdx.x; f < a.dpad; f += 128) {
    float mean = 0.0f, iv = 0.0f;
    if (f < a.d) {
      mean = a.loc[static_cast<size_t>(c) * a.d + f];
      const float sc = a.scale[static_cast<size_t>(c) * a.d + f];
      const float var = sc * sc;
      iv = 1.0f / var;
      lv += logf(var);
      lsig += logf(sc);
    }
    tm[f] = mean;
    tv[f] = iv;
  }
  red[0][threadIdx.x] = lv;
  red[1][threadIdx.x] = lsig;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* cs = a.consts + static_cast<size_t>(c) * 4;
    cs[0] = 0.5f * red[0][0];
    cs[2] = red[1][0] + static_cast<float>(a.d) * 0.91893853320467274178f;  // log sqrt(2 pi)
    if (a.weights) {
      float wsum = 0.0f;
      for (int i = 0; i < a.K; ++i) wsum += a.weights[i];
      const float pr = a.weights[c] / wsum;
      cs[1] = logf(pr);
      // Categorical(probs).logits = log(clamp(probs, eps, 1-eps)); MixtureSameFamily adds log_softmax of it
      const float eps = 1.1920928955078125e-07f;
      float lse_m = -INFINITY, lse_s = 0.0f;
      for (int i = 0; i < a.K; ++i) {
        const float li = logf(fminf(fmaxf(a.weights[i] / wsum, eps), 1.0f - eps));
        const float m = fmaxf(lse_m, li);
        lse_s = lse_s * expf(lse_m - m) + expf(li - m);
        lse_m = m;
      }
      const float lc = logf(fminf(fmaxf(pr, eps), 1.0f - eps));
      cs[3] = lc - (lse_m + logf(lse_s));
    } else {
      cs[1] = 0.0f;
      cs[3] = 0.0f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// log-density / score of a distribution, one thread per particle (runs once per trajectory or in
// unit tests; the in-loop scores live in the simulate kernels).
// ------------------------------------------------------------------------------------------------
// distr/gauss.py:217-221 -> MixtureSameFamily.log_prob: Normal.log_prob summed + log mixture, logsumexp -- over the components
// c0, c0 + cs, ...: the running (maximum, sum) pair, (-inf, 0) when there is none
__device__ void gmm_logp_partial(const DistEvalArgs& a, const float* x, int c0, int cs, float& m_run, float& l_run) {
  const DistDev& ds = a.ds;
  m_run = -INFINITY;
  l_run = 0.0f;
  for (int c = c0; c < ds.k; c += cs) {
    const float* tm = ds.tab + static_cast<size_t>(c) * 2 * a.dpad;
    const float* tv = tm + a.dpad;
    double acc = 0.0;  // once per trajectory: accumulate the 100+ term sum in fp64 (fp32 ulp of |log p| ~ 1.5e-5)
    for (int f = 0; f < a.d; ++f) {  // (fetching the wave-uniform table entries 16 at a time was measured: 2x slower, 82 registers)
      const float dl = x[f] - tm[f];
      acc += static_cast<double>(dl * dl) * tv[f];
    }
    const float part = static_cast<float>(acc);
    float lp = (-0.5f * part) - ds.consts[4 * c + 2];
    if (ds.kind == SDENG_DIST_GAUSS_DIAG) { m_run = lp; l_run = 1.0f; break; }
    lp += ds.consts[4 * c + 3];
    const float m_new = fmaxf(m_run, lp);
    l_run = l_run * expf(m_run - m_new) + expf(lp - m_new);
    m_run = m_new;
  }
}

// distr/logistic_regression.py:41-61; aux0 = X [n,d-1], aux1 = y [n]; p0 = weight_scale, p1 = intercept_mean,
// p2 = intercept_scale, p3 = threshold.  Likelihood over the data rows n0, n0 + ns, ...; the prior apart.
__device__ float logreg_loglik_partial(const DistEvalArgs& a, const float* x, int n0, int ns) {
  const DistDev& ds = a.ds;
  const int dw = a.d - 1;
  const float c = x[dw];
  const float eps = 1.1920928955078125e-07f;
  float ll = 0.0f;
  for (int n = n0; n < ds.k; n += ns) {
    const float* Xn = ds.aux0 + static_cast<size_t>(n) * dw;
    float lg = 0.0f;
    for (int f = 0; f < dw; ++f) lg = __builtin_fmaf(Xn[f], x[f], lg);
    lg += c;
    float p = 1.0f / (1.0f + expf(-lg));
    p = fminf(fmaxf(p, ds.p3), 1.0f - ds.p3);
    p = fminf(fmaxf(p, eps), 1.0f - eps);
    const float l2 = logf(p) - log1pf(-p);
    // binary_cross_entropy_with_logits = (1-y)*l + log(1+exp(-|l|)) + max(-l,0)
    const float y = ds.aux1[n];
    ll -= (1.0f - y) * l2 + (fmaxf(-l2, 0.0f) + log1pf(expf(-fabsf(l2))));
  }
  return ll;
}

__device__ float logreg_logprior(const DistEvalArgs& a, const float* x) {
  const DistDev& ds = a.ds;
  const int dw = a.d - 1;
  float prior = 0.0f;
  const float lws = logf(ds.p0), two_ws2 = 2.0f * (ds.p0 * ds.p0);
  for (int f = 0; f < dw; ++f) prior += (-(x[f] * x[f]) / two_ws2 - lws) - 0.91893853320467274178f;
  const float dc = x[dw] - ds.p1;
  prior += (-(dc * dc) / (2.0f * (ds.p2 * ds.p2)) - logf(ds.p2)) - 0.91893853320467274178f;
  return prior;
}

// |L^-1 (x - mu)|^2 over the rows i0, i0 + is, ... of L^-1 (aux0 = loc [d], aux1 = L^-1 [d,d], lower triangular)
__device__ float gauss_full_m2_partial(const DistEvalArgs& a, const float* x, int i0, int is) {
  const DistDev& ds = a.ds;
  float m2 = 0.0f;
  for (int i = i0; i < a.d; i += is) {
    float yi = 0.0f;
    const float* Li = ds.aux1 + static_cast<size_t>(i) * a.d;
    for (int j = 0; j <= i; ++j) yi = __builtin_fmaf(Li[j], x[j] - ds.aux0[j], yi);
    m2 = __builtin_fmaf(yi, yi, m2);
  }
  return m2;
}

__device__ float dist_logp_row(const DistEvalArgs& a, const float* x) {
  const int d = a.d;
  const DistDev& ds = a.ds;
  float out = 0.0f;
  if (ds.kind == SDENG_DIST_GMM_DIAG || ds.kind == SDENG_DIST_GAUSS_DIAG) {
    float m_run, l_run;
    gmm_logp_partial(a, x, 0, 1, m_run, l_run);
    out = m_run + logf(l_run);
  } else if (ds.kind == SDENG_DIST_ISO_GAUSS) {
    // distr/gauss.py:757-762; p0 = loc, p1 = scale, p2 = -0.5*d*log(2 pi var) (host), p3 = var
    double acc = 0.0;
    for (int f = 0; f < d; ++f) {
      const float dl = x[f] - ds.p0;
      acc += static_cast<double>(dl) * dl;
    }
    const float part = static_cast<float>(acc);
    out = ds.p2 - (0.5f * part) / ds.p3;
  } else if (ds.kind == SDENG_DIST_PHI4) {
    // distr/phi_four.py:44-79, 92-93
    const float coef = ds.p0 * static_cast<float>(d);
    double gsum = 0.0, vsum = 0.0;
    float prev = 0.0f;
    for (int f = 0; f < d; ++f) {
      const float xv = x[f];
      const float dl = xv - prev;
      gsum += (dl * dl) * 0.5f;
      const float om = 1.0f - xv * xv;
      vsum += (om * om) * 0.25f + ds.p1 * xv;
      prev = xv;
    }
    gsum += (prev * prev) * 0.5f;
    const float grad = static_cast<float>(gsum), V = static_cast<float>(vsum);
    out = (-ds.p2) * (grad * coef + V / coef);
  } else if (ds.kind == SDENG_DIST_GAUSS_FULL) {
    // distr/gauss.py:677-699 -> MultivariateNormal.log_prob: -0.5*(d log 2pi + |L^-1 (x-mu)|^2) - sum log diag L
    // aux0 = loc [d], aux1 = L^-1 [d,d] (lower), p0 = sum log diag L
    const float m2 = gauss_full_m2_partial(a, x, 0, 1);
    out = -0.5f * (static_cast<float>(d) * 1.8378770664093453f + m2) - ds.p0;
  } else if (ds.kind == SDENG_DIST_LOGREG) {
    out = logreg_loglik_partial(a, x, 0, 1) + logreg_logprior(a, x);
  }
  else if (ds.kind == SDENG_DIST_RINGS) {
    // distr/rings.py:93-98: log MixtureSameFamily(Categorical(w), Normal(rad, scale))(r) + log Uniform(0,2pi) - log r
    const float r = sqrtf(x[0] * x[0] + x[1] * x[1]);
    float wsum = 0.0f;
    for (int k = 0; k < ds.k; ++k) wsum += ds.aux1[k];
    const float ls = logf(ds.p0), var = ds.p0 * ds.p0;
    float m_run = -INFINITY, l_run = 0.0f;
    for (int k = 0; k < ds.k; ++k) {
      const float dl = r - ds.aux0[k];
      // Normal.log_prob: -((v - loc)^2)/(2 var) - log(scale) - log(sqrt(2 pi))
      const float lp = ((-(dl * dl) / (2.0f * var)) - ls) - 0.91893853320467274178f + logf(ds.aux1[k] / wsum);
      const float m_new = fmaxf(m_run, lp);
      l_run = l_run * expf(m_run - m_new) + expf(lp - m_new);
      m_run = m_new;
    }
    out = ((m_run + logf(l_run)) - 1.8378770664093453f) - logf(r);
  }
  if (ds.clip > 0.0f) out = clampf(out, ds.clip);
  return out;
}

__device__ void dist_score_row(const DistEvalArgs& a, const float* x, float* sc) {
  const int d = a.d;
  const DistDev& ds = a.ds;
  if (ds.kind == SDENG_DIST_GMM_DIAG) {  // distr/gauss.py:97-107
    float m_run = -INFINITY, l_run = 0.0f;
    for (int f = 0; f < d; ++f) sc[f] = 0.0f;
    for (int c = 0; c < ds.k; ++c) {
      const float* tm = ds.tab + static_cast<size_t>(c) * 2 * a.dpad;
      const float* tv = tm + a.dpad;
      float part = 0.0f;
      for (int f = 0; f < d; ++f) {
        const float dl = x[f] - tm[f];
        part = __builtin_fmaf(dl * dl, tv[f], part);
      }
      float lp = ((-0.5f * part) - ds.p0) - ds.consts[4 * c + 0];
      lp = ds.consts[4 * c + 1] + lp;
      const float m_new = fmaxf(m_run, lp);
      const float so = expf(m_run - m_new), pk = expf(lp - m_new);
      l_run = l_run * so + pk;
      m_run = m_new;
      for (int f = 0; f < d; ++f) sc[f] = __builtin_fmaf(pk, (tm[f] - x[f]) * tv[f], sc[f] * so);
    }
    const float inv = 1.0f / l_run;
    for (int f = 0; f < d; ++f) sc[f] *= inv;
  } else if (ds.kind == SDENG_DIST_GAUSS_DIAG) {  // distr/gauss.py:124-126
    for (int f = 0; f < d; ++f) sc[f] = -((x[f] - ds.tab[f]) * ds.tab[a.dpad + f]);
  } else if (ds.kind == SDENG_DIST_ISO_GAUSS) {  // distr/gauss.py:764-766
    for (int f = 0; f < d; ++f) sc[f] = (ds.p0 - x[f]) / (ds.p1 * ds.p1);
  } else if (ds.kind == SDENG_DIST_PHI4) {  // distr/phi_four.py:81-96
    const float coef = ds.p0 * static_cast<float>(d);
    for (int f = 0; f < d; ++f) {
      const float xv = x[f];
      const float xl = f > 0 ? x[f - 1] : 0.0f, xr = f < d - 1 ? x[f + 1] : 0.0f;
      float g = (ds.p1 - xv * (1.0f - xv * xv)) / coef;
      g = g + coef * ((2.0f * xv - xr) - xl);
      sc[f] = (-ds.p2) * g;
    }
  } else if (ds.kind == SDENG_DIST_GAUSS_FULL) {  // distr/gauss.py:129-135; scale ptr in tab = precision [d,d]
    for (int i = 0; i < d; ++i) {
      float acc = 0.0f;
      const float* Pi = ds.tab + static_cast<size_t>(i) * d;
      for (int j = 0; j < d; ++j) acc = __builtin_fmaf(Pi[j], x[j] - ds.aux0[j], acc);
      sc[i] = -acc;
    }
  } else if (ds.kind == SDENG_DIST_LOGREG) {
    // the reference differentiates posterior_log_prob by autograd (distr/base.py:146-154); closed form of that
    // gradient through sigmoid -> clip(thr) -> clamp(eps) -> logit -> BCE-with-logits
    const int dw = d - 1;
    const float c = x[dw];
    const float eps = 1.1920928955078125e-07f;
    for (int f = 0; f < dw; ++f) sc[f] = -x[f] / (ds.p0 * ds.p0);
    float gc = -(c - ds.p1) / (ds.p2 * ds.p2);
    for (int n = 0; n < ds.k; ++n) {
      const float* Xn = ds.aux0 + static_cast<size_t>(n) * dw;
      float lg = 0.0f;
      for (int f = 0; f < dw; ++f) lg = __builtin_fmaf(Xn[f], x[f], lg);
      lg += c;
      const float p = 1.0f / (1.0f + expf(-lg));
      const float pc = fminf(fmaxf(p, ds.p3), 1.0f - ds.p3);
      const float pcc = fminf(fmaxf(pc, eps), 1.0f - eps);
      const bool pass = (p >= ds.p3) && (p <= 1.0f - ds.p3) && (pc >= eps) && (pc <= 1.0f - eps);
      const float l2 = logf(pcc) - log1pf(-pcc);
      const float sg = 1.0f / (1.0f + expf(-l2));
      float r = (ds.aux1[n] - sg) * (1.0f / pcc + 1.0f / (1.0f - pcc)) * (p * (1.0f - p));
      r = pass ? r : 0.0f;
      for (int f = 0; f < dw; ++f) sc[f] = __builtin_fmaf(r, Xn[f], sc[f]);
      gc += r;
    }
    sc[dw] = gc;
  } else if (ds.kind == SDENG_DIST_RINGS) {
    const f32x4 r = rings_score(f32x4{x[0], x[1], 0.0f, 0.0f}, ds, 0);
    sc[0] = r[0];
    sc[1] = r[1];
  }
}

// The block's 64 rows are brought in by all four waves with coalesced loads -- several per thread in flight, 16 bytes wide when the rows
// are -- and parked in LDS with an odd row stride: thread (wave w, lane l) then walks row l conflict-free.  (A thread reading its own row
// straight from HBM touches 64 different cache lines per wave-instruction; a one-wave block staging row after row had two loads in flight
// per CU-slot and ran at 190 GB/s: 340 us for the 64 MB of cfg 2's x_N.)
#define SD_EVAL_ROWS 64
#define SD_EVAL_THREADS 256
#define SD_EVAL_RED_FLOATS (SD_EVAL_THREADS * 2)

__device__ inline const float* stage_rows(const float* x, int B, int d, float* sh) {
  const int row0 = blockIdx.x * SD_EVAL_ROWS, stride = d | 1, tid = threadIdx.x;
  const int rows = B - row0 < SD_EVAL_ROWS ? B - row0 : SD_EVAL_ROWS;
  const float* src = x + static_cast<size_t>(row0) * d;  // rows * d contiguous floats; row0 * d * 4 bytes is a multiple of 256
  if ((d & 3) == 0) {
    const int d4 = d >> 2, n4 = rows * d4;
    const float inv = 1.0f / static_cast<float>(d4);
    const f32x4* src4 = reinterpret_cast<const f32x4*>(src);
    for (int base = tid; base < n4; base += 4 * SD_EVAL_THREADS) {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i4 = base + j * SD_EVAL_THREADS;
        v[j] = i4 < n4 ? src4[i4] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i4 = base + j * SD_EVAL_THREADS;
        if (i4 < n4) {
          const int r = static_cast<int>((static_cast<float>(i4) + 0.5f) * inv);  // i4 / d4: exact, the fraction is >= 0.5 / d4 away from an integer
          float* dst = sh + r * stride + 4 * (i4 - r * d4);
          dst[0] = v[j][0]; dst[1] = v[j][1]; dst[2] = v[j][2]; dst[3] = v[j][3];
        }
      }
    }
  } else {
    const int n = rows * d;
    const float inv = 1.0f / static_cast<float>(d);
    for (int base = tid; base < n; base += 8 * SD_EVAL_THREADS) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = base + j * SD_EVAL_THREADS;
        v[j] = i < n ? src[i] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = base + j * SD_EVAL_THREADS;
        if (i < n) {
          const int r = static_cast<int>((static_cast<float>(i) + 0.5f) * inv);
          sh[r * stride + (i - r * d)] = v[j];
        }
      }
    }
  }
  __syncthreads();
  return sh + (tid & 63) * stride;
}

// log-density of the block's staged rows: thread (wave w, lane l) works on row l.  The components of a mixture and the data rows of the
// logistic-regression likelihood are dealt over the four waves and merged through LDS; every other kind is one pass over the features by
// wave 0.  The value is returned in wave 0.  Every thread of the block calls it (barriers inside).
__device__ float dist_logp_block(const DistEvalArgs& a, const float* x, float* red) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = SD_EVAL_THREADS / 64;
  const DistDev& ds = a.ds;
  if (ds.kind == SDENG_DIST_GMM_DIAG && ds.k > 1) {
    float m_run, l_run;
    gmm_logp_partial(a, x, w, NW, m_run, l_run);
    red[2 * threadIdx.x] = m_run;
    red[2 * threadIdx.x + 1] = l_run;
    __syncthreads();
    float out = 0.0f;
    if (w == 0) {  // wave 0 holds component 0: its running maximum is finite
      for (int j = 1; j < NW; ++j) {
        const float mj = red[2 * (64 * j + lane)], lj = red[2 * (64 * j + lane) + 1];
        const float m_new = fmaxf(m_run, mj);
        l_run = l_run * expf(m_run - m_new) + lj * expf(mj - m_new);
        m_run = m_new;
      }
      out = m_run + logf(l_run);
      if (ds.clip > 0.0f) out = clampf(out, ds.clip);
    }
    __syncthreads();
    return out;
  }
  if (ds.kind == SDENG_DIST_LOGREG) {
    const float ll = logreg_loglik_partial(a, x, w, NW);
    red[threadIdx.x] = ll;
    __syncthreads();
    float out = 0.0f;
    if (w == 0) {
      float sum = ll;
      for (int j = 1; j < NW; ++j) sum += red[64 * j + lane];
      out = sum + logreg_logprior(a, x);
      if (ds.clip > 0.0f) out = clampf(out, ds.clip);
    }
    __syncthreads();
    return out;
  }
  if (ds.kind == SDENG_DIST_GAUSS_FULL) {
    const float m2 = gauss_full_m2_partial(a, x, w, NW);
    red[threadIdx.x] = m2;
    __syncthreads();
    float out = 0.0f;
    if (w == 0) {
      float sum = m2;
      for (int j = 1; j < NW; ++j) sum += red[64 * j + lane];
      out = -0.5f * (static_cast<float>(a.d) * 1.8378770664093453f + sum) - ds.p0;
      if (ds.clip > 0.0f) out = clampf(out, ds.clip);
    }
    __syncthreads();
    return out;
  }
  return w == 0 ? dist_logp_row(a, x) : 0.0f;
}

__global__ void __launch_bounds__(SD_EVAL_THREADS) k_dist_eval(DistEvalArgs a) {
  extern __shared__ float sh[];
  float* red = sh + SD_EVAL_ROWS * (a.d | 1);
  const float* x = stage_rows(a.x, a.B, a.d, sh);
  const int row = blockIdx.x * SD_EVAL_ROWS + (threadIdx.x & 63);
  const bool writer = threadIdx.x < 64 && row < a.B;
  if (a.logp_out) {
    const float lp = dist_logp_block(a, x, red);
    if (writer) a.logp_out[row] = lp;
  }
  if (a.score_out && writer) dist_score_row(a, x, a.score_out + static_cast<size_t>(row) * a.d);
}

// ------------------------------------------------------------------------------------------------
// Langevin moves of the annealed samplers: one thread per chain, the chain's state / gradient / proposal rows in LDS, all K
// moves of a level in ONE launch -- the reference (and the host composition this replaces) spends ~30 small launches per move.
// ------------------------------------------------------------------------------------------------
#define SD_MOVE_ROWS 5  // x, grad, proposal, gradient at the proposal, scratch score row
__device__ inline void tempered_eval(const MovesArgs& a, float w, const float* y, float* g_out, float* tmp, float& lp_out) {
  const bool both = a.prior.ds.kind != SDENG_DIST_NONE;
  const float lp1 = dist_logp_row(a.target, y);
  dist_score_row(a.target, y, g_out);
  if (!both) {
    lp_out = lp1;
    return;
  }
  const float lp0 = dist_logp_row(a.prior, y);
  dist_score_row(a.prior, y, tmp);
  lp_out = (1.0f - w) * lp0 + w * lp1;   // ebm_mle.py's tempered density (hip_tempered_log_prob_and_grads)
  for (int f = 0; f < a.d; ++f) g_out[f] = (1.0f - w) * tmp[f] + w * g_out[f];
}

__global__ void __launch_bounds__(64) k_langevin_moves(MovesArgs a, int chains_per_block) {
  extern __shared__ float sh[];
  const int tid = threadIdx.x, d = a.d, stride = d | 1;
  const int chain = blockIdx.x * chains_per_block + tid;
  if (tid >= chains_per_block || chain >= a.B) return;  // (no barriers below: a chain is one thread)
  float* xr = sh + (tid * SD_MOVE_ROWS + 0) * stride;
  float* gr = xr + stride;
  float* pr = gr + stride;
  float* gp = pr + stride;
  float* tmp = gp + stride;
  for (int f = 0; f < d; ++f) {
    xr[f] = a.x[static_cast<size_t>(chain) * d + f];
    gr[f] = a.grad[static_cast<size_t>(chain) * d + f];
  }
  float lp = a.lp[chain], step = a.step[chain], acc = 0.0f, last = 1.0f;
  const float w = a.t ? a.t[chain] : 1.0f;
  const uint32_t cidx = static_cast<uint32_t>(a.chain0 + chain);
  const float log_target = a.target_acc > 0.0f ? logf(a.target_acc) : 0.0f;
  for (int m = 0; m < a.K; ++m) {
    const float sd = sqrtf(2.0f * step);
    // proposal: sample_multivariate_normal_diag(mean = y + h grad, variance = 2 h)  (mcmc.py:19-21, :97-99)
    for (int f0 = 0; f0 < d; f0 += 4) {
      f32x4 z;
      if (a.z) {
        for (int r = 0; r < 4; ++r) z[r] = f0 + r < d ? a.z[(static_cast<size_t>(m) * a.B + chain) * d + f0 + r] : 0.0f;
      } else {
        z = philox_normal4(cidx, static_cast<uint32_t>(m), static_cast<uint32_t>(f0 >> 2), 2u, a.seed_lo, a.seed_hi);
      }
      for (int r = 0; r < 4 && f0 + r < d; ++r) pr[f0 + r] = sd * z[r] + (xr[f0 + r] + step * gr[f0 + r]);
    }
    float lp_p;
    tempered_eval(a, w, pr, gp, tmp, lp_p);
    bool accept = true;
    float log_acc = 0.0f;
    if (!a.ula) {
      // log q(prop | y) and log q(y | prop), unnormalised: -0.5 sum (samples - mean)^2 / variance  (mcmc.py:24-31)
      float qf = 0.0f, qb = 0.0f;
      for (int f = 0; f < d; ++f) {
        const float df = pr[f] - (xr[f] + step * gr[f]);
        const float db = xr[f] - (pr[f] + step * gp[f]);
        qf += df * df;
        qb += db * db;
      }
      const float var = 2.0f * step;
      const float joint_prop = lp_p - (-0.5f * qf) / var, joint_orig = lp - (-0.5f * qb) / var;
      log_acc = joint_prop - joint_orig;
      float uu;
      if (a.u) {
        uu = a.u[static_cast<size_t>(m) * a.B + chain];
      } else {
        uint32_t r4[4];
        philox4x32_10(cidx, 0u, static_cast<uint32_t>(m), 3u, a.seed_lo, a.seed_hi, r4);
        uu = u01(r4[0]);
      }
      accept = logf(uu) < log_acc;
    }
    if (accept) {
      for (int f = 0; f < d; ++f) {
        xr[f] = pr[f];
        gr[f] = gp[f];
      }
      lp = lp_p;
    }
    if (!a.ula && a.target_acc > 0.0f) {  // heuristics_step_size (mcmc.py:55-74), per chain
      if (log_acc - log_target > 0.04879016416943205f) step = step * 1.01f;        // log1p(0.05)
      if (log_target - log_acc > 0.05129329438755058f) step = step / 1.01f;        // -log1p(-0.05)
    }
    if (!a.ula) last = expf(fminf(0.0f, log_acc));
    if (m >= a.keep_from) {
      if (!a.ula) acc += last;
      if (a.samples) {
        float* dst = a.samples + (static_cast<size_t>(m - a.keep_from) * a.B + chain) * d;
        for (int f = 0; f < d; ++f) dst[f] = xr[f];
      }
    }
  }
  for (int f = 0; f < d; ++f) {
    a.x[static_cast<size_t>(chain) * d + f] = xr[f];
    a.grad[static_cast<size_t>(chain) * d + f] = gr[f];
  }
  a.lp[chain] = lp;
  a.step[chain] = step;
  if (a.acc_sum) a.acc_sum[chain] = acc;
  if (a.acc_last) a.acc_last[chain] = last;
}

// terminal cost, in place on rnd (losses/oc.py:290: rnd += ref_logp(x) - target_logp(x); :973: rnd -= target_logp(x))
__global__ void __launch_bounds__(SD_EVAL_THREADS) k_terminal(TerminalArgs a) {
  extern __shared__ float sh[];
  float* red = sh + SD_EVAL_ROWS * (a.d | 1);
  const float* x = stage_rows(a.x, a.B, a.d, sh);
  const int row = blockIdx.x * SD_EVAL_ROWS + (threadIdx.x & 63);
  float term = 0.0f;
  if (a.use_ref) {
    DistEvalArgs e;
    e.ds = a.ref; e.B = a.B; e.d = a.d; e.dpad = a.dpad;
    term = dist_logp_block(e, x, red);
  }
  if (a.use_target) {
    DistEvalArgs e;
    e.ds = a.target; e.B = a.B; e.d = a.d; e.dpad = a.dpad;
    term = term - dist_logp_block(e, x, red);
  }
  if (threadIdx.x < 64 && row < a.B) a.rnd[row] += term;
}

// ------------------------------------------------------------------------------------------------
// estimators (losses/oc.py:150-161; eval/metrics.py:135-140).  Two passes: block partials, then combine.
// ------------------------------------------------------------------------------------------------
__device__ inline float block_reduce(float v, float* sh, bool is_max) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, t) : v + t;
  }
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  __syncthreads();
  return r;
}

// partials[b] = (max(-rnd), sum(-rnd), sum rnd^2 shifted, count) per block; pass 2 combines
__global__ void __launch_bounds__(256) k_logz_partial(const float* rnd, long long B, float* partials) {
  __shared__ float sh[4];
  float mx = -INFINITY, s1 = 0.0f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < B; i += gridDim.x * 256ll) {
    const float v = -rnd[i];
    mx = fmaxf(mx, v);
    s1 += v;
  }
  mx = block_reduce(mx, sh, true);
  s1 = block_reduce(s1, sh, false);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x + 0] = mx;
    partials[2 * blockIdx.x + 1] = s1;
  }
}

__global__ void __launch_bounds__(256) k_logz_partial2(const float* rnd, long long B, const float* partials, int nb,
                                                      float* partials2) {
  __shared__ float sh[4];
  float mx = -INFINITY, s1 = 0.0f;
  for (int i = 0; i < nb; ++i) {
    mx = fmaxf(mx, partials[2 * i]);
    s1 += partials[2 * i + 1];
  }
  const float mean = s1 / static_cast<float>(B);  // mean of -rnd
  float se = 0.0f, se2 = 0.0f, sv = 0.0f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < B; i += gridDim.x * 256ll) {
    const float v = -rnd[i];
    const float ex = expf(v - mx);
    se += ex;
    se2 = __builtin_fmaf(ex, ex, se2);
    const float dv = v - mean;
    sv = __builtin_fmaf(dv, dv, sv);
  }
  se = block_reduce(se, sh, false);
  se2 = block_reduce(se2, sh, false);
  sv = block_reduce(sv, sh, false);
  if (threadIdx.x == 0) {
    partials2[3 * blockIdx.x + 0] = se;
    partials2[3 * blockIdx.x + 1] = se2;
    partials2[3 * blockIdx.x + 2] = sv;
  }
}

__global__ void k_logz_final(const float* partials, const float* partials2, int nb, long long B, float* stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mx = -INFINITY, s1 = 0.0f, se = 0.0f, se2 = 0.0f, sv = 0.0f;
  for (int i = 0; i < nb; ++i) {
    mx = fmaxf(mx, partials[2 * i]);
    s1 += partials[2 * i + 1];
    se += partials2[3 * i];
    se2 += partials2[3 * i + 1];
    sv += partials2[3 * i + 2];
  }
  const float fB = static_cast<float>(B);
  stats[0] = s1 / fB;
  stats[1] = (mx + logf(se)) - logf(fB);
  stats[2] = B > 1 ? sv / (fB - 1.0f) : 0.0f;
  stats[3] = (se * se) / se2 / fB;
  stats[4] = mx;
  stats[5] = se;
  stats[6] = se2;
  stats[7] = s1;
}

__global__ void k_softmax_weights(const float* rnd, long long B, const float* stats, float* w) {
  const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (i >= B) return;
  w[i] = expf((-rnd[i]) - stats[4]) / stats[5];
}

// grid.y = number of consecutive steps: block row y writes out[y][B][d] for step `step + y`
__global__ void k_philox(unsigned seed_lo, unsigned seed_hi, int step, long long particle0, int B, int d, unsigned stream_id,
                         float* out) {
  const int nj = (d + 3) / 4;
  const long long idx = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (idx >= static_cast<long long>(B) * nj) return;
  const int row = static_cast<int>(idx / nj), jb = static_cast<int>(idx % nj);
  const f32x4 z = philox_normal4(static_cast<uint32_t>(particle0 + row), static_cast<uint32_t>(step) + blockIdx.y, static_cast<uint32_t>(jb),
                                 stream_id, seed_lo, seed_hi);
  float* o = out + static_cast<size_t>(blockIdx.y) * B * d;
  for (int e = 0; e < 4; ++e)
    if (4 * jb + e < d) o[static_cast<size_t>(row) * d + 4 * jb + e] = z[e];
}

// ------------------------------------------------------------------------------------------------
// prior.sample((B,)) on the device: the x0 the step-loop kernels draw in registers, materialised.
//   ISO_GAUSS / GAUSS_DIAG: one thread per (particle, quad of features), x0_quad() of sim_device.hpp;
//   GAUSS_FULL (GaussFull.sample, distr/gauss.py:709-713: MultivariateNormal = loc + L z): 16 particles per block, their
//   normals staged in LDS, then x0_i = loc_i + sum_{j <= i} L[i][j] z_j.
// ------------------------------------------------------------------------------------------------
__global__ void k_sample_x0(X0Dev q, unsigned seed_lo, unsigned seed_hi, long long particle0, int B, int d, float* out) {
  const int nj = (d + 3) / 4;
  const long long idx = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (idx >= static_cast<long long>(B) * nj) return;
  const int row = static_cast<int>(idx / nj), jb = static_cast<int>(idx % nj);
  const f32x4 x = x0_quad(q, static_cast<uint32_t>(particle0 + row), jb, d, seed_lo, seed_hi);
  for (int e = 0; e < 4; ++e)
    if (4 * jb + e < d) out[static_cast<size_t>(row) * d + 4 * jb + e] = x[e];
}
#define SD_X0_FULL_ROWS 64
__global__ void __launch_bounds__(256) k_sample_x0_full(const float* loc, const float* L, unsigned seed_lo, unsigned seed_hi, long long particle0,
                                                       int B, int d, float* out) {
  // 64 particles per block: their normals z[r][j] and the transposed factor Lt[j][f] (odd stride) in LDS, so that the lanes of a wave
  // -- consecutive features f of one or two particles -- read consecutive words of Lt and broadcast z; L itself is read once per
  // block, coalesced (a lane walking its own row of L in global memory touched 64 cache lines per load).
  extern __shared__ float sh[];
  const int nj = (d + 3) / 4, zs = 4 * nj + 1, ls = d | 1;
  float* z = sh;                             // [64][zs]
  float* Lt = sh + SD_X0_FULL_ROWS * zs;     // [d][ls]
  const int row0 = blockIdx.x * SD_X0_FULL_ROWS;
  for (int i = threadIdx.x; i < d * d; i += 256) {
    const int f = i / d, j = i - f * d;
    Lt[j * ls + f] = L[i];
  }
  for (int i = threadIdx.x; i < SD_X0_FULL_ROWS * nj; i += 256) {
    const int r = i / nj, jb = i - r * nj;
    const f32x4 v = philox_normal4(static_cast<uint32_t>(particle0 + row0 + r), 0u, static_cast<uint32_t>(jb), SD_X0_STREAM, seed_lo, seed_hi);
    for (int e = 0; e < 4; ++e) z[r * zs + 4 * jb + e] = v[e];
  }
  __syncthreads();
  const int rows = B - row0 < SD_X0_FULL_ROWS ? B - row0 : SD_X0_FULL_ROWS;
  for (int i = threadIdx.x; i < rows * d; i += 256) {
    const int r = i / d, f = i - r * d;
    float acc = 0.0f;
    for (int j = 0; j <= f; ++j) acc = __builtin_fmaf(Lt[j * ls + f], z[r * zs + j], acc);
    out[static_cast<size_t>(row0) * d + i] = loc[f] + acc;
  }
}
int sd_launch_sample_x0(const sdeng_dist& ds, unsigned lo, unsigned hi, long long p0, int B, int d, float* out, hipStream_t s) {
  if (ds.kind == SDENG_DIST_GAUSS_FULL) {
    const size_t lds = (static_cast<size_t>(SD_X0_FULL_ROWS) * (4 * ((d + 3) / 4) + 1) + static_cast<size_t>(d) * (d | 1)) * sizeof(float);
    if (lds > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sample_x0_full), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(k_sample_x0_full, dim3((B + SD_X0_FULL_ROWS - 1) / SD_X0_FULL_ROWS), dim3(256), lds, s, ds.loc, ds.aux, lo, hi, p0, B, d, out);
  } else {
    X0Dev q;
    q.kind = ds.kind; q.loc = ds.loc; q.scale = ds.scale; q.p0 = ds.p0; q.p1 = ds.p1;
    const long long n = static_cast<long long>(B) * ((d + 3) / 4);
    hipLaunchKernelGGL(k_sample_x0, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, q, lo, hi, p0, B, d, out);
  }
  return static_cast<int>(hipGetLastError());
}

// ---- host-side launch wrappers -------------------------------------------------------------------
int sd_launch_pack(const PackArgs& a, hipStream_t s) {
  const int total = sd_lds_weight_floats(a.NT) * 2 + (a.transpose ? 0 : 3 * 64 + 16 * a.NT);
  if (!a.transpose) hipLaunchKernelGGL(k_weight_scales, dim3(4), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_pack_mlp, dim3((total + 255) / 256), dim3(256), 0, s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_time_embed(const TimeEmbedArgs& a, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_time_embed, dim3(N), dim3(64), 0, s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_ref_tables(const RefTabArgs& a, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_same_var, dim3(1), dim3(128), 0, s, a.vars, a.K, a.d, a.same_var);
  hipLaunchKernelGGL(k_ref_tables, dim3(N * a.K), dim3(128), 0, s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_ref_full_tables(const RefFullArgs& a, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_ref_full_tables, dim3(N * a.K), dim3(256), 0, s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_ref_mm_tables(const RefMMArgs& a, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_same_var, dim3(1), dim3(128), 0, s, a.vars, a.K, a.d, a.same_var);
  hipLaunchKernelGGL(k_ref_mm_tables, dim3(N), dim3(256), 0, s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_dist_tables(const DistTabArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_dist_tables, dim3(a.K), dim3(128), 0, s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_dist_eval(const DistEvalArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_dist_eval, dim3((a.B + SD_EVAL_ROWS - 1) / SD_EVAL_ROWS), dim3(SD_EVAL_THREADS), (SD_EVAL_ROWS * (a.d | 1) + SD_EVAL_RED_FLOATS) * sizeof(float), s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_moves(const MovesArgs& a, hipStream_t s) {
  const int stride = a.d | 1;
  int cpb = 64;
  while (cpb > 8 && static_cast<size_t>(cpb) * SD_MOVE_ROWS * stride * sizeof(float) > 150 * 1024) cpb /= 2;  // d > ~115: 32 chains per block
  const size_t lds = static_cast<size_t>(cpb) * SD_MOVE_ROWS * stride * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_langevin_moves), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(k_langevin_moves, dim3((a.B + cpb - 1) / cpb), dim3(64), lds, s, a, cpb);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_terminal(const TerminalArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_terminal, dim3((a.B + SD_EVAL_ROWS - 1) / SD_EVAL_ROWS), dim3(SD_EVAL_THREADS), (SD_EVAL_ROWS * (a.d | 1) + SD_EVAL_RED_FLOATS) * sizeof(float), s, a);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_logz(const float* rnd, long long B, float* stats, float* weights, float* scratch, hipStream_t s) {
  int nb = static_cast<int>((B + 256 * 8 - 1) / (256 * 8));
  if (nb > SD_LOGZ_MAX_BLOCKS) nb = SD_LOGZ_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  float* p1 = scratch;
  float* p2 = scratch + 2 * SD_LOGZ_MAX_BLOCKS;
  hipLaunchKernelGGL(k_logz_partial, dim3(nb), dim3(256), 0, s, rnd, B, p1);
  hipLaunchKernelGGL(k_logz_partial2, dim3(nb), dim3(256), 0, s, rnd, B, p1, nb, p2);
  hipLaunchKernelGGL(k_logz_final, dim3(1), dim3(64), 0, s, p1, p2, nb, B, stats);
  if (weights) hipLaunchKernelGGL(k_softmax_weights, dim3(static_cast<unsigned>((B + 255) / 256)), dim3(256), 0, s, rnd, B, stats, weights);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_philox(unsigned lo, unsigned hi, int step, int n_steps, long long p0, int B, int d, unsigned stream_id, float* out, hipStream_t s) {
  const long long n = static_cast<long long>(B) * ((d + 3) / 4);
  for (int done = 0; done < n_steps; done += 65535) {  // grid.y limit
    const int chunk = n_steps - done < 65535 ? n_steps - done : 65535;
    hipLaunchKernelGGL(k_philox, dim3(static_cast<unsigned>((n + 255) / 256), chunk), dim3(256), 0, s, lo, hi, step + done, p0, B, d, stream_id,
                       out + static_cast<size_t>(done) * B * d);
  }
  return static_cast<int>(hipGetLastError());
}
