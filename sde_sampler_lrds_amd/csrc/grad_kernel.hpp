// Fused forward + backward of the drift net over M = N * B (time, state) rows: the batched control pass of the training direction
// (SURVEY 8f-1).  Reference: the autograd graph of FourierMLP.forward (models/mlp.py:135-143) under ClippedCtrl (models/reparam.py:33-43)
// that losses/oc.py:83-103 (generative_and_sde_ctrl) builds once per step and loss.backward() walks.  Here the states are constants (the
// trajectory came from the step-loop kernel), so all N * B evaluations are independent rows:
//
//   forward   h0 = W_in x + b_in + e_t ; a0 = gelu(h0) ; h1 = W_1 a0 + b_1 ; a1 = gelu(h1) ; h2 = W_2 a1 + b_2 ; a2 = gelu(h2) ;
//             u = W_out a2 + b_out                                                  (u only decides the clip mask)
//   backward  d_out = cot * [|u| <= clip] ; d2 = (W_out^T d_out) gelu'(h2) ; d1 = (W_2^T d2) gelu'(h1) ; d0 = (W_1^T d1) gelu'(h0)
//             [gx = W_in^T d0 : the gradient w.r.t. the state, for the KL adjoint]
//
// and the parameter gradients are six skinny GEMMs over the rows -- dW_out = d_out^T a2, dW_2 = d2^T a1, dW_1 = d1^T a0, dW_in = d0^T x,
// bias gradients = column sums, time-embedding cotangent = sum_b d0 -- which the host forms from the per-row arrays this kernel writes.
// One wave = 16 rows, same register layout as the step loop: every product is a split-f16 MFMA chain; the transposed products read
// transposed weight images (k_pack_mlp with `transpose`: W_out^T is an input-layer-shaped matrix, W_in^T an output-layer-shaped one).
// Range: a cotangent row is scaled by a per-row power of two to ~2^9 before each transposed product and scaled back after (the backward
// pass is linear in the cotangent) -- gradients of 1e-6 magnitude keep fp32's relative accuracy through the f16 split.
#pragma once
#include "sim_device.hpp"

struct VjpArgs {
  int M, B, d, N;          // rows, rows per time (row r is evaluated at time r / B), dimension, number of times
  const float* x;          // [M, d]
  const float* cot;        // [M, d] cotangent of u; nullptr: forward only (u_out)
  const float* wpack;      // forward image (+ biases + layer scales behind it)
  const float* wpack_t;    // transposed image: [W_out^T | W_2^T | W_1^T | W_in^T] in the slots of [W_in | W_1 | W_2 | W_out]
  const float* temb;       // [N][64]
  float clip_model;
  float *a0, *a1, *a2;     // [M, 64] gelu(h0), gelu(h1), gelu(h2)
  float *d0, *d1, *d2;     // [M, 64] cotangents of h0, h1, h2
  float* dout;             // [M, d]  cotangent of the net's output (after the clip mask)
  float* gx;               // [M, d]  W_in^T d0, or nullptr
  float* u_out;            // [M, d]  the control itself (after the clip), or nullptr
  float* trash;
  int ntiles;
};

// gelu'(v) = Phi(v) + v phi(v), Phi from the same polynomial as gelu_fast (2^P(|v|) = Phi(-|v|))
SD_INLINE float gelu_grad(float v) {
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
  const float pt = __builtin_amdgcn_exp2f(r);                     // Phi(-|v|)
  const float Phi = v >= 0.0f ? 1.0f - pt : pt;
  const float phi = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * (v * v));  // exp(-v^2/2) / sqrt(2 pi)
  return __builtin_fmaf(v, phi, Phi);
}

// per-row power of two that brings the row's largest |entry| to [2^8, 2^9) -- up or down; 1 for an all-zero or non-finite row
template <int T>
SD_INLINE float row_normalise(const f32x4 (&v)[T]) {
  float m = 0.0f;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, __builtin_fabsf(v[t][r]));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (!(m > 0.0f) || !(m <= 3.0e38f)) return 1.0f;
  int ex;
  frexpf(m, &ex);
  int e = 9 - ex;
  e = e > 120 ? 120 : (e < -120 ? -120 : e);
  return ldexpf(1.0f, e);
}

SD_INLINE void store_h(float* dst, float* trash, uint32_t row, bool live, int g, const f32x4 (&v)[SD_HT]) {
#pragma unroll
  for (int t = 0; t < SD_HT; ++t) *reinterpret_cast<f32x4*>(live ? dst + static_cast<size_t>(row) * SD_H + 16 * t + 4 * g : trash) = v[t];
}

template <int NT, bool GX>
__global__ void __launch_bounds__(SD_THREADS, SD_WAVES / 4) k_ctrl_vjp(const VjpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS: forward W_in, W_1, W_2, then the transposed W_out^T, W_2^T, W_1^T (same three slots of the transposed image); the forward
  // W_out (clip mask only) and W_in^T (state gradient only) are read through L2
  const int NF = sd_off_wout(NT);
  {
    const f32x4* s0 = reinterpret_cast<const f32x4*>(a.wpack);
    const f32x4* s1 = reinterpret_cast<const f32x4*>(a.wpack_t);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = tid; i < NF / 4; i += SD_THREADS) {
      dst[i] = s0[i];
      dst[NF / 4 + i] = s1[i];
    }
  }
  __syncthreads();
  const float* lds_t = lds + NF;
  const float* bias = a.wpack + sd_off_bias(NT);
  const NetScale ns = load_net_scale(bias, NT);
  const int p = lane & 15, g = lane >> 4;
  float* trash = a.trash + tid * 4;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int tile = blockIdx.x + gridDim.x * wave; tile < a.ntiles; tile += gridDim.x * SD_WAVES) {
    const uint32_t row = static_cast<uint32_t>(tile) * 16u + p;
    const bool live = row < static_cast<uint32_t>(a.M);
    const int kt = live ? static_cast<int>(row / static_cast<uint32_t>(a.B)) : 0;
    const float* te = a.temb + static_cast<size_t>(kt) * SD_H;
    f32x4 h0[SD_HT], h1[SD_HT], h2[SD_HT], act[SD_HT];
    {
      f32x4 x[NT];
      load_rows<NT>(a.x, row, a.d, live, g, x);
#pragma unroll
      for (int t = 0; t < SD_HT; ++t) h0[t] = load_tile4(bias, t, g);
      dense<NT, SD_HT>(x, h0, lds + sd_off_win(NT), lane);
    }
#pragma unroll
    for (int t = 0; t < SD_HT; ++t) {
      const f32x4 e = load_tile4(te, t, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h0[t][r] = __builtin_fmaf(h0[t][r], ns.inv_in, e[r]);
        act[t][r] = gelu_fast(h0[t][r]);
      }
      h1[t] = load_tile4(bias + 64, t, g);
    }
    if (a.a0) store_h(a.a0, trash, row, live, g, act);
    dense<SD_HT, SD_HT>(act, h1, lds + sd_off_wh1(NT), lane);
#pragma unroll
    for (int t = 0; t < SD_HT; ++t) {
      h1[t] = h1[t] * ns.inv_h1;
#pragma unroll
      for (int r = 0; r < 4; ++r) act[t][r] = gelu_fast(h1[t][r]);
      h2[t] = load_tile4(bias + 128, t, g);
    }
    if (a.a1) store_h(a.a1, trash, row, live, g, act);
    dense<SD_HT, SD_HT>(act, h2, lds + sd_off_wh2(NT), lane);
#pragma unroll
    for (int t = 0; t < SD_HT; ++t) {
      h2[t] = h2[t] * ns.inv_h2;
#pragma unroll
      for (int r = 0; r < 4; ++r) act[t][r] = gelu_fast(h2[t][r]);
    }
    if (a.a2) store_h(a.a2, trash, row, live, g, act);
    __builtin_amdgcn_sched_barrier(0);
    // ---- output layer: only the clip mask of ClippedCtrl (torch.clip passes the gradient where min <= u <= max; NaN: nowhere) ----
    f32x4 dl[NT];
    {
      const HidSplit hs = split_hidden(act);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 c = a.cot ? load_quad(a.cot, row, a.d, live, t, g) : zero;
        if (a.clip_model > 0.0f || a.u_out) {
          f32x4 u[1];
          mlp_out_tiles<NT, 1>(hs, a.wpack, bias, t, lane, u, ns.inv_out);  // A operands through L2
          if (a.clip_model > 0.0f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              c[r] = (__builtin_fabsf(u[0][r]) <= a.clip_model) ? c[r] : 0.0f;
              u[0][r] = clampf(u[0][r], a.clip_model);
            }
          }
          if (a.u_out) store_quad(a.u_out, trash, row, a.d, live, t, g, u[0]);
        }
        dl[t] = c;
        if (a.dout) store_quad(a.dout, trash, row, a.d, live, t, g, c);
        __builtin_amdgcn_sched_barrier(0);  // one output tile at a time: hoisted, the A operands of all tiles (read through L2) spill
      }
    }
    if (!a.cot) continue;  // forward only (wave-uniform)
    __builtin_amdgcn_sched_barrier(0);
    // ---- backward through the three hidden activations ----
    f32x4 dh[SD_HT], gacc[SD_HT];
    {
      const float sg = row_normalise<NT>(dl), back = ns.inv_out / sg;
#pragma unroll
      for (int t = 0; t < SD_HT; ++t) gacc[t] = zero;
      dense<NT, SD_HT, true>(dl, gacc, lds_t + sd_off_win(NT), lane, sg);  // (W_out^T 2^e) (d_out sigma)
#pragma unroll
      for (int t = 0; t < SD_HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] = (gacc[t][r] * back) * gelu_grad(h2[t][r]);
    }
    store_h(a.d2, trash, row, live, g, dh);
    {
      const float sg = row_normalise<SD_HT>(dh), back = ns.inv_h2 / sg;
#pragma unroll
      for (int t = 0; t < SD_HT; ++t) gacc[t] = zero;
      dense<SD_HT, SD_HT, true>(dh, gacc, lds_t + sd_off_wh1(NT), lane, sg);  // W_2^T
#pragma unroll
      for (int t = 0; t < SD_HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] = (gacc[t][r] * back) * gelu_grad(h1[t][r]);
    }
    store_h(a.d1, trash, row, live, g, dh);
    {
      const float sg = row_normalise<SD_HT>(dh), back = ns.inv_h1 / sg;
#pragma unroll
      for (int t = 0; t < SD_HT; ++t) gacc[t] = zero;
      dense<SD_HT, SD_HT, true>(dh, gacc, lds_t + sd_off_wh2(NT), lane, sg);  // W_1^T
#pragma unroll
      for (int t = 0; t < SD_HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] = (gacc[t][r] * back) * gelu_grad(h0[t][r]);
    }
    store_h(a.d0, trash, row, live, g, dh);
    if constexpr (GX) {  // gradient w.r.t. the state: W_in^T d0, an output-layer-shaped product (A operands through L2)
      const float sg = row_normalise<SD_HT>(dh), back = ns.inv_in / sg;
      f32x4 sc[SD_HT];
#pragma unroll
      for (int t = 0; t < SD_HT; ++t) sc[t] = dh[t] * sg;
      const HidSplit hs = split_hidden(sc);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 o[1] = {zero}, mx[1] = {zero};
        dense_pre<2, 1>(hs.h, hs.l, o, mx, reinterpret_cast<const f16x8*>(a.wpack_t + sd_off_wout(NT) + t * 2 * 512), lane);
        fold_lo<1>(o, mx);
        store_quad(a.gx, trash, row, a.d, live, t, g, o[0] * back);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

template <int NT>
static int launch_ctrl_vjp(const VjpArgs& a, int grid, hipStream_t stream) {
  const size_t lds_bytes = static_cast<size_t>(2 * sd_off_wout(NT)) * sizeof(float);
  hipError_t e;
  if (a.gx) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctrl_vjp<NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL((k_ctrl_vjp<NT, true>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctrl_vjp<NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL((k_ctrl_vjp<NT, false>), dim3(grid), dim3(SD_THREADS), lds_bytes, stream, a);
  }
  return static_cast<int>(hipGetLastError());
}
#define SD_DEFINE_VJP(NT) \
  int sd_launch_vjp_##NT(const VjpArgs& a, int grid, hipStream_t s) { return launch_ctrl_vjp<NT>(a, grid, s); }
