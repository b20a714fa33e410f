#include "cmcd_kernel.hpp"
// (the k_simulate_cmcd instantiations live in gen/cmcd_<tiles>.hip, one translation unit per tile count)

// one 16-bit half of a packed split-f16 A-operand image (layout of k_pack_mlp): block (to, kb), part, lane, j
//   -> (o, i) = (16 to + (lane & 15), 16 (2 kb + j/4) + 4 (lane >> 4) + j%4)
__device__ inline void block_coords(int local, int KB, int& part, int& o, int& i) {
  const int j = local & 7, lane = (local >> 3) & 63, blk = local >> 10;
  part = (local >> 9) & 1;
  const int kb = blk % KB, to = blk / KB;
  o = 16 * to + (lane & 15);
  i = 16 * (2 * kb + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
}
__device__ inline _Float16 split_part(float w, int part) {
  const _Float16 hi = static_cast<_Float16>(w);
  return part == 0 ? hi : static_cast<_Float16>((w - static_cast<float>(hi)) * 2048.0f);
}
// augmented design matrix Xa = [X | 1] (n rows, dw + 1 columns) -> logits image [row tiles][KB(NT)] and grad image
// [NT][row K-blocks] (cmcd_kernel.hpp), zero outside the data; labels padded with zeros
__global__ void k_logreg_images(const float* X, const float* y, int n, int dw, int NT, float* image, float* y_pad) {
  const int KB = sd_kb(NT), RKB = sd_lr_row_kb(n);
  const int n_logit = sd_lr_logit_floats(NT, n) * 2, n_grad = sd_lr_grad_floats(NT, n) * 2;
  _Float16* img = reinterpret_cast<_Float16*>(image);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_logit + n_grad; idx += gridDim.x * blockDim.x) {
    int part, o, i, row, col;
    if (idx < n_logit) {
      block_coords(idx, KB, part, o, i);
      row = o; col = i;
    } else {
      block_coords(idx - n_logit, RKB, part, o, i);
      row = i; col = o;
    }
    float v = 0.0f;
    if (row < n) v = (col < dw) ? X[static_cast<size_t>(row) * dw + col] : (col == dw ? 1.0f : 0.0f);
    img[idx] = split_part(v, part);
    if (idx < 32 * RKB) y_pad[idx] = (idx < n) ? y[idx] : 0.0f;
  }
}
// [d x d] matrix -> packed split-f16 A operands (same image as a dense layer with NT in / NT out tiles), + padded mean
__global__ void k_pack_square(const float* P, const float* loc, int d, int NT, float* out, float* loc_pad) {
  const int KB = sd_kb(NT);
  _Float16* img = reinterpret_cast<_Float16*>(out);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < NT * KB * 1024; idx += gridDim.x * blockDim.x) {
    int part, o, i;
    block_coords(idx, KB, part, o, i);
    img[idx] = split_part((o < d && i < d) ? P[static_cast<size_t>(o) * d + i] : 0.0f, part);
    if (idx < 16 * NT) loc_pad[idx] = idx < d ? loc[idx] : 0.0f;
  }
}
int sd_launch_logreg_images(const float* X, const float* y, int n, int dw, int NT, float* image, float* y_pad, hipStream_t s) {
  hipLaunchKernelGGL(k_logreg_images, dim3(96), dim3(256), 0, s, X, y, n, dw, NT, image, y_pad);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_pack_square(const float* P, const float* loc, int d, int NT, float* out, float* loc_pad, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_square, dim3(NT * sd_kb(NT) * 4), dim3(256), 0, s, P, loc, d, NT, out, loc_pad);
  return static_cast<int>(hipGetLastError());
}
