#include "cmcd_kernel.hpp"
int sd_launch_cmcd_1(const CmcdArgs& a, int grid, hipStream_t s) { return launch_cmcd<1>(a, grid, s); }
int sd_launch_cmcd_2(const CmcdArgs& a, int grid, hipStream_t s) { return launch_cmcd<2>(a, grid, s); }
int sd_launch_cmcd_4(const CmcdArgs& a, int grid, hipStream_t s) { return launch_cmcd<4>(a, grid, s); }

// design matrix -> LDS image [SD_LR_ROWS][SD_LR_STRIDE]: X in columns 0..d-2, a column of ones at d-1 (intercept), zeros elsewhere
__global__ void k_logreg_image(const float* X, const float* y, int n, int dw, float* image, float* y_pad) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < SD_LR_ROWS * SD_LR_STRIDE; idx += gridDim.x * blockDim.x) {
    const int r = idx / SD_LR_STRIDE, c = idx % SD_LR_STRIDE;
    float v = 0.0f;
    if (r < n) v = (c < dw) ? X[static_cast<size_t>(r) * dw + c] : (c == dw ? 1.0f : 0.0f);
    image[idx] = v;
    if (c == 0) y_pad[r] = (r < n) ? y[r] : 0.0f;
  }
}
// [d x d] matrix -> packed MFMA A operands (same image as a dense layer with DT in / DT out tiles), + padded mean
__global__ void k_pack_square(const float* P, const float* loc, int d, int NT, float* out, float* loc_pad) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < NT * NT * 256; idx += gridDim.x * blockDim.x) {
    const int r = idx & 3, lane = (idx >> 2) & 63, pair = idx >> 8;
    const int ti = pair % NT, to = pair / NT;
    const int o = 16 * to + (lane & 15), i = feat(ti, r, lane >> 4);
    out[idx] = (o < d && i < d) ? P[static_cast<size_t>(o) * d + i] : 0.0f;
    if (idx < 16 * NT) loc_pad[idx] = idx < d ? loc[idx] : 0.0f;
  }
}
int sd_launch_logreg_image(const float* X, const float* y, int n, int dw, float* image, float* y_pad, hipStream_t s) {
  hipLaunchKernelGGL(k_logreg_image, dim3(49), dim3(256), 0, s, X, y, n, dw, image, y_pad);
  return static_cast<int>(hipGetLastError());
}
int sd_launch_pack_square(const float* P, const float* loc, int d, int NT, float* out, float* loc_pad, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_square, dim3(NT * NT), dim3(256), 0, s, P, loc, d, NT, out, loc_pad);
  return static_cast<int>(hipGetLastError());
}
