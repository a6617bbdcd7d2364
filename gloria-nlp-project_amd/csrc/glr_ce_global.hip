// K2: dual cross-entropy over a B x B similarity matrix (labels = arange), forward + backward.
//     Replaces the nn.CrossEntropyLoss pairs of /root/reference/gloria/loss/gloria_loss.py:86-87
//     and :167-170.  HBM/latency bound: 2*B*B*4 bytes.
// K3: global similarity matrix, forward + backward.  Replaces the norm/bmm/clamp/scale of
//     global_loss (gloria_loss.py:75-80).  100 MFLOP at B = 256: latency bound, plain FMA.
#include "glr_common.h"

namespace {

// ---------------------------------------------------------------- K2
// blocks [0, B): row b; blocks [B, 2B): column i.  256 threads.
__global__ void __launch_bounds__(256) k_ce_lse(const float* __restrict__ sim, int B, float* __restrict__ lse_row,
                                                float* __restrict__ lse_col) {
  __shared__ float sh[4];
  const int which = blockIdx.x >= B;
  const int k = which ? blockIdx.x - B : blockIdx.x;
  const size_t stride = which ? (size_t)B : 1, base = which ? (size_t)k : (size_t)k * B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < B; j += 256) m = fmaxf(m, sim[base + j * stride]);
  m = wave_max(m);
  if (lane == 0) sh[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float s = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) s += expf(sim[base + j * stride] - m);
  s = wave_sum(s);
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float lse = m + logf(sh[0] + sh[1] + sh[2] + sh[3]);
    (which ? lse_col : lse_row)[k] = lse;
  }
}

// one block: losses[0] = mean_b(lse_row[b] - sim[b,b]), losses[1] = mean_i(lse_col[i] - sim[i,i]);
// fixed summation order => bitwise reproducible
__global__ void __launch_bounds__(256) k_ce_finish(const float* __restrict__ sim, int B,
                                                   const float* __restrict__ lse_row,
                                                   const float* __restrict__ lse_col, float* __restrict__ losses) {
  __shared__ float sh[2][4];
  float a = 0.f, c = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float dg = sim[(size_t)j * B + j];
    a += lse_row[j] - dg;
    c += lse_col[j] - dg;
  }
  a = wave_sum(a);
  c = wave_sum(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wave] = a; sh[1][wave] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    losses[0] = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / (float)B;
    losses[1] = (sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) / (float)B;
  }
}

__global__ void __launch_bounds__(256) k_ce_bwd(const float* __restrict__ sim, int B,
                                                const float* __restrict__ lse_row,
                                                const float* __restrict__ lse_col, const float* __restrict__ g,
                                                int row0, int n_rows, float* __restrict__ dsim) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n_rows * B) return;
  const int r = row0 + (int)(idx / B), i = (int)(idx % B);
  const float s = sim[(size_t)r * B + i];
  const float dg = (r == i) ? 1.f : 0.f;
  const float invB = 1.f / (float)B;
  dsim[idx] = g[0] * invB * (expf(s - lse_row[r]) - dg) + g[1] * invB * (expf(s - lse_col[i]) - dg);
}

// ---------------------------------------------------------------- K3
// one wave per row: L2 norm
__global__ void __launch_bounds__(256) k_row_norms(const float* __restrict__ x, int rows, int D,
                                                   float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[(size_t)row * D + d]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) out[row] = sqrtf(s);
}

// block per image row b; each wave walks sentences i = wave, wave+4, ...
__global__ void __launch_bounds__(256) k_global_sim(const float* __restrict__ img, const float* __restrict__ txt,
                                                    const float* __restrict__ ni, const float* __restrict__ nt,
                                                    int B_txt, int D, float temp3, float eps,
                                                    float* __restrict__ sim, int ld_sim) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < B_txt; i += 4) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += img[(size_t)b * D + d] * txt[(size_t)i * D + d];
    s = wave_sum(s);
    if (lane == 0) sim[(size_t)b * ld_sim + i] = s / fmaxf(ni[b] * nt[i], eps) * temp3;
  }
}

// gradient wrt the rows of X given partner rows Y:  for fixed x-row a,
//   dX[a] = temp3 * sum_j g(a,j) * ( Y_j / den  -  [nx*ny > eps] * dot * ny / (den^2 * nx) * X_a )
// transposed = 0: a indexes images (g(a,j) = dsim[a, j]); transposed = 1: a indexes sentences.
// block per a; 256 threads; dynamic LDS: coef[n_other]
__global__ void __launch_bounds__(256) k_global_bwd(const float* __restrict__ X, const float* __restrict__ Y,
                                                    const float* __restrict__ nx, const float* __restrict__ ny,
                                                    const float* __restrict__ dsim, int ld_sim, int transposed,
                                                    int n_other, int D, float temp3, float eps,
                                                    float* __restrict__ dX) {
  extern __shared__ float coef[];          // [n_other] then 4 partials
  float* part = coef + n_other;
  const int a = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float nxa = nx[a];
  float self = 0.f;                        // coefficient of X_a (summed per wave by lane 0)
  for (int j = wave; j < n_other; j += 4) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += X[(size_t)a * D + d] * Y[(size_t)j * D + d];
    s = wave_sum(s);
    const float g = transposed ? dsim[(size_t)j * ld_sim + a] : dsim[(size_t)a * ld_sim + j];
    const float prod = nxa * ny[j];
    const float den = fmaxf(prod, eps);
    if (lane == 0) {
      coef[j] = temp3 * g / den;
      if (prod >= eps && nxa > 0.f) self += temp3 * g * s * ny[j] / (den * den * nxa);
    }
  }
  if (lane == 0) part[wave] = self;
  __syncthreads();
  const float cself = part[0] + part[1] + part[2] + part[3];
  for (int d = threadIdx.x; d < D; d += 256) {
    // four independent chains (the loop is latency bound: one dependent FMA per L2 load otherwise), summed in a
    // fixed order
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    for (; j + 4 <= n_other; j += 4) {
      a0 += coef[j] * Y[(size_t)j * D + d];
      a1 += coef[j + 1] * Y[(size_t)(j + 1) * D + d];
      a2 += coef[j + 2] * Y[(size_t)(j + 2) * D + d];
      a3 += coef[j + 3] * Y[(size_t)(j + 3) * D + d];
    }
    for (; j < n_other; ++j) a0 += coef[j] * Y[(size_t)j * D + d];
    dX[(size_t)a * D + d] = ((a0 + a1) + (a2 + a3)) - cself * X[(size_t)a * D + d];
  }
}

}  // namespace

extern "C" int glr_dual_ce_fwd(const float* sim, int B, float* lse_row, float* lse_col, float* losses,
                               void* stream) {
  if (!sim || !lse_row || !lse_col || !losses || B <= 0) return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ce_lse, dim3(2 * B), dim3(256), 0, st, sim, B, lse_row, lse_col);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_ce_finish, dim3(1), dim3(256), 0, st, sim, B, lse_row, lse_col, losses);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_dual_ce_bwd(const float* sim, int B, const float* lse_row, const float* lse_col,
                               const float* g, int row0, int n_rows, float* dsim, void* stream) {
  if (!sim || !lse_row || !lse_col || !g || !dsim || B <= 0 || row0 < 0 || n_rows <= 0 || row0 + n_rows > B)
    return GLR_EINVAL;
  const size_t n = (size_t)n_rows * B;
  hipLaunchKernelGGL(k_ce_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sim, B,
                     lse_row, lse_col, g, row0, n_rows, dsim);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_global_sim_fwd(const float* img, const float* txt, int B_img, int B_txt, int D, float temp3,
                                  float eps, float* sim, int ld_sim, float* ni, float* nt, void* stream) {
  if (!img || !txt || !sim || !ni || !nt || B_img <= 0 || B_txt <= 0 || D <= 0 || ld_sim < B_txt) return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_row_norms, dim3((B_img + 3) / 4), dim3(256), 0, st, img, B_img, D, ni);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_row_norms, dim3((B_txt + 3) / 4), dim3(256), 0, st, txt, B_txt, D, nt);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_global_sim, dim3(B_img), dim3(256), 0, st, img, txt, ni, nt, B_txt, D, temp3, eps, sim,
                     ld_sim);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_global_sim_bwd(const float* img, const float* txt, const float* ni, const float* nt,
                                  const float* dsim, int ld_sim, int B_img, int B_txt, int D, float temp3,
                                  float eps, float* dimg, float* dtxt, void* stream) {
  if (!img || !txt || !ni || !nt || !dsim || !dimg || !dtxt || B_img <= 0 || B_txt <= 0 || D <= 0) return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_global_bwd, dim3(B_img), dim3(256), (B_txt + 4) * sizeof(float), st, img, txt, ni, nt, dsim,
                     ld_sim, 0, B_txt, D, temp3, eps, dimg);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_global_bwd, dim3(B_txt), dim3(256), (B_img + 4) * sizeof(float), st, txt, img, nt, ni, dsim,
                     ld_sim, 1, B_img, D, temp3, eps, dtxt);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
