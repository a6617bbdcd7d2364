// Fused training-mode BatchNorm2d + (residual add) + ReLU for channels-last bf16 activations (the ResNet-50
// bottleneck epilogues feeding the local embedder; reference: torchvision resnet50 via
// /root/reference/gloria/models/cnn_backbones.py:31-35, vision_model.py:67-86).
//
// HBM-bound.  torch runs this as BN (2 + 2 passes) + add + relu (+ relu backward) kernels; fused, an
// activation tensor of E elements moves
//   forward   stats: read x (2E bytes)            apply: read x (+ residual), write y (4E..6E)
//   backward  reduce: read x, dy (4E)             apply: read x, dy, write dx (6E)      [mask recomputed from x]
//             (with a residual the ReLU mask comes from y: +2E per pass)
// x is viewed [R = N*H*W, C], C contiguous; a thread owns 8 consecutive channels (16-byte loads).
// Reductions: per-block partial sums [n_part][C] in fixed order, then a second tiny kernel: bitwise reproducible.
#include "glr_common.h"

namespace {

constexpr int BN_NT = 256;
constexpr int BN_MAX_PART = 512;

struct bf8 { unsigned short v[8]; };

__device__ __forceinline__ void ld8(const unsigned short* p, float (&f)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void st8(unsigned short* p, const float (&f)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- pass 1 of both directions: per-channel partial sums of two quantities
//   MODE 0 (forward stats)  q0 = x,  q1 = x^2
//   MODE 1 (backward)       q0 = dz, q1 = dz * xhat,  dz = dy * [z > 0]
template <int MODE, bool RESID>
__global__ void __launch_bounds__(BN_NT) k_bn_reduce(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                     const unsigned short* __restrict__ y, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, long long R, int C, int relu,
                                                     float* __restrict__ part) {
  const int cg_per_blk = min(C / 8, 32), rl_per_blk = BN_NT / cg_per_blk;
  const int cg = blockIdx.x * cg_per_blk + threadIdx.x % cg_per_blk, rl = threadIdx.x / cg_per_blk;
  const int c0 = cg * 8;
  float a0[8], a1[8], mu[8], is[8], ga[8], be[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a0[i] = a1[i] = 0.f;
    if (MODE == 1) { mu[i] = mean[c0 + i]; is[i] = invstd[c0 + i]; ga[i] = gamma[c0 + i]; be[i] = beta[c0 + i]; }
  }
  const long long rstep = (long long)gridDim.y * rl_per_blk;
  constexpr int UN = 4;                                   // rows in flight per thread (independent 16-byte loads)
  for (long long r = (long long)blockIdx.y * rl_per_blk + rl; r < R; r += UN * rstep) {
    float xv[UN][8], dv[UN][8], yv[UN][8];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long ru = r + u * rstep;
      if (ru < R) {
        ld8(x + ru * C + c0, xv[u]);
        if (MODE == 1) ld8(dy + ru * C + c0, dv[u]);
        if (MODE == 1 && RESID) ld8(y + ru * C + c0, yv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (r + u * rstep < R) {
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { a0[i] += xv[u][i]; a1[i] += xv[u][i] * xv[u][i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float xh = (xv[u][i] - mu[i]) * is[i];
            const bool on = !relu || (RESID ? yv[u][i] > 0.f : xh * ga[i] + be[i] > 0.f);
            const float dz = on ? dv[u][i] : 0.f;
            a0[i] += dz;
            a1[i] += dz * xh;
          }
        }
      }
    }
  }
  // reduce over the row lanes of the block (fixed order), one partial per (blockIdx.y, channel)
  __shared__ float red[2][BN_NT][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = a0[i]; red[1][threadIdx.x][i] = a1[i]; }
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < rl_per_blk; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a0[i] += red[0][k * cg_per_blk + threadIdx.x][i];
        a1[i] += red[1][k * cg_per_blk + threadIdx.x][i];
      }
    float* p0 = part + ((size_t)blockIdx.y * 2 + 0) * C + c0;
    float* p1 = part + ((size_t)blockIdx.y * 2 + 1) * C + c0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { p0[i] = a0[i]; p1[i] = a1[i]; }
  }
}

// forward finish: mean / invstd (+ running statistics, momentum form of nn.BatchNorm2d)
__global__ void k_bn_stats_finish(const float* __restrict__ part, int n_part, int C, long long R, float eps, float momentum,
                                  float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                  float* __restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < n_part; ++k) { s += part[((size_t)k * 2) * C + c]; ss += part[((size_t)k * 2 + 1) * C + c]; }
  const double m = s / (double)R;
  double var = ss / (double)R - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}

// backward finish: sums -> dgamma, dbeta and the two per-channel means the dx formula needs
__global__ void k_bn_bwd_finish(const float* __restrict__ part, int n_part, int C, long long R, float* __restrict__ dgamma,
                                float* __restrict__ dbeta, float* __restrict__ m_dz, float* __restrict__ m_dzx) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, sx = 0.0;
  for (int k = 0; k < n_part; ++k) { s += part[((size_t)k * 2) * C + c]; sx += part[((size_t)k * 2 + 1) * C + c]; }
  dbeta[c] = (float)s;
  dgamma[c] = (float)sx;
  m_dz[c] = (float)(s / (double)R);
  m_dzx[c] = (float)(sx / (double)R);
}

// forward apply: y = relu?( (x - mean) invstd gamma + beta (+ residual) )
template <bool RESID>
__global__ void __launch_bounds__(BN_NT) k_bn_apply(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    long long n_vec, int C, int relu, unsigned short* __restrict__ y) {
  // the launcher makes the grid stride a multiple of the vectors per row, so a thread stays on its 8 channels
  const unsigned cpr = (unsigned)C / 8u;
  const int c0 = (int)(((unsigned)blockIdx.x * BN_NT + threadIdx.x) % cpr) * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = invstd[c0 + i] * gamma[c0 + i];
    sh[i] = beta[c0 + i] - mean[c0 + i] * sc[i];
  }
  const long long vstep = (long long)gridDim.x * BN_NT;
  constexpr int UN = 4;
  for (long long v = (long long)blockIdx.x * BN_NT + threadIdx.x; v < n_vec; v += UN * vstep) {
    float xv[UN][8], rv[UN][8];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long vu = v + u * vstep;
      if (vu < n_vec) {
        ld8(x + vu * 8, xv[u]);
        if (RESID) ld8(res + vu * 8, rv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long vu = v + u * vstep;
      if (vu < n_vec) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float z = __builtin_fmaf(xv[u][i], sc[i], sh[i]);
          if (RESID) z += rv[u][i];
          o[i] = relu ? fmaxf(z, 0.f) : z;
        }
        st8(y + vu * 8, o);
      }
    }
  }
}

// backward apply: dx = (dz - mean(dz) - xhat mean(dz xhat)) invstd gamma;  dres = dz
template <bool RESID>
__global__ void __launch_bounds__(BN_NT) k_bn_bwd_apply(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                        const unsigned short* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ m_dz,
                                                        const float* __restrict__ m_dzx, long long n_vec, int C, int relu,
                                                        unsigned short* __restrict__ dx, unsigned short* __restrict__ dres) {
  const unsigned cpr = (unsigned)C / 8u;
  const int c0 = (int)(((unsigned)blockIdx.x * BN_NT + threadIdx.x) % cpr) * 8;
  float mu[8], is[8], ga[8], be[8], md[8], mx[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mu[i] = mean[c0 + i]; is[i] = invstd[c0 + i]; ga[i] = gamma[c0 + i]; be[i] = beta[c0 + i];
    md[i] = m_dz[c0 + i]; mx[i] = m_dzx[c0 + i];
  }
  const long long vstep = (long long)gridDim.x * BN_NT;
  constexpr int UN = 2;
  for (long long v = (long long)blockIdx.x * BN_NT + threadIdx.x; v < n_vec; v += UN * vstep) {
    float xv[UN][8], dv[UN][8], yv[UN][8];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long vu = v + u * vstep;
      if (vu < n_vec) {
        ld8(x + vu * 8, xv[u]);
        ld8(dy + vu * 8, dv[u]);
        if (RESID) ld8(y + vu * 8, yv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long vu = v + u * vstep;
      if (vu < n_vec) {
        float o[8], dr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[u][i] - mu[i]) * is[i];
          const bool on = !relu || (RESID ? yv[u][i] > 0.f : xh * ga[i] + be[i] > 0.f);
          const float dz = on ? dv[u][i] : 0.f;
          o[i] = (dz - md[i] - xh * mx[i]) * is[i] * ga[i];
          dr[i] = dz;
        }
        st8(dx + vu * 8, o);
        if (RESID) st8(dres + vu * 8, dr);
      }
    }
  }
}

// grid of the element-wise passes: grid * BN_NT must be a multiple of the vectors per row (C / 8) so that every
// thread keeps its channels; a multiple of lcm(C/8, 256) / 256 blocks does that
int apply_grid(long long n_vec, int C) {
  const int cpr = C / 8;
  int a = cpr, b = BN_NT;
  while (b) { const int t = a % b; a = b; b = t; }            // a = gcd(cpr, 256)
  const int unit = cpr / a;                                   // blocks per period
  long long blocks = (n_vec + BN_NT - 1) / BN_NT;
  if (blocks > 8192) blocks = 8192;
  blocks = (blocks + unit - 1) / unit * unit;
  return (int)blocks;
}

int n_parts(long long R, int C) {
  const int cg_per_blk = C / 8 < 32 ? C / 8 : 32, rl = BN_NT / cg_per_blk;
  const int col_blocks = (C / 8) / cg_per_blk;
  long long want = 2048 / col_blocks;                       // ~2048 workgroups in flight
  const long long max_rows = (R + rl - 1) / rl;
  if (want > max_rows) want = max_rows;
  if (want > BN_MAX_PART) want = BN_MAX_PART;
  if (want < 1) want = 1;
  return (int)want;
}

}  // namespace

extern "C" int glr_bn_workspace_floats(long long R, int C) { return n_parts(R, C) * 2 * C; }

extern "C" int glr_bn_act_fwd(const void* x, const void* residual, const float* gamma, const float* beta, long long R,
                              int C, float eps, float momentum, int relu, float* run_mean, float* run_var, float* mean,
                              float* invstd, float* workspace, void* y, void* stream) {
  if (!x || !gamma || !beta || !mean || !invstd || !workspace || !y || R <= 0 || C <= 0 || C % 8 != 0 || (C / 8) % (C / 8 < 32 ? C / 8 : 32) != 0)
    return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int cg_per_blk = C / 8 < 32 ? C / 8 : 32, np = n_parts(R, C);
  hipLaunchKernelGGL((k_bn_reduce<0, false>), dim3((C / 8) / cg_per_blk, np), dim3(BN_NT), 0, st, (const unsigned short*)x,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, R, C, 0, workspace);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_stats_finish, dim3((C + 255) / 256), dim3(256), 0, st, workspace, np, C, R, eps, momentum, mean, invstd,
                     run_mean, run_var);
  GLR_CHECK_LAUNCH();
  const long long n_vec = R * C / 8;
  const int grid = apply_grid(n_vec, C);
  if (residual)
    hipLaunchKernelGGL((k_bn_apply<true>), dim3(grid), dim3(BN_NT), 0, st, (const unsigned short*)x, (const unsigned short*)residual,
                       mean, invstd, gamma, beta, n_vec, C, relu, (unsigned short*)y);
  else
    hipLaunchKernelGGL((k_bn_apply<false>), dim3(grid), dim3(BN_NT), 0, st, (const unsigned short*)x, nullptr, mean, invstd, gamma,
                       beta, n_vec, C, relu, (unsigned short*)y);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_bn_act_bwd(const void* x, const void* dy, const void* y, const float* gamma, const float* beta,
                              const float* mean, const float* invstd, long long R, int C, int relu, int has_residual,
                              float* workspace, float* dgamma, float* dbeta, float* tmp2c, void* dx, void* dres,
                              void* stream) {
  if (!x || !dy || !gamma || !beta || !mean || !invstd || !workspace || !dgamma || !dbeta || !tmp2c || !dx || R <= 0 || C <= 0 ||
      C % 8 != 0 || (has_residual && (!y || !dres)))
    return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int cg_per_blk = C / 8 < 32 ? C / 8 : 32, np = n_parts(R, C);
  const dim3 rgrid((C / 8) / cg_per_blk, np);
  if (has_residual)
    hipLaunchKernelGGL((k_bn_reduce<1, true>), rgrid, dim3(BN_NT), 0, st, (const unsigned short*)x, (const unsigned short*)dy,
                       (const unsigned short*)y, mean, invstd, gamma, beta, R, C, relu, workspace);
  else
    hipLaunchKernelGGL((k_bn_reduce<1, false>), rgrid, dim3(BN_NT), 0, st, (const unsigned short*)x, (const unsigned short*)dy,
                       nullptr, mean, invstd, gamma, beta, R, C, relu, workspace);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_bwd_finish, dim3((C + 255) / 256), dim3(256), 0, st, workspace, np, C, R, dgamma, dbeta, tmp2c, tmp2c + C);
  GLR_CHECK_LAUNCH();
  const long long n_vec = R * C / 8;
  const int grid = apply_grid(n_vec, C);
  if (has_residual)
    hipLaunchKernelGGL((k_bn_bwd_apply<true>), dim3(grid), dim3(BN_NT), 0, st, (const unsigned short*)x, (const unsigned short*)dy,
                       (const unsigned short*)y, mean, invstd, gamma, beta, tmp2c, tmp2c + C, n_vec, C, relu, (unsigned short*)dx,
                       (unsigned short*)dres);
  else
    hipLaunchKernelGGL((k_bn_bwd_apply<false>), dim3(grid), dim3(BN_NT), 0, st, (const unsigned short*)x, (const unsigned short*)dy,
                       nullptr, mean, invstd, gamma, beta, tmp2c, tmp2c + C, n_vec, C, relu, (unsigned short*)dx, nullptr);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
