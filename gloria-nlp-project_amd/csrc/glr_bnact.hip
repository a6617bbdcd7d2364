// Fused training-mode BatchNorm2d + (residual add) + ReLU for channels-last bf16 activations: the 53 normalisation
// sites of the ResNet-50 image encoder (reference: torchvision resnet50 driven by
// /root/reference/gloria/models/cnn_backbones.py:31-35 and vision_model.py:67-86; Bottleneck = conv-bn-relu x2,
// conv-bn, + identity, relu).
//
// HBM-bound.  torch runs a site as BatchNorm (mean/variance pass + normalise pass; backward: scale/bias-gradient pass
// + dx pass) plus separate add / relu / relu-backward kernels: 13 tensor passes per site and direction pair without a
// skip connection, 19 with one.  Fused, an activation tensor of E bytes moves
//   forward   stats : read x                          E        apply: read x (+ skip), write y          2E (3E)
//   backward  reduce: read x, dy (+ y, write dz)      2E (4E)  apply: read x, dy (dz), write dx         3E
// (the ReLU mask is recomputed from x; with a skip connection it comes from y, and the masked gradient dz - which
// IS the skip connection's gradient - is written by the reduce pass and re-read by the apply pass).
//
// x is viewed [R = N*H*W, C], C contiguous; a thread owns 8 consecutive channels (one 16-byte load per row).
// * reduce passes: as many workgroups as the chip holds at once, each walking slabs of rows round-robin with 8 (4)
//   independent 16-byte loads per thread in flight (~100 KB per CU, what ~6 TB/s needs at HBM latency).
//   Per-workgroup partial sums [2][C][n_part] are combined in a fixed order by a second kernel (one wave per
//   channel): bitwise reproducible.
// * apply passes: short workgroups (256 threads x 4 vectors), and they walk the tensor in the OPPOSITE direction of the
//   reduce pass in front of them, so the tail the reduce pass touched last is still in the 256 MB Infinity Cache.
#include "glr_common.h"

namespace {

constexpr int BN_NT = 256;

__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 ldv(const unsigned short* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stv(unsigned short* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

// eight consecutive channels of one row, by element type: bf16 (16 bytes, the training configuration) or fp32 (32 bytes,
// the fp32 parity configuration of BASELINE config 1; round 3).  UNROLL scales the independent loads per thread so that
// the bytes in flight (and the registers holding them) are the same for both.
template <typename E> struct Vec8;
template <> struct Vec8<unsigned short> { uint4 v; };
template <> struct Vec8<float> { float4 a, b; };
template <typename E> struct ElemTraits;
template <> struct ElemTraits<unsigned short> { static constexpr int SHIFT = 0; };
template <> struct ElemTraits<float> { static constexpr int SHIFT = 1; };
__device__ __forceinline__ Vec8<unsigned short> ld8(const unsigned short* p) { return {ldv(p)}; }
__device__ __forceinline__ Vec8<float> ld8(const float* p) {
  return {*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)};
}
__device__ __forceinline__ void un8(const Vec8<unsigned short>& v, float (&f)[8]) { unpack8(v.v, f); }
__device__ __forceinline__ void un8(const Vec8<float>& v, float (&f)[8]) {
  f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
}
__device__ __forceinline__ void st8(unsigned short* p, const float (&f)[8]) { stv(p, pack8(f)); }
__device__ __forceinline__ void st8(float* p, const float (&f)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

// ---- pass 1 of both directions: per-channel partial sums of two quantities over a slab of rows
//   MODE 0 (forward stats)  q0 = x,  q1 = x^2
//   MODE 1 (backward)       q0 = dz, q1 = dz * xhat,  dz = dy * [z > 0]   (RESID: mask from y, dz also written out)
template <typename E, int MODE, bool RESID>
__global__ void __launch_bounds__(BN_NT) k_bn_reduce(const E* __restrict__ x, const E* __restrict__ dy,
                                                     const E* __restrict__ y, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, long long R, int C, int relu,
                                                     int n_part, float* __restrict__ part,
                                                     E* __restrict__ dz_out, const E* __restrict__ dy2) {
  constexpr int UN = (MODE == 0 ? 8 : 4) >> ElemTraits<E>::SHIFT;
  const int cg_per_blk = min(C / 8, 32), rl_per_blk = BN_NT / cg_per_blk;
  const int cg = blockIdx.x * cg_per_blk + threadIdx.x % cg_per_blk, rl = threadIdx.x / cg_per_blk;
  const int c0 = cg * 8;
  float a0[8], a1[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a0[i] = a1[i] = 0.f;
    if (MODE == 1) {
      mu[i] = mean[c0 + i]; is[i] = invstd[c0 + i];
      sc[i] = is[i] * gamma[c0 + i];
      sh[i] = beta[c0 + i] - mu[i] * sc[i];          // z = fma(x, sc, sh): the forward's own expression (same mask)
    }
  }
  // slabs of UN * rl_per_blk consecutive rows (one unrolled step of the workgroup = 8..32 KB of contiguous memory),
  // dealt round-robin to the gridDim.y workgroups of this channel block: every workgroup finishes at the same time
  const long long r_end = R;
  const long long slab = (long long)UN * rl_per_blk;
  for (long long r = (long long)blockIdx.y * slab + rl; r < r_end; r += slab * gridDim.y) {
    Vec8<E> xr[UN], dr[UN], yr[UN], d2[UN];
    const bool two = MODE == 1 && RESID && dy2 != nullptr;     // the gradient arrives as two tensors (main + skip consumer)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long ru = r + (long long)u * rl_per_blk;
      if (ru < r_end) {
        xr[u] = ld8(x + ru * C + c0);
        if (MODE == 1) dr[u] = ld8(dy + ru * C + c0);
        if (MODE == 1 && RESID) yr[u] = ld8(y + ru * C + c0);
        if (two) d2[u] = ld8(dy2 + ru * C + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long ru = r + (long long)u * rl_per_blk;
      if (ru < r_end) {
        float xv[8];
        un8(xr[u], xv);
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { a0[i] += xv[i]; a1[i] = __builtin_fmaf(xv[i], xv[i], a1[i]); }
        } else {
          float dv[8], yv[8], dzv[8];
          un8(dr[u], dv);
          if (RESID) un8(yr[u], yv);
          if (two) {
            float d2v[8];
            un8(d2[u], d2v);
#pragma unroll
            for (int i = 0; i < 8; ++i) dv[i] += d2v[i];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool on = !relu || (RESID ? yv[i] > 0.f : __builtin_fmaf(xv[i], sc[i], sh[i]) > 0.f);
            const float dz = on ? dv[i] : 0.f;
            dzv[i] = dz;
            a0[i] += dz;
            a1[i] = __builtin_fmaf(dz, (xv[i] - mu[i]) * is[i], a1[i]);
          }
          if (RESID) st8(dz_out + ru * C + c0, dzv);
        }
      }
    }
  }
  // reduce over the row lanes of the block (fixed order), one partial per (blockIdx.y, channel)
  __shared__ float red[2][BN_NT][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = a0[i]; red[1][threadIdx.x][i] = a1[i]; }
  __syncthreads();
  // thread t < 8 cg_per_blk sums channel column t of both quantities over the row lanes
  const int ncol = cg_per_blk * 8;
  if ((int)threadIdx.x < ncol) {
    const int col = threadIdx.x, tcg = col >> 3, ti = col & 7;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < rl_per_blk; ++k) { s0 += red[0][k * cg_per_blk + tcg][ti]; s1 += red[1][k * cg_per_blk + tcg][ti]; }
    const size_t ch = (size_t)blockIdx.x * ncol + col;
    part[ch * n_part + blockIdx.y] = s0;                       // [2][C][n_part]: the second stage reads along the parts
    part[((size_t)C + ch) * n_part + blockIdx.y] = s1;
  }
}

// second stage: one wave per channel sums the channel's partials of both quantities (lane-strided, then a fixed
// butterfly) in double
//   FWD: mean / invstd (+ running statistics, momentum form of nn.BatchNorm2d)
//   BWD: dgamma, dbeta and the two per-channel means the dx formula needs
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
template <bool FWD>
__global__ void __launch_bounds__(BN_NT) k_bn_finish(const float* __restrict__ part, int n_part, int C, long long R, float eps,
                                                     float momentum, float* __restrict__ o0, float* __restrict__ o1,
                                                     float* __restrict__ o2, float* __restrict__ o3, long long* __restrict__ counter) {
  const int lane = threadIdx.x & 63;
  if (counter != nullptr && blockIdx.x == 0 && threadIdx.x == 0) counter[0] += 1;     // nn.BatchNorm2d.num_batches_tracked
  const int c = blockIdx.x * (BN_NT / 64) + (threadIdx.x >> 6);
  if (c >= C) return;
  const float* p0 = part + (size_t)c * n_part;
  const float* p1 = part + ((size_t)C + c) * n_part;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < n_part; k += 64) { s += (double)p0[k]; ss += (double)p1[k]; }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (lane != 0) return;
  if (FWD) {
    const double m = s / (double)R;
    double var = ss / (double)R - m * m;
    if (var < 0.0) var = 0.0;
    o0[c] = (float)m;                                        // mean
    o1[c] = (float)(1.0 / sqrt(var + (double)eps));          // invstd
    if (o2) {                                                // running mean / var
      const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
      o2[c] = (1.f - momentum) * o2[c] + momentum * (float)m;
      o3[c] = (1.f - momentum) * o3[c] + momentum * (float)unb;
    }
  } else {
    o0[c] = (float)ss;                                       // dgamma
    o1[c] = (float)s;                                        // dbeta
    o2[c] = (float)(s / (double)R);                          // mean(dz)
    o3[c] = (float)(ss / (double)R);                         // mean(dz xhat)
  }
}

// forward apply: y = relu?( (x - mean) invstd gamma + beta (+ residual) ); blocks walk the tensor backwards
template <typename E, bool RESID>
__global__ void __launch_bounds__(BN_NT) k_bn_apply(const E* __restrict__ x, const E* __restrict__ res,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    long long n_vec, int C, int relu, E* __restrict__ y) {
  constexpr int UN = 4 >> ElemTraits<E>::SHIFT;
  const unsigned cpr = (unsigned)C / 8u;                     // a power of two <= 256: a thread keeps its channels
  const int c0 = (int)(threadIdx.x & (cpr - 1)) * 8;
  const long long v0 = (long long)(gridDim.x - 1 - blockIdx.x) * (BN_NT * UN) + threadIdx.x;
  Vec8<E> xr[UN], rr[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const long long v = v0 + u * BN_NT;
    if (v < n_vec) {
      xr[u] = ld8(x + v * 8);
      if (RESID) rr[u] = ld8(res + v * 8);
    }
  }
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = invstd[c0 + i] * gamma[c0 + i];
    sh[i] = beta[c0 + i] - mean[c0 + i] * sc[i];
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const long long v = v0 + u * BN_NT;
    if (v < n_vec) {
      float xv[8], rv[8], o[8];
      un8(xr[u], xv);
      if (RESID) un8(rr[u], rv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float z = __builtin_fmaf(xv[i], sc[i], sh[i]);
        if (RESID) z += rv[i];
        o[i] = relu ? fmaxf(z, 0.f) : z;
      }
      st8(y + v * 8, o);
    }
  }
}

// backward apply: dx = (dz - mean(dz) - xhat mean(dz xhat)) invstd gamma.  RESID: `dy` is the dz the reduce pass wrote.
template <typename E, bool RESID>
__global__ void __launch_bounds__(BN_NT) k_bn_bwd_apply(const E* __restrict__ x, const E* __restrict__ dy,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ m_dz, const float* __restrict__ m_dzx,
                                                        long long n_vec, int C, int relu, E* __restrict__ dx) {
  constexpr int UN = 4 >> ElemTraits<E>::SHIFT;
  const unsigned cpr = (unsigned)C / 8u;
  const int c0 = (int)(threadIdx.x & (cpr - 1)) * 8;
  const long long v0 = (long long)(gridDim.x - 1 - blockIdx.x) * (BN_NT * UN) + threadIdx.x;
  Vec8<E> xr[UN], dr[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const long long v = v0 + u * BN_NT;
    if (v < n_vec) {
      xr[u] = ld8(x + v * 8);
      dr[u] = ld8(dy + v * 8);
    }
  }
  // dx = dz * k1 - (k2 + x * k3):  k1 = invstd gamma, k3 = invstd mean(dz xhat) k1, k2 = mean(dz) k1 - mean k3
  float k1[8], k2[8], k3[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float is = invstd[c0 + i], mu = mean[c0 + i];
    k1[i] = is * gamma[c0 + i];
    k3[i] = is * m_dzx[c0 + i] * k1[i];
    k2[i] = m_dz[c0 + i] * k1[i] - mu * k3[i];
    sh[i] = beta[c0 + i] - mu * k1[i];
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const long long v = v0 + u * BN_NT;
    if (v < n_vec) {
      float xv[8], dv[8], o[8];
      un8(xr[u], xv);
      un8(dr[u], dv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool on = RESID || !relu || __builtin_fmaf(xv[i], k1[i], sh[i]) > 0.f;
        const float dz = on ? dv[i] : 0.f;
        o[i] = __builtin_fmaf(dz, k1[i], -__builtin_fmaf(xv[i], k3[i], k2[i]));
      }
      st8(dx + v * 8, o);
    }
  }
}

struct BnPlan { int cg_per_blk, col_blocks, n_part; };

// workgroups of a reduce pass = what the chip holds AT ONCE (CUs x resident workgroups of that kernel: a grid-stride
// pass with one and a bit rounds of workgroups ends with most of the chip idle), split over the channel blocks;
// never more parts than slabs
constexpr int BN_MAX_WG = 2048;
int bn_resident(GlrOccupancy& occ, const void* fn) {
  const int n = glr_dev_cus() * occ.get(fn, BN_NT);     // per-device tables (glr_common.h): no library-global state
  return n > BN_MAX_WG ? BN_MAX_WG : n;
}
BnPlan bn_plan(long long R, int C, int resident, int un) {
  BnPlan p;
  p.cg_per_blk = C / 8 < 32 ? C / 8 : 32;
  p.col_blocks = (C / 8) / p.cg_per_blk;
  const int rl = BN_NT / p.cg_per_blk;
  long long want = resident / p.col_blocks;
  const long long slabs = (R + (long long)rl * un - 1) / ((long long)rl * un);
  if (want > slabs) want = slabs;
  if (want < 1) want = 1;
  p.n_part = (int)want;
  return p;
}

bool bn_shape_ok(long long R, int C) { return R > 0 && C >= 8 && C <= 2048 && (C & (C - 1)) == 0; }

}  // namespace

extern "C" int glr_bn_workspace_floats(long long R, int C) {
  if (!bn_shape_ok(R, C)) return 0;
  return bn_plan(R, C, BN_MAX_WG, 1).n_part * 2 * C;          // upper bound over both directions
}

template <typename E>
int bn_fwd_launch(const void* x, const void* residual, const float* gamma, const float* beta, long long R, int C, float eps,
                  float momentum, int relu, float* run_mean, float* run_var, long long* num_batches_tracked, float* mean,
                  float* invstd, float* workspace, void* y, hipStream_t st) {
  constexpr int UNA = 4 >> ElemTraits<E>::SHIFT;
  static GlrOccupancy occ0;
  const int res0 = bn_resident(occ0, (const void*)k_bn_reduce<E, 0, false>);
  const BnPlan pl = bn_plan(R, C, res0, 8 >> ElemTraits<E>::SHIFT);
  hipLaunchKernelGGL((k_bn_reduce<E, 0, false>), dim3(pl.col_blocks, pl.n_part), dim3(BN_NT), 0, st, (const E*)x,
                     (const E*)nullptr, (const E*)nullptr, nullptr, nullptr, nullptr, nullptr, R, C, 0, pl.n_part, workspace,
                     (E*)nullptr, (const E*)nullptr);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL((k_bn_finish<true>), dim3((C + 3) / 4), dim3(BN_NT), 0, st, workspace, pl.n_part, C, R, eps, momentum,
                     mean, invstd, run_mean, run_var, num_batches_tracked);
  GLR_CHECK_LAUNCH();
  const long long n_vec = R * C / 8;
  const int grid = (int)((n_vec + BN_NT * UNA - 1) / (BN_NT * UNA));
  if (residual)
    hipLaunchKernelGGL((k_bn_apply<E, true>), dim3(grid), dim3(BN_NT), 0, st, (const E*)x, (const E*)residual,
                       mean, invstd, gamma, beta, n_vec, C, relu, (E*)y);
  else
    hipLaunchKernelGGL((k_bn_apply<E, false>), dim3(grid), dim3(BN_NT), 0, st, (const E*)x, (const E*)nullptr, mean, invstd, gamma,
                       beta, n_vec, C, relu, (E*)y);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

template <typename E>
int bn_bwd_launch(const void* x, const void* dy, const void* dy2, const void* y, const float* gamma, const float* beta,
                  const float* mean, const float* invstd, long long R, int C, int relu, int has_residual, float* workspace,
                  float* out4c, void* dx, void* dres, hipStream_t st) {
  constexpr int UNA = 4 >> ElemTraits<E>::SHIFT;
  static GlrOccupancy occ1, occ2;
  const int res1 = bn_resident(occ1, (const void*)k_bn_reduce<E, 1, false>);
  const int res2 = bn_resident(occ2, (const void*)k_bn_reduce<E, 1, true>);
  const BnPlan pl = bn_plan(R, C, has_residual ? res2 : res1, 4 >> ElemTraits<E>::SHIFT);
  const dim3 rgrid(pl.col_blocks, pl.n_part);
  if (has_residual)
    hipLaunchKernelGGL((k_bn_reduce<E, 1, true>), rgrid, dim3(BN_NT), 0, st, (const E*)x, (const E*)dy,
                       (const E*)y, mean, invstd, gamma, beta, R, C, relu, pl.n_part, workspace,
                       (E*)dres, (const E*)dy2);
  else
    hipLaunchKernelGGL((k_bn_reduce<E, 1, false>), rgrid, dim3(BN_NT), 0, st, (const E*)x, (const E*)dy,
                       (const E*)nullptr, mean, invstd, gamma, beta, R, C, relu, pl.n_part, workspace, (E*)nullptr, (const E*)nullptr);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL((k_bn_finish<false>), dim3((C + 3) / 4), dim3(BN_NT), 0, st, workspace, pl.n_part, C, R, 0.f, 0.f, out4c,
                     out4c + C, out4c + 2 * C, out4c + 3 * C, (long long*)nullptr);
  GLR_CHECK_LAUNCH();
  const long long n_vec = R * C / 8;
  const int grid = (int)((n_vec + BN_NT * UNA - 1) / (BN_NT * UNA));
  if (has_residual)
    hipLaunchKernelGGL((k_bn_bwd_apply<E, true>), dim3(grid), dim3(BN_NT), 0, st, (const E*)x, (const E*)dres,
                       mean, invstd, gamma, beta, out4c + 2 * C, out4c + 3 * C, n_vec, C, relu, (E*)dx);
  else
    hipLaunchKernelGGL((k_bn_bwd_apply<E, false>), dim3(grid), dim3(BN_NT), 0, st, (const E*)x, (const E*)dy,
                       mean, invstd, gamma, beta, out4c + 2 * C, out4c + 3 * C, n_vec, C, relu, (E*)dx);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_bn_act_fwd(const void* x, const void* residual, const float* gamma, const float* beta, long long R,
                              int C, float eps, float momentum, int relu, float* run_mean, float* run_var,
                              long long* num_batches_tracked, float* mean, float* invstd, float* workspace, void* y,
                              int dtype, void* stream) {
  if (!x || !gamma || !beta || !mean || !invstd || !workspace || !y || !bn_shape_ok(R, C)) return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GLR_BF16)
    return bn_fwd_launch<unsigned short>(x, residual, gamma, beta, R, C, eps, momentum, relu, run_mean, run_var,
                                         num_batches_tracked, mean, invstd, workspace, y, st);
  if (dtype == GLR_F32)
    return bn_fwd_launch<float>(x, residual, gamma, beta, R, C, eps, momentum, relu, run_mean, run_var, num_batches_tracked,
                                mean, invstd, workspace, y, st);
  return GLR_EDTYPE;
}

// out4c = [dgamma | dbeta | mean(dz) | mean(dz xhat)], 4*C floats.  has_residual: `y` (the forward's output) gives
// the ReLU mask and `dres` receives the masked gradient (the skip connection's gradient).
extern "C" int glr_bn_act_bwd(const void* x, const void* dy, const void* dy2, const void* y, const float* gamma, const float* beta,
                              const float* mean, const float* invstd, long long R, int C, int relu, int has_residual,
                              float* workspace, float* out4c, void* dx, void* dres, int dtype, void* stream) {
  if (!x || !dy || !gamma || !beta || !mean || !invstd || !workspace || !out4c || !dx || !bn_shape_ok(R, C) ||
      (has_residual && (!y || !dres)) || (dy2 && !has_residual))
    return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GLR_BF16)
    return bn_bwd_launch<unsigned short>(x, dy, dy2, y, gamma, beta, mean, invstd, R, C, relu, has_residual, workspace, out4c, dx,
                                         dres, st);
  if (dtype == GLR_F32)
    return bn_bwd_launch<float>(x, dy, dy2, y, gamma, beta, mean, invstd, R, C, relu, has_residual, workspace, out4c, dx, dres, st);
  return GLR_EDTYPE;
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, padding 1) of the ResNet stem on channels-last bf16, forward and backward.  torch's NHWC kernels
// keep an int64 argmax per output element (0.74 GB for the 0.18 GB output here) and the backward zero-fills the 0.74 GB
// gradient before scattering into it: 1.9 ms per step for one layer.  Here the forward stores the argmax as the window
// position 0..8 in ONE byte (first maximum in scan order, torch's rule) and the backward GATHERS: every input pixel looks
// at the <= 4 windows that contain it.  Traffic 0.92 + 0.28 GB forward, 0.28 + 0.74 GB backward.
namespace {
__global__ void __launch_bounds__(256) k_maxpool3s2_fwd(const unsigned short* __restrict__ x, int H, int W, int C, int Ho, int Wo,
                                                        long long total, unsigned short* __restrict__ y, unsigned char* __restrict__ idx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // (n, oh, ow, channel group of 8)
  if (i >= total) return;
  const int cg = C >> 3;
  const int g = (int)(i % cg);
  long long t = i / cg;
  const int ow = (int)(t % Wo); t /= Wo;
  const int oh = (int)(t % Ho);
  const long long n = t / Ho;
  float best[8];
  unsigned bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0u; }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int h = 2 * oh - 1 + kh;
    if (h < 0 || h >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int w = 2 * ow - 1 + kw;
      if (w < 0 || w >= W) continue;
      float v[8];
      unpack8(ldv(x + ((n * H + h) * W + w) * C + g * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = (unsigned)(kh * 3 + kw); }
    }
  }
  stv(y + i * 8, pack8(best));
  *reinterpret_cast<uint2*>(idx + i * 8) =
      make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
}

__global__ void __launch_bounds__(256) k_maxpool3s2_bwd(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                        int H, int W, int C, int Ho, int Wo, long long total,
                                                        unsigned short* __restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // (n, h, w, channel group of 8)
  if (i >= total) return;
  const int cg = C >> 3;
  const int g = (int)(i % cg);
  long long t = i / cg;
  const int w = (int)(t % W); t /= W;
  const int h = (int)(t % H);
  const long long n = t / H;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int oh1 = min(Ho - 1, (h + 1) >> 1), ow1 = min(Wo - 1, (w + 1) >> 1);
  for (int oh = h >> 1; oh <= oh1; ++oh)
    for (int ow = w >> 1; ow <= ow1; ++ow) {
      const unsigned pos = (unsigned)((h - (2 * oh - 1)) * 3 + (w - (2 * ow - 1)));
      const long long o = ((n * Ho + oh) * Wo + ow) * cg + g;
      const uint2 b = *reinterpret_cast<const uint2*>(idx + o * 8);
      float d[8];
      unpack8(ldv(dy + o * 8), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned be = ((e < 4 ? b.x : b.y) >> (8 * (e & 3))) & 0xffu;
        acc[e] += be == pos ? d[e] : 0.f;
      }
    }
  stv(dx + i * 8, pack8(acc));
}
}  // namespace

extern "C" int glr_maxpool3s2_fwd(const void* x, int B, int H, int W, int C, void* y, uint8_t* idx, void* stream) {
  if (!x || !y || !idx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return GLR_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(k_maxpool3s2_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, H, W, C, Ho, Wo, total, (unsigned short*)y, idx);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_maxpool3s2_bwd(const void* dy, const uint8_t* idx, int B, int H, int W, int C, void* dx, void* stream) {
  if (!dy || !idx || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return GLR_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * H * W * (C / 8);
  hipLaunchKernelGGL(k_maxpool3s2_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dy, idx, H, W, C, Ho, Wo, total, (unsigned short*)dx);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
