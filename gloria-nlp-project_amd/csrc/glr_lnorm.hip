// Fused  y = LayerNorm(dropout(h) + inp)  of the BERT sub-layer outputs (BertSelfOutput / BertOutput of the text encoder,
// reference: transformers' BertModel built by /root/reference/gloria/models/text_model.py:18-20 and run under native
// AMP), forward and backward.
//
// Under autocast torch runs this as dropout (bf16) -> add (bf16 + fp32 -> fp32) -> layer_norm (fp32) -> a cast of the
// result to bf16 for the next Linear: 29 bytes of HBM traffic per element forward, ~47 backward (dropout backward,
// layer-norm input and weight gradients, the casts and adds that join the fp32 and bf16 gradient streams).  Fused:
//   forward   read h (bf16) + inp (fp32); write y (fp32, the residual stream) + y (bf16, the next GEMM's operand)   12 B
//   backward  read dy (fp32) + dy (bf16) + h + inp; write d_inp (fp32) + d_h (bf16)                                  18 B
// plus 1 bit per element of dropout mask and 8 bytes per row of statistics.  z = dropout(h) + inp is recomputed in the
// backward instead of being saved.  HBM-bound; one wave per row (H / 64 elements per lane in registers, two-pass
// variance), no LDS in the forward.  Weight gradients: per-wave register accumulation over a grid-stride row loop,
// per-workgroup partials, fixed-order second stage - bitwise reproducible.
// Dropout bits come from Philox4x32-10 keyed by (seed, offset) with the element-quad index as counter: the mask is a
// pure function of (seed, offset, row, column) - not torch's stream, the same Bernoulli(1 - p) law.
#include "glr_common.h"

namespace {

constexpr int LN_NT = 256;

struct U4 { unsigned x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ void unpack4(const uint2 u, float (&f)[4]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ uint2 pack4(const float (&f)[4]) {
  return make_uint2((unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16), (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16));
}

// element e = 4 * (lane + 64 i) + c of a row; its dropout bit is bit `lane` of mask word (i * 4 + c) of the row
template <int NQ>
__global__ void __launch_bounds__(LN_NT) k_drop_add_ln_fwd(const unsigned short* __restrict__ h, const float* __restrict__ inp,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           long long R, float eps, float p_drop, unsigned seed_lo, unsigned seed_hi,
                                                           unsigned off_lo, unsigned off_hi, const unsigned long long* __restrict__ rng,
                                                           float* __restrict__ out32, unsigned short* __restrict__ out16,
                                                           float* __restrict__ stats, unsigned long long* __restrict__ mask) {
  constexpr int H = NQ * 256;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (LN_NT / 64) + (threadIdx.x >> 6);
  if (row >= R) return;
  if (rng != nullptr) {        // key from device memory (hipGraph replays: the host rewrites the cell before each replay)
    const unsigned long long s = rng[0], o = rng[1] + (((unsigned long long)off_hi << 32) | off_lo);
    seed_lo = (unsigned)s; seed_hi = (unsigned)(s >> 32); off_lo = (unsigned)o; off_hi = (unsigned)(o >> 32);
  }
  const bool drop = p_drop > 0.f;
  const unsigned thr = drop ? (unsigned)fminf(p_drop * 4294967296.f, 4294967040.f) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  float z[NQ][4];
  unsigned long long myword = 0ull;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int quad = lane + 64 * i;
    float hv[4];
    unpack4(*reinterpret_cast<const uint2*>(h + row * H + 4 * quad), hv);
    const float4 iv = *reinterpret_cast<const float4*>(inp + row * H + 4 * quad);
    const float ivv[4] = {iv.x, iv.y, iv.z, iv.w};
    if (drop) {
      const unsigned long long ctr = (unsigned long long)row * (H / 4) + quad;
      const U4 rnd = philox4x32((unsigned)ctr, (unsigned)(ctr >> 32), off_lo, off_hi, seed_lo, seed_hi);
      const unsigned rr[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool keep = rr[c] >= thr;
        const unsigned long long word = __ballot(keep);
        if (lane == i * 4 + c) myword = word;
        z[i][c] = (keep ? hv[c] * inv_keep : 0.f) + ivv[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) z[i][c] = hv[c] + ivv[c];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NQ; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) s += z[i][c];
  const float mean = wave_sum_f(s) * (1.f / H);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NQ; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float d = z[i][c] - mean; v = __builtin_fmaf(d, d, v); }
  const float rstd = rsqrtf(wave_sum_f(v) * (1.f / H) + eps);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int quad = lane + 64 * i;
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * quad);
    const float4 b = *reinterpret_cast<const float4*>(beta + 4 * quad);
    float y[4];
    y[0] = (z[i][0] - mean) * rstd * g.x + b.x;
    y[1] = (z[i][1] - mean) * rstd * g.y + b.y;
    y[2] = (z[i][2] - mean) * rstd * g.z + b.z;
    y[3] = (z[i][3] - mean) * rstd * g.w + b.w;
    *reinterpret_cast<float4*>(out32 + row * H + 4 * quad) = make_float4(y[0], y[1], y[2], y[3]);
    *reinterpret_cast<uint2*>(out16 + row * H + 4 * quad) = pack4(y);
  }
  if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
  if (drop && lane < NQ * 4) mask[row * (NQ * 4) + lane] = myword;
}

// backward: dz = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)); d_inp = dz; d_h = dz * keep / (1 - p);
// dgamma += dy xhat, dbeta += dy, dhsum += d_h (per-wave registers -> per-workgroup partial [3][H][n_part]); dhsum is
// the bias gradient of the dense Linear that produced h (its column sums would otherwise be one more pass over d_h)
template <int NQ>
__global__ void __launch_bounds__(LN_NT) k_drop_add_ln_bwd(const float* __restrict__ dy32, const unsigned short* __restrict__ dy16,
                                                           const unsigned short* __restrict__ h, const float* __restrict__ inp,
                                                           const float* __restrict__ gamma, const float* __restrict__ stats,
                                                           const unsigned long long* __restrict__ mask, long long R, float p_drop,
                                                           float* __restrict__ d_inp, unsigned short* __restrict__ d_h,
                                                           float* __restrict__ part) {
  constexpr int H = NQ * 256;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  float g[NQ][4], dg[NQ][4], db[NQ][4], dhs[NQ][4];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const float4 gv = *reinterpret_cast<const float4*>(gamma + 4 * (lane + 64 * i));
    g[i][0] = gv.x; g[i][1] = gv.y; g[i][2] = gv.z; g[i][3] = gv.w;
#pragma unroll
    for (int c = 0; c < 4; ++c) { dg[i][c] = 0.f; db[i][c] = 0.f; dhs[i][c] = 0.f; }
  }
  for (long long row = (long long)blockIdx.x * (LN_NT / 64) + wv; row < R; row += (long long)gridDim.x * (LN_NT / 64)) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    unsigned long long myword = 0ull;
    if (drop && lane < NQ * 4) myword = mask[row * (NQ * 4) + lane];
    float xh[NQ][4], gd[NQ][4];
    unsigned keepbits[NQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int quad = lane + 64 * i;
      float hv[4], dv[4] = {0.f, 0.f, 0.f, 0.f};
      unpack4(*reinterpret_cast<const uint2*>(h + row * H + 4 * quad), hv);
      const float4 iv = *reinterpret_cast<const float4*>(inp + row * H + 4 * quad);
      const float ivv[4] = {iv.x, iv.y, iv.z, iv.w};
      if (dy16 != nullptr) unpack4(*reinterpret_cast<const uint2*>(dy16 + row * H + 4 * quad), dv);
      if (dy32 != nullptr) {
        const float4 d4 = *reinterpret_cast<const float4*>(dy32 + row * H + 4 * quad);
        dv[0] += d4.x; dv[1] += d4.y; dv[2] += d4.z; dv[3] += d4.w;
      }
      unsigned kb = 0xfu;
      if (drop) {
        kb = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned long long word = __shfl(myword, i * 4 + c, 64);
          kb |= (unsigned)((word >> lane) & 1ull) << c;
        }
      }
      keepbits[i] = kb;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float z = (((kb >> c) & 1u) ? hv[c] * inv_keep : 0.f) + ivv[c];
        const float x = (z - mean) * rstd;
        xh[i][c] = x;
        gd[i][c] = g[i][c] * dv[c];
        s1 += gd[i][c];
        s2 = __builtin_fmaf(gd[i][c], x, s2);
        dg[i][c] = __builtin_fmaf(dv[c], x, dg[i][c]);
        db[i][c] += dv[c];
      }
    }
    const float m1 = wave_sum_f(s1) * (1.f / H), m2 = wave_sum_f(s2) * (1.f / H);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int quad = lane + 64 * i;
      float dz[4], dh[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dz[c] = rstd * (gd[i][c] - m1 - xh[i][c] * m2);
        dh[c] = ((keepbits[i] >> c) & 1u) ? dz[c] * inv_keep : 0.f;
        dhs[i][c] += dh[c];
      }
      *reinterpret_cast<float4*>(d_inp + row * H + 4 * quad) = make_float4(dz[0], dz[1], dz[2], dz[3]);
      *reinterpret_cast<uint2*>(d_h + row * H + 4 * quad) = pack4(dh);
    }
  }
  // the workgroup's four waves -> one partial per column (fixed order)
  __shared__ float red[3][LN_NT / 64][H];
#pragma unroll
  for (int i = 0; i < NQ; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      red[0][wv][4 * (lane + 64 * i) + c] = dg[i][c];
      red[1][wv][4 * (lane + 64 * i) + c] = db[i][c];
      red[2][wv][4 * (lane + 64 * i) + c] = dhs[i][c];
    }
  __syncthreads();
  for (int col = threadIdx.x; col < H; col += LN_NT) {
    float a = 0.f, b = 0.f, e = 0.f;
#pragma unroll
    for (int k = 0; k < LN_NT / 64; ++k) { a += red[0][k][col]; b += red[1][k][col]; e += red[2][k][col]; }
    part[(size_t)col * gridDim.x + blockIdx.x] = a;
    part[((size_t)H + col) * gridDim.x + blockIdx.x] = b;
    part[((size_t)2 * H + col) * gridDim.x + blockIdx.x] = e;
  }
}

// second stage: one wave per column sums its partials of the three quantities in a fixed order
__global__ void __launch_bounds__(LN_NT) k_ln_finish(const float* __restrict__ part, int n_part, int H, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, void* __restrict__ dhsum, int dhsum_bf16) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * (LN_NT / 64) + (threadIdx.x >> 6);
  if (col >= H) return;
  float a = 0.f, b = 0.f, e = 0.f;
  for (int k = lane; k < n_part; k += 64) {
    a += part[(size_t)col * n_part + k];
    b += part[((size_t)H + col) * n_part + k];
    if (dhsum) e += part[((size_t)2 * H + col) * n_part + k];
  }
  a = wave_sum_f(a);
  b = wave_sum_f(b);
  e = wave_sum_f(e);
  if (lane == 0) {
    dgamma[col] = a;
    dbeta[col] = b;
    if (dhsum) {
      if (dhsum_bf16) reinterpret_cast<unsigned short*>(dhsum)[col] = f2bf(e);
      else reinterpret_cast<float*>(dhsum)[col] = e;
    }
  }
}

// Column sums of a bf16 [R, C] matrix (bias gradient of a Linear: sum over tokens of dy): thread = 8 columns x every 8th
// row of its workgroup's row slab, LDS reduction over the 8 row lanes, partial [C][n_part], fixed-order second stage.
// HBM-bound, 2 bytes per element; aten's generic reduce_kernel runs these shapes at 0.7 - 2.7 TB/s.
constexpr int CS_COLS = 256;           // columns per workgroup (32 threads x 8)
__global__ void __launch_bounds__(LN_NT) k_colsum(const unsigned short* __restrict__ x, long long R, int C, long long rows_per,
                                                   float* __restrict__ part) {
  const int ct = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col0 = blockIdx.x * CS_COLS + ct * 8;
  const long long r0 = (long long)blockIdx.y * rows_per;
  const long long r1 = r0 + rows_per < R ? r0 + rows_per : R;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned short* px = x + col0;
  long long r = r0 + rl;
  for (; r + 24 < r1; r += 32) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(px + (r + 8 * u) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] += __uint_as_float(v[u].x << 16); acc[1] += __uint_as_float(v[u].x & 0xffff0000u);
      acc[2] += __uint_as_float(v[u].y << 16); acc[3] += __uint_as_float(v[u].y & 0xffff0000u);
      acc[4] += __uint_as_float(v[u].z << 16); acc[5] += __uint_as_float(v[u].z & 0xffff0000u);
      acc[6] += __uint_as_float(v[u].w << 16); acc[7] += __uint_as_float(v[u].w & 0xffff0000u);
    }
  }
  for (; r < r1; r += 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(px + r * C);
    acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
    acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
    acc[4] += __uint_as_float(v.z << 16); acc[5] += __uint_as_float(v.z & 0xffff0000u);
    acc[6] += __uint_as_float(v.w << 16); acc[7] += __uint_as_float(v.w & 0xffff0000u);
  }
  __shared__ float red[8][CS_COLS + 8];
#pragma unroll
  for (int c = 0; c < 8; ++c) red[rl][ct * 8 + c] = acc[c];
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
  part[(size_t)(blockIdx.x * CS_COLS + threadIdx.x) * gridDim.y + blockIdx.y] = s;
}

__global__ void __launch_bounds__(LN_NT) k_colsum_finish(const float* __restrict__ part, int n_part, int C, void* __restrict__ out,
                                                          int out_bf16) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * (LN_NT / 64) + (threadIdx.x >> 6);
  if (col >= C) return;
  float a = 0.f;
  for (int k = lane; k < n_part; k += 64) a += part[(size_t)col * n_part + k];
  a = wave_sum_f(a);
  if (lane == 0) {
    if (out_bf16) reinterpret_cast<unsigned short*>(out)[col] = f2bf(a);
    else reinterpret_cast<float*>(out)[col] = a;
  }
}
int colsum_parts(long long R, int C) {
  long long np = 1024 / (C / CS_COLS);                 // ~4 workgroups per CU over the whole grid
  if (np < 1) np = 1;
  const long long max_np = (R + 31) / 32;             // at least 32 rows per workgroup
  return (int)(np < max_np ? np : max_np);
}
bool colsum_shape_ok(long long R, int C) { return R > 0 && C >= CS_COLS && C % CS_COLS == 0 && C <= 16384; }

constexpr int LN_MAX_PART = 1024;      // workgroups of the backward pass (4 per CU: all resident at once)
int ln_parts(long long R) {
  const long long blocks = (R + LN_NT / 64 - 1) / (LN_NT / 64);
  return (int)(blocks < LN_MAX_PART ? blocks : LN_MAX_PART);
}
bool ln_shape_ok(long long R, int H) { return R > 0 && H >= 256 && H <= 1024 && H % 256 == 0; }

}  // namespace

extern "C" int glr_ln_workspace_floats(long long R, int H) { return ln_shape_ok(R, H) ? ln_parts(R) * 3 * H : 0; }

extern "C" int glr_colsum_workspace_floats(long long R, int C) { return colsum_shape_ok(R, C) ? colsum_parts(R, C) * C : 0; }

extern "C" int glr_colsum_bf16(const void* x16, long long R, int C, float* workspace, void* out, int out_bf16, void* stream) {
  if (!x16 || !workspace || !out || !colsum_shape_ok(R, C)) return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int np = colsum_parts(R, C);
  const long long rows_per = (R + np - 1) / np;
  hipLaunchKernelGGL(k_colsum, dim3(C / CS_COLS, np), dim3(LN_NT), 0, st, (const unsigned short*)x16, R, C, rows_per, workspace);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_colsum_finish, dim3((C + LN_NT / 64 - 1) / (LN_NT / 64)), dim3(LN_NT), 0, st, workspace, np, C, out, out_bf16);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_drop_add_ln_fwd(const void* h16, const float* inp32, const float* gamma, const float* beta, long long R, int H,
                                   float eps, float p_drop, unsigned long long seed, unsigned long long offset,
                                   const unsigned long long* rng_cell, float* out32, void* out16, float* stats,
                                   unsigned long long* mask, void* stream) {
  if (!h16 || !inp32 || !gamma || !beta || !out32 || !out16 || !stats || !ln_shape_ok(R, H)) return GLR_EINVAL;
  if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && !mask)) return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((R + LN_NT / 64 - 1) / (LN_NT / 64));
#define GLR_LN_FWD(NQ)                                                                                                      \
  hipLaunchKernelGGL((k_drop_add_ln_fwd<NQ>), dim3(grid), dim3(LN_NT), 0, st, (const unsigned short*)h16, inp32, gamma, beta, R, \
                     eps, p_drop, (unsigned)seed, (unsigned)(seed >> 32), (unsigned)offset, (unsigned)(offset >> 32), rng_cell, out32, (unsigned short*)out16, stats, mask)
  switch (H / 256) {
    case 1: GLR_LN_FWD(1); break;
    case 2: GLR_LN_FWD(2); break;
    case 3: GLR_LN_FWD(3); break;
    default: GLR_LN_FWD(4); break;
  }
#undef GLR_LN_FWD
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_drop_add_ln_bwd(const float* dy32, const void* dy16, const void* h16, const float* inp32, const float* gamma,
                                   const float* stats, const unsigned long long* mask, long long R, int H, float p_drop,
                                   float* d_inp32, void* d_h16, float* workspace, float* dgamma, float* dbeta, void* dhsum,
                                   int dhsum_bf16, void* stream) {
  if ((!dy32 && !dy16) || !h16 || !inp32 || !gamma || !stats || !d_inp32 || !d_h16 || !workspace || !dgamma || !dbeta ||
      !ln_shape_ok(R, H))
    return GLR_EINVAL;
  if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && !mask)) return GLR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int np = ln_parts(R);
#define GLR_LN_BWD(NQ)                                                                                                      \
  hipLaunchKernelGGL((k_drop_add_ln_bwd<NQ>), dim3(np), dim3(LN_NT), 0, st, dy32, (const unsigned short*)dy16,               \
                     (const unsigned short*)h16, inp32, gamma, stats, mask, R, p_drop, d_inp32, (unsigned short*)d_h16, workspace)
  switch (H / 256) {
    case 1: GLR_LN_BWD(1); break;
    case 2: GLR_LN_BWD(2); break;
    case 3: GLR_LN_BWD(3); break;
    default: GLR_LN_BWD(4); break;
  }
#undef GLR_LN_BWD
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_ln_finish, dim3((H + LN_NT / 64 - 1) / (LN_NT / 64)), dim3(LN_NT), 0, st, workspace, np, H, dgamma, dbeta, dhsum,
                     dhsum_bf16);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
