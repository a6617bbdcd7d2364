// Parameter-sized work of one optimisation step as THREE launches over flat buffers (the
// reference leaves it to Lightning: torch.optim.Adam(betas=(0.5, 0.999)) + gradient_clip_val 0.25 under native AMP,
// /root/reference/gloria/builder.py:84-87, run.py:172-207):
//   glr_sumsq_partial   per-workgroup sums of squares of a flat gradient buffer (bf16 or fp32), fixed order
//   glr_clip_coef       total norm over all partials -> { norm, min(1, max_norm / (norm + 1e-6)) } on the device
//                       (torch.nn.utils.clip_grad_norm_'s coefficient, no host round trip)
//   glr_adam_step       Adam (torch semantics: L2 weight decay added to the gradient, bias corrections) on the fp32
//                       MASTER weights with the clipped gradient, writing the bf16 SHADOW the forward reads in the
//                       same pass: no per-step autocast weight casts, no gradient casts, one launch per dtype group
//                       instead of ~600 small kernels.  HBM-bound: 4 + 4 + 4 + 2 read, 4 + 4 + 4 + 2 written per
//                       shadowed parameter.
#include "glr_common.h"

namespace {

constexpr int OPT_NT = 256;
constexpr int OPT_VEC = 8;          // elements per thread per trip

__device__ __forceinline__ void load8(const void* p, int dtype, size_t i, float (&v)[8]) {
  if (dtype == GLR_BF16) {
    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + i);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = __uint_as_float(w[k] << 16);
      v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
  } else {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}

// partial[blockIdx.x] = sum of squares of this block's contiguous share (n is a multiple of 8: buffers are padded)
__global__ void __launch_bounds__(OPT_NT) k_sumsq(const void* __restrict__ x, int dtype, size_t n, float* __restrict__ partial) {
  __shared__ float red[OPT_NT / 64];
  const size_t per = ((n / OPT_VEC + gridDim.x - 1) / gridDim.x) * OPT_VEC;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  for (size_t i = lo + (size_t)threadIdx.x * OPT_VEC; i < hi; i += (size_t)OPT_NT * OPT_VEC) {
    float v[8];
    load8(x, dtype, i, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) s = __builtin_fmaf(v[k], v[k], s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = sqrt(sum partial), out[1] = clip coefficient (1 when max_norm <= 0)
__global__ void __launch_bounds__(OPT_NT) k_clip_coef(const float* __restrict__ partial, int n, float max_norm,
                                                      float* __restrict__ out) {
  __shared__ float red[OPT_NT / 64];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += OPT_NT) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    out[0] = norm;
    out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  }
}

__global__ void __launch_bounds__(OPT_NT) k_adam(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                 const void* __restrict__ grad, int grad_dtype,
                                                 unsigned short* __restrict__ shadow, size_t n, float lr, float b1, float b2,
                                                 float eps, float wd, float bc1, float bc2_sqrt,
                                                 const float* __restrict__ clip) {
  const size_t i = ((size_t)blockIdx.x * OPT_NT + threadIdx.x) * OPT_VEC;
  if (i >= n) return;
  const float c = clip ? clip[1] : 1.f;
  float g[8], p[8], a[8], b[8];
  load8(grad, grad_dtype, i, g);
  load8(master, GLR_F32, i, p);
  load8(m, GLR_F32, i, a);
  load8(v, GLR_F32, i, b);
  const float step = lr / bc1;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float gk = __builtin_fmaf(wd, p[k], g[k] * c);             // torch.optim.Adam: grad + weight_decay * param
    a[k] = __builtin_fmaf(b1, a[k], (1.f - b1) * gk);
    b[k] = __builtin_fmaf(b2, b[k], (1.f - b2) * gk * gk);
    const float denom = sqrtf(b[k]) / bc2_sqrt + eps;
    p[k] = p[k] - step * (a[k] / denom);
  }
  *reinterpret_cast<float4*>(master + i) = make_float4(p[0], p[1], p[2], p[3]);
  *reinterpret_cast<float4*>(master + i + 4) = make_float4(p[4], p[5], p[6], p[7]);
  *reinterpret_cast<float4*>(m + i) = make_float4(a[0], a[1], a[2], a[3]);
  *reinterpret_cast<float4*>(m + i + 4) = make_float4(a[4], a[5], a[6], a[7]);
  *reinterpret_cast<float4*>(v + i) = make_float4(b[0], b[1], b[2], b[3]);
  *reinterpret_cast<float4*>(v + i + 4) = make_float4(b[4], b[5], b[6], b[7]);
  if (shadow) {
    uint4 u;
    u.x = f2bf(p[0]) | ((unsigned)f2bf(p[1]) << 16);
    u.y = f2bf(p[2]) | ((unsigned)f2bf(p[3]) << 16);
    u.z = f2bf(p[4]) | ((unsigned)f2bf(p[5]) << 16);
    u.w = f2bf(p[6]) | ((unsigned)f2bf(p[7]) << 16);
    *reinterpret_cast<uint4*>(shadow + i) = u;
  }
}

// ---- pointer-table forms: gradients stay where autograd left them (one tensor per parameter, no accumulation
// into a flat buffer: that costs an add launch per parameter).  One workgroup per chunk of <= OPT_CHUNK elements.
//   chunk[c] = { parameter index, element offset inside the parameter, element offset in the flat buffers, count }
//   gptr[param] = device address of the parameter's gradient this step (0: no gradient -> the parameter is skipped,
//   like torch.optim.Adam skips a None gradient)
struct ChunkEnt { int param; int count; long long poff; long long foff; };

__device__ __forceinline__ float ld1(const void* p, int dtype, size_t i) {
  return dtype == GLR_BF16 ? bf2f(reinterpret_cast<const unsigned short*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}

__global__ void __launch_bounds__(OPT_NT) k_sumsq_mt(const ChunkEnt* __restrict__ chunk, const unsigned long long* __restrict__ gptr,
                                                     int dtype, float* __restrict__ partial) {
  __shared__ float red[OPT_NT / 64];
  const ChunkEnt e = chunk[blockIdx.x];
  const void* g = reinterpret_cast<const void*>(gptr[e.param]);
  float s = 0.f;
  if (g != nullptr) {
    const int nv = e.count & ~7;
    for (int i = threadIdx.x * OPT_VEC; i < nv; i += OPT_NT * OPT_VEC) {
      float v[8];
      load8(g, dtype, (size_t)e.poff + i, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) s = __builtin_fmaf(v[k], v[k], s);
    }
    for (int i = nv + threadIdx.x; i < e.count; i += OPT_NT) { const float x = ld1(g, dtype, (size_t)e.poff + i); s = __builtin_fmaf(x, x, s); }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(OPT_NT) k_adam_mt(const ChunkEnt* __restrict__ chunk, const unsigned long long* __restrict__ gptr,
                                                    int grad_dtype, float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v, unsigned short* __restrict__ shadow, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ clip) {
  const ChunkEnt e = chunk[blockIdx.x];
  const void* gp = reinterpret_cast<const void*>(gptr[e.param]);
  if (gp == nullptr) return;
  const float c = clip ? clip[1] : 1.f;
  const float step = lr / bc1;
  // chunks start 8-aligned in the flat buffers (parameters are padded to 8) and 8-aligned inside the parameter
  for (int i = threadIdx.x * OPT_VEC; i < e.count; i += OPT_NT * OPT_VEC) {
    const size_t f = (size_t)e.foff + i;
    float g[8], p[8], a[8], b[8];
    if (i + 8 <= e.count) {
      load8(gp, grad_dtype, (size_t)e.poff + i, g);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = i + k < e.count ? ld1(gp, grad_dtype, (size_t)e.poff + i + k) : 0.f;
    }
    load8(master, GLR_F32, f, p);
    load8(m, GLR_F32, f, a);
    load8(v, GLR_F32, f, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float gk = __builtin_fmaf(wd, p[k], g[k] * c);
      a[k] = __builtin_fmaf(b1, a[k], (1.f - b1) * gk);
      b[k] = __builtin_fmaf(b2, b[k], (1.f - b2) * gk * gk);
      p[k] = p[k] - step * (a[k] / (sqrtf(b[k]) / bc2_sqrt + eps));
    }
    *reinterpret_cast<float4*>(master + f) = make_float4(p[0], p[1], p[2], p[3]);
    *reinterpret_cast<float4*>(master + f + 4) = make_float4(p[4], p[5], p[6], p[7]);
    *reinterpret_cast<float4*>(m + f) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(m + f + 4) = make_float4(a[4], a[5], a[6], a[7]);
    *reinterpret_cast<float4*>(v + f) = make_float4(b[0], b[1], b[2], b[3]);
    *reinterpret_cast<float4*>(v + f + 4) = make_float4(b[4], b[5], b[6], b[7]);
    if (shadow) {
      uint4 u;
      u.x = f2bf(p[0]) | ((unsigned)f2bf(p[1]) << 16);
      u.y = f2bf(p[2]) | ((unsigned)f2bf(p[3]) << 16);
      u.z = f2bf(p[4]) | ((unsigned)f2bf(p[5]) << 16);
      u.w = f2bf(p[6]) | ((unsigned)f2bf(p[7]) << 16);
      *reinterpret_cast<uint4*>(shadow + f) = u;
    }
  }
}

// gather: the gradients of a bucket's parameters (one tensor each) -> their slots of the flat gradient buffer, one
// launch per bucket (data-parallel: the bucket is then all-reduced as one slice)
__global__ void __launch_bounds__(OPT_NT) k_gather_mt(const ChunkEnt* __restrict__ chunk, const unsigned long long* __restrict__ gptr,
                                                      int dtype, void* __restrict__ flat) {
  const ChunkEnt e = chunk[blockIdx.x];
  const void* g = reinterpret_cast<const void*>(gptr[e.param]);
  if (g == nullptr) return;                        // the slot keeps the zeros of zero_grad
  const int esz = dtype == GLR_BF16 ? 2 : 4, per16 = 16 / esz;
  const unsigned char* src = reinterpret_cast<const unsigned char*>(g) + (size_t)e.poff * esz;
  unsigned char* dst = reinterpret_cast<unsigned char*>(flat) + (size_t)e.foff * esz;
  const int nv = e.count / per16;
  for (int i = threadIdx.x; i < nv; i += OPT_NT)
    reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  for (int i = nv * per16 + threadIdx.x; i < e.count; i += OPT_NT) {
    if (esz == 2) reinterpret_cast<unsigned short*>(dst)[i] = reinterpret_cast<const unsigned short*>(src)[i];
    else reinterpret_cast<float*>(dst)[i] = reinterpret_cast<const float*>(src)[i];
  }
}

}  // namespace

extern "C" int glr_gather_mt(const void* chunk_table, int n_chunks, const uint64_t* grad_ptrs, int dtype, void* flat,
                             void* stream) {
  if (!chunk_table || !grad_ptrs || !flat || n_chunks <= 0) return GLR_EINVAL;
  if (dtype != GLR_F32 && dtype != GLR_BF16) return GLR_EDTYPE;
  hipLaunchKernelGGL(k_gather_mt, dim3(n_chunks), dim3(OPT_NT), 0, (hipStream_t)stream, (const ChunkEnt*)chunk_table,
                     (const unsigned long long*)grad_ptrs, dtype, flat);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_sumsq_mt(const void* chunk_table, int n_chunks, const uint64_t* grad_ptrs, int dtype, float* partial,
                            void* stream) {
  if (!chunk_table || !grad_ptrs || !partial || n_chunks <= 0) return GLR_EINVAL;
  if (dtype != GLR_F32 && dtype != GLR_BF16) return GLR_EDTYPE;
  hipLaunchKernelGGL(k_sumsq_mt, dim3(n_chunks), dim3(OPT_NT), 0, (hipStream_t)stream, (const ChunkEnt*)chunk_table,
                     (const unsigned long long*)grad_ptrs, dtype, partial);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_adam_step_mt(const void* chunk_table, int n_chunks, const uint64_t* grad_ptrs, int grad_dtype,
                                float* master, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, const float* clip, void* stream) {
  if (!chunk_table || !grad_ptrs || !master || !exp_avg || !exp_avg_sq || n_chunks <= 0 || step < 1) return GLR_EINVAL;
  if (grad_dtype != GLR_F32 && grad_dtype != GLR_BF16) return GLR_EDTYPE;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(k_adam_mt, dim3(n_chunks), dim3(OPT_NT), 0, (hipStream_t)stream, (const ChunkEnt*)chunk_table,
                     (const unsigned long long*)grad_ptrs, grad_dtype, master, exp_avg, exp_avg_sq,
                     (unsigned short*)shadow_bf16, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), clip);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_sumsq_blocks(long long n) {
  if (n <= 0) return GLR_EINVAL;
  const long long b = (n / OPT_VEC + OPT_NT * 8 - 1) / (OPT_NT * 8);       // >= 8 trips per thread
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int glr_sumsq_partial(const void* x, int dtype, long long n, float* partial, void* stream) {
  if (!x || !partial || n <= 0 || n % OPT_VEC != 0) return GLR_EINVAL;
  if (dtype != GLR_F32 && dtype != GLR_BF16) return GLR_EDTYPE;
  hipLaunchKernelGGL(k_sumsq, dim3(glr_sumsq_blocks(n)), dim3(OPT_NT), 0, (hipStream_t)stream, x, dtype, (size_t)n, partial);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_clip_coef(const float* partial, int n_partial, float max_norm, float* out, void* stream) {
  if (!partial || !out || n_partial <= 0) return GLR_EINVAL;
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(OPT_NT), 0, (hipStream_t)stream, partial, n_partial, max_norm, out);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_adam_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_dtype,
                             void* shadow_bf16, long long n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, const float* clip, void* stream) {
  if (!master || !exp_avg || !exp_avg_sq || !grad || n <= 0 || n % OPT_VEC != 0 || step < 1) return GLR_EINVAL;
  if (grad_dtype != GLR_F32 && grad_dtype != GLR_BF16) return GLR_EDTYPE;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const size_t threads = (size_t)n / OPT_VEC;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((threads + OPT_NT - 1) / OPT_NT)), dim3(OPT_NT), 0, (hipStream_t)stream,
                     master, exp_avg, exp_avg_sq, grad, grad_dtype, (unsigned short*)shadow_bf16, (size_t)n, lr, beta1,
                     beta2, eps, weight_decay, bc1, sqrtf(bc2), clip);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
