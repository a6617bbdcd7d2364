// K7: image half of the collate function on the GPU.
//
// Reference: /root/reference/gloria/datasets/mimic_for_gloria.py
//   :36-42   normalize / original_tensor_to_numpy_image  - (x - min) / (max - min) * 255 in fp32, truncated to uint8
//   :136-181 _resize_img   - long side -> `scale` with cv2.INTER_AREA (aspect kept), short side zero padded
//   :120-133 process_img   - PIL "L" -> "RGB" (3 equal channels), transform, stack
// and /root/reference/gloria/builder.py:159-201 (RandomCrop / CenterCrop, ToTensor, Normalize(0.5, 0.5)).
//
// Two kernels, both HBM-bound byte work (no MFMA):
//   k_image_minmax   one read of the ragged source batch -> per-image min / max (order-preserving uint keys, atomics)
//   k_collate        one thread per OUTPUT pixel of the crop window: it evaluates cv2's INTER_AREA cell of that
//                    pixel straight from the source (quantising each source pixel to uint8 on the fly), so the
//                    uint8 image, the resized image and the padded 256x256 frame are never materialised and only
//                    the source pixels under the crop window are read.  Every fp32 / fp64 operation is written
//                    with explicit rounding (no FMA contraction) in the order OpenCV's scalar code performs it,
//                    which makes the result bit-identical to the CPU restatement (oracle/collate_oracle.py).
// The upscaling branch of cv2.INTER_AREA is not built: the host mirror rejects images whose long side is < scale.

#include "glr_common.h"

namespace {

constexpr int MM_BLOCKS = 64;
constexpr double GLR_DBL_EPS = 2.220446049250313e-16;

__device__ __forceinline__ unsigned f2key(float f) {
  unsigned b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(b);
}

template <typename T> struct Src;
template <> struct Src<float> { static constexpr int VEC = 4; };
template <> struct Src<short> { static constexpr int VEC = 8; };
template <> struct Src<unsigned char> { static constexpr int VEC = 16; };

__global__ void k_minmax_init(unsigned* state, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    state[2 * i] = 0xFFFFFFFFu;
    state[2 * i + 1] = 0u;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_image_minmax(const unsigned char* __restrict__ src,
                                                      const long long* __restrict__ offset,
                                                      const int* __restrict__ desc, unsigned* __restrict__ state) {
  constexpr int VEC = Src<T>::VEC;
  const int b = blockIdx.y;
  const long long off = offset[b];
  const long long n = (long long)desc[8 * b] * desc[8 * b + 1];
  const T* p = reinterpret_cast<const T*>(src + off);
  float mn = INFINITY, mx = -INFINITY;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nthr = (long long)gridDim.x * blockDim.x;
  long long done = 0;
  if ((off & 15) == 0) {                       // 16-byte vector loads over the aligned body
    const long long nv = n / VEC;
    const uint4* pv = reinterpret_cast<const uint4*>(p);
    for (long long i = tid; i < nv; i += nthr) {
      uint4 raw = pv[i];
      const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = (float)e[j];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
    }
    done = nv * VEC;
  }
  for (long long i = done + tid; i < n; i += nthr) {
    float v = (float)p[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0 && mn <= mx) {
    atomicMin(&state[2 * b], f2key(mn));
    atomicMax(&state[2 * b + 1], f2key(mx));
  }
}

// ordered taps of one destination index along one axis (OpenCV computeResizeAreaTab, evaluated per index)
struct Taps {
  int s1, s2;                  // full-weight source range [s1, s2)
  float a_first, a_mid, a_last;
  bool has_first, has_last;    // partial taps at s1 - 1 and s2
};

// Everything below must round every operation separately (OpenCV's scalar code does): the HIP `__f*_rn` header
// intrinsics still carry the header's contraction flags after inlining, so plain operators are used under this pragma.
#pragma clang fp contract(off)
__device__ __forceinline__ float mul_(float a, float b) { return a * b; }
__device__ __forceinline__ float add_(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_(float a, float b) { return a - b; }
__device__ __forceinline__ float div_(float a, float b) { return a / b; }
__device__ __forceinline__ double mul_(double a, double b) { return a * b; }
__device__ __forceinline__ double add_(double a, double b) { return a + b; }
__device__ __forceinline__ double sub_(double a, double b) { return a - b; }
__device__ __forceinline__ double div_(double a, double b) { return a / b; }

__device__ __forceinline__ Taps make_taps(int d, int ssize, double scale) {
  Taps t;
  const double fs1 = mul_((double)d, scale);
  const double fs2 = add_(fs1, scale);
  const double cell = fmin(scale, sub_((double)ssize, fs1));
  int s1 = (int)ceil(fs1), s2 = (int)floor(fs2);
  s2 = min(s2, ssize - 1);
  s1 = min(s1, s2);
  t.s1 = s1;
  t.s2 = s2;
  const double d1 = sub_((double)s1, fs1);
  const double d2 = sub_(fs2, (double)s2);
  t.has_first = d1 > 1e-3;
  t.has_last = d2 > 1e-3;
  t.a_first = (float)div_(d1, cell);
  t.a_mid = (float)div_(1.0, cell);
  t.a_last = (float)div_(fmin(fmin(d2, 1.0), cell), cell);
  return t;
}

template <typename T, bool MINMAX>
struct Quant {
  float mn, range;
  __device__ __forceinline__ float operator()(T raw) const {
    if (!MINMAX) return (float)raw;                       // already the 8-bit image
    float y = mul_(div_(sub_((float)raw, mn), range), 255.0f);
    int q = (int)y;                                       // C cast: truncation (NaN when max == min -> 0)
    q = y != y ? 0 : min(max(q, 0), 255);
    return (float)q;
  }
};

template <typename T, bool MINMAX>
__global__ __launch_bounds__(256) void k_collate(const unsigned char* __restrict__ src,
                                                 const long long* __restrict__ offset, const int* __restrict__ desc,
                                                 const unsigned* __restrict__ state, int crop,
                                                 float* __restrict__ out) {
  const int b = blockIdx.y;
  const int oy = blockIdx.x, ox = threadIdx.x;
  if (ox >= crop) return;
  const int* ds = desc + 8 * b;
  const int H = ds[0], W = ds[1], dH = ds[2], dW = ds[3];
  const int dy = ds[6] + oy - ds[4], dx = ds[7] + ox - ds[5];     // position inside the resized image
  const T* S = reinterpret_cast<const T*>(src + offset[b]);
  Quant<T, MINMAX> q;
  q.mn = 0.f;
  q.range = 1.f;
  if (MINMAX) {
    q.mn = key2f(state[2 * b]);
    q.range = sub_(key2f(state[2 * b + 1]), q.mn);
  }
  int v = 0;                                                       // zero padding outside the resized image
  if (dy >= 0 && dy < dH && dx >= 0 && dx < dW) {
    if (dH == H && dW == W) {
      v = (int)q(S[(long long)dy * W + dx]);                       // cv2.resize copies when the sizes are equal
    } else {
      const double scale_x = div_(1.0, div_((double)dW, (double)W));
      const double scale_y = div_(1.0, div_((double)dH, (double)H));
      const int ix = __double2int_rn(scale_x), iy = __double2int_rn(scale_y);
      if (fabs(scale_x - ix) < GLR_DBL_EPS && fabs(scale_y - iy) < GLR_DBL_EPS) {
        int sum = 0;                                               // integer scale: exact box sums
        for (int r = 0; r < iy; ++r) {
          const T* row = S + (long long)(dy * iy + r) * W + dx * ix;
          for (int c = 0; c < ix; ++c) sum += (int)q(row[c]);
        }
        if (ix == 2 && iy == 2) {
          v = (sum + 2) >> 2;                                      // OpenCV's 8u 2x2 SIMD rounding
        } else {
          const float inv = div_(1.0f, (float)(ix * iy));
          v = __float2int_rn(mul_((float)sum, inv));
        }
      } else {
        const Taps tx = make_taps(dx, W, scale_x);
        const Taps ty = make_taps(dy, H, scale_y);
        float total = 0.f;
        bool first_row = true;
        const int r0 = ty.has_first ? ty.s1 - 1 : ty.s1;
        const int r1 = ty.has_last ? ty.s2 + 1 : ty.s2;
        for (int sy = r0; sy < r1; ++sy) {
          const float beta = sy < ty.s1 ? ty.a_first : (sy < ty.s2 ? ty.a_mid : ty.a_last);
          const T* row = S + (long long)sy * W;
          float buf = 0.f;
          if (tx.has_first) buf = add_(buf, mul_(q(row[tx.s1 - 1]), tx.a_first));
          for (int sx = tx.s1; sx < tx.s2; ++sx) buf = add_(buf, mul_(q(row[sx]), tx.a_mid));
          if (tx.has_last) buf = add_(buf, mul_(q(row[tx.s2]), tx.a_last));
          const float term = mul_(beta, buf);
          total = first_row ? term : add_(total, term);
          first_row = false;
        }
        v = __float2int_rn(total);                                 // saturate_cast<uchar>: round half to even
      }
      v = min(max(v, 0), 255);
    }
  }
  const float t = div_((float)v, 255.0f);                      // ToTensor
  const float o = div_(sub_(t, 0.5f), 0.5f);              // Normalize(0.5, 0.5)
  const long long plane = (long long)crop * crop;
  float* dst = out + (long long)b * 3 * plane + (long long)oy * crop + ox;
  dst[0] = o;
  dst[plane] = o;
  dst[2 * plane] = o;
}

}  // namespace

extern "C" int glr_image_minmax(const void* src, const int64_t* offset, const int32_t* desc, int B, int src_dtype,
                                uint32_t* state, void* stream) {
  if (!src || !offset || !desc || !state || B <= 0) return GLR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_minmax_init, dim3((B + 255) / 256), dim3(256), 0, st, state, B);
  const unsigned char* s = static_cast<const unsigned char*>(src);
  const long long* off = reinterpret_cast<const long long*>(offset);
  dim3 grid(MM_BLOCKS, B);
  switch (src_dtype) {
    case GLR_SRC_U8: hipLaunchKernelGGL(k_image_minmax<unsigned char>, grid, dim3(256), 0, st, s, off, desc, state); break;
    case GLR_SRC_I16: hipLaunchKernelGGL(k_image_minmax<short>, grid, dim3(256), 0, st, s, off, desc, state); break;
    case GLR_SRC_F32: hipLaunchKernelGGL(k_image_minmax<float>, grid, dim3(256), 0, st, s, off, desc, state); break;
    default: return GLR_EDTYPE;
  }
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_collate_images(const void* src, const int64_t* offset, const int32_t* desc,
                                  const uint32_t* state, int B, int src_dtype, int crop, float* out, void* stream) {
  if (!src || !offset || !desc || !out || B <= 0 || crop <= 0 || crop > 256) return GLR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned char* s = static_cast<const unsigned char*>(src);
  const long long* off = reinterpret_cast<const long long*>(offset);
  dim3 grid(crop, B);
#define GLR_COLLATE(T)                                                                                         \
  do {                                                                                                         \
    if (state) hipLaunchKernelGGL((k_collate<T, true>), grid, dim3(256), 0, st, s, off, desc, state, crop, out); \
    else hipLaunchKernelGGL((k_collate<T, false>), grid, dim3(256), 0, st, s, off, desc, state, crop, out);      \
  } while (0)
  switch (src_dtype) {
    case GLR_SRC_U8: GLR_COLLATE(unsigned char); break;
    case GLR_SRC_I16: GLR_COLLATE(short); break;
    case GLR_SRC_F32: GLR_COLLATE(float); break;
    default: return GLR_EDTYPE;
  }
#undef GLR_COLLATE
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
