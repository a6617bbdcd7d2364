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
//   k_collate        one workgroup per output row of the crop window, one thread per OUTPUT pixel: the source rows
//                    under that output row are staged through LDS (16-byte coalesced loads, quantised to uint8 on
//                    the way) and each thread evaluates cv2's INTER_AREA cell of its pixel from LDS, so the
//                    uint8 image, the resized image and the padded 256x256 frame are never materialised and only
//                    the source pixels under the crop window are read.  Every fp32 / fp64 operation is written
//                    with explicit rounding (no FMA contraction) in the order OpenCV's scalar code performs it,
//                    which makes the result bit-identical to the CPU restatement (oracle/collate_oracle.py).
// Images whose long side is below `scale` are ENLARGED: cv::resize then emulates INTER_AREA with its fixed-point bilinear
// code and area-style coordinates (mode 3 below; restated in oracle/collate_oracle.py resize_area_up_u8).
//
// Random transforms of builder.py:167-186 (RandomHorizontalFlip, RandomAffine, ColorJitter; torchvision 0.8.2 on PIL
// images) as device passes over the cropped 8-bit batch with HOST-DRAWN parameters: k_collate can leave the crop as
// uint8 [B, crop, crop]; k_aug_geom = flip + PIL's nearest-neighbour AFFINE transform (Geometry.c: the scaling special
// case with positions accumulated in double, or 16.16 fixed point); k_img_sum + k_aug_blend = ImageEnhance brightness /
// contrast (Blend.c, fp32, truncating); k_u8_to_tensor = ToTensor + Normalize(0.5, 0.5) on three equal channels.

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
    for (long long i = tid; i < nv; i += 4 * nthr) {           // four independent 16-byte loads in flight per thread
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long iu = i + u * nthr;
        raw[u] = pv[iu < nv ? iu : i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T* e = reinterpret_cast<const T*>(&raw[u]);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = (float)e[j];
          mn = fminf(mn, v);
          mx = fmaxf(mx, v);
        }
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

template <typename T> struct IsIntSrc { static constexpr bool value = true; };
template <> struct IsIntSrc<float> { static constexpr bool value = false; };

template <typename T, bool MINMAX>
struct Quant {
  float mn, range, rcp;
  __device__ __forceinline__ void set(float lo, float hi) {
    mn = lo;
    range = sub_(hi, lo);
    // loop-invariant half of the compiler's own fp32 division (v_rcp_f32 + one Newton step), see quotient()
    const float y = __builtin_amdgcn_rcpf(range);
    rcp = __builtin_fmaf(__builtin_fmaf(-range, y, 1.0f), y, y);
    // max == min: the reference divides 0 / 0 -> NaN -> uint8 cast; defined as grey level 0 here.  With range 1 and a
    // zero reciprocal every quotient below is exactly 0 without a per-pixel test.
    if (!(range > 0.f)) { range = 1.f; rcp = 0.f; }
  }
  // n / range, bit-identical to the `/` operator.  For 8/16-bit sources n and range are INTEGERS with
  // 0 <= n <= range < 2^17: the v_div_scale / v_div_fixup steps of the compiler's expansion are identities, and of its
  // two residual corrections ONE already lands on the correctly rounded quotient - checked EXHAUSTIVELY for every such
  // pair (glr_selftest_quotient, tests/test_gpu_collate.py): 3 instructions with the reciprocal hoisted instead of 13.
  __device__ __forceinline__ float quotient(float n) const {
    if (!IsIntSrc<T>::value) return div_(n, range);
    const float q0 = mul_(n, rcp);
    const float e0 = __builtin_fmaf(-range, q0, n);
    return __builtin_fmaf(e0, rcp, q0);
  }
  // ((x - min) / (max - min)) * 255 in fp32, before the C cast
  __device__ __forceinline__ float scaled(T raw) const { return mul_(quotient(sub_((float)raw, mn)), 255.0f); }
  // the 8-bit grey level of a raw pixel: the C cast (truncation) of scaled()
  __device__ __forceinline__ int level(T raw) const {
    if (!MINMAX) return (int)raw;                          // already the 8-bit image
    const float y = scaled(raw);
    int q = (int)y;
    if (!IsIntSrc<T>::value) q = y != y ? 0 : q;          // float sources may hold NaN pixels; integer ones cannot
    return min(max(q, 0), 255);
  }
  // level(raw) inserted as byte `sel` of `old`: floor (the values are >= 0: truncation) + v_cvt_pk_u8_f32, which
  // saturates to [0, 255] and turns NaN into 0 - two instructions for the cast, both clamps, the shift and the or
  __device__ __forceinline__ unsigned pack(T raw, unsigned sel, unsigned old) const {
    if (!MINMAX) return old | ((unsigned)raw << (8 * sel));
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(scaled(raw)), sel, old);
  }
  __device__ __forceinline__ float operator()(T raw) const { return (float)level(raw); }
};

// one source row as the resampler sees it: quantised pixel `col` as a float
template <typename T, bool MINMAX>
struct GlobalRow {
  const T* row;
  Quant<T, MINMAX> q;
  __device__ __forceinline__ float at(int col) const { return q(row[col]); }
};
struct LdsRow {
  const unsigned char* row;                 // staged, already quantised bytes; row[col] valid for the block's span
  __device__ __forceinline__ float at(int col) const { return (float)row[col]; }
};
template <typename T, bool MINMAX>
struct GlobalRows {
  const T* S;
  int W;
  Quant<T, MINMAX> q;
  __device__ __forceinline__ GlobalRow<T, MINMAX> operator()(int sy) const {
    return GlobalRow<T, MINMAX>{S + (long long)sy * W, q};
  }
};
struct LdsRows {                            // rows [r, r + nr) staged at `pitch` bytes each, vector aligned per row
  const unsigned char* lds;
  int r, pitch, W, c_lo, vmask;
  __device__ __forceinline__ LdsRow operator()(int sy) const {
    const int mis = (int)(((long long)sy * W + c_lo) & vmask);
    return LdsRow{lds + (sy - r) * pitch + mis - c_lo};
  }
};

// running state of one output pixel while its source rows are visited in order
struct Acc {
  float total;
  bool first_row;
  int isum;
};

// rows [ra, rb) of the cell of one output pixel, in OpenCV's order (mode 0 copy, 1 integer scale, 2 general)
template <typename RowOf>
__device__ __forceinline__ Acc accumulate(const RowOf row_of, int ra, int rb, int mode, int dx, int ix, const Taps tx,
                                          const Taps ty, Acc a) {
  for (int sy = ra; sy < rb; ++sy) {
    const auto row = row_of(sy);
    if (mode == 0) {
      a.isum = (int)row.at(dx);
    } else if (mode == 1) {
      for (int c = 0; c < ix; ++c) a.isum += (int)row.at(dx * ix + c);
    } else {
      const float beta = sy < ty.s1 ? ty.a_first : (sy < ty.s2 ? ty.a_mid : ty.a_last);
      float buf = 0.f;
      if (tx.has_first) buf = add_(buf, mul_(row.at(tx.s1 - 1), tx.a_first));
      for (int sx = tx.s1; sx < tx.s2; ++sx) buf = add_(buf, mul_(row.at(sx), tx.a_mid));
      if (tx.has_last) buf = add_(buf, mul_(row.at(tx.s2), tx.a_last));
      const float term = mul_(beta, buf);
      a.total = a.first_row ? term : add_(a.total, term);
      a.first_row = false;
    }
  }
  return a;
}

constexpr int LDS_BYTES = 40960;            // staged source rows of one output row (4 workgroups per CU)

// One workgroup per (output row, image).  The source rows under that output row are staged into LDS with 16-byte
// coalesced loads (all loads of a thread in flight together), quantised to uint8 once per source pixel, and every
// thread then walks the taps of its own output pixel out of LDS in OpenCV's summation order.  Rows are staged in
// chunks when they do not fit; images whose packed offset is not 16-byte aligned, or whose single-row span exceeds
// LDS, take the same arithmetic straight from global memory.
template <typename T, bool MINMAX>
__global__ __launch_bounds__(256) void k_collate(const unsigned char* __restrict__ src,
                                                 const long long* __restrict__ offset, const int* __restrict__ desc,
                                                 const unsigned* __restrict__ state, int crop,
                                                 float* __restrict__ out, unsigned char* __restrict__ out_u8) {
  constexpr int VEC = Src<T>::VEC;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int b = blockIdx.y;
  const int oy = blockIdx.x, ox = threadIdx.x;
  const int* ds = desc + 8 * b;
  const int H = ds[0], W = ds[1], dH = ds[2], dW = ds[3];
  const int dy = ds[6] + oy - ds[4], dx = ds[7] + ox - ds[5];     // position inside the resized image
  const long long off = offset[b];
  const T* S = reinterpret_cast<const T*>(src + off);
  Quant<T, MINMAX> q;
  q.set(0.f, 1.f);
  if (MINMAX) q.set(key2f(state[2 * b]), key2f(state[2 * b + 1]));
  const bool active = ox < crop;
  const bool ok = active && dx >= 0 && dx < dW;
  const int dx_lo = max(0, ds[7] - ds[5]), dx_hi = min(dW, ds[7] + crop - ds[5]);   // columns this row needs
  int v = 0;                                                       // zero padding outside the resized image
  if (dy >= 0 && dy < dH && dx_hi > dx_lo) {                       // uniform over the workgroup
    int mode = 0, ix = 1, iy = 1;
    double scale_x = 1.0, scale_y = 1.0;
    if (!(dH == H && dW == W)) {                                   // cv2.resize copies when the sizes are equal
      scale_x = div_(1.0, div_((double)dW, (double)W));
      scale_y = div_(1.0, div_((double)dH, (double)H));
      ix = __double2int_rn(scale_x);
      iy = __double2int_rn(scale_y);
      mode = (fabs(scale_x - ix) < GLR_DBL_EPS && fabs(scale_y - iy) < GLR_DBL_EPS) ? 1 : 2;
      if (dH > H || dW > W) mode = 3;                              // not (scale_x >= 1 && scale_y >= 1): bilinear emulation
    }
    if (mode == 3) {
      // cv::resize, area_mode with ksize 2 and fixed point: coefficient loop, HResizeLinear (int32), VResizeLinear 8u.
      // Small images (long side < scale): straight from global memory.
      if (ok) {
        const double inv_x = div_((double)dW, (double)W), inv_y = div_((double)dH, (double)H);
        int sx = (int)floor(mul_((double)dx, scale_x));
        float fx = (float)sub_((double)(dx + 1), mul_((double)(sx + 1), inv_x));
        fx = fx <= 0.f ? 0.f : sub_(fx, (float)(int)floorf(fx));
        bool past = false;                                         // dx >= xmax: D = S[sx] * ONE
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx + 1 >= W) {
          past = true;
          if (sx >= W - 1) { fx = 0.f; sx = W - 1; }
        }
        const int a0 = min(max(__float2int_rn(mul_(sub_(1.f, fx), 2048.f)), -32768), 32767);
        const int a1 = min(max(__float2int_rn(mul_(fx, 2048.f)), -32768), 32767);
        const int sy = (int)floor(mul_((double)dy, scale_y));
        float fy = (float)sub_((double)(dy + 1), mul_((double)(sy + 1), inv_y));
        fy = fy <= 0.f ? 0.f : sub_(fy, (float)(int)floorf(fy));
        const int b0 = min(max(__float2int_rn(mul_(sub_(1.f, fy), 2048.f)), -32768), 32767);
        const int b1 = min(max(__float2int_rn(mul_(fy, 2048.f)), -32768), 32767);
        const int ra = min(max(sy, 0), H - 1), rb = min(max(sy + 1, 0), H - 1);
        const int sx1 = min(sx + 1, W - 1);
        const T* Ra = S + (long long)ra * W;
        const T* Rb = S + (long long)rb * W;
        const int h0 = past ? (int)q(Ra[sx]) * 2048 : (int)q(Ra[sx]) * a0 + (int)q(Ra[sx1]) * a1;
        const int h1 = past ? (int)q(Rb[sx]) * 2048 : (int)q(Rb[sx]) * a0 + (int)q(Rb[sx1]) * a1;
        v = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 255;
      }
    } else {
    Taps ty = {}, tx = {};
    int r0 = dy, r1 = dy + 1, c_lo = dx_lo, c_hi = dx_hi;
    if (mode == 1) {
      r0 = dy * iy;
      r1 = r0 + iy;
      c_lo = dx_lo * ix;
      c_hi = dx_hi * ix;
    } else if (mode == 2) {
      ty = make_taps(dy, H, scale_y);
      r0 = ty.has_first ? ty.s1 - 1 : ty.s1;
      r1 = ty.has_last ? ty.s2 + 1 : ty.s2;
      c_lo = max(0, (int)floor(mul_((double)dx_lo, scale_x)) - 1);
      c_hi = min(W, (int)ceil(mul_((double)dx_hi, scale_x)) + 1);
      if (ok) tx = make_taps(dx, W, scale_x);
    }
    const int pitch = (c_hi - c_lo + 2 * VEC + 15) & ~15;
    const int R = LDS_BYTES / pitch;
    Acc acc = {0.f, true, 0};
    if ((off & 15) == 0 && R > 0) {
      const int vpr = pitch / VEC;                                 // 16-byte vectors staged per row
      const long long n_pad = ((long long)H * W + VEC - 1) / VEC * VEC;   // images are padded to 16 bytes in `src`
      for (int r = r0; r < r1; r += R) {
        const int nr = min(R, r1 - r);
        if (r != r0) __syncthreads();
        for (int v0 = ox; v0 < nr * vpr; v0 += 4 * 256) {         // up to four loads in flight per thread
          uint4 raw[4];
          int where[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int vi = v0 + u * 256;
            where[u] = -1;
            if (vi < nr * vpr) {
              const int rl = vi / vpr, k = vi - rl * vpr;
              const long long idx = (((long long)(r + rl) * W + c_lo) & ~(long long)(VEC - 1)) + (long long)k * VEC;
              if (idx + VEC <= n_pad) {
                raw[u] = *reinterpret_cast<const uint4*>(S + idx);
                where[u] = rl * pitch + k * VEC;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (where[u] >= 0) {
              const T* e = reinterpret_cast<const T*>(&raw[u]);
              unsigned w[VEC / 4] = {};
#pragma unroll
              for (int j = 0; j < VEC; ++j) w[j / 4] = q.pack(e[j], j % 4, w[j / 4]);
              unsigned* d = reinterpret_cast<unsigned*>(lds + where[u]);
#pragma unroll
              for (int j = 0; j < VEC / 4; ++j) d[j] = w[j];
            }
          }
        }
        __syncthreads();
        if (ok) acc = accumulate(LdsRows{lds, r, pitch, W, c_lo, VEC - 1}, r, r + nr, mode, dx, ix, tx, ty, acc);
      }
    } else if (ok) {
      acc = accumulate(GlobalRows<T, MINMAX>{S, W, q}, r0, r1, mode, dx, ix, tx, ty, acc);
    }
    if (ok) {
      if (mode == 0) {
        v = acc.isum;
      } else if (mode == 1) {
        if (ix == 2 && iy == 2) {
          v = (acc.isum + 2) >> 2;                                 // OpenCV's 8u 2x2 SIMD rounding
        } else {
          const float inv = div_(1.0f, (float)(ix * iy));
          v = __float2int_rn(mul_((float)acc.isum, inv));
        }
      } else {
        v = __float2int_rn(acc.total);                             // saturate_cast<uchar>: round half to even
      }
      v = min(max(v, 0), 255);
    }
    }
  }
  if (!active) return;
  if (out_u8 != nullptr) {                                         // the 8-bit crop, for the transform passes
    out_u8[((long long)b * crop + oy) * crop + ox] = (unsigned char)v;
    return;
  }
  const float t = div_((float)v, 255.0f);                          // ToTensor
  const float o = div_(sub_(t, 0.5f), 0.5f);                       // Normalize(0.5, 0.5)
  const long long plane = (long long)crop * crop;
  float* dst = out + (long long)b * 3 * plane + (long long)oy * crop + ox;
  dst[0] = o;
  dst[plane] = o;
  dst[2 * plane] = o;
}

// ---- random transforms on the cropped 8-bit batch [B, n, n] (n <= 256: one thread per pixel of a row) ----
__device__ __forceinline__ long long pil_fix(double v) {          // Geometry.c FIX: FLOOR(v * 65536.0 + 0.5)
  const double t = add_(mul_(v, 65536.0), 0.5);
  return t < 0.0 ? (long long)floor(t) : (long long)t;
}
__device__ __forceinline__ int pil_coord(double v, int n) {       // COORD: v < 0 ? -1 : (int)v  (>= n is outside anyway)
  return v < 0.0 ? -1 : (v >= (double)n ? n : (int)v);
}

// flip[b] != 0: RandomHorizontalFlip fired; matrix[b][0..5]: PIL AFFINE coefficients (NaN in [0]: no affine).
// out(y, x) = flipped(yin, xin) or 0 outside, flipped(y, x) = src(y, n - 1 - x).
__global__ __launch_bounds__(256) void k_aug_geom(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int n,
                                                  const int* __restrict__ flip, const double* __restrict__ matrix) {
  const int b = blockIdx.y, y = blockIdx.x, x = threadIdx.x;
  if (x >= n) return;
  const unsigned char* S = src + (long long)b * n * n;
  int xi = x, yi = y;
  const double* m = matrix ? matrix + 6 * b : nullptr;
  if (m != nullptr && m[0] == m[0]) {
    if (m[1] == 0.0 && m[3] == 0.0) {                               // ImagingScaleAffine: xo += a[0] per pixel, in double
      double xo = add_(m[2], mul_(m[0], 0.5)), yo = add_(m[5], mul_(m[4], 0.5));
      for (int i = 0; i < x; ++i) xo = add_(xo, m[0]);
      for (int i = 0; i < y; ++i) yo = add_(yo, m[4]);
      xi = pil_coord(xo, n);
      yi = pil_coord(yo, n);
    } else {                                                        // affine_fixed: 16.16 fixed point
      const long long a0 = pil_fix(m[0]), a1 = pil_fix(m[1]), a3 = pil_fix(m[3]), a4 = pil_fix(m[4]);
      const long long a2 = pil_fix(add_(add_(m[2], mul_(m[0], 0.5)), mul_(m[1], 0.5)));
      const long long a5 = pil_fix(add_(add_(m[5], mul_(m[3], 0.5)), mul_(m[4], 0.5)));
      const long long xx = (a2 + y * a1 + x * a0) >> 16, yy = (a5 + y * a4 + x * a3) >> 16;
      xi = xx < 0 ? -1 : (xx >= n ? n : (int)xx);
      yi = yy < 0 ? -1 : (yy >= n ? n : (int)yy);
    }
  }
  unsigned char v = 0;
  if (xi >= 0 && xi < n && yi >= 0 && yi < n) v = S[(long long)yi * n + (flip[b] ? n - 1 - xi : xi)];
  dst[((long long)b * n + y) * n + x] = v;
}

__global__ __launch_bounds__(256) void k_img_sum(const unsigned char* __restrict__ img, int count, unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  const unsigned char* S = img + (long long)b * count;
  unsigned long long s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) s += S[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(&sums[b], s);          // integer adds: exact in any order
}

// kind[b]: 0 none, 1 brightness (degenerate = 0), 2 contrast (degenerate = int(mean + 0.5)); alpha[b]: the factor as a C
// float.  PIL Blend.c: out = (UINT8)((int)deg + alpha * ((int)in - (int)deg)), clipped when alpha is outside [0, 1].
__global__ __launch_bounds__(256) void k_aug_blend(unsigned char* __restrict__ img, int count, const int* __restrict__ kind,
                                                   const float* __restrict__ alpha, const unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y, k = kind[b];
  if (k == 0) return;
  const float a = alpha[b];
  if (a == 1.0f) return;                                            // ImagingCopy(image)
  int g = 0;
  if (k == 2) g = (int)add_(div_((double)sums[b], (double)count), 0.5);
  unsigned char* S = img + (long long)b * count;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const int v = S[i];
    int o;
    if (a == 0.0f) {
      o = g;
    } else {
      const float t = add_((float)g, mul_(a, (float)(v - g)));
      if (a >= 0.f && a <= 1.f) o = (int)t;
      else o = t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
    }
    S[i] = (unsigned char)o;
  }
}

__global__ __launch_bounds__(256) void k_u8_to_tensor(const unsigned char* __restrict__ img, int count, float* __restrict__ out) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float t = div_((float)img[(long long)b * count + i], 255.0f);       // ToTensor
    const float o = div_(sub_(t, 0.5f), 0.5f);                                 // Normalize(0.5, 0.5)
    float* d = out + (long long)b * 3 * count + i;
    d[0] = o;
    d[count] = o;
    d[2 * (long long)count] = o;
  }
}

// exhaustive check of Quant::quotient's shortened division: every integer pair 0 <= n <= d, d in [d_lo, d_hi)
__global__ __launch_bounds__(256) void k_selftest_quotient(int d_lo, int d_hi, unsigned long long* __restrict__ bad) {
  const int d = d_lo + blockIdx.x;
  if (d >= d_hi) return;
  Quant<short, true> q;
  q.set(0.f, (float)d);
  unsigned long long n_bad = 0;
  for (int n = threadIdx.x; n <= d; n += 256) {
    const float a = q.quotient((float)n), b = div_((float)n, (float)d);
    n_bad += __float_as_uint(a) != __float_as_uint(b);
  }
  if (n_bad) atomicAdd(bad, n_bad);
}

}  // namespace

extern "C" int glr_selftest_quotient(int d_lo, int d_hi, uint64_t* bad, void* stream) {
  if (!bad || d_lo < 1 || d_hi <= d_lo || d_hi > (1 << 17) + 1) return GLR_EINVAL;
  hipLaunchKernelGGL(k_selftest_quotient, dim3(d_hi - d_lo), dim3(256), 0, static_cast<hipStream_t>(stream), d_lo, d_hi,
                     reinterpret_cast<unsigned long long*>(bad));
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_aug_geom(const uint8_t* src, uint8_t* dst, int B, int size, const int32_t* flip, const double* matrix,
                            void* stream) {
  if (!src || !dst || src == dst || !flip || B <= 0 || size <= 0 || size > 256) return GLR_EINVAL;
  hipLaunchKernelGGL(k_aug_geom, dim3(size, B), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, size, flip, matrix);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_aug_jitter(uint8_t* img, int B, int size, const int32_t* kind, const float* alpha, uint64_t* sums_ws,
                              void* stream) {
  if (!img || !kind || !alpha || !sums_ws || B <= 0 || size <= 0 || size > 256) return GLR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int count = size * size;
  if (hipMemsetAsync(sums_ws, 0, sizeof(uint64_t) * B, st) != hipSuccess) return GLR_ELAUNCH;
  const int blocks = (count + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(k_img_sum, dim3(blocks, B), dim3(256), 0, st, img, count, reinterpret_cast<unsigned long long*>(sums_ws));
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_aug_blend, dim3(blocks, B), dim3(256), 0, st, img, count, kind, alpha,
                     reinterpret_cast<const unsigned long long*>(sums_ws));
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_u8_to_tensor(const uint8_t* img, int B, int size, float* out, void* stream) {
  if (!img || !out || B <= 0 || size <= 0 || size > 256) return GLR_EINVAL;
  const int count = size * size;
  hipLaunchKernelGGL(k_u8_to_tensor, dim3((count + 2047) / 2048, B), dim3(256), 0, static_cast<hipStream_t>(stream), img, count, out);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

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
                                  const uint32_t* state, int B, int src_dtype, int crop, float* out, uint8_t* out_u8,
                                  void* stream) {
  if (!src || !offset || !desc || (!out == !out_u8) || B <= 0 || crop <= 0 || crop > 256) return GLR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned char* s = static_cast<const unsigned char*>(src);
  const long long* off = reinterpret_cast<const long long*>(offset);
  dim3 grid(crop, B);
#define GLR_COLLATE(T)                                                                                         \
  do {                                                                                                         \
    if (state) hipLaunchKernelGGL((k_collate<T, true>), grid, dim3(256), 0, st, s, off, desc, state, crop, out, out_u8); \
    else hipLaunchKernelGGL((k_collate<T, false>), grid, dim3(256), 0, st, s, off, desc, state, crop, out, out_u8);      \
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
