// Exact selection on fp32 scores: the two places where the reference turns scores
// into indices / thresholds,
//   * retrieval ranking      np.argsort(similarities)[::-1][:top_k]            (gloria/models/retrival_model.py:118)
//   * percentile threshold   torch.topk(preds, total - top_k, largest=False).values.max()
//                                                                              (gloria/lightning/callbacks.py:56)
// Both are order statistics, computed here by MSD radix select on the order-preserving bit pattern of the
// float: the result is the SAME element the CPU sort picks - bit exact, no tolerance.  Ties between equal
// scores are ranked by DESCENDING index (what reversing a stable ascending argsort gives); -0.0 == +0.0;
// NaNs rank above +inf (torch.topk's order).  One 1024-thread workgroup per row; HBM-bound (4 or 8 passes of
// n * 4 bytes, L2-resident for the sizes of the path: 224 x 224 overlays, a few thousand retrieval targets).
#include "glr_common.h"

namespace {

constexpr int SEL_NT = 1024;

__device__ __forceinline__ unsigned f2key(float v) {
  unsigned u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;                                  // -0.0 == +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);             // ascending unsigned order == ascending float order
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// k-th smallest (1-indexed) of the composite keys comp(i), i in [0, n): BITS = 32 (value only) or 64 (value, index).
// Every thread returns the same result.  hist: 256 counters + 2 words of scratch in LDS.
template <int BITS, typename F>
__device__ unsigned long long radix_select(F comp, long long n, unsigned long long k, unsigned* hist) {
  unsigned long long prefix = 0, mask = 0;
  for (int shift = BITS - 8; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += SEL_NT) hist[i] = 0;
    __syncthreads();
    for (long long i = threadIdx.x; i < n; i += SEL_NT) {
      const unsigned long long c = comp(i);
      if ((c & mask) == prefix) atomicAdd(&hist[(unsigned)(c >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long cum = 0;
      int sel = 255;
      for (int b = 0; b < 256; ++b) {
        const unsigned c = hist[b];
        if (cum + c >= k) { sel = b; break; }
        cum += c;
      }
      hist[256] = (unsigned)sel;
      hist[257] = (unsigned)(k - cum);         // rank inside the selected bucket (fits: bucket count < 2^32)
    }
    __syncthreads();
    prefix |= (unsigned long long)hist[256] << shift;
    mask |= 0xffull << shift;
    k = hist[257];
    __syncthreads();
  }
  return prefix;
}

__global__ void __launch_bounds__(SEL_NT) k_kth_value(const float* __restrict__ x, long long n, long long k,
                                                      float* __restrict__ out) {
  __shared__ unsigned hist[258];
  const float* row = x + (size_t)blockIdx.x * n;
  const unsigned key = (unsigned)radix_select<32>([&](long long i) { return (unsigned long long)f2key(row[i]); }, n,
                                                  (unsigned long long)k, hist);
  if (threadIdx.x == 0) out[blockIdx.x] = key2f(key);
}

// indices (and values) of the k largest scores of each row, in descending order of (value, index)
__global__ void __launch_bounds__(SEL_NT) k_topk_desc(const float* __restrict__ x, long long n, int k,
                                                      long long* __restrict__ idx_out, float* __restrict__ val_out) {
  __shared__ unsigned hist[258];
  __shared__ unsigned long long cand[SEL_NT];
  __shared__ unsigned ncand;
  const float* row = x + (size_t)blockIdx.x * n;
  auto comp = [&](long long i) { return ((unsigned long long)f2key(row[i]) << 32) | (unsigned long long)(unsigned)i; };
  // threshold = k-th largest composite = (n - k + 1)-th smallest; composites are distinct, so exactly k pass
  const unsigned long long thr = radix_select<64>(comp, n, (unsigned long long)(n - k + 1), hist);
  int kp = 1;
  while (kp < k) kp <<= 1;
  if (threadIdx.x == 0) ncand = 0;
  for (int i = threadIdx.x; i < kp; i += SEL_NT) cand[i] = 0ull;          // padding sorts last
  __syncthreads();
  for (long long i = threadIdx.x; i < n; i += SEL_NT) {
    const unsigned long long c = comp(i);
    if (c >= thr) cand[atomicAdd(&ncand, 1u)] = c;                        // any order: sorted below
  }
  __syncthreads();
  // bitonic sort, descending
  for (int size = 2; size <= kp; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < kp / 2; t += SEL_NT) {
        const int lo = (t / stride) * stride * 2 + (t % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = cand[lo], b = cand[hi];
        if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int j = threadIdx.x; j < k; j += SEL_NT) {
    const unsigned long long c = cand[j];
    idx_out[(size_t)blockIdx.x * k + j] = (long long)(c & 0xffffffffull);
    if (val_out) val_out[(size_t)blockIdx.x * k + j] = row[(long long)(c & 0xffffffffull)];
  }
}

// counts behind the reference's percentile metrics (callbacks.py:57-61): per row
//   out[row] = { #(pred > thr & target), #(pred > thr), #(target), #(pred > thr | target) }
__global__ void __launch_bounds__(256) k_threshold_counts(const float* __restrict__ pred, const unsigned char* __restrict__ target,
                                                          const float* __restrict__ thr, long long n,
                                                          unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[4][4];
  const float* p = pred + (size_t)blockIdx.x * n;
  const unsigned char* t = target + (size_t)blockIdx.x * n;
  const float th = thr[blockIdx.x];
  unsigned tp = 0, pp = 0, tt = 0, un = 0;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const bool a = p[i] > th, b = t[i] != 0;
    tp += a && b; pp += a; tt += b; un += a || b;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned v[4] = {tp, pp, tt, un};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    unsigned s = v[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[wave][c] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4) out[(size_t)blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Nearest-upsampling footprint counts (the localization metrics WITHOUT the 224 x 224 overlay).
// nn.Upsample(size=(Hl, Wl)) of an ih x iw map (callbacks.py:319, mode 'nearest') copies cell
// (min(floor(y ih / Hl), ih - 1), min(floor(x iw / Wl), iw - 1)) - torch's fp32 rule - to pixel (y, x): every
// pixel statistic of the overlay is a statistic of <= ih * iw cell values weighted by how many pixels (npix) and how
// many LABEL pixels (cnt) each cell covers.  One workgroup per (image, row band of one cell row): reads its band
// of the label once (HBM-bound: Hl * Wl bytes per image), integer LDS counters.
__global__ void __launch_bounds__(256) k_cell_counts(const unsigned char* __restrict__ labels, int Hl, int Wl, int ih,
                                                     int iw, int* __restrict__ cnt, int* __restrict__ npix) {
  extern __shared__ int cs[];                     // [iw] label pixels, [iw] pixels
  const int b = blockIdx.y, ry = blockIdx.x;
  for (int i = threadIdx.x; i < 2 * iw; i += 256) cs[i] = 0;
  __syncthreads();
  const float sy = (float)ih / (float)Hl, sx = (float)iw / (float)Wl;
  // rows of this band: a superset [y0, y1) from the real-valued bounds, membership decided by the fp32 rule itself
  const int y0 = max(0, (int)floorf((float)ry / sy) - 2), y1 = min(Hl, (int)ceilf((float)(ry + 1) / sy) + 2);
  const unsigned char* lab = labels + (size_t)b * Hl * Wl;
  for (int y = y0; y < y1; ++y) {
    if (min((int)floorf((float)y * sy), ih - 1) != ry) continue;
    for (int x = threadIdx.x; x < Wl; x += 256) {
      const int rx = min((int)floorf((float)x * sx), iw - 1);
      atomicAdd(&cs[iw + rx], 1);
      if (lab[(size_t)y * Wl + x]) atomicAdd(&cs[rx], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < iw; i += 256) {
    cnt[((size_t)b * ih + ry) * iw + i] = cs[i];
    npix[((size_t)b * ih + ry) * iw + i] = cs[iw + i];
  }
}

}  // namespace

extern "C" int glr_cell_counts(const uint8_t* labels, int B, int Hl, int Wl, int ih, int iw, int32_t* cnt,
                               int32_t* npix, void* stream) {
  if (!labels || !cnt || !npix || B <= 0 || Hl <= 0 || Wl <= 0 || ih <= 0 || iw <= 0 || iw > 4096) return GLR_EINVAL;
  hipLaunchKernelGGL(k_cell_counts, dim3(ih, B), dim3(256), 2 * iw * sizeof(int), (hipStream_t)stream, labels, Hl, Wl,
                     ih, iw, cnt, npix);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_kth_value(const float* x, int rows, long long n, long long k, float* out, void* stream) {
  if (!x || !out || rows <= 0 || n <= 0 || k < 1 || k > n || n >= (1ll << 32)) return GLR_EINVAL;
  hipLaunchKernelGGL(k_kth_value, dim3(rows), dim3(SEL_NT), 0, (hipStream_t)stream, x, n, k, out);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_topk_desc(const float* x, int rows, long long n, int k, int64_t* idx, float* val, void* stream) {
  if (!x || !idx || rows <= 0 || n <= 0 || k < 1 || k > n || k > SEL_NT || n >= (1ll << 32)) return GLR_EINVAL;
  hipLaunchKernelGGL(k_topk_desc, dim3(rows), dim3(SEL_NT), 0, (hipStream_t)stream, x, n, k, (long long*)idx, val);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_threshold_counts(const float* pred, const uint8_t* target, const float* thr, int rows, long long n,
                                    uint64_t* out, void* stream) {
  if (!pred || !target || !thr || !out || rows <= 0 || n <= 0) return GLR_EINVAL;
  hipLaunchKernelGGL(k_threshold_counts, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, target, thr, n,
                     (unsigned long long*)out);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
