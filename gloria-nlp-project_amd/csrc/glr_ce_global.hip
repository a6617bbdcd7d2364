// K2: dual cross-entropy over a B x B similarity matrix (labels = arange), forward + backward.
//     Replaces the nn.CrossEntropyLoss pairs of /root/reference/gloria/loss/gloria_loss.py:86-87
//     and :167-170.  HBM/latency bound: 2*B*B*4 bytes.
// K3: global similarity matrix, forward + backward, on the matrix cores (fp32 MFMA, exact fp32).  Replaces the
//     norm/bmm/clamp/scale of global_loss (gloria_loss.py:75-80) and autograd through them.
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
// Global similarity on the matrix cores, one launch per direction.
//
// The operands arrive in fp32 (the reference runs global_loss in fp32, gloria_loss.py:66-88), so the
// contraction uses v_mfma_f32_32x32x2_f32: an exact fp32 fma chain at the fp32 vector rate - 100 MFLOP at
// B = 256 is ~3 us of matrix time; the kernel is latency bound, not rate bound, and keeps the 1e-4 bar.
// Operand lane map (cdna guide section 3): lane l feeds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31].
// A lane loads 4 consecutive k of its row (one 16-byte load) per 8-wide k group; MFMA t of the group then
// contracts k = k0 + t (lanes 0-31) and k0 + 4 + t (lanes 32-63) - the same permutation on both operands.
// A workgroup = one 32 x 32 output tile, its 8 waves split K (interleaved 8-wide groups) and are summed in LDS
// in a fixed order (bitwise reproducible).
constexpr int GS_NW = 8;              // waves per workgroup = K split
constexpr int GS_NT = 64 * GS_NW;

// forward: sim[a, j] = temp3 * <I_a, T_j> / max(|I_a| |T_j|, eps); the row norms (gloria_loss.py:75-76) are
// accumulated from the operand registers of the same loop and written by the first tile row / column.
__global__ void __launch_bounds__(GS_NT) k_global_sim_mfma(const float* __restrict__ img, const float* __restrict__ txt,
                                                           int B_img, int B_txt, int D, float temp3, float eps,
                                                           float* __restrict__ sim, int ld_sim, float* __restrict__ ni,
                                                           float* __restrict__ nt) {
  __shared__ float red[GS_NW][16][64];
  __shared__ float nrm[2][GS_NW][64];
  __shared__ float nfin[2][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int a0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  const bool va = a0 + r < B_img, vb = j0 + r < B_txt;
  const float* pa = img + (size_t)(va ? a0 + r : 0) * D;
  const float* pb = txt + (size_t)(vb ? j0 + r : 0) * D;
  const bool vec = (D & 3) == 0;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float na = 0.f, nb = 0.f;
  const int ngroups = (D + 7) / 8;
  auto load4 = [&](const float* p, bool valid, int c) {
    const int k = c * 8 + 4 * h;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && c < ngroups) {
      if (vec && k + 3 < D) v = *reinterpret_cast<const float4*>(p + k);
      else {
        if (k < D) v.x = p[k];
        if (k + 1 < D) v.y = p[k + 1];
        if (k + 2 < D) v.z = p[k + 2];
        if (k + 3 < D) v.w = p[k + 3];
      }
    }
    return v;
  };
  // the next group's operands are in flight while the current group's four dependent MFMAs run
  float4 x = load4(pa, va, w), y = load4(pb, vb, w);
  for (int c = w; c < ngroups; c += GS_NW) {
    const float4 xn = load4(pa, va, c + GS_NW), yn = load4(pb, vb, c + GS_NW);
    na += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    nb += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc, 0, 0, 0);
    x = xn;
    y = yn;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[w][q][lane] = acc[q];
  nrm[0][w][lane] = na;
  nrm[1][w][lane] = nb;
  __syncthreads();
  if (threadIdx.x < 64) {                       // squared norms: waves x 2 lane halves, fixed order
    const int which = threadIdx.x >> 5, row = threadIdx.x & 31;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < GS_NW; ++k) s += nrm[which][k][row] + nrm[which][k][row + 32];
    nfin[which][row] = sqrtf(s);
  }
  __syncthreads();
#pragma unroll
  for (int e = threadIdx.x; e < 1024; e += GS_NT) {
    const int q = e >> 6, ln = e & 63;
    const int row = (q & 3) + 8 * (q >> 2) + 4 * (ln >> 5), col = ln & 31;
    float sdot = 0.f;
#pragma unroll
    for (int k = 0; k < GS_NW; ++k) sdot += red[k][q][ln];
    if (a0 + row < B_img && j0 + col < B_txt)
      sim[(size_t)(a0 + row) * ld_sim + j0 + col] = sdot / fmaxf(nfin[0][row] * nfin[1][col], eps) * temp3;
  }
  if (threadIdx.x < 32) {
    if (blockIdx.y == 0 && a0 + threadIdx.x < B_img) ni[a0 + threadIdx.x] = nfin[0][threadIdx.x];
    if (blockIdx.x == 0 && j0 + threadIdx.x < B_txt) nt[j0 + threadIdx.x] = nfin[1][threadIdx.x];
  }
}

// backward, both directions in one launch (blockIdx.z = 0: d/d img, 1: d/d txt).  For a row a of X with partner
// rows Y_j, den = max(|X_a| |Y_j|, eps), g = dsim, s = <X_a, Y_j> = sim den / temp3 (from the saved forward):
//   dX[a] = sum_j (temp3 g / den) Y_j  -  X_a * sum_j [|X_a||Y_j| >= eps, |X_a| > 0] g sim |Y_j| / (den |X_a|)
// (the second term is torch's sub-gradient of the clamp, zero where the clamp is active).  The first term is a
// GEMM whose A operand (the coefficients) is formed in registers from g and the norms while it is loaded; the
// self coefficient is accumulated by the same lanes.  Output tile 32 rows x 32 features, K = partner rows.
__global__ void __launch_bounds__(GS_NT) k_global_bwd_mfma(const float* __restrict__ img, const float* __restrict__ txt,
                                                           const float* __restrict__ ni, const float* __restrict__ nt,
                                                           const float* __restrict__ sim, const float* __restrict__ dsim,
                                                           int ld_sim, int B_img, int B_txt, int D, float temp3, float eps,
                                                           float* __restrict__ dimg, float* __restrict__ dtxt) {
  __shared__ float red[GS_NW][16][64];
  __shared__ float selfp[GS_NW][64];
  __shared__ float selff[32];
  const int tr = blockIdx.z;                         // 0: X = img, 1: X = txt
  const int n_x = tr ? B_txt : B_img, n_o = tr ? B_img : B_txt;
  const int a0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  if (a0 >= n_x) return;
  const float* X = tr ? txt : img;
  const float* Y = tr ? img : txt;
  const float* nx = tr ? nt : ni;
  const float* ny = tr ? ni : nt;
  float* dX = tr ? dtxt : dimg;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool va = a0 + r < n_x, vd = d0 + r < D;
  const int a = va ? a0 + r : 0;
  const float nxa = nx[a];
  // element (a, j) of sim / dsim
  const size_t sa = tr ? 1 : (size_t)ld_sim, sj = tr ? (size_t)ld_sim : 1;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float self = 0.f;
  const int ngroups = (n_o + 7) / 8;
  struct Grp { float g[4], sv[4], ny[4], y[4]; };
  auto load = [&](int c) {
    Grp v;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = c * 8 + 4 * h + t;
      const bool vj = j < n_o;
      const int jj = vj ? j : 0;
      const size_t e = (size_t)a * sa + (size_t)jj * sj;
      v.g[t] = (va && vj) ? dsim[e] : 0.f;
      v.sv[t] = sim[e];
      v.ny[t] = ny[jj];
      v.y[t] = (vj && vd) ? Y[(size_t)jj * D + d0 + r] : 0.f;
    }
    return v;
  };
  Grp cur = load(w);
  for (int c = w; c < ngroups; c += GS_NW) {
    const Grp nxt = load(c + GS_NW);                 // out-of-range groups load element 0 with g = 0
    float ca[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float prod = nxa * cur.ny[t];
      const float den = fmaxf(prod, eps);
      ca[t] = temp3 * cur.g[t] / den;
      if (prod >= eps && nxa > 0.f) self += cur.g[t] * cur.sv[t] * cur.ny[t] / (den * nxa);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[t], cur.y[t], acc, 0, 0, 0);
    cur = nxt;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[w][q][lane] = acc[q];
  selfp[w][lane] = self;
  __syncthreads();
  if (threadIdx.x < 32) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < GS_NW; ++k) s += selfp[k][threadIdx.x] + selfp[k][threadIdx.x + 32];
    selff[threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int e = threadIdx.x; e < 1024; e += GS_NT) {
    const int q = e >> 6, ln = e & 63;
    const int row = (q & 3) + 8 * (q >> 2) + 4 * (ln >> 5), col = ln & 31;
    if (a0 + row < n_x && d0 + col < D) {
      const size_t o = (size_t)(a0 + row) * D + d0 + col;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < GS_NW; ++k) v += red[k][q][ln];
      dX[o] = v - selff[row] * X[o];
    }
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
  hipLaunchKernelGGL(k_global_sim_mfma, dim3((B_img + 31) / 32, (B_txt + 31) / 32), dim3(GS_NT), 0, (hipStream_t)stream,
                     img, txt, B_img, B_txt, D, temp3, eps, sim, ld_sim, ni, nt);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_global_sim_bwd(const float* img, const float* txt, const float* ni, const float* nt,
                                  const float* sim, const float* dsim, int ld_sim, int B_img, int B_txt, int D,
                                  float temp3, float eps, float* dimg, float* dtxt, void* stream) {
  if (!img || !txt || !ni || !nt || !sim || !dsim || !dimg || !dtxt || B_img <= 0 || B_txt <= 0 || D <= 0 ||
      ld_sim < B_txt)
    return GLR_EINVAL;
  const int nb = (B_img > B_txt ? B_img : B_txt);
  hipLaunchKernelGGL(k_global_bwd_mfma, dim3((nb + 31) / 32, (D + 31) / 32, 2), dim3(GS_NT), 0, (hipStream_t)stream, img,
                     txt, ni, nt, sim, dsim, ld_sim, B_img, B_txt, D, temp3, eps, dimg, dtxt);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
