// K5  glr_wordpiece_segsum_fwd/bwd - word-piece -> word segment-sum fused with the sum over the last
//     BERT layers and the mean over the L slots.  Replaces BertEncoder.aggregate_tokens and the
//     post-processing of BertEncoder.forward (/root/reference/gloria/models/text_model.py:32-90, 96-131:
//     a Python double loop with one .item() device sync per token).  HBM-bound:
//     n_layers*B*L*D*e read + B*D*L*4 write.
// K4  glr_attn_sup_fwd - attention-supervision loss on the diagonal attention maps without ever
//     materialising the 224x224 upsampled maps.  Replaces gloria_model.py:143-147 (mean over words,
//     nearest interpolate, normalise, -log sum(label * map)).  HBM-bound: B*Hl*Wl label bytes.
// a-2 glr_cosine_fwd/bwd - row-wise cosine similarity with the reference's clamp on the PRODUCT of the
//     norms (gloria_loss.py:11-16).
#include "glr_common.h"

namespace {

// ------------------------------------------------------------------ K5
constexpr int K5_MAX_LAYERS = 16;     // BERT-base exposes 13 hidden states; `last_n_layers` may ask for all of them
struct LayerPtrs {
  const void* h[K5_MAX_LAYERS];
};

// grid (ceil(D/64), B), 256 threads = 64 features x 4 token phases; features past D are masked (any D).
// dst[b][t] = word slot of token t (monotone non-decreasing over the kept tokens) or -1.
__global__ void __launch_bounds__(256) k_segsum_fwd(LayerPtrs hp, int n_layers, int in_dtype,
                                                    const int* __restrict__ dst, float* __restrict__ word_emb,
                                                    float* __restrict__ sent_emb, int L, int D, float layer_scale) {
  extern __shared__ float tile[];                 // [64][L + 1]  word slots of 64 features
  const int b = blockIdx.y, d0 = blockIdx.x * 64;
  const int dx = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int LP = L + 1;
  for (int i = threadIdx.x; i < 64 * LP; i += 256) tile[i] = 0.f;
  __syncthreads();
  // one thread per feature walks the tokens in order (fixed summation order => bitwise reproducible);
  // the other three quarter-blocks only help with zeroing and the transposed write-out
  if (ph == 0 && d0 + dx < D) {
    for (int t = 0; t < L; ++t) {
      const int slot = dst[(size_t)b * L + t];
      if (slot < 0) continue;
      float v = 0.f;
      for (int l = 0; l < n_layers; ++l) v += ld_any(hp.h[l], ((size_t)b * L + t) * D + d0 + dx, in_dtype);
      tile[dx * LP + slot] += v * layer_scale;
    }
  }
  __syncthreads();
  // word_emb [B, D, L]: rows of L floats per feature, written coalesced along L
  for (int i = threadIdx.x; i < 64 * L; i += 256) {
    const int dl = i / L, w = i % L;
    if (d0 + dl < D) word_emb[((size_t)b * D + d0 + dl) * L + w] = tile[dl * LP + w];
  }
  if (threadIdx.x < 64 && d0 + threadIdx.x < D) {                          // sentence embedding: mean over ALL L slots (text_model.py:110)
    float s = 0.f;
    for (int w = 0; w < L; ++w) s += tile[threadIdx.x * LP + w];
    sent_emb[(size_t)b * D + d0 + threadIdx.x] = s / (float)L;
  }
}

// d_hidden[b][t][d] (same for every layer) = layer_scale * (d_word[b][d][dst] + d_sent[b][d] / L) if dst >= 0
__global__ void __launch_bounds__(256) k_segsum_bwd(const float* __restrict__ d_word, const float* __restrict__ d_sent,
                                                    const int* __restrict__ dst, void* __restrict__ d_hidden,
                                                    int out_dtype, int L, int D, float layer_scale) {
  extern __shared__ float tile[];                 // [64][L + 1]
  const int b = blockIdx.y, d0 = blockIdx.x * 64;
  const int dx = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int LP = L + 1;
  for (int i = threadIdx.x; i < 64 * L; i += 256) {
    const int dl = i / L, w = i % L;
    tile[dl * LP + w] = (d_word && d0 + dl < D) ? d_word[((size_t)b * D + d0 + dl) * L + w] : 0.f;
  }
  __syncthreads();
  if (d0 + dx >= D) return;
  const float gs = d_sent ? d_sent[(size_t)b * D + d0 + dx] / (float)L : 0.f;
  for (int t = ph; t < L; t += 4) {
    const int slot = dst[(size_t)b * L + t];
    const float v = slot >= 0 ? (tile[dx * LP + slot] + gs) * layer_scale : 0.f;
    st_any(d_hidden, ((size_t)b * L + t) * D + d0 + dx, out_dtype, v);
  }
}

// ------------------------------------------------------------------ K4
// block per image: cnt[b][r] = number of label pixels whose nearest-neighbour source is region r
// (torch F.interpolate default 'nearest': src = min(floor(dst * scale), in - 1), scale = in / out in fp32),
// then M[r] = mean_w a2[w, r], loss_b = -log( sum_r M cnt / sum_r M npix ), and the unit gradient
// dmap[w, r] = (npix[r]/den - cnt[r]/num) / n  (loss_b differentiated w.r.t. every map element).
__global__ void __launch_bounds__(256) k_attn_sup(const float* __restrict__ attn, const long long* __restrict__ attn_off,
                                                  const int* __restrict__ cap_lens, int img_offset,
                                                  const unsigned char* __restrict__ labels, int Hl, int Wl, int ih,
                                                  int iw, float* __restrict__ loss_b, float* __restrict__ dmap) {
  extern __shared__ float sh[];                   // cnt[S], npix[S], M[S], red[8]
  const int S = ih * iw;
  float* cnt = sh;
  float* npix = sh + S;
  float* M = sh + 2 * S;
  float* red = sh + 3 * S;
  const int b = blockIdx.x, sent = img_offset + b;
  for (int i = threadIdx.x; i < 2 * S; i += 256) sh[i] = 0.f;
  __syncthreads();
  const float sy = (float)ih / (float)Hl, sx = (float)iw / (float)Wl;
  const unsigned char* lab = labels + (size_t)b * Hl * Wl;
  for (int i = threadIdx.x; i < Hl * Wl; i += 256) {
    const int y = i / Wl, x = i % Wl;
    const int ry = min((int)floorf((float)y * sy), ih - 1), rx = min((int)floorf((float)x * sx), iw - 1);
    atomicAdd(&npix[ry * iw + rx], 1.f);
    if (lab[i]) atomicAdd(&cnt[ry * iw + rx], 1.f);
  }
  const int n = cap_lens[sent];
  const float* a = attn + attn_off[sent];
  for (int r = threadIdx.x; r < S; r += 256) {
    float s = 0.f;
    for (int w = 0; w < n; ++w) s += a[(size_t)w * S + r];
    M[r] = s / (float)n;
  }
  __syncthreads();
  float num = 0.f, den = 0.f;
  for (int r = threadIdx.x; r < S; r += 256) { num += M[r] * cnt[r]; den += M[r] * npix[r]; }
  num = wave_sum(num);
  den = wave_sum(den);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave] = num; red[4 + wave] = den; }
  __syncthreads();
  num = red[0] + red[1] + red[2] + red[3];
  den = red[4] + red[5] + red[6] + red[7];
  if (threadIdx.x == 0) loss_b[b] = -logf(num / den);
  if (dmap) {
    float* d = dmap + attn_off[sent];
    for (int i = threadIdx.x; i < n * S; i += 256) {
      const int r = i % S;
      d[i] = (npix[r] / den - cnt[r] / num) / (float)n;
    }
  }
}

// ------------------------------------------------------------------ cosine
// one wave per row
__global__ void __launch_bounds__(256) k_cosine_fwd(const float* __restrict__ x1, const float* __restrict__ x2,
                                                    int rows, int D, float eps, float* __restrict__ out,
                                                    float* __restrict__ stats) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float d = 0.f, a = 0.f, c = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float u = x1[(size_t)row * D + k], v = x2[(size_t)row * D + k];
    d += u * v; a += u * u; c += v * v;
  }
  d = wave_sum(d); a = wave_sum(a); c = wave_sum(c);
  if (lane == 0) {
    const float n1 = sqrtf(a), n2 = sqrtf(c);
    out[row] = d / fmaxf(n1 * n2, eps);
    if (stats) { stats[row * 3] = d; stats[row * 3 + 1] = n1; stats[row * 3 + 2] = n2; }
  }
}

__global__ void __launch_bounds__(256) k_cosine_bwd(const float* __restrict__ x1, const float* __restrict__ x2,
                                                    const float* __restrict__ stats, const float* __restrict__ g,
                                                    int rows, int D, float eps, float* __restrict__ dx1,
                                                    float* __restrict__ dx2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float d = stats[row * 3], n1 = stats[row * 3 + 1], n2 = stats[row * 3 + 2], gr = g[row];
  const float prod = n1 * n2, den = fmaxf(prod, eps);
  const bool un = prod >= eps;
  // cos = d/den: d cos/dx1 = x2/den - [un] d * n2 / (den^2 n1) * x1   (torch: clamp passes grad where input >= min)
  const float c1 = (un && n1 > 0.f) ? d * n2 / (den * den * n1) : 0.f;
  const float c2 = (un && n2 > 0.f) ? d * n1 / (den * den * n2) : 0.f;
  for (int k = lane; k < D; k += 64) {
    const float u = x1[(size_t)row * D + k], v = x2[(size_t)row * D + k];
    dx1[(size_t)row * D + k] = gr * (v / den - c1 * u);
    dx2[(size_t)row * D + k] = gr * (u / den - c2 * v);
  }
}

// ------------------------------------------------------------------ K6: attention regularisers
// Inputs: amean[b, i, r] = word-mean attention of image b for sentence i (K1 forward).  With a no-attention
// column (shift = 1) the reference works on P = [1 - sum_{r>=1} A[r], A[1:]] (gloria_loss.py:131-135), else P = A.
//   entropy[b, i] = -sum_r P log P                               (:95-96, :136-137)
//   kl[b, i]      = 1/2 sum_r (P_d - P_i)(log P_d - log P_i),  d = img_offset + b   (:91-92, :180-190)
//   na[b]         = log(1 - sum_{r>=shift} A_d[r])               (:130)
// One workgroup (4 waves) per image; wave w takes sentences w, w+4, ...; a row is 6 values per lane.
constexpr int REG_K = GLR_MAX_SPAD / 64;

struct RegRow { float p[REG_K], lp[REG_K]; };

__device__ __forceinline__ void reg_load_row(const float* __restrict__ a, int S_eff, int shift, int lane, RegRow& row) {
  float rest = 0.f;
#pragma unroll
  for (int k = 0; k < REG_K; ++k) {
    const int r = lane + 64 * k;
    row.p[k] = (r < S_eff) ? a[r] : 0.f;
    if (r >= 1) rest += row.p[k];
  }
  if (shift) {
    rest = wave_sum(rest);
    if (lane == 0) row.p[0] = 1.f - rest;
  }
#pragma unroll
  for (int k = 0; k < REG_K; ++k) row.lp[k] = (lane + 64 * k < S_eff) ? __logf(row.p[k]) : 0.f;
}

__global__ void __launch_bounds__(256) k_attn_reg_fwd(const float* __restrict__ amean, int n_sent, int S_pad, int S_eff,
                                                      int shift, int img_offset, float* __restrict__ out) {
  __shared__ float red[4][2];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = img_offset + b;
  const float* base = amean + (size_t)b * n_sent * S_pad;
  RegRow pd;
  reg_load_row(base + (size_t)d * S_pad, S_eff, shift, lane, pd);
  float ent = 0.f, kl = 0.f;
  for (int i = wave; i < n_sent; i += 4) {
    RegRow pi;
    reg_load_row(base + (size_t)i * S_pad, S_eff, shift, lane, pi);
    float e = 0.f, k2 = 0.f;
#pragma unroll
    for (int k = 0; k < REG_K; ++k) {
      if (lane + 64 * k < S_eff) {
        e -= pi.p[k] * pi.lp[k];
        k2 += (pd.p[k] - pi.p[k]) * (pd.lp[k] - pi.lp[k]);
      }
    }
    ent += wave_sum(e);
    if (i != d) kl += 0.5f * wave_sum(k2);
  }
  if (lane == 0) { red[wave][0] = ent; red[wave][1] = kl; }
  __syncthreads();
  if (threadIdx.x < 64) {
    // no-attention score of the diagonal pair: log(1 - mean_w sum_{r >= shift} a2)
    float sa = 0.f;
#pragma unroll
    for (int k = 0; k < REG_K; ++k) {
      const int r = lane + 64 * k;
      if (r >= shift && r < S_eff) sa += base[(size_t)d * S_pad + r];
    }
    sa = wave_sum(sa);
    if (lane == 0) {
      out[b * 4 + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      out[b * 4 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
      out[b * 4 + 2] = __logf(1.f - sa);
      out[b * 4 + 3] = 0.f;
    }
  }
}

// gradient w.r.t. amean of  coef[0] * sum entropy + coef[1] * sum kl + coef[2] * sum na
__global__ void __launch_bounds__(256) k_attn_reg_bwd(const float* __restrict__ amean, int n_sent, int S_pad, int S_eff,
                                                      int shift, int img_offset, const float* __restrict__ coef,
                                                      float* __restrict__ damean) {
  __shared__ float dacc[4][GLR_MAX_SPAD];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = img_offset + b;
  const float ce = coef[0], ck = coef[1], cn = coef[2];
  const float* base = amean + (size_t)b * n_sent * S_pad;
  float* dbase = damean + (size_t)b * n_sent * S_pad;
  RegRow pd;
  reg_load_row(base + (size_t)d * S_pad, S_eff, shift, lane, pd);
  float dd[REG_K];                                    // this wave's share of d kl / d P_d
#pragma unroll
  for (int k = 0; k < REG_K; ++k) dd[k] = 0.f;
  for (int i = wave; i < n_sent; i += 4) {
    if (i == d) continue;
    RegRow pi;
    reg_load_row(base + (size_t)i * S_pad, S_eff, shift, lane, pi);
    float dp[REG_K];
#pragma unroll
    for (int k = 0; k < REG_K; ++k) {
      dp[k] = 0.f;
      if (lane + 64 * k < S_eff) {
        const float dl = pi.lp[k] - pd.lp[k];          // log P_i - log P_d
        dp[k] = -ce * (pi.lp[k] + 1.f) + 0.5f * ck * (dl - pd.p[k] / pi.p[k] + 1.f);
        dd[k] += 0.5f * ck * (-dl + 1.f - pi.p[k] / pd.p[k]);
      }
    }
    const float dp0 = __shfl(dp[0], 0, 64);            // d / d P[0] (the "1 - rest" entry when shift)
#pragma unroll
    for (int k = 0; k < REG_K; ++k) {
      const int r = lane + 64 * k;
      float v = 0.f;
      if (r < S_eff) v = shift ? (r >= 1 ? dp[k] - dp0 : 0.f) : dp[k];
      if (r < S_pad) dbase[(size_t)i * S_pad + r] = v;
    }
  }
#pragma unroll
  for (int k = 0; k < REG_K; ++k) dacc[wave][lane + 64 * k] = dd[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    // diagonal row: entropy term + the kl terms of all other sentences (fixed wave order) + no-attention term
    float sa = 0.f;
#pragma unroll
    for (int k = 0; k < REG_K; ++k) {
      const int r = lane + 64 * k;
      if (r >= shift && r < S_eff) sa += base[(size_t)d * S_pad + r];
    }
    sa = wave_sum(sa);
    const float dna = cn != 0.f ? -cn / (1.f - sa) : 0.f;   // d na / d A_d[r], r >= shift (1 - sa ~ 0 without the extra column)
    float dp[REG_K];
#pragma unroll
    for (int k = 0; k < REG_K; ++k) {
      const int r = lane + 64 * k;
      dp[k] = 0.f;
      if (r < S_eff)
        dp[k] = -ce * (pd.lp[k] + 1.f) + ((dacc[0][r] + dacc[1][r]) + (dacc[2][r] + dacc[3][r]));
    }
    const float dp0 = __shfl(dp[0], 0, 64);
#pragma unroll
    for (int k = 0; k < REG_K; ++k) {
      const int r = lane + 64 * k;
      float v = 0.f;
      if (r < S_eff) {
        v = shift ? (r >= 1 ? dp[k] - dp0 : 0.f) : dp[k];
        if (r >= shift) v += dna;
      }
      if (r < S_pad) dbase[(size_t)d * S_pad + r] = v;
    }
  }
}

// ------------------------------------------------------------------ embedding-table gradients (BertEmbeddings)
// torch's embedding_dense_backward sorts the indices on the device and runs ~17 launches per table; its
// `sum_and_scatter` takes 0.4 - 0.6 ms per table at 24 832 tokens because one token id ([PAD], or the single token type)
// owns most rows.  Here the WORD table's segments come from the host (the trainer keeps the caption ids on the host
// anyway: a stable argsort, padding rows dropped): one workgroup per distinct token sums that token's rows in the fixed
// order of the sort - D / 4 column threads x 4 row lanes, four loads in flight per lane, lanes combined in order.  The
// TOKEN-TYPE table has two rows: both are masked column sums of dy (two-level, fixed order), out[0] = all - out[1].
constexpr int EMB_LANES = 4;
__global__ void __launch_bounds__(1024) k_embedding_bwd(const float* __restrict__ dy, const int* __restrict__ order,
                                                        const int* __restrict__ seg_lo, const int* __restrict__ seg_hi,
                                                        const int* __restrict__ seg_tok, int D, float* __restrict__ dW) {
  extern __shared__ float4 emb_red[];               // [EMB_LANES][D / 4]
  const int nc = D / 4, ct = threadIdx.x % nc, rl = threadIdx.x / nc;
  const int u = blockIdx.x, k0 = seg_lo[u], k1 = seg_hi[u];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = k0 + rl;
  for (; k + 3 * EMB_LANES < k1; k += 4 * EMB_LANES) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(dy + (size_t)order[k + j * EMB_LANES] * D + 4 * ct);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  }
  for (; k < k1; k += EMB_LANES) {
    const float4 v = *reinterpret_cast<const float4*>(dy + (size_t)order[k] * D + 4 * ct);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  emb_red[rl * nc + ct] = acc;
  __syncthreads();
  if (rl == 0) {
    float4 s = emb_red[ct];
#pragma unroll
    for (int j = 1; j < EMB_LANES; ++j) {
      const float4 t = emb_red[j * nc + ct];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4*>(dW + (size_t)seg_tok[u] * D + 4 * ct) = s;
  }
}

constexpr int TE_COLS = 256;                         // columns per workgroup: 64 threads x float4
__global__ void __launch_bounds__(256) k_type_emb_partial(const float* __restrict__ dy, const long long* __restrict__ tt, long long R,
                                                          int D, long long rows_per, float* __restrict__ part) {
  const int ct = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col0 = blockIdx.x * TE_COLS + 4 * ct;
  const long long r0 = (long long)blockIdx.y * rows_per, r1 = r0 + rows_per < R ? r0 + rows_per : R;
  float4 all = make_float4(0.f, 0.f, 0.f, 0.f), one = all;
  for (long long r = r0 + rl; r < r1; r += 4) {
    const float4 v = *reinterpret_cast<const float4*>(dy + r * D + col0);
    const float m = tt[r] != 0 ? 1.f : 0.f;
    all.x += v.x; all.y += v.y; all.z += v.z; all.w += v.w;
    one.x += m * v.x; one.y += m * v.y; one.z += m * v.z; one.w += m * v.w;
  }
  __shared__ float4 red[2][4][64];
  red[0][rl][ct] = all;
  red[1][rl][ct] = one;
  __syncthreads();
  if (rl < 2) {                                     // wave 0: class "all", wave 1: class "one"
    float4 s = red[rl][0][ct];
#pragma unroll
    for (int j = 1; j < 4; ++j) { const float4 t = red[rl][j][ct]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) part[((size_t)rl * D + col0 + i) * gridDim.y + blockIdx.y] = sv[i];
  }
}
__global__ void __launch_bounds__(256) k_type_emb_finish(const float* __restrict__ part, int n_part, int D, float* __restrict__ dW2) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= D) return;
  float a = 0.f, o = 0.f;
  for (int k = lane; k < n_part; k += 64) { a += part[(size_t)col * n_part + k]; o += part[((size_t)D + col) * n_part + k]; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); o += __shfl_xor(o, m, 64); }
  if (lane == 0) { dW2[col] = a - o; dW2[D + col] = o; }
}
int type_emb_parts(long long R, int D) {
  long long np = 1024 / (D / TE_COLS);
  const long long mx = (R + 15) / 16;
  if (np > mx) np = mx;
  return (int)(np < 1 ? 1 : np);
}

}  // namespace

extern "C" int glr_embedding_bwd(const float* dy, const int32_t* order, const int32_t* seg_lo, const int32_t* seg_hi,
                                 const int32_t* seg_tok, int n_seg, int D, float* dW, void* stream) {
  if (!dy || !order || !seg_lo || !seg_hi || !seg_tok || !dW || n_seg < 0 || D <= 0 || D % 4 != 0 || D > 1024) return GLR_EINVAL;
  if (n_seg == 0) return GLR_OK;
  const int nc = D / 4;
  hipLaunchKernelGGL(k_embedding_bwd, dim3(n_seg), dim3(nc * EMB_LANES), (size_t)EMB_LANES * nc * sizeof(float4), (hipStream_t)stream,
                     dy, order, seg_lo, seg_hi, seg_tok, D, dW);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_type_embedding_workspace_floats(long long R, int D) {
  return (R > 0 && D >= TE_COLS && D % TE_COLS == 0) ? 2 * D * type_emb_parts(R, D) : 0;
}

extern "C" int glr_type_embedding_bwd(const float* dy, const int64_t* token_type, long long R, int D, float* workspace, float* dW2,
                                      void* stream) {
  if (!dy || !token_type || !workspace || !dW2 || R <= 0 || D < TE_COLS || D % TE_COLS != 0) return GLR_EINVAL;
  const int np = type_emb_parts(R, D);
  const long long rows_per = (R + np - 1) / np;
  hipLaunchKernelGGL(k_type_emb_partial, dim3(D / TE_COLS, np), dim3(256), 0, (hipStream_t)stream, dy, (const long long*)token_type, R, D,
                     rows_per, workspace);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_type_emb_finish, dim3((D + 3) / 4), dim3(256), 0, (hipStream_t)stream, workspace, np, D, dW2);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_attn_reg_fwd(const float* amean, int B_img, int n_sent, int S_pad, int S_eff, int shift,
                                int img_offset, float* out, void* stream) {
  if (!amean || !out || B_img <= 0 || n_sent <= 0 || S_pad <= 0 || S_pad > GLR_MAX_SPAD || S_pad % 64 != 0) return GLR_EINVAL;
  if (S_eff <= shift || S_eff > S_pad || (shift != 0 && shift != 1) || img_offset < 0 || img_offset + B_img > n_sent)
    return GLR_EINVAL;
  hipLaunchKernelGGL(k_attn_reg_fwd, dim3(B_img), dim3(256), 0, (hipStream_t)stream, amean, n_sent, S_pad, S_eff, shift,
                     img_offset, out);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_attn_reg_bwd(const float* amean, int B_img, int n_sent, int S_pad, int S_eff, int shift,
                                int img_offset, const float* coef, float* damean, void* stream) {
  if (!amean || !coef || !damean || B_img <= 0 || n_sent <= 0 || S_pad <= 0 || S_pad > GLR_MAX_SPAD || S_pad % 64 != 0)
    return GLR_EINVAL;
  if (S_eff <= shift || S_eff > S_pad || (shift != 0 && shift != 1) || img_offset < 0 || img_offset + B_img > n_sent)
    return GLR_EINVAL;
  hipLaunchKernelGGL(k_attn_reg_bwd, dim3(B_img), dim3(256), 0, (hipStream_t)stream, amean, n_sent, S_pad, S_eff, shift,
                     img_offset, coef, damean);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_wordpiece_segsum_fwd(const void* const* hidden, int n_layers, int in_dtype, const int32_t* dst,
                                        float* word_emb, float* sent_emb, int B, int L, int D, int mean_layers,
                                        void* stream) {
  if (!hidden || !dst || !word_emb || !sent_emb || n_layers < 1 || n_layers > K5_MAX_LAYERS || B <= 0 || L <= 0 || D <= 0)
    return GLR_EINVAL;
  if (in_dtype != GLR_F32 && in_dtype != GLR_BF16) return GLR_EDTYPE;
  LayerPtrs hp;
  for (int l = 0; l < K5_MAX_LAYERS; ++l) hp.h[l] = l < n_layers ? hidden[l] : nullptr;
  for (int l = 0; l < n_layers; ++l) if (!hp.h[l]) return GLR_EINVAL;
  const size_t lds = (size_t)64 * (L + 1) * sizeof(float);
  if (lds > 64 * 1024) return GLR_EINVAL;
  hipLaunchKernelGGL(k_segsum_fwd, dim3((D + 63) / 64, B), dim3(256), lds, (hipStream_t)stream, hp, n_layers, in_dtype, dst,
                     word_emb, sent_emb, L, D, mean_layers ? 1.f / (float)n_layers : 1.f);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_wordpiece_segsum_bwd(const float* d_word, const float* d_sent, const int32_t* dst, void* d_hidden,
                                        int out_dtype, int B, int L, int D, int n_layers, int mean_layers,
                                        void* stream) {
  if (!dst || !d_hidden || B <= 0 || L <= 0 || D <= 0 || n_layers < 1) return GLR_EINVAL;
  if (out_dtype != GLR_F32 && out_dtype != GLR_BF16) return GLR_EDTYPE;
  const size_t lds = (size_t)64 * (L + 1) * sizeof(float);
  if (lds > 64 * 1024) return GLR_EINVAL;
  hipLaunchKernelGGL(k_segsum_bwd, dim3((D + 63) / 64, B), dim3(256), lds, (hipStream_t)stream, d_word, d_sent, dst, d_hidden,
                     out_dtype, L, D, mean_layers ? 1.f / (float)n_layers : 1.f);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_attn_sup_fwd(const float* attn, const int64_t* attn_off, const int32_t* cap_lens, int img_offset,
                                const uint8_t* labels, int B, int Hl, int Wl, int ih, int iw, float* loss_b,
                                float* dmap, void* stream) {
  if (!attn || !attn_off || !cap_lens || !labels || !loss_b || B <= 0 || Hl <= 0 || Wl <= 0 || ih <= 0 || iw <= 0)
    return GLR_EINVAL;
  const size_t lds = (size_t)(3 * ih * iw + 8) * sizeof(float);
  if (lds > 64 * 1024) return GLR_EINVAL;
  hipLaunchKernelGGL(k_attn_sup, dim3(B), dim3(256), lds, (hipStream_t)stream, attn, (const long long*)attn_off,
                     cap_lens, img_offset, labels, Hl, Wl, ih, iw, loss_b, dmap);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_cosine_fwd(const float* x1, const float* x2, int rows, int D, float eps, float* out, float* stats,
                              void* stream) {
  if (!x1 || !x2 || !out || rows <= 0 || D <= 0) return GLR_EINVAL;
  hipLaunchKernelGGL(k_cosine_fwd, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x1, x2, rows, D, eps, out,
                     stats);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_cosine_bwd(const float* x1, const float* x2, const float* stats, const float* g, int rows, int D,
                              float eps, float* dx1, float* dx2, void* stream) {
  if (!x1 || !x2 || !stats || !g || !dx1 || !dx2 || rows <= 0 || D <= 0) return GLR_EINVAL;
  hipLaunchKernelGGL(k_cosine_bwd, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x1, x2, stats, g, rows, D,
                     eps, dx1, dx2);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

// ------------------------------------------------------------------------------------------
// Input resize of the image encoder: F.interpolate(x, (299, 299), mode="bilinear", align_corners=True)
// (/root/reference/gloria/models/vision_model.py:68) fused with what autocast and channels-last do to its result in
// front of conv1 - fp32 NCHW / NHWC in, bf16 NHWC out in one pass (torch: upsample in fp32, a layout copy, a cast:
// 1.3 ms for 256 images; this: read 154 MB, write 137 MB).  Same interpolation arithmetic as torch's kernel
// (scale = (in - 1) / (out - 1), source index = scale * destination index, lambdas from its fraction).
namespace {
__global__ void __launch_bounds__(256) k_upsample_bilinear_cl(const float* __restrict__ x, long long sn, long long sc, long long sh,
                                                              long long sw, int C, int Hi, int Wi, int Ho, int Wo, float rh, float rw,
                                                              long long total, unsigned short* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // output pixel (n, oh, ow)
  if (i >= total) return;
  const int ow = (int)(i % Wo);
  const long long t = i / Wo;
  const int oh = (int)(t % Ho);
  const long long n = t / Ho;
  const float fh = rh * (float)oh, fw = rw * (float)ow;
  const int h1 = (int)fh, w1 = (int)fw;
  const int hp = h1 < Hi - 1 ? 1 : 0, wp = w1 < Wi - 1 ? 1 : 0;
  const float lh1 = fh - (float)h1, lh0 = 1.f - lh1, lw1 = fw - (float)w1, lw0 = 1.f - lw1;
  const float* b0 = x + n * sn + (long long)h1 * sh + (long long)w1 * sw;
  unsigned short* dst = y + i * C;
  for (int c = 0; c < C; ++c) {
    const float* pc = b0 + c * sc;
    const float v = lh0 * (lw0 * pc[0] + lw1 * pc[wp * sw]) + lh1 * (lw0 * pc[hp * sh] + lw1 * pc[hp * sh + wp * sw]);
    dst[c] = f2bf(v);
  }
}
}  // namespace

extern "C" int glr_upsample_bilinear_cl(const float* x, long long sn, long long sc, long long sh, long long sw, int B, int C, int Hi,
                                        int Wi, int Ho, int Wo, void* y, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GLR_EINVAL;
  const float rh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, rw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const long long total = (long long)B * Ho * Wo;
  hipLaunchKernelGGL(k_upsample_bilinear_cl, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, sn, sc, sh, sw,
                     C, Hi, Wi, Ho, Wo, rh, rw, total, (unsigned short*)y);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
