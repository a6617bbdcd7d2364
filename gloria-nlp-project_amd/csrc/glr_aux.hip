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
struct LayerPtrs {
  const void* h[4];
};

// grid (D/64, B), 256 threads = 64 features x 4 token phases.
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
  if (ph == 0) {
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
    word_emb[((size_t)b * D + d0 + dl) * L + w] = tile[dl * LP + w];
  }
  if (threadIdx.x < 64) {                          // sentence embedding: mean over ALL L slots (text_model.py:110)
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
    tile[dl * LP + w] = d_word ? d_word[((size_t)b * D + d0 + dl) * L + w] : 0.f;
  }
  __syncthreads();
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

}  // namespace

extern "C" int glr_wordpiece_segsum_fwd(const void* const* hidden, int n_layers, int in_dtype, const int32_t* dst,
                                        float* word_emb, float* sent_emb, int B, int L, int D, int mean_layers,
                                        void* stream) {
  if (!hidden || !dst || !word_emb || !sent_emb || n_layers < 1 || n_layers > 4 || B <= 0 || L <= 0 || D % 64 != 0)
    return GLR_EINVAL;
  if (in_dtype != GLR_F32 && in_dtype != GLR_BF16) return GLR_EDTYPE;
  LayerPtrs hp;
  for (int l = 0; l < 4; ++l) hp.h[l] = l < n_layers ? hidden[l] : nullptr;
  for (int l = 0; l < n_layers; ++l) if (!hp.h[l]) return GLR_EINVAL;
  const size_t lds = (size_t)64 * (L + 1) * sizeof(float);
  if (lds > 64 * 1024) return GLR_EINVAL;
  hipLaunchKernelGGL(k_segsum_fwd, dim3(D / 64, B), dim3(256), lds, (hipStream_t)stream, hp, n_layers, in_dtype, dst,
                     word_emb, sent_emb, L, D, mean_layers ? 1.f / (float)n_layers : 1.f);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_wordpiece_segsum_bwd(const float* d_word, const float* d_sent, const int32_t* dst, void* d_hidden,
                                        int out_dtype, int B, int L, int D, int n_layers, int mean_layers,
                                        void* stream) {
  if (!dst || !d_hidden || B <= 0 || L <= 0 || D % 64 != 0 || n_layers < 1) return GLR_EINVAL;
  if (out_dtype != GLR_F32 && out_dtype != GLR_BF16) return GLR_EDTYPE;
  const size_t lds = (size_t)64 * (L + 1) * sizeof(float);
  if (lds > 64 * 1024) return GLR_EINVAL;
  hipLaunchKernelGGL(k_segsum_bwd, dim3(D / 64, B), dim3(256), lds, (hipStream_t)stream, d_word, d_sent, dst, d_hidden,
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
