// K1: fused region x word attention + cosine similarity + log-sum-exp (forward).
//
// Replaces, for every (image b, sentence i) pair at once, the body of the reference sentence
// loop: attention_fn (/root/reference/gloria/loss/gloria_loss.py:19-63), cosine_similarity
// (:11-16) and the exp/sum/log of local_loss (:150-158, :164).
//
// One workgroup (8 waves) = one image b x one tile of 64 word slots (whole sentences):
//
//   phase 1  scores^T[w, r] = sum_d T[w, d] V[r, d]          MFMA 32x32, K = D
//            A = packed words  tp [slot][d]   (64 rows), B^T = vt[b] [region][d] (S_pad rows);
//            both streamed HBM/L2 -> registers -> LDS in K chunks, the next chunk's global
//            loads are issued before the current chunk's MFMAs (issue-early / write-late).
//   phase 2  scores -> LDS fp32 tile; one thread per region column walks the words of each
//            sentence: softmax over the sentence's words (gloria_loss.py:42-43), * temp1,
//            exp for the softmax over regions (:51-52, normalised later by Z_w); writes the
//            e2 image [word][region] in the operand dtype (B^T operand of phase 3).
//            Z_w = sum_r e2[w, r] is summed from the image as stored, so the weights used by
//            the MFMA and their normaliser agree exactly.
//   phase 3  ctx~[d, w] = sum_r V[d, r] e2[w, r]              MFMA 32x32, K = S_pad
//            A = vd[b] [d][region] streamed in K chunks (two passes of 384 features),
//            epilogue straight from the accumulators: |ctx~_w|^2 and <T_w, ctx~_w>.
//   phase 4  cos_w = <T_w, c_w> / max(|T_w||c_w|, eps), c_w = ctx~_w / Z_w (:13-16, :150);
//            per sentence: sim[b, i] = temp3 * log(agg_w exp(temp2 cos_w)) (:153-164).
//            Optional outputs for the diagonal pair: attention map (:141-143) and the
//            weighted context (:59).
//
// Sentences longer than one tile (n > 64 words, e.g. the 96-word reports or the 256-word
// stress case) own ceil(n/64) consecutive tiles and are handled by the workgroup of their first
// tile in two sweeps: sweep 0 runs phase 1 over every sub-tile and keeps running (max, sum)
// statistics of the word softmax per region; sweep 1 repeats phase 1 per sub-tile and finishes
// phases 2-4 with those statistics, accumulating the per-sentence aggregate across sub-tiles.
//
// Wave w: wm = w & 1 picks the 32-word block, wg = w >> 1 the group of 32-row blocks
// {wg, wg+4, wg+8} of the big operand (regions in phase 1, features in phase 3).
#include "glr_common.h"

namespace {

struct LaParams {
  const unsigned char* vt;
  const unsigned char* vd;
  const unsigned char* tp;
  const float* tnorm;
  const int* sent_slot0;
  const int* cap_lens;
  const int* tile_first;
  const int* order;
  const int* tile_nsub;
  int n_tiles, n_sent, B_img, D, S_eff, S_pad;
  float temp1, temp2, temp3;
  int agg;
  float eps;
  float* sim;
  int ld_sim;
  float* attn;
  const long long* attn_off;
  int strip;
  float* wctx;
  int ld_wctx;
  int pair_only, img_offset;
  // LDS carve (bytes)
  int off_stage, off_sc, off_e2, off_small;
};

constexpr int TW = GLR_TILE_WORDS;  // 64 word slots per tile
constexpr int NTHR = 512;

template <typename O>
__global__ void __launch_bounds__(NTHR) k_local_attn_fwd(LaParams p) {
  constexpr int ESZ = O::ESZ, CB = O::CB, PITCH = CB + 16, PPR = CB / 16, KSTEPS = CB / 32;
  constexpr int NPMAX = ((GLR_MAX_SPAD + TW) * PPR + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wg = wave >> 1;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- block -> (image, tile).  Blocks that share blockIdx % 8 share an XCD (speed only):
  // all tiles of one image go to one XCD so vt[b]/vd[b] stay in that XCD's L2.
  int b, tile0;
  if (p.pair_only) {
    b = blockIdx.x;
    tile0 = p.sent_slot0[p.img_offset + b] / TW;
  } else {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    b = (q / p.n_tiles) * 8 + xcd;
    tile0 = q % p.n_tiles;
    if (b >= p.B_img) return;
  }
  int nsub = p.tile_nsub[tile0];      // 0: ordinary tile, k > 1: head of a k-tile sentence, < 0: continuation
  if (nsub < 0) return;
  if (nsub == 0) nsub = 1;

  const int S_pad = p.S_pad, D = p.D;
  const int SCP = S_pad + 4;                 // fp32 score tile pitch (floats)
  const int E2P = S_pad * ESZ + 16;          // e2 image pitch (bytes)
  const int nrb = S_pad >> 5;                // 32-region blocks

  unsigned char* stage = smem + p.off_stage;
  float* sc = reinterpret_cast<float*>(smem + p.off_sc);
  unsigned char* e2 = smem + p.off_e2;
  int* seg_w0 = reinterpret_cast<int*>(smem + p.off_small);
  int* seg_n = seg_w0 + TW;
  int* seg_sent = seg_n + TW;
  float* zsum = reinterpret_cast<float*>(seg_sent + TW);
  float* red = zsum + TW;        // [2][4][TW]
  float* exs = red + 8 * TW;     // [TW]
  int* diag = reinterpret_cast<int*>(exs + TW);   // [0] = w0, [1] = n of the diagonal sentence
  float* aggv = reinterpret_cast<float*>(diag + 2);   // running aggregate of a multi-tile sentence
  float* mrun = aggv + 2;                         // [S_pad] running max   (multi-tile sentences)
  float* srun = mrun + GLR_MAX_SPAD;              // [S_pad] running sum

  const int seg_first = p.tile_first[tile0];
  const int long_sent = p.order[seg_first];
  const int long_n = p.cap_lens[long_sent];
  const size_t rowbytes1 = (size_t)D * ESZ;
  const unsigned char* vt_b = p.vt + (size_t)b * S_pad * rowbytes1;
  if (nsub > 1) {
    if (tid < S_pad) { mrun[tid] = -INFINITY; srun[tid] = 0.f; }
    if (tid == 0) aggv[0] = 0.f;
  }

  uint4 pre[NPMAX];
  f32x16 acc[3];

  for (int sweep = (nsub > 1 ? 0 : 1); sweep < 2; ++sweep)
  for (int sub = 0; sub < nsub; ++sub) {
  const int tile = tile0 + sub;
  __syncthreads();          // previous iteration's LDS readers are done
  int nseg;
  if (nsub > 1) {
    nseg = 1;
    if (tid == 0) { seg_sent[0] = long_sent; seg_w0[0] = 0; seg_n[0] = min(TW, long_n - sub * TW); }
  } else {
    nseg = p.tile_first[tile + 1] - seg_first;
    if (tid < nseg) {
      const int sent = p.order[seg_first + tid];
      seg_sent[tid] = sent;
      seg_w0[tid] = p.sent_slot0[sent] - tile * TW;
      seg_n[tid] = p.cap_lens[sent];
    }
  }
  if (tid == 0) { diag[0] = 0; diag[1] = 0; }
  const unsigned char* tp_tile = p.tp + (size_t)tile * TW * rowbytes1;

  // ================= phase 1: scores^T = T . V^T =================
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  const int rows1 = S_pad + TW;
  const int nch1 = (int)(rowbytes1 / CB);

#define G1_ISSUE(c)                                                                          \
  _Pragma("unroll") for (int i = 0; i < NPMAX; ++i) {                                        \
    const int idx = tid + i * NTHR;                                                          \
    const int row = idx / PPR, pc = idx % PPR;                                               \
    if (row < rows1) {                                                                       \
      const unsigned char* src = (row < TW) ? (tp_tile + (size_t)row * rowbytes1)            \
                                            : (vt_b + (size_t)(row - TW) * rowbytes1);       \
      pre[i] = *reinterpret_cast<const uint4*>(src + (size_t)(c) * CB + pc * 16);            \
    }                                                                                        \
  }
#define ST_WRITE(nrows)                                                                      \
  _Pragma("unroll") for (int i = 0; i < NPMAX; ++i) {                                        \
    const int idx = tid + i * NTHR;                                                          \
    const int row = idx / PPR, pc = idx % PPR;                                               \
    if (row < (nrows)) *reinterpret_cast<uint4*>(stage + row * PITCH + pc * 16) = pre[i];    \
  }

  G1_ISSUE(0)
  ST_WRITE(rows1)
  __syncthreads();
  for (int c = 0; c < nch1; ++c) {
    if (c + 1 < nch1) { G1_ISSUE(c + 1) }
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      const typename O::frag a = O::ld(stage + (wm * 32 + l31) * PITCH + kk * 32 + h * 16);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int rb = wg + 4 * j;
        if (rb < nrb) {
          const typename O::frag bb = O::ld(stage + (TW + rb * 32 + l31) * PITCH + kk * 32 + h * 16);
          O::mma(a, bb, acc[j]);
        }
      }
    }
    __syncthreads();
    if (c + 1 < nch1) {
      ST_WRITE(rows1)
      __syncthreads();
    }
  }

  // scores -> LDS fp32 tile sc[word][region] (may alias the staging buffer: all reads are done)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int rb = wg + 4 * j;
    if (rb < nrb) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int word = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        sc[word * SCP + rb * 32 + l31] = acc[j][q];
      }
    }
  }
  __syncthreads();

  // ================= phase 2: the two softmaxes, one thread per region column =================
  if (sweep == 0) {
    // multi-tile sentence, statistics sweep: fold this sub-tile into the running (max, sum)
    if (tid < S_pad) {
      const int r = tid, n = seg_n[0];
      float m = mrun[r];
      for (int w = 0; w < n; ++w) m = fmaxf(m, sc[w * SCP + r]);
      float sum = srun[r] * __expf(mrun[r] - m);
      for (int w = 0; w < n; ++w) sum += __expf(sc[w * SCP + r] - m);
      mrun[r] = m;
      srun[r] = sum;
    }
    continue;            // next sub-tile (the loop head synchronises)
  }
  if (tid < S_pad) {
    const int r = tid;
    const bool live = r < p.S_eff;
    int used = 0;
    for (int s = 0; s < nseg; ++s) {
      const int w0 = seg_w0[s], n = seg_n[s];
      float m, sum;
      if (nsub > 1) {
        m = mrun[r];
        sum = srun[r];
        for (int w = 0; w < n; ++w) sc[(w0 + w) * SCP + r] = __expf(sc[(w0 + w) * SCP + r] - m);
      } else {
        m = -INFINITY;
        for (int w = 0; w < n; ++w) m = fmaxf(m, sc[(w0 + w) * SCP + r]);
        sum = 0.f;
        for (int w = 0; w < n; ++w) {
          const float e = __expf(sc[(w0 + w) * SCP + r] - m);
          sc[(w0 + w) * SCP + r] = e;
          sum += e;
        }
      }
      const float scale = p.temp1 / sum;     // temp1 * softmax_w
      for (int w = 0; w < n; ++w) {
        const float v = live ? __expf(sc[(w0 + w) * SCP + r] * scale) : 0.f;
        O::from_f32(e2 + (w0 + w) * E2P + r * ESZ, v);
      }
      used = w0 + n;
    }
    for (int w = used; w < TW; ++w) O::from_f32(e2 + w * E2P + r * ESZ, 0.f);
  }
  // which segment is the diagonal pair (image b <-> sentence img_offset + b)?
  if (tid < nseg && seg_sent[tid] == p.img_offset + b) { diag[0] = seg_w0[tid]; diag[1] = seg_n[tid]; }
  const int wbase = (nsub > 1) ? sub * TW : 0;   // index of slot 0's word inside its sentence (multi-tile)
  __syncthreads();

  // Z_w = sum_r e2[w, r], 8 threads per word
  {
    const int word = tid >> 3, part = tid & 7;
    const int span = S_pad >> 3;
    float z = 0.f;
    for (int r = part * span; r < (part + 1) * span; ++r) z += O::to_f32(e2 + word * E2P + r * ESZ);
    z += __shfl_xor(z, 1, 64);
    z += __shfl_xor(z, 2, 64);
    z += __shfl_xor(z, 4, 64);
    if (part == 0) zsum[word] = z;
  }
  __syncthreads();

  // ================= phase 3: ctx~ = V . e2^T, epilogue in registers =================
  const size_t rowbytes2 = (size_t)S_pad * ESZ;
  const unsigned char* vd_b = p.vd + (size_t)b * D * rowbytes2;
  const int nch2 = (int)(rowbytes2 / CB);
  const int myword = wm * 32 + l31;
  const size_t myslot = (size_t)tile * TW + myword;
  const int dw0 = diag[0], dn = diag[1];
  const float myz = zsum[myword];
  const float inv_z = myz > 0.f ? 1.f / myz : 0.f;
  float nrm = 0.f, dt = 0.f;

  for (int d0 = 0; d0 < D; d0 += 384) {
    const int rows2 = min(384, D - d0);
    const int ndb = rows2 >> 5;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

#define G2_ISSUE(c)                                                                          \
  _Pragma("unroll") for (int i = 0; i < NPMAX; ++i) {                                        \
    const int idx = tid + i * NTHR;                                                          \
    const int row = idx / PPR, pc = idx % PPR;                                               \
    if (row < rows2)                                                                         \
      pre[i] = *reinterpret_cast<const uint4*>(vd_b + (size_t)(d0 + row) * rowbytes2 +       \
                                               (size_t)(c) * CB + pc * 16);                  \
  }
    G2_ISSUE(0)
    ST_WRITE(rows2)
    __syncthreads();
    for (int c = 0; c < nch2; ++c) {
      if (c + 1 < nch2) { G2_ISSUE(c + 1) }
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const typename O::frag bb = O::ld(e2 + myword * E2P + c * CB + kk * 32 + h * 16);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int db = wg + 4 * j;
          if (db < ndb) {
            const typename O::frag a = O::ld(stage + (db * 32 + l31) * PITCH + kk * 32 + h * 16);
            O::mma(a, bb, acc[j]);
          }
        }
      }
      __syncthreads();
      if (c + 1 < nch2) {
        ST_WRITE(rows2)
        __syncthreads();
      }
    }
    // epilogue of this pass: acc[j][q] = ctx~[d, myword], d = d0 + db*32 + (q&3) + 8*(q>>2) + 4h
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int db = wg + 4 * j;
      if (db < ndb) {
#pragma unroll
        for (int qg = 0; qg < 4; ++qg) {
          const int d = d0 + db * 32 + 8 * qg + 4 * h;
          float t[4];
          if (ESZ == 4) {
            const float4 tv = *reinterpret_cast<const float4*>(p.tp + (myslot * D + d) * 4);
            t[0] = tv.x; t[1] = tv.y; t[2] = tv.z; t[3] = tv.w;
          } else {
            const uint2 tv = *reinterpret_cast<const uint2*>(p.tp + (myslot * D + d) * 2);
            t[0] = bf2f((unsigned short)(tv.x & 0xffff)); t[1] = bf2f((unsigned short)(tv.x >> 16));
            t[2] = bf2f((unsigned short)(tv.y & 0xffff)); t[3] = bf2f((unsigned short)(tv.y >> 16));
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = acc[j][qg * 4 + k];
            nrm += v * v;
            dt += v * t[k];
          }
          if (p.wctx != nullptr && myword >= dw0 && myword < dw0 + dn) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              p.wctx[((size_t)b * D + d + k) * p.ld_wctx + (wbase + myword - dw0)] = acc[j][qg * 4 + k] * inv_z;
          }
        }
      }
    }
  }

  nrm += __shfl_xor(nrm, 32, 64);
  dt += __shfl_xor(dt, 32, 64);
  if (h == 0) {
    red[(0 * 4 + wg) * TW + myword] = nrm;
    red[(1 * 4 + wg) * TW + myword] = dt;
  }
  __syncthreads();

  // ================= phase 4: cosine, per-sentence aggregate =================
  if (tid < TW) {
    const float n2 = red[tid] + red[TW + tid] + red[2 * TW + tid] + red[3 * TW + tid];
    const float dd = red[4 * TW + tid] + red[5 * TW + tid] + red[6 * TW + tid] + red[7 * TW + tid];
    const float z = zsum[tid];
    float cosv = 0.f;
    if (z > 0.f) {
      const float iz = 1.f / z;
      const float cn = sqrtf(n2) * iz;                       // |c_w|
      const float den = fmaxf(p.tnorm[(size_t)tile * TW + tid] * cn, p.eps);
      cosv = dd * iz / den;
    }
    exs[tid] = __expf(p.temp2 * cosv);
  }
  __syncthreads();
  if (tid < nseg) {
    const int w0 = seg_w0[tid], n = seg_n[tid];
    float v = 0.f;
    if (p.agg == GLR_AGG_MAX) {
      for (int w = 0; w < n; ++w) v = fmaxf(v, exs[w0 + w]);
    } else {
      for (int w = 0; w < n; ++w) v += exs[w0 + w];
    }
    bool emit = true;
    int ntot = n;
    if (nsub > 1) {        // accumulate across the sub-tiles of a multi-tile sentence (tid == 0 only)
      v = (p.agg == GLR_AGG_MAX) ? fmaxf(v, aggv[0]) : v + aggv[0];
      aggv[0] = v;
      emit = (sub == nsub - 1);
      ntot = long_n;
    }
    if (p.agg == GLR_AGG_MEAN) v /= (float)ntot;
    const int sent = seg_sent[tid];
    if (emit && (!p.pair_only || sent == p.img_offset + b))
      p.sim[(size_t)b * p.ld_sim + sent] = p.temp3 * __logf(v);
  }
  // attention map of the diagonal pair: a2[w, r] = e2[w, r] / Z_w, no-attention column stripped
  if (p.attn != nullptr && dn > 0) {
    const int sout = p.S_eff - p.strip;
    float* out = p.attn + p.attn_off[p.img_offset + b] + (size_t)wbase * sout;
    for (int idx = tid; idx < dn * sout; idx += NTHR) {
      const int w = idx / sout, r = idx % sout + p.strip;
      out[idx] = O::to_f32(e2 + (dw0 + w) * E2P + r * ESZ) / zsum[dw0 + w];
    }
  }
  }  // sub / sweep loops
#undef G1_ISSUE
#undef G2_ISSUE
#undef ST_WRITE
}

}  // namespace

extern "C" int glr_region_pad(int s_eff) { return (s_eff + 63) / 64 * 64; }

extern "C" int glr_local_attn_fwd(const void* vt, const void* vd, const void* tp, const float* tnorm,
                                  const int32_t* sent_slot0, const int32_t* cap_lens,
                                  const int32_t* tile_first, const int32_t* order,
                                  const int32_t* tile_nsub, int n_tiles, int n_sent, int B_img, int D, int S_eff, float temp1, float temp2, float temp3, int agg,
                                  float eps, float* sim, int ld_sim, float* attn, const int64_t* attn_off,
                                  int strip, float* wctx, int ld_wctx, int pair_only, int img_offset,
                                  int op_dtype, void* stream) {
  if (!vt || !vd || !tp || !tnorm || !sent_slot0 || !cap_lens || !tile_first || !order || !tile_nsub || !sim) return GLR_EINVAL;
  if (op_dtype != GLR_F32 && op_dtype != GLR_BF16) return GLR_EDTYPE;
  if (n_tiles <= 0 || n_sent <= 0 || B_img <= 0 || S_eff <= 0 || D <= 0) return GLR_EINVAL;
  if (D % 64 != 0) return GLR_EINVAL;
  const int S_pad = glr_region_pad(S_eff);
  if (S_pad > GLR_MAX_SPAD) return GLR_EINVAL;
  if (agg < 0 || agg > 2) return GLR_EINVAL;
  if (attn && !attn_off) return GLR_EINVAL;
  if (pair_only && img_offset + B_img > n_sent) return GLR_EINVAL;

  LaParams p;
  p.vt = (const unsigned char*)vt; p.vd = (const unsigned char*)vd; p.tp = (const unsigned char*)tp;
  p.tnorm = tnorm; p.sent_slot0 = sent_slot0; p.cap_lens = cap_lens; p.tile_first = tile_first; p.order = order; p.tile_nsub = tile_nsub;
  p.n_tiles = n_tiles; p.n_sent = n_sent; p.B_img = B_img; p.D = D; p.S_eff = S_eff; p.S_pad = S_pad;
  p.temp1 = temp1; p.temp2 = temp2; p.temp3 = temp3; p.agg = agg; p.eps = eps;
  p.sim = sim; p.ld_sim = ld_sim; p.attn = attn; p.attn_off = (const long long*)attn_off; p.strip = strip;
  p.wctx = wctx; p.ld_wctx = ld_wctx; p.pair_only = pair_only; p.img_offset = img_offset;

  const int esz = op_dtype == GLR_F32 ? 4 : 2;
  const int pitch = (op_dtype == GLR_F32 ? OpF32::CB : OpBF16::CB) + 16;
  const int stage_rows = max(S_pad + TW, min(D, 384));
  const int stage_bytes = stage_rows * pitch;
  const int sc_bytes = TW * (S_pad + 4) * 4;
  const int e2_bytes = TW * (S_pad * esz + 16);
  const int small_bytes = 8192;   // segment table, reductions, multi-tile running stats
  if (op_dtype == GLR_BF16) {
    p.off_stage = 0; p.off_sc = 0;
    p.off_e2 = max(stage_bytes, sc_bytes);
    p.off_small = p.off_e2 + e2_bytes;
  } else {  // fp32: e2 overwrites the score tile in place (same pitch), staging must not alias it
    p.off_sc = 0; p.off_e2 = 0;
    p.off_stage = sc_bytes;
    p.off_small = p.off_stage + stage_bytes;
  }
  const int lds = p.off_small + small_bytes;
  if (lds > 160 * 1024) return GLR_EINVAL;

  const int grid = pair_only ? B_img : ((B_img + 7) / 8) * 8 * n_tiles;
  hipStream_t st = (hipStream_t)stream;
  if (op_dtype == GLR_BF16) {
    if (hipFuncSetAttribute((const void*)k_local_attn_fwd<OpBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return GLR_ELAUNCH;
    hipLaunchKernelGGL(k_local_attn_fwd<OpBF16>, dim3(grid), dim3(NTHR), lds, st, p);
  } else {
    if (hipFuncSetAttribute((const void*)k_local_attn_fwd<OpF32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return GLR_ELAUNCH;
    hipLaunchKernelGGL(k_local_attn_fwd<OpF32>, dim3(grid), dim3(NTHR), lds, st, p);
  }
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
