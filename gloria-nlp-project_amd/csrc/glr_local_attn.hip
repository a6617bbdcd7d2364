// K1: fused region x word attention + cosine similarity + log-sum-exp, forward AND backward.
//
// Replaces, for every (image b, sentence i) pair at once, the body of the reference sentence loop:
// attention_fn (/root/reference/gloria/loss/gloria_loss.py:19-63), cosine_similarity (:11-16) and the
// exp/sum/log of local_loss (:150-158, :164) - and, in the backward variant, autograd through them.
//
// Maths, per image b and word w of sentence i, regions r:
//   s[w,r]  = <T_w, V_r>                          a1 = softmax over the words of the sentence
//   e2[w,r] = exp(temp1 * a1[w,r])                Z_w = sum_r e2,   a2 = e2 / Z_w
//   c_w     = sum_r a2[w,r] V_r                   dot_w = <T_w, c_w> = sum_r a2[w,r] s[w,r]
//   |c_w|^2 = sum_r a2[w,r] u[w,r],  u[w,r] = <V_r, c_w> = sum_r' a2[w,r'] G[r,r'],  G = V^T V (Gram)
// so the weighted context itself is never formed: the second contraction runs against the
// S x S Gram matrix of the image (K = S instead of K = D) and both cosine ingredients are
// region-sums of products that already sit in the MFMA accumulators.
//
// One workgroup (8 waves, 512 threads) = one image b x one tile of 64 word slots (whole sentences).
//   P1  acc[w,r] = T_tile . V_b^T            MFMA 32x32, K = D.  T tile (64 rows) and vt[b] (S_pad
//       rows, [region][feature]) stream HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into a
//       2-deep ring; the LDS image is lane-linear, bank conflicts are removed by an XOR swizzle of
//       the 16-byte slot applied on the SOURCE address and again on the fragment read.
//   P2  forward: scores -> LDS fp32 tile; one thread per region column walks the words of each
//       sentence for log-sum-exp statistics lse[r, sentence] (kept for backward).
//       Then, from the scores still in registers: a1 = exp(s - lse), e2 = exp(temp1 a1) -> LDS image
//       [word][region] in the operand dtype; per-word Z and dot~ by butterfly reductions.
//       (backward: lse comes from the forward, no walk.)
//   P3  acc[w,r] (+)= E . G_b^T              MFMA 32x32, K = S_pad, A operand = the LDS image,
//       gram[b] rows streamed through the same ring.
//   P4  forward: |c|^2 from acc, cosine, per-sentence aggregate -> sim[b, i]; optional diagonal
//       attention maps.
//       backward: with per-word scalars alpha, beta, kappa derived from dsim and the saved stats the
//       accumulator is initialised with -alpha*s and the image holds beta*a2, so P3 leaves -da2;
//       softmax backward over regions (needs kappa only) and over words (segment sums by LDS
//       atomics) gives ds; outputs X = ds + alpha*a2, beta*a2 and a2 for the gradient GEMMs.
//
// Sentences longer than one tile own ceil(n/64) consecutive tiles, processed by the workgroup of
// their first tile in two sweeps (statistics, then results).
//
// Wave w: wm = w & 1 -> 32-word block; wg = w >> 1 -> region blocks {wg, wg+4, wg+8}.
// Accumulator element q of a block: word row (q&3) + 8*(q>>2) + 4*(lane>>5), region column lane&31.
#include "glr_common.h"

extern "C" int glr_region_pad(int s_eff) { return (s_eff + 63) / 64 * 64; }
// populated word slots per 64-slot tile: the fp32 mode keeps a 32-word score tile + fp32 image in LDS
extern "C" int glr_tile_capacity(int op_dtype) { return op_dtype == GLR_F32 ? 32 : GLR_TILE_WORDS; }

namespace {

constexpr int TW = GLR_TILE_WORDS;  // 64 word slots per tile
constexpr int NTHR = 512;
constexpr int WSTAT = 4;            // floats per (image, slot) saved by forward: Z, cos, |c|^2, unused

struct LaParams {
  const unsigned char* vt;      // [B_img][S_pad][D]
  const unsigned char* gram;    // [B_img][S_pad][S_pad]
  const unsigned char* tp;      // [n_slots][D]
  const float* tnorm;           // [n_slots]
  const int* sent_slot0;
  const int* cap_lens;
  const int* tile_first;
  const int* order;
  const int* tile_nsub;
  int n_tiles, n_sent, n_slots, B_img, D, S_eff, S_pad;
  int tw;                       // populated word slots per tile: 64 (bf16) or 32 (fp32, LDS budget)
  float temp1, temp2, temp3;
  int agg;
  float eps;
  float* sim;                   // [B_img][ld_sim]   (fwd: out, bwd: in)
  int ld_sim;
  float* lse;                   // [B_img][n_sent][S_pad]  (fwd: optional out, bwd: in)
  float* wstat;                 // [B_img][n_slots][WSTAT] (fwd: optional out, bwd: in)
  float* attn;                  // fwd optional out
  const long long* attn_off;
  int strip;
  int pair_only, img_offset;
  // backward only
  const float* dsim;            // [B_img][ld_sim]
  unsigned char* xout;          // [n_slots][B_img][S_pad] op dtype
  unsigned char* bout;          // [B_img][n_slots][S_pad] beta*a2
  unsigned char* aout;          // [B_img][n_slots][S_pad] a2
  float* gamma;                 // [B_img][n_slots]
  // LDS carve (bytes)
  int off_img, off_small;
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
// one LDS-DMA piece: lane i writes 16 B at lds_dst + 16*i, read from its own gsrc (asm: hipcc's
// waitcnt pass would otherwise drain every in-flight DMA before each ds_read)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_dyn(int n) {   // n is wave-uniform
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    default: wait_vm<7>(); break;
  }
}
__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): own LDS traffic done (vmcnt untouched)
  __builtin_amdgcn_s_barrier();
}

// acc[j] (+)= A(64 words x K) . Bt(brows x K)^T for this wave's blocks.
//   A_RES = false: A rows stream from `asrc` (64 rows, pitch apitch bytes) together with B.
//   A_RES = true : A fragments come from the LDS image `aimg` ([word][k], pitch aimg_pitch bytes).
// B rows stream from `bsrc` (brows rows, pitch bpitch bytes).  K bytes = nchunk * CB.
template <typename O, bool A_RES>
__device__ __forceinline__ void stream_gemm(f32x16 (&acc)[3], unsigned char* ring, int buf_bytes,
                                            const unsigned char* asrc, size_t apitch,
                                            const unsigned char* bsrc, size_t bpitch, int brows, int nchunk,
                                            const unsigned char* aimg, int aimg_pitch, int wave, int lane, int wm,
                                            int wg, int nrb, int tw) {
  constexpr int CB = O::CB, PPR = CB / 16, RPB = 256 / CB, KSTEPS = CB / 32, RPI = 64 / PPR;
  constexpr int NPW = ((GLR_MAX_SPAD + TW) / RPI + 7) / 8;   // DMA pieces per wave and chunk (upper bound)
  const int l31 = lane & 31, h = lane >> 5;
  const int arows = A_RES ? 0 : tw;
  const bool active = wm * 32 < tw;                    // fp32 tiles hold 32 words: odd waves only move data
  const int winstr = (arows + brows) / RPI;            // 1-KiB DMA pieces per chunk
  const int nw = (winstr - wave + 7) / 8;              // pieces this wave issues per chunk
  const unsigned ring_lds = lds_addr(ring);
  const int prow = lane / PPR, pslot = lane % PPR;
  int koff[KSTEPS];                                    // byte offset of k-step kk inside a swizzled row
#pragma unroll
  for (int kk = 0; kk < KSTEPS; ++kk) koff[kk] = ((kk * 2 + h) ^ ((l31 / RPB) & (PPR - 1))) * 16;

  auto issue = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int k = wave + 8 * i;
      if (k < winstr) {
        const int row = k * RPI + prow;
        const int g = pslot ^ ((row / RPB) & (PPR - 1));
        const unsigned char* src = (row < arows) ? (asrc + (size_t)row * apitch)
                                                 : (bsrc + (size_t)(row - arows) * bpitch);
        glds16(src + (size_t)c * CB + g * 16, ring_lds + buf * buf_bytes + k * 1024);
      }
    }
  };

  issue(0, 0);
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) {
      issue(c + 1, (c + 1) & 1);
      wait_vm_dyn(nw);                                   // chunk c landed, chunk c+1 may be in flight
    } else {
      wait_vm<0>();
    }
    wg_barrier();
    const unsigned char* rb = ring + (c & 1) * buf_bytes;
    if (active) {
      // every row this lane reads is (multiple of 16) + l31, so its swizzle term depends on l31 only;
      // fragment addresses are one lane base + koff[kk] + an immediate row-block offset
      const unsigned char* bb0 = rb + (arows + wg * 32 + l31) * CB;
      const unsigned char* aa0 = A_RES ? (aimg + (wm * 32 + l31) * aimg_pitch + c * CB + h * 16)
                                       : (rb + (wm * 32 + l31) * CB);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const typename O::frag a = A_RES ? O::ld(aa0 + kk * 32) : O::ld(aa0 + koff[kk]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (wg + 4 * j < nrb) {
            const typename O::frag bb = O::ld(bb0 + koff[kk] + j * (4 * 32 * CB));
            O::mma(a, bb, acc[j]);
          }
        }
      }
    }
    wg_barrier();                                        // ring buffer (c&1) free for chunk c+2
  }
}

__device__ __forceinline__ float half_sum32(float v) {   // sum over the 32 lanes sharing lane>>5
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename O, bool BWD>
__global__ void __launch_bounds__(NTHR) k_local_attn(LaParams p) {
  constexpr int ESZ = O::ESZ, CB = O::CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wg = wave >> 1;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- block -> (image, tile).  Blocks that share blockIdx % 8 share an XCD (speed only):
  // all tiles of one image go to one XCD so vt[b] / gram[b] stay in that XCD's L2.
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
  // LDS tile pitches are compile-time constants (sized for GLR_MAX_SPAD) so that every per-element
  // LDS address is one lane-dependent base + an immediate offset
  constexpr int SCP = GLR_MAX_SPAD;                  // fp32 score tile pitch (floats)
  constexpr int IMP = GLR_MAX_SPAD * ESZ + 16;       // LDS image pitch (bytes)
  const int nrb = S_pad >> 5;
  const int tw = p.tw;
  const bool wactive = wm * 32 < tw;

  unsigned char* ring = smem;                // ring buffers / score tile / rho share [0, off_img)
  float* sc = reinterpret_cast<float*>(smem);
  unsigned char* img = smem + p.off_img;
  int* seg_w0 = reinterpret_cast<int*>(smem + p.off_small);
  int* seg_n = seg_w0 + TW;
  int* seg_sent = seg_n + TW;
  int* wseg = seg_sent + TW;                          // [TW] segment of each word slot (-1 = empty)
  float* red = reinterpret_cast<float*>(wseg + TW);   // [3][4][TW]  (bwd: al, be, ka, zi per word)
  float* zsum = red + 12 * TW;                        // [TW]
  float* exs = zsum + TW;                             // [TW]
  int* diag = reinterpret_cast<int*>(exs + TW);       // [0] = w0, [1] = n of the diagonal sentence
  float* aggv = reinterpret_cast<float*>(diag + 2);
  float* mrun = aggv + 2;                             // [S_pad] running max / lse (multi-tile)
  float* srun = mrun + GLR_MAX_SPAD;                  // [S_pad] running sum / rho (multi-tile bwd)
  float* w_al = red;                                  // bwd per-word scalars
  float* w_be = red + TW;
  float* w_ka = red + 2 * TW;
  float* w_zi = red + 3 * TW;

  const int seg_first = p.tile_first[tile0];
  const int long_sent = p.order[seg_first];
  const int long_n = p.cap_lens[long_sent];
  const size_t rowbytes1 = (size_t)D * ESZ;
  const size_t rowbytes2 = (size_t)S_pad * ESZ;
  const unsigned char* vt_b = p.vt + (size_t)b * S_pad * rowbytes1;
  const unsigned char* gram_b = p.gram + (size_t)b * S_pad * rowbytes2;
  const int buf1 = (tw + S_pad) * CB, buf2 = S_pad * CB;
  const int nch1 = (int)(rowbytes1 / CB), nch2 = (int)(rowbytes2 / CB);
  if (nsub > 1) {
    if (tid < S_pad) { mrun[tid] = -INFINITY; srun[tid] = 0.f; }
    if (tid == 0) aggv[0] = 0.f;
  }

  f32x16 acc[3];
  f32x16 a1r[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) a1r[j][q] = 0.f;

  for (int sweep = (nsub > 1 ? 0 : 1); sweep < 2; ++sweep)
  for (int sub = 0; sub < nsub; ++sub) {
    const int tile = tile0 + sub;
    __syncthreads();          // previous iteration's LDS readers are done
    int nseg;
    if (nsub > 1) {
      nseg = 1;
      if (tid == 0) { seg_sent[0] = long_sent; seg_w0[0] = 0; seg_n[0] = min(tw, long_n - sub * tw); }
    } else {
      nseg = p.tile_first[tile + 1] - seg_first;
      if (tid < nseg) {
        const int sent = p.order[seg_first + tid];
        seg_sent[tid] = sent;
        seg_w0[tid] = p.sent_slot0[sent] - tile * TW;
        seg_n[tid] = p.cap_lens[sent];
      }
    }
    if (tid < TW) wseg[tid] = -1;
    if (tid == 0) { diag[0] = 0; diag[1] = 0; }
    __syncthreads();
    if (tid < nseg) {
      const int w0 = seg_w0[tid], n = seg_n[tid];
      for (int w = 0; w < n; ++w) wseg[w0 + w] = tid;
      if (seg_sent[tid] == p.img_offset + b) { diag[0] = w0; diag[1] = n; }
    }
    __syncthreads();
    const int wbase = (nsub > 1) ? sub * tw : 0;   // index of slot 0's word inside its sentence (multi-tile)

    if (BWD) {
      // per-word scalars from dsim and the forward's saved statistics
      if (tid < TW) {
        const int sg = wseg[tid];
        float al = 0.f, be = 0.f, ka = 0.f, zi = 0.f, ga = 0.f;
        const size_t slot = (size_t)tile * TW + tid;
        if (sg >= 0) {
          const int sent = seg_sent[sg];
          const float g = p.dsim[(size_t)b * p.ld_sim + sent];
          const float* ws = p.wstat + ((size_t)b * p.n_slots + slot) * WSTAT;
          const float Z = ws[0], cosv = ws[1], nc2 = ws[2];
          const float tn = p.tnorm[slot];
          float A = __expf(p.sim[(size_t)b * p.ld_sim + sent] / p.temp3);
          if (p.agg == GLR_AGG_MEAN) A *= (float)p.cap_lens[sent];
          const float q = g * p.temp3 * p.temp2 * __expf(p.temp2 * cosv) / A;
          const float nc = sqrtf(nc2);
          const float prod = tn * nc;
          const float den = fmaxf(prod, p.eps);
          al = q / den;
          if (prod >= p.eps) { be = q * cosv / nc2; ga = q * cosv / (tn * tn); }
          ka = al * (cosv * den) - be * nc2;
          zi = Z > 0.f ? 1.f / Z : 0.f;
        }
        w_al[tid] = al; w_be[tid] = be; w_ka[tid] = ka; w_zi[tid] = zi;
        if (sweep == 1) p.gamma[(size_t)b * p.n_slots + slot] = ga;
      }
    }

    // ================= P1: acc[w, r] = T . V^T =================
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    stream_gemm<O, false>(acc, ring, buf1, p.tp + (size_t)tile * TW * rowbytes1, rowbytes1, vt_b, rowbytes1, S_pad,
                          nch1, nullptr, 0, wave, lane, wm, wg, nrb, tw);

    if (!BWD) {
      // scores -> LDS fp32 tile sc[word][region] (aliases the ring: every wave passed the last barrier)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int blk = wg + 4 * j;
        if (blk < nrb && wactive) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int word = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            sc[word * SCP + blk * 32 + l31] = acc[j][q];
          }
        }
      }
      __syncthreads();
      // column walk: log-sum-exp over the words of each sentence, one thread per region.
      // The result overwrites row `s` of the thread's own column (row s <= first word of segment s).
      if (tid < S_pad) {
        const int r = tid;
        if (nsub > 1) {
          if (sweep == 0) {
            const int n = seg_n[0];
            float m = mrun[r];
            for (int w = 0; w < n; ++w) m = fmaxf(m, sc[w * SCP + r]);
            float sum = srun[r] * __expf(mrun[r] - m);
            for (int w = 0; w < n; ++w) sum += __expf(sc[w * SCP + r] - m);
            mrun[r] = m;
            srun[r] = sum;
            if (sub == nsub - 1) {                       // final statistics of the sentence
              const float l = m + __logf(sum);
              mrun[r] = l;
              if (p.lse) p.lse[((size_t)b * p.n_sent + long_sent) * S_pad + r] = l;
            }
          }
        } else {
          for (int s = 0; s < nseg; ++s) {
            const int w0 = seg_w0[s], n = seg_n[s];
            float m = -INFINITY;
#pragma unroll 4
            for (int w = 0; w < n; ++w) m = fmaxf(m, sc[(w0 + w) * SCP + r]);
            float sum = 0.f;
#pragma unroll 4
            for (int w = 0; w < n; ++w) sum += __expf(sc[(w0 + w) * SCP + r] - m);
            const float l = m + __logf(sum);
            sc[s * SCP + r] = l;
            if (p.lse) p.lse[((size_t)b * p.n_sent + seg_sent[s]) * S_pad + r] = l;
          }
        }
      }
      __syncthreads();
      if (sweep == 0) continue;          // statistics sweep of a multi-tile sentence: next sub-tile
    }

    // ================= P2: a1, e2 from the scores in registers; LDS image =================
    // (word-row loop outermost: the per-word scalars are live for one row at a time)
    float zq[16], dq[16];
    if (wactive) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int word = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int sg = wseg[word];
        const int sgc = max(sg, 0);
        const int sent = seg_sent[sgc];
        float zi = 0.f, be = 0.f, al = 0.f;
        if (BWD) { zi = w_zi[word]; be = w_be[word]; al = w_al[word]; }
        float zacc = 0.f, dacc = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int blk = wg + 4 * j;
          if (blk < nrb) {
            const int region = blk * 32 + l31;
            // branch-free: invalid (empty slot / padded region) elements read a valid dummy and are zeroed
            const bool ok = sg >= 0 && region < p.S_eff;
            float l;
            if (BWD) l = p.lse[((size_t)b * p.n_sent + sent) * S_pad + region];
            else l = (nsub > 1) ? mrun[region] : sc[sgc * SCP + region];
            const float a1 = ok ? __expf(acc[j][q] - l) : 0.f;
            const float e2 = ok ? __expf(p.temp1 * a1) : 0.f;
            if (BWD) {
              O::from_f32(img + word * IMP + region * ESZ, be * (e2 * zi));
              a1r[j][q] = a1;
              acc[j][q] = -al * acc[j][q];
            } else {
              O::from_f32(img + word * IMP + region * ESZ, e2);
              const float e2r = ESZ == 4 ? e2 : bf2f(f2bf(e2));        // as the MFMA will see it
              zacc += e2r;
              dacc += e2r * acc[j][q];
              acc[j][q] = 0.f;
            }
          }
        }
        zq[q] = zacc;
        dq[q] = dacc;
      }
    }
    if (!BWD) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float z = half_sum32(zq[q]), d = half_sum32(dq[q]);
        if (l31 == 0 && wactive) {
          const int word = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          red[(0 * 4 + wg) * TW + word] = z;
          red[(1 * 4 + wg) * TW + word] = d;
        }
      }
    }
    __syncthreads();        // image complete (and the score tile is dead: the ring may be reused)

    // ================= P3: acc[w, r] (+)= image . G^T =================
    stream_gemm<O, true>(acc, ring, buf2, nullptr, 0, gram_b, rowbytes2, S_pad, nch2, img, IMP, wave, lane, wm, wg,
                         nrb, tw);

    if (!BWD) {
      // |c~|^2 = sum_r e2[w, r] * u~[w, r]
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int word = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        float v = 0.f;
        if (wactive) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int blk = wg + 4 * j;
            if (blk < nrb) v += O::to_f32(img + word * IMP + (blk * 32 + l31) * ESZ) * acc[j][q];
          }
        }
        v = half_sum32(v);
        if (l31 == 0 && wactive) red[(2 * 4 + wg) * TW + word] = v;
      }
      __syncthreads();

      // ================= P4 (forward): cosine, per-sentence aggregate =================
      if (tid < tw) {
        const float z = red[tid] + red[TW + tid] + red[2 * TW + tid] + red[3 * TW + tid];
        const float dd = red[4 * TW + tid] + red[5 * TW + tid] + red[6 * TW + tid] + red[7 * TW + tid];
        const float nn = red[8 * TW + tid] + red[9 * TW + tid] + red[10 * TW + tid] + red[11 * TW + tid];
        float cosv = 0.f, nc2 = 0.f;
        if (z > 0.f) {
          const float iz = 1.f / z;
          nc2 = fmaxf(nn, 0.f) * iz * iz;                      // |c_w|^2
          const float den = fmaxf(p.tnorm[(size_t)tile * TW + tid] * sqrtf(nc2), p.eps);
          cosv = dd * iz / den;
        }
        zsum[tid] = z;
        exs[tid] = __expf(p.temp2 * cosv);
        if (p.wstat) {
          float* ws = p.wstat + ((size_t)b * p.n_slots + (size_t)tile * TW + tid) * WSTAT;
          ws[0] = z; ws[1] = cosv; ws[2] = nc2; ws[3] = 0.f;
        }
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
      const int dw0 = diag[0], dn = diag[1];
      if (p.attn != nullptr && dn > 0) {
        const int sout = p.S_eff - p.strip;
        float* out = p.attn + p.attn_off[p.img_offset + b] + (size_t)wbase * sout;
        for (int idx = tid; idx < dn * sout; idx += NTHR) {
          const int w = idx / sout, r = idx % sout + p.strip;
          out[idx] = O::to_f32(img + (dw0 + w) * IMP + r * ESZ) / zsum[dw0 + w];
        }
      }
    } else {
      // ================= P4 (backward) =================
      // acc = beta*u - alpha*s = -da2.  softmax-over-regions backward needs only kappa_w;
      // softmax-over-words backward needs rho[r, sentence] = sum_w a1*da1 (LDS atomics).
      float* rho = (nsub > 1) ? srun : reinterpret_cast<float*>(smem);       // [nseg][S_pad]
      if (nsub == 1) {
        for (int i = tid; i < nseg * S_pad; i += NTHR) rho[i] = 0.f;
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int word = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int sg = wseg[word];
        const int rrow = (nsub > 1) ? 0 : max(sg, 0);
        const float zi = w_zi[word], ka = w_ka[word];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int blk = wg + 4 * j;
          if (blk < nrb) {
            const int region = blk * 32 + l31;
            const float a1 = a1r[j][q];
            const bool ok = sg >= 0 && region < p.S_eff;
            const float a2 = __expf(p.temp1 * a1) * zi;
            const float da1 = ok ? p.temp1 * a2 * (-acc[j][q] - ka) : 0.f;
            if (nsub == 1 || sweep == 0) atomicAdd(&rho[rrow * S_pad + region], ok ? a1 * da1 : 0.f);
            acc[j][q] = da1;
          }
        }
      }
      __syncthreads();
      if (sweep == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int word = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          const int sg = wseg[word];
          const size_t slot = (size_t)tile * TW + word;
          const int rrow = (nsub > 1) ? 0 : max(sg, 0);
          const float zi = w_zi[word], al = w_al[word], be = w_be[word];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int blk = wg + 4 * j;
            if (blk < nrb) {
              const int region = blk * 32 + l31;
              const bool ok = sg >= 0 && region < p.S_eff;
              const float a1 = a1r[j][q];
              const float a2 = ok ? __expf(p.temp1 * a1) * zi : 0.f;
              const float ds = a1 * (acc[j][q] - rho[rrow * S_pad + region]);
              const float x = ok ? ds + al * a2 : 0.f;
              const float ba = be * a2;
              O::from_f32(p.xout + ((slot * p.B_img + b) * S_pad + region) * ESZ, x);
              O::from_f32(p.bout + (((size_t)b * p.n_slots + slot) * S_pad + region) * ESZ, ba);
              O::from_f32(p.aout + (((size_t)b * p.n_slots + slot) * S_pad + region) * ESZ, a2);
            }
          }
        }
      }
    }
  }  // sub / sweep loops
}

int carve(LaParams& p, int op_dtype, int S_pad) {
  const int esz = op_dtype == GLR_F32 ? 4 : 2;
  const int cb = op_dtype == GLR_F32 ? OpF32::CB : OpBF16::CB;
  const int ring1 = 2 * (p.tw + S_pad) * cb;    // P1 ring
  const int ring2 = 2 * S_pad * cb;             // P3 ring (must not overlap the image)
  const int sc_bytes = p.tw * GLR_MAX_SPAD * 4; // score tile / rho (alias the ring, dead when it runs)
  const int img_bytes = p.tw * (GLR_MAX_SPAD * esz + 16);
  p.off_img = max(ring2, sc_bytes);             // the P1 ring may run over the image: it is dead then
  p.off_small = max(p.off_img + img_bytes, ring1);
  return p.off_small + 8192;
}

template <bool BWD>
int launch(LaParams& p, int op_dtype, void* stream) {
  const int lds = carve(p, op_dtype, p.S_pad);
  if (lds > 160 * 1024) return GLR_EINVAL;
  const int grid = p.pair_only ? p.B_img : ((p.B_img + 7) / 8) * 8 * p.n_tiles;
  hipStream_t st = (hipStream_t)stream;
  if (op_dtype == GLR_BF16) {
    if (hipFuncSetAttribute((const void*)k_local_attn<OpBF16, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return GLR_ELAUNCH;
    hipLaunchKernelGGL((k_local_attn<OpBF16, BWD>), dim3(grid), dim3(NTHR), lds, st, p);
  } else {
    if (hipFuncSetAttribute((const void*)k_local_attn<OpF32, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return GLR_ELAUNCH;
    hipLaunchKernelGGL((k_local_attn<OpF32, BWD>), dim3(grid), dim3(NTHR), lds, st, p);
  }
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

int fill_common(LaParams& p, const void* vt, const void* gram, const void* tp, const float* tnorm,
                const int32_t* sent_slot0, const int32_t* cap_lens, const int32_t* tile_first, const int32_t* order,
                const int32_t* tile_nsub, int n_tiles, int n_sent, int B_img, int D, int S_eff, float temp1,
                float temp2, float temp3, int agg, float eps, int op_dtype) {
  if (!vt || !gram || !tp || !tnorm || !sent_slot0 || !cap_lens || !tile_first || !order || !tile_nsub) return GLR_EINVAL;
  if (op_dtype != GLR_F32 && op_dtype != GLR_BF16) return GLR_EDTYPE;
  if (n_tiles <= 0 || n_sent <= 0 || B_img <= 0 || S_eff <= 0 || D <= 0 || D % 64 != 0) return GLR_EINVAL;
  const int S_pad = glr_region_pad(S_eff);
  if (S_pad > GLR_MAX_SPAD) return GLR_EINVAL;
  if (agg < 0 || agg > 2) return GLR_EINVAL;
  p.vt = (const unsigned char*)vt; p.gram = (const unsigned char*)gram; p.tp = (const unsigned char*)tp;
  p.tnorm = tnorm; p.sent_slot0 = sent_slot0; p.cap_lens = cap_lens; p.tile_first = tile_first; p.order = order;
  p.tile_nsub = tile_nsub; p.n_tiles = n_tiles; p.n_sent = n_sent; p.n_slots = n_tiles * TW; p.B_img = B_img;
  p.D = D; p.S_eff = S_eff; p.S_pad = S_pad; p.temp1 = temp1; p.temp2 = temp2; p.temp3 = temp3; p.agg = agg;
  p.eps = eps;
  p.tw = glr_tile_capacity(op_dtype);
  p.attn = nullptr; p.attn_off = nullptr; p.strip = 0; p.pair_only = 0; p.img_offset = 0;
  p.dsim = nullptr; p.xout = nullptr; p.bout = nullptr; p.aout = nullptr; p.gamma = nullptr;
  p.lse = nullptr; p.wstat = nullptr; p.sim = nullptr; p.ld_sim = 0;
  return GLR_OK;
}

}  // namespace



extern "C" int glr_local_attn_fwd(const void* vt, const void* gram, const void* tp, const float* tnorm,
                                  const int32_t* sent_slot0, const int32_t* cap_lens,
                                  const int32_t* tile_first, const int32_t* order, const int32_t* tile_nsub,
                                  int n_tiles, int n_sent, int B_img, int D, int S_eff, float temp1, float temp2,
                                  float temp3, int agg, float eps, float* sim, int ld_sim, float* lse, float* wstat,
                                  float* attn, const int64_t* attn_off, int strip, int pair_only, int img_offset,
                                  int op_dtype, void* stream) {
  LaParams p;
  const int rc = fill_common(p, vt, gram, tp, tnorm, sent_slot0, cap_lens, tile_first, order, tile_nsub, n_tiles,
                             n_sent, B_img, D, S_eff, temp1, temp2, temp3, agg, eps, op_dtype);
  if (rc != GLR_OK) return rc;
  if (!sim || (attn && !attn_off)) return GLR_EINVAL;
  if (pair_only && img_offset + B_img > n_sent) return GLR_EINVAL;
  p.sim = sim; p.ld_sim = ld_sim; p.lse = lse; p.wstat = wstat; p.attn = attn;
  p.attn_off = (const long long*)attn_off; p.strip = strip; p.pair_only = pair_only; p.img_offset = img_offset;
  return launch<false>(p, op_dtype, stream);
}

extern "C" int glr_local_attn_bwd(const void* vt, const void* gram, const void* tp, const float* tnorm,
                                  const int32_t* sent_slot0, const int32_t* cap_lens,
                                  const int32_t* tile_first, const int32_t* order, const int32_t* tile_nsub,
                                  int n_tiles, int n_sent, int B_img, int D, int S_eff, float temp1, float temp2,
                                  float temp3, int agg, float eps, const float* sim, const float* dsim, int ld_sim,
                                  const float* lse, const float* wstat, void* xout, void* bout, void* aout,
                                  float* gamma, int op_dtype, void* stream) {
  LaParams p;
  const int rc = fill_common(p, vt, gram, tp, tnorm, sent_slot0, cap_lens, tile_first, order, tile_nsub, n_tiles,
                             n_sent, B_img, D, S_eff, temp1, temp2, temp3, agg, eps, op_dtype);
  if (rc != GLR_OK) return rc;
  if (!sim || !dsim || !lse || !wstat || !xout || !bout || !aout || !gamma) return GLR_EINVAL;
  if (agg == GLR_AGG_MAX) return GLR_EINVAL;      // max aggregation is inference-only (gloria_model.py:199)
  p.sim = const_cast<float*>(sim); p.dsim = dsim; p.ld_sim = ld_sim; p.lse = const_cast<float*>(lse);
  p.wstat = const_cast<float*>(wstat); p.xout = (unsigned char*)xout; p.bout = (unsigned char*)bout;
  p.aout = (unsigned char*)aout; p.gamma = gamma;
  return launch<true>(p, op_dtype, stream);
}
