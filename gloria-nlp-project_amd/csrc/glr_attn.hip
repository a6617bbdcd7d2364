// Self-attention of the BERT text encoder for short sequences (L <= 128 tokens, head size 64), forward and backward:
// softmax(Q K^T / sqrt(d) + key mask) -> dropout -> . V, one workgroup per (sentence, head), everything in LDS.
// Replaces BertSelfAttention's matmul / softmax / dropout / matmul (transformers' modeling_bert as built by
// /root/reference/gloria/models/text_model.py:18-20; 97 tokens per caption in imagenome_pretrain_config.yaml).
//
// Why not the library kernel: at 97 tokens the flash-attention kernels torch dispatches take 95 us forward and 390 us
// backward per layer for 7 + 18 GFLOP - they tile for long sequences.  Here a head is ONE 128 x 128 score tile:
//   forward   Q, K row-major and V transposed in LDS (53 KB: three workgroups per CU); wave w owns query rows
//             32w..32w+31: S (16 MFMA 32x32x16),
//             row softmax across the 32 lanes of a lane half (DPP butterflies), dropout, P -> the wave's LDS slab,
//             O = P V (2 x L/16 MFMA).  Saves lse per row and the dropout keep bits (1 bit per score).
//   backward  two passes, each a kernel of its own (56 / 72 KB of LDS: two workgroups per CU), waves independent after
//             the cooperative load, no atomics, no cross-wave reductions:
//             pass Q (wave owns 32 queries): S, dP = dO V^T, dS -> slab, dQ = dS K;
//             pass K (wave owns 32 keys):    S^T = K Q^T, dP^T = V dO^T (the same products with the operands
//             swapped land key-major in the accumulators), P^T -> slab, dV = P^T dO, dS^T -> slab, dK = dS^T Q.
//             Every product is an "A . B^T" of two k-contiguous operands: the wave's own rows come straight from
//             global memory as MFMA fragments, the other side sits in LDS; the three operands that are contracted
//             over tokens (K for dQ, dO for dV, Q for dK) get a transposed LDS copy at load time.
// Tensors are the [B, L, n_heads * 64] outputs of the query / key / value Linears read in place (row stride = hidden
// size); O and the gradients are written in the same layout, so no permute / contiguous copies exist around the op.
// Dropout bits: a keyed counter hash of (head, query row, key): a pure function of (seed, offset, position); the
// backward reads the stored bits and never runs the generator.
#include "glr_common.h"

namespace {

constexpr int AT_NT = 256;
constexpr int AT_RP = 144;            // bytes per row of a row-major [token][64] LDS operand (128 + 16: bank spread)
constexpr int AT_SP = 272;            // bytes per row of a wave's score / staging slab: 128 keys + 16
constexpr float AT_LOG2E = 1.4426950408889634f;

struct AttnParams {
  const unsigned short* q; const unsigned short* k; const unsigned short* v;   // [B, L, ld]
  const unsigned char* key_mask;    // [B, L] nonzero = attend; NULL = all
  int B, nh, L, ld, ld_o;           // row strides (elements) of q / k / v / dq / dk / dv and of o / d_o
  float scale, p_drop;
  unsigned seed_lo, seed_hi, off_lo, off_hi;
  const unsigned long long* rng;   // NULL, or device cell {seed, offset base}: key = (rng[0], rng[1] + offset) (hipGraph replays)
  unsigned short* o;                // fwd out / bwd in
  float* lse;                       // [B * nh, 128]
  unsigned* keep;                   // [B * nh, 128, 4] keep bits: key 32 j + i of query row r = bit i of word (r, j)
  const unsigned short* d_o;        // bwd
  unsigned short* dq; unsigned short* dk; unsigned short* dv;
  int tp;                           // bytes per row of a token-contiguous LDS operand: max(2 * ceil16(L) + 16, 144)
};

// dropout bits: a counter-based hash (two rounds of a 32-bit multiply-xorshift mixer, keyed by seed and offset) of the
// score's position - 16 bits per score.  (Philox4x32-10 cost 40 quarter-rate integer multiplies per 8 scores here.)
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned hash32(unsigned ctr, unsigned k0, unsigned k1) { return mix32(mix32(ctr ^ k0) + k1); }

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// reductions over the 32 lanes of a lane half (every lane ends up with the result)
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, dpp<0xB1>(v));       // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp<0x4E>(v));       // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp<0x141>(v));      // row_half_mirror
  v = fmaxf(v, dpp<0x140>(v));      // row_mirror
  return fmaxf(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ float half_sum(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v + __shfl_xor(v, 16, 64);
}

__device__ __forceinline__ bf16x8 ldf(const unsigned char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ void mma(const bf16x8& a, const bf16x8& b, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = 0.f;
}
// accumulator register q of a 32x32 block -> row inside the block (column = lane & 31)
__device__ __forceinline__ int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// A [L x 64] bf16 head tile (row stride ld elements) is moved in two steps so that ALL global loads of a workgroup are
// in flight together (a load -> LDS-store loop per operand exposes the full memory latency once per iteration: with one
// workgroup per CU that was 12 (forward) / 28 (backward) serial round trips and most of the kernel time):
// item i = tid + 256 t, t = 0..3: token row i & 127, 16-byte piece i >> 7 (rows >= L read as zero): the lanes of a wave
// hold consecutive tokens of one piece, so the transposed LDS stores below are contiguous 2-byte runs (conflict-free).
__device__ __forceinline__ void tile_load(const unsigned short* g, int ld, int L, uint4 (&r)[4], int tid) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + AT_NT * t, row = i & 127, pc = i >> 7;
    r[t] = make_uint4(0u, 0u, 0u, 0u);
    if (row < L) r[t] = *reinterpret_cast<const uint4*>(g + (size_t)row * ld + pc * 8);
  }
}
// -> LDS row-major, 128 rows of AT_RP bytes
__device__ __forceinline__ void tile_store_rows(const uint4 (&r)[4], unsigned char* dst, int tid) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + AT_NT * t, row = i & 127, pc = i >> 7;
    *reinterpret_cast<uint4*>(dst + row * AT_RP + pc * 16) = r[t];
  }
}
// -> LDS transposed: dst[d][token], tp bytes per row, tokens [L, ceil16(L)) zero
__device__ __forceinline__ void tile_store_transposed(const uint4 (&r)[4], int lk, unsigned char* dst, int tp, int tid) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + AT_NT * t, row = i & 127, pc = i >> 7;
    if (row < lk) {
      const unsigned w[4] = {r[t].x, r[t].y, r[t].z, r[t].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<unsigned short*>(dst + (pc * 8 + 2 * e) * tp + row * 2) = (unsigned short)(w[e] & 0xffffu);
        *reinterpret_cast<unsigned short*>(dst + (pc * 8 + 2 * e + 1) * tp + row * 2) = (unsigned short)(w[e] >> 16);
      }
    }
  }
}
__device__ __forceinline__ float dot8(const uint4 a, const uint4 c) {
  const unsigned aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
  float d = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    d = __builtin_fmaf(__uint_as_float(aw[e] << 16), __uint_as_float(cw[e] << 16), d);
    d = __builtin_fmaf(__uint_as_float(aw[e] & 0xffff0000u), __uint_as_float(cw[e] & 0xffff0000u), d);
  }
  return d;
}
// acc[j] = A_rows(32 rows at arow0) . B_rows(block j)^T over 64 features, both row-major AT_RP operands
__device__ __forceinline__ void gemm_rows64(f32x16 (&acc)[4], const unsigned char* A, int arow0, const unsigned char* Bm, int nblk,
                                            int l31, int h) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 fa = ldf(A + (arow0 + l31) * AT_RP + ks * 32 + h * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nblk) mma(fa, ldf(Bm + (32 * j + l31) * AT_RP + ks * 32 + h * 16), acc[j]);
  }
}
// out[c] = slab_rows(32) . Bt_rows(block c)^T over kt 16-token steps, both token-contiguous operands (tp bytes per row)
__device__ __forceinline__ void gemm_tokens(f32x16 (&out)[2], const unsigned char* slab, const unsigned char* Bt, int tp, int kt,
                                            int l31, int h) {
  for (int ks = 0; ks < kt; ++ks) {
    const bf16x8 fa = ldf(slab + l31 * AT_SP + ks * 32 + h * 16);
#pragma unroll
    for (int c = 0; c < 2; ++c) mma(fa, ldf(Bt + (32 * c + l31) * tp + ks * 32 + h * 16), out[c]);
  }
}
// a wave's 32 x 64 fp32 result -> bf16 rows of a [B, L, ld] tensor, through the wave's slab (16-byte stores)
__device__ __forceinline__ void store_rows(const f32x16 (&out)[2], unsigned char* slab, int tp, unsigned short* g, int ld, int row0, int L,
                                           int lane, int l31, int h) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      *reinterpret_cast<unsigned short*>(slab + acc_row(q, h) * AT_SP + (32 * c + l31) * 2) = f2bf(out[c][q]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = lane + 64 * t, row = i >> 3, pc = i & 7;
    if (row0 + row < L)
      *reinterpret_cast<uint4*>(g + (size_t)(row0 + row) * ld + pc * 8) = *reinterpret_cast<const uint4*>(slab + row * AT_SP + pc * 16);
  }
}

__global__ void __launch_bounds__(AT_NT) k_attn_fwd(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
  const int L = p.L, tp = p.tp, lk = (L + 15) & ~15, kt = lk >> 4, nkb = (L + 31) >> 5;
  unsigned char* Qs = smem;
  unsigned char* Ks = Qs + 128 * AT_RP;
  unsigned char* Vt = Ks + 128 * AT_RP;
  unsigned char* Ps = smem;                      // the P slabs take the place of Q / K once every wave has its scores
  float* kb = reinterpret_cast<float*>(Vt + 64 * tp);
  const size_t base = (size_t)b * L * p.ld + (size_t)hd * 64;
  {
    uint4 rq[4], rk[4], rv[4];
    tile_load(p.q + base, p.ld, L, rq, tid);
    tile_load(p.k + base, p.ld, L, rk, tid);
    tile_load(p.v + base, p.ld, L, rv, tid);
    tile_store_rows(rq, Qs, tid);
    tile_store_rows(rk, Ks, tid);
    tile_store_transposed(rv, lk, Vt, tp, tid);
  }
  if (tid < 128) kb[tid] = (tid < L && (p.key_mask == nullptr || p.key_mask[(size_t)b * L + tid] != 0)) ? 0.f : -INFINITY;
  __syncthreads();
  const int row0 = 32 * wave;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) zero16(acc[j]);
  if (row0 < L) gemm_rows64(acc, Qs, row0, Ks, nkb, l31, h);
  __syncthreads();                              // Q / K are dead: their space becomes the P slabs
  if (row0 >= L) return;                        // no barrier below: a wave without query rows is done
  float kbv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) kbv[j] = kb[32 * j + l31];
  const bool drop = p.p_drop > 0.f;
  const unsigned thr16 = drop ? (unsigned)(p.p_drop * 65536.f + 0.5f) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  const float sl = p.scale * AT_LOG2E;
  unsigned char* slab = Ps + wave * 32 * AT_SP;
  unsigned mw0 = 0u, mw1 = 0u;
  unsigned sd_lo = p.seed_lo, sd_hi = p.seed_hi, of_lo = p.off_lo, of_hi = p.off_hi;
  if (p.rng != nullptr) {
    const unsigned long long s = p.rng[0], o = p.rng[1] + (((unsigned long long)p.off_hi << 32) | p.off_lo);
    sd_lo = (unsigned)s; sd_hi = (unsigned)(s >> 32); of_lo = (unsigned)o; of_hi = (unsigned)(o >> 32);
  }
  const unsigned hk0 = sd_lo ^ (of_lo * 0x9E3779B9u), hk1 = sd_hi ^ (of_hi * 0x85EBCA6Bu) ^ 0xC2B2AE35u;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = acc_row(q, h);
    float v[4], m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = __builtin_fmaf(acc[j][q], sl, kbv[j]); m = fmaxf(m, v[j]); }
    m = half_max(m);
    const float ms = m == -INFINITY ? 0.f : m;                     // a sentence with every key masked: all-zero row
    float e[4], s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { e[j] = __builtin_amdgcn_exp2f(v[j] - ms); s += e[j]; }
    s = half_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    if (l31 == 0) p.lse[(size_t)bh * 128 + row0 + r] = (ms + __builtin_amdgcn_logf(fmaxf(s, 1e-37f))) * (1.f / AT_LOG2E);
    unsigned r01 = 0u, r23 = 0u;
    if (drop) {
      const unsigned ctr = (((unsigned)bh * 128u + (unsigned)(row0 + r)) * 32u + (unsigned)l31) * 2u;
      r01 = hash32(ctr, hk0, hk1);
      r23 = hash32(ctr + 1u, hk0, hk1);
    }
    const unsigned r16[4] = {r01 & 0xffffu, r01 >> 16, r23 & 0xffffu, r23 >> 16};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool keepb = !drop || r16[j] >= thr16;
      if (drop) {
        const unsigned long long word = __ballot(keepb);
        if (lane == q * 4 + j) { mw0 = (unsigned)word; mw1 = (unsigned)(word >> 32); }
      }
      const float pd = keepb ? e[j] * inv * inv_keep : 0.f;
      *reinterpret_cast<unsigned short*>(slab + r * AT_SP + (32 * j + l31) * 2) = f2bf(pd);
    }
  }
  if (drop) {
    const int q = lane >> 2, j = lane & 3;
    const int r0 = row0 + acc_row(q, 0);
    p.keep[((size_t)bh * 128 + r0) * 4 + j] = mw0;
    p.keep[((size_t)bh * 128 + r0 + 4) * 4 + j] = mw1;
  }
  f32x16 out[2];
  zero16(out[0]); zero16(out[1]);
  gemm_tokens(out, slab, Vt, tp, kt, l31, h);
  store_rows(out, slab, tp, p.o + (size_t)b * L * p.ld_o + (size_t)hd * 64, p.ld_o, row0, L, lane, l31, h);
}

// ---- backward, split by pass so that two workgroups fit a CU (56 / 72 KB of LDS instead of 155 KB in one kernel:
// at one workgroup of four waves per CU every latency of the long dependent chains is exposed).  The wave's own 32
// rows - the A operand of its score products - go straight from global memory into MFMA fragments; the B operands
// are staged in LDS, and their space becomes the wave slabs once every wave has its two score tiles.
__device__ __forceinline__ void frag_rows_load(const unsigned short* g, int ld, int row, int L, int h, bf16x8 (&f)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < L) v = *reinterpret_cast<const uint4*>(g + (size_t)row * ld + ks * 16 + h * 8);
    f[ks] = __builtin_bit_cast(bf16x8, v);
  }
}
__device__ __forceinline__ void gemm_frag64(f32x16 (&acc)[4], const bf16x8 (&fa)[4], const unsigned char* Bm, int nblk, int l31, int h) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nblk) mma(fa[ks], ldf(Bm + (32 * j + l31) * AT_RP + ks * 32 + h * 16), acc[j]);
}

// pass Q: workgroup = (sentence, head), wave = 32 query rows: S, dP = dO V^T, dS -> slab, dQ = dS K
__global__ void __launch_bounds__(AT_NT, 2) k_attn_bwd_q(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
  const int L = p.L, tp = p.tp, lk = (L + 15) & ~15, kt = lk >> 4, nkb = (L + 31) >> 5;
  unsigned char* Ks = smem;
  unsigned char* Vs = Ks + 128 * AT_RP;
  unsigned char* Kt = Vs + 128 * AT_RP;
  unsigned char* Ps = smem;                      // slabs take the place of K / V
  float* kb = reinterpret_cast<float*>(Kt + 64 * tp);
  float* lse = kb + 128;
  float* delta = lse + 128;                                       // [128], rows of wave w written by wave w
  unsigned* keep = reinterpret_cast<unsigned*>(delta + 128);      // [128][4]
  const size_t base = (size_t)b * L * p.ld + (size_t)hd * 64, base_o = (size_t)b * L * p.ld_o + (size_t)hd * 64;
  const bool drop = p.p_drop > 0.f;
  const int row0 = 32 * wave;
  const bool active = row0 < L;
  bf16x8 fq[4], fg[4];
  {
    uint4 rk[4], rv[4];
    bf16x8 fo[4];
    tile_load(p.k + base, p.ld, L, rk, tid);
    tile_load(p.v + base, p.ld, L, rv, tid);
    frag_rows_load(p.q + base, p.ld, row0 + l31, L, h, fq);
    frag_rows_load(p.d_o + base_o, p.ld_o, row0 + l31, L, h, fg);
    frag_rows_load(p.o + base_o, p.ld_o, row0 + l31, L, h, fo);
    if (tid < 128) {
      kb[tid] = (tid < L && (p.key_mask == nullptr || p.key_mask[(size_t)b * L + tid] != 0)) ? 0.f : -INFINITY;
      lse[tid] = tid < L ? p.lse[(size_t)bh * 128 + tid] * AT_LOG2E : 0.f;
    }
    for (int i = tid; i < 512; i += AT_NT) keep[i] = (drop && (i >> 2) < L) ? p.keep[(size_t)bh * 512 + i] : 0xffffffffu;
    tile_store_rows(rk, Ks, tid);
    tile_store_rows(rv, Vs, tid);
    tile_store_transposed(rk, lk, Kt, tp, tid);
    // delta[row] = <dO[row], O[row]>: a lane holds half of its row (the k pieces of its lane half)
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) d += dot8(__builtin_bit_cast(uint4, fg[ks]), __builtin_bit_cast(uint4, fo[ks]));
    d += __shfl_xor(d, 32, 64);
    if (h == 0) delta[row0 + l31] = d;
  }
  __syncthreads();
  f32x16 acc[4], acc2[4], out[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) { zero16(acc[j]); zero16(acc2[j]); }
  if (active) {
    gemm_frag64(acc, fq, Ks, nkb, l31, h);                 // S
    gemm_frag64(acc2, fg, Vs, nkb, l31, h);                // dP (before the dropout scaling) = dO V^T
  }
  __syncthreads();                              // K / V rows are dead: their space becomes the slabs
  if (!active) return;                          // no barrier below
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  const float sl = p.scale * AT_LOG2E;
  unsigned char* slab = Ps + wave * 32 * AT_SP;
  {
    float kbv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kbv[j] = kb[32 * j + l31];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = acc_row(q, h);
      const float lr = lse[row0 + r], dl = delta[row0 + r];
      const uint4 kw = *reinterpret_cast<const uint4*>(keep + (row0 + r) * 4);
      const unsigned kwj[4] = {kw.x, kw.y, kw.z, kw.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[j][q], sl, kbv[j]) - lr);
        const float dp = ((kwj[j] >> l31) & 1u) ? acc2[j][q] * inv_keep : 0.f;
        *reinterpret_cast<unsigned short*>(slab + r * AT_SP + (32 * j + l31) * 2) = f2bf(pr * (dp - dl) * p.scale);
      }
    }
  }
  zero16(out[0]); zero16(out[1]);
  gemm_tokens(out, slab, Kt, tp, kt, l31, h);             // dQ = dS K
  store_rows(out, slab, tp, p.dq + base, p.ld, row0, L, lane, l31, h);
}

// pass K: wave = 32 key rows: S^T = K Q^T, dP^T = V dO^T, P^T -> slab, dV = P^T dO, dS^T -> slab, dK = dS^T Q
__global__ void __launch_bounds__(AT_NT, 2) k_attn_bwd_kv(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
  const int L = p.L, tp = p.tp, lk = (L + 15) & ~15, kt = lk >> 4, nkb = (L + 31) >> 5;
  unsigned char* Qs = smem;
  unsigned char* Gs = Qs + 128 * AT_RP;          // dO
  unsigned char* Qt = Gs + 128 * AT_RP;
  unsigned char* Gt = Qt + 64 * tp;
  unsigned char* Ps = smem;                      // slabs take the place of Q / dO rows
  float* kb = reinterpret_cast<float*>(Gt + 64 * tp);
  float* lse = kb + 128;
  float* delta = lse + 128;                                       // [2][128] partial sums (the two thread halves)
  unsigned* keep = reinterpret_cast<unsigned*>(delta + 256);      // [128][4]
  const size_t base = (size_t)b * L * p.ld + (size_t)hd * 64, base_o = (size_t)b * L * p.ld_o + (size_t)hd * 64;
  const bool drop = p.p_drop > 0.f;
  const int row0 = 32 * wave;
  const bool active = row0 < L;
  bf16x8 fk[4], fv[4];
  {
    uint4 rq[4], rg[4], ro[4];
    tile_load(p.q + base, p.ld, L, rq, tid);
    tile_load(p.d_o + base_o, p.ld_o, L, rg, tid);
    tile_load(p.o + base_o, p.ld_o, L, ro, tid);
    frag_rows_load(p.k + base, p.ld, row0 + l31, L, h, fk);
    frag_rows_load(p.v + base, p.ld, row0 + l31, L, h, fv);
    if (tid < 128) {
      kb[tid] = (tid < L && (p.key_mask == nullptr || p.key_mask[(size_t)b * L + tid] != 0)) ? 0.f : -INFINITY;
      lse[tid] = tid < L ? p.lse[(size_t)bh * 128 + tid] * AT_LOG2E : 0.f;
    }
    for (int i = tid; i < 512; i += AT_NT) keep[i] = (drop && (i >> 2) < L) ? p.keep[(size_t)bh * 512 + i] : 0xffffffffu;
    tile_store_rows(rq, Qs, tid);
    tile_store_rows(rg, Gs, tid);
    tile_store_transposed(rq, lk, Qt, tp, tid);
    tile_store_transposed(rg, lk, Gt, tp, tid);
    delta[tid] = (dot8(rg[0], ro[0]) + dot8(rg[1], ro[1])) + (dot8(rg[2], ro[2]) + dot8(rg[3], ro[3]));
  }
  __syncthreads();
  f32x16 acc[4], acc2[4], out[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) { zero16(acc[j]); zero16(acc2[j]); }
  float lq[4], dq_[4], kbk16[16];
  unsigned kwq[4];
  if (active) {
    gemm_frag64(acc, fk, Qs, nkb, l31, h);                 // S^T
    gemm_frag64(acc2, fv, Gs, nkb, l31, h);                // dP^T = V dO^T
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lq[j] = lse[32 * j + l31];
    dq_[j] = delta[32 * j + l31] + delta[128 + 32 * j + l31];
    kwq[j] = keep[(32 * j + l31) * 4 + wave];            // keys of this wave's block, of query 32 j + l31
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) kbk16[q] = kb[row0 + acc_row(q, h)];
  __syncthreads();                              // Q / dO rows are dead: their space becomes the slabs
  if (!active) return;                          // no barrier below
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  const float sl = p.scale * AT_LOG2E;
  unsigned char* slab = Ps + wave * 32 * AT_SP;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = acc_row(q, h);                         // key row inside the block = bit index
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[j][q], sl, kbk16[q]) - lq[j]);
      const bool keepb = (kwq[j] >> r) & 1u;
      *reinterpret_cast<unsigned short*>(slab + r * AT_SP + (32 * j + l31) * 2) = f2bf(keepb ? pr * inv_keep : 0.f);
      const float dp = keepb ? acc2[j][q] * inv_keep : 0.f;
      acc2[j][q] = pr * (dp - dq_[j]) * p.scale;          // dS^T, kept for the second product
    }
  }
  zero16(out[0]); zero16(out[1]);
  gemm_tokens(out, slab, Gt, tp, kt, l31, h);             // dV = Pd^T dO
  {
    f32x16 dvv[2] = {out[0], out[1]};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = acc_row(q, h);
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<unsigned short*>(slab + r * AT_SP + (32 * j + l31) * 2) = f2bf(acc2[j][q]);
    }
    zero16(out[0]); zero16(out[1]);
    gemm_tokens(out, slab, Qt, tp, kt, l31, h);           // dK = dS^T Q
    store_rows(out, slab, tp, p.dk + base, p.ld, row0, L, lane, l31, h);
    store_rows(dvv, slab, tp, p.dv + base, p.ld, row0, L, lane, l31, h);
  }
}

// bytes per row of the token-contiguous (transposed) LDS operands
int attn_tp(int L) { return 2 * ((L + 15) & ~15) + 16; }
size_t attn_lds_fwd(int L) { return (size_t)2 * 128 * AT_RP + (size_t)64 * attn_tp(L) + 512; }   // slabs alias Q / K: 128 AT_SP <= 256 AT_RP
size_t attn_lds_bwd_q(int L) { return (size_t)2 * 128 * AT_RP + (size_t)64 * attn_tp(L) + 3 * 512 + 2048; }
size_t attn_lds_bwd_kv(int L) { return (size_t)2 * 128 * AT_RP + (size_t)2 * 64 * attn_tp(L) + 4 * 512 + 2048; }
size_t attn_lds_bwd(int L) { return attn_lds_bwd_kv(L) > attn_lds_bwd_q(L) ? attn_lds_bwd_kv(L) : attn_lds_bwd_q(L); }

int attn_fill(AttnParams& p, const void* q, const void* k, const void* v, const unsigned char* key_mask, int B, int nh, int L, int ld,
              int ld_o, float scale, float p_drop, unsigned long long seed, unsigned long long offset) {
  if (!q || !k || !v || B <= 0 || nh <= 0 || L <= 0 || L > 128 || ld < nh * 64 || ld % 8 != 0 || ld_o < nh * 64 || ld_o % 8 != 0)
    return GLR_EINVAL;
  if (p_drop < 0.f || p_drop >= 1.f) return GLR_EINVAL;
  p.q = (const unsigned short*)q; p.k = (const unsigned short*)k; p.v = (const unsigned short*)v; p.key_mask = key_mask;
  p.B = B; p.nh = nh; p.L = L; p.ld = ld; p.ld_o = ld_o; p.scale = scale; p.p_drop = p_drop;
  p.seed_lo = (unsigned)seed; p.seed_hi = (unsigned)(seed >> 32); p.off_lo = (unsigned)offset; p.off_hi = (unsigned)(offset >> 32);
  p.rng = nullptr;
  p.o = nullptr; p.lse = nullptr; p.keep = nullptr; p.d_o = nullptr; p.dq = p.dk = p.dv = nullptr;
  p.tp = attn_tp(L);
  return GLR_OK;
}

}  // namespace

extern "C" int glr_attn_max_tokens(int backward) {
  for (int L = 128; L > 0; L -= 16)
    if ((backward ? attn_lds_bwd(L) : attn_lds_fwd(L)) <= 160 * 1024) return L;
  return 0;
}

extern "C" int glr_attn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, int B, int n_heads, int L, int ld,
                            int ld_o, float scale, float p_drop, unsigned long long seed, unsigned long long offset,
                            const unsigned long long* rng_cell, void* o, float* lse, uint32_t* keep, void* stream) {
  AttnParams p;
  const int rc = attn_fill(p, q, k, v, key_mask, B, n_heads, L, ld, ld_o, scale, p_drop, seed, offset);
  if (rc != GLR_OK) return rc;
  if (!o || !lse || (p_drop > 0.f && !keep) || L > glr_attn_max_tokens(0)) return GLR_EINVAL;
  p.o = (unsigned short*)o; p.lse = lse; p.keep = keep; p.rng = rng_cell;
  const int lds = (int)attn_lds_fwd(L);
  static GlrLdsAttr lds_fwd;
  if (glr_ensure_lds(lds_fwd, (const void*)k_attn_fwd, lds) != GLR_OK) return GLR_ELAUNCH;
  hipLaunchKernelGGL(k_attn_fwd, dim3(B * n_heads), dim3(AT_NT), lds, (hipStream_t)stream, p);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const uint8_t* key_mask,
                            const float* lse, const uint32_t* keep, int B, int n_heads, int L, int ld, int ld_o, float scale, float p_drop,
                            void* dq, void* dk, void* dv, void* stream) {
  AttnParams p;
  const int rc = attn_fill(p, q, k, v, key_mask, B, n_heads, L, ld, ld_o, scale, p_drop, 0, 0);
  if (rc != GLR_OK) return rc;
  if (!o || !d_o || !lse || !dq || !dk || !dv || (p_drop > 0.f && !keep) || L > glr_attn_max_tokens(1)) return GLR_EINVAL;
  p.o = (unsigned short*)const_cast<void*>(o); p.d_o = (const unsigned short*)d_o; p.lse = const_cast<float*>(lse);
  p.keep = const_cast<unsigned*>(keep); p.dq = (unsigned short*)dq; p.dk = (unsigned short*)dk; p.dv = (unsigned short*)dv;
  const int lds_q = (int)attn_lds_bwd_q(L), lds_kv = (int)attn_lds_bwd_kv(L);
  static GlrLdsAttr lds_bq, lds_bkv;
  if (glr_ensure_lds(lds_bq, (const void*)k_attn_bwd_q, lds_q) != GLR_OK) return GLR_ELAUNCH;
  if (glr_ensure_lds(lds_bkv, (const void*)k_attn_bwd_kv, lds_kv) != GLR_OK) return GLR_ELAUNCH;
  hipLaunchKernelGGL(k_attn_bwd_q, dim3(B * n_heads), dim3(AT_NT), lds_q, (hipStream_t)stream, p);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_attn_bwd_kv, dim3(B * n_heads), dim3(AT_NT), lds_kv, (hipStream_t)stream, p);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
