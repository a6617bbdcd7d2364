// Host-side tile planning + HBM-bound operand packing kernels (layout change + dtype convert).
//
// glr_plan_tiles   replaces the per-sentence slice words_emb[i, :, :cap_lens[i]] of the reference
//                  loop (/root/reference/gloria/loss/gloria_loss.py:116-123) by a packed slot table.
// glr_pack_regions context.view(B, D, S) + optional no_attn_vec column + the transpose-copy of
//                  attention_fn (gloria_loss.py:30-35), done ONCE per step instead of once per sentence.
// glr_pack_words   word slices of every sentence, word-major, plus their L2 norms (:14, :122).
#include <string.h>

#include <algorithm>
#include <vector>

#include "glr_k1.h"

extern "C" int glr_version(void) { return 2; }

extern "C" int glr_plan_tiles_bound(const int32_t* cap_lens, int n_sent, int capacity) {
  if (!cap_lens || n_sent <= 0 || (capacity != 32 && capacity != GLR_TILE_WORDS)) return GLR_EINVAL;
  int total = 0;
  for (int i = 0; i < n_sent; ++i) {
    if (cap_lens[i] < 1 || cap_lens[i] > GLR_MAX_WORDS) return GLR_EINVAL;
    total += (cap_lens[i] + capacity - 1) / capacity;
  }
  return total;     // >= number of tiles and >= number of `order` entries
}

extern "C" int glr_plan_tiles(const int32_t* cap_lens, int n_sent, int capacity, int max_pair_seg, int32_t* sent_slot0,
                              int32_t* tile_first, int32_t* order, int32_t* tile_nsub) {
  if (!cap_lens || !sent_slot0 || !tile_first || !order || !tile_nsub || n_sent <= 0) return GLR_EINVAL;
  if (capacity != 32 && capacity != GLR_TILE_WORDS) return GLR_EINVAL;
  for (int i = 0; i < n_sent; ++i)
    if (cap_lens[i] < 1 || cap_lens[i] > GLR_MAX_WORDS) return GLR_EINVAL;
  std::vector<int> fill;                       // used slots per tile (capacity = closed)
  std::vector<std::vector<int>> members;
  std::vector<int> nsub;
  // first fit in caption order, at most `max_sent` sentences per ordinary tile
  auto pack = [&](int max_sent) {
    fill.clear(); members.clear(); nsub.clear();
    for (int i = 0; i < n_sent; ++i) {
      const int n = cap_lens[i];
      if (n > capacity) {                      // multi-tile sentence: its own run of consecutive tiles
        const int k = (n + capacity - 1) / capacity;
        for (int s = 0; s < k; ++s) {
          fill.push_back(capacity);
          members.emplace_back(1, i);
          nsub.push_back(s == 0 ? k : -1);
        }
        continue;
      }
      size_t t = 0;
      while (t < fill.size() && (fill[t] + n > capacity || (int)members[t].size() >= max_sent)) ++t;
      if (t == fill.size()) { fill.push_back(0); members.emplace_back(); nsub.push_back(0); }
      fill[t] += n;
      members[t].push_back(i);
    }
  };
  // work items the pair kernels would need for the current packing: ordinary tiles sorted by sentence count and
  // paired fewest-with-most (the order built below), a pair holding at most max_pair_seg sentences
  auto items = [&]() {
    std::vector<int> cnt;
    int n_items = 0;
    for (size_t t = 0; t < members.size(); ++t) {
      if (nsub[t] == 0) cnt.push_back((int)members[t].size());
      else if (nsub[t] > 0) ++n_items;
    }
    std::sort(cnt.begin(), cnt.end());
    std::vector<int> seq;
    for (size_t lo = 0, hi = cnt.size(); lo < hi;) {
      seq.push_back(cnt[lo++]);
      if (lo < hi) seq.push_back(cnt[--hi]);
    }
    for (size_t t = 0; t < seq.size();) {
      if (t + 1 < seq.size() && seq[t] + seq[t + 1] <= max_pair_seg) t += 2; else t += 1;
      ++n_items;
    }
    return n_items;
  };
  // Plain first fit leaves the many short sentences of a length-sorted batch in the last tiles, which then hold more
  // sentences than a pair may (max_pair_seg) and run as single tiles - a workgroup each, like a whole pair.  With
  // pairing in view, a cap on the sentences per tile is chosen that minimises the number of work items (ties: fewer
  // tiles): a tile more usually costs less than the pairs it unlocks.
  int best_cap = n_sent;
  if (max_pair_seg > 1 && capacity == GLR_TILE_WORDS) {
    long best_cost = -1;
    for (int cap = max_pair_seg; cap >= max_pair_seg / 2; --cap) {
      pack(cap == max_pair_seg ? n_sent : cap);
      const long cost = (long)items() * 4096 + (long)members.size();
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_cap = cap == max_pair_seg ? n_sent : cap; }
    }
  }
  pack(best_cap);
  // Tile ORDER is free (a tile is just 64 consecutive slots): multi-tile sentences first (their runs stay
  // together), then the ordinary tiles, contiguous - the pair kernels take two CONSECUTIVE ordinary tiles - and
  // balanced: fewest sentences next to most, second fewest next to second most, ...
  {
    std::vector<size_t> by_cnt, seq;
    for (size_t t = 0; t < members.size(); ++t) if (nsub[t] != 0) seq.push_back(t);
    for (size_t t = 0; t < members.size(); ++t) if (nsub[t] == 0) by_cnt.push_back(t);
    std::stable_sort(by_cnt.begin(), by_cnt.end(), [&](size_t a, size_t b) { return members[a].size() < members[b].size(); });
    for (size_t lo = 0, hi = by_cnt.size(); lo < hi;) {
      seq.push_back(by_cnt[lo++]);
      if (lo < hi) seq.push_back(by_cnt[--hi]);
    }
    std::vector<std::vector<int>> m2(members.size());
    std::vector<int> f2(fill.size()), n2(nsub.size());
    for (size_t i = 0; i < seq.size(); ++i) { m2[i] = members[seq[i]]; f2[i] = fill[seq[i]]; n2[i] = nsub[seq[i]]; }
    members.swap(m2);
    fill.swap(f2);
    nsub.swap(n2);
    for (size_t t = 0; t < members.size(); ++t) {
      if (nsub[t] != 0) continue;
      int pos = 0;
      for (int s : members[t]) { sent_slot0[s] = (int)t * GLR_TILE_WORDS + pos; pos += cap_lens[s]; }
    }
  }
  for (size_t t = 0; t < members.size(); ++t)
    if (nsub[t] > 0) sent_slot0[members[t][0]] = (int)t * GLR_TILE_WORDS;
  int k = 0;
  for (size_t t = 0; t < members.size(); ++t) {
    tile_first[t] = k;
    tile_nsub[t] = nsub[t];
    for (int s : members[t]) order[k++] = s;
  }
  tile_first[members.size()] = k;
  return (int)members.size();
}

extern "C" int glr_plan_items(const int32_t* tile_nsub, const int32_t* tile_first, int n_tiles, int allow_pairs,
                              int max_pair_seg, int32_t* single_tile, int32_t* pair_tile, int32_t* all_tile,
                              int32_t* counts) {
  if (!tile_nsub || !tile_first || !single_tile || !pair_tile || !all_tile || !counts || n_tiles <= 0) return GLR_EINVAL;
  int ns = 0, np = 0, na = 0;
  auto pairable = [&](int t) { return t < n_tiles && tile_nsub[t] == 0; };
  for (int t = 0; t < n_tiles;) {
    if (tile_nsub[t] < 0) return GLR_EINVAL;         // a continuation tile cannot start an item
    if (tile_nsub[t] == 2 && allow_pairs) {          // a 65..128-word sentence owns exactly one pair of tiles
      pair_tile[np++] = t; all_tile[na++] = t;       // forward: pair kernel; backward: multi-tile path of the head
      t += 2;
    } else if (tile_nsub[t] > 1) {                   // longer sentence: one item, handled in sweeps
      single_tile[ns++] = t; all_tile[na++] = t;
      t += tile_nsub[t];
    } else if (allow_pairs && pairable(t) && pairable(t + 1) && tile_first[t + 2] - tile_first[t] <= max_pair_seg) {
      pair_tile[np++] = t; all_tile[na++] = t; all_tile[na++] = t + 1;
      t += 2;
    } else {
      single_tile[ns++] = t; all_tile[na++] = t;
      t += 1;
    }
  }
  counts[0] = ns; counts[1] = np; counts[2] = na;
  return GLR_OK;
}

// Row flags of the forward pair kernel.  There a wave holds ALL 64 word slots of a tile for its region columns:
// lane half h (lane >> 5) owns the slots w with ((w >> 2) & 1) == h, 32 rows in word order, row index
// k(w) = 16 (w >> 5) + 4 ((w & 31) >> 3) + (w & 3) (the accumulator register, second 32-word block at k >= 16).
// The words of a sentence are a run of rows in each half; the kernel walks the rows once per pass and only acts
// where a run starts or ends, which is the same for all lanes of a half: per tile and half one bit per row.
//   flags[tile][0..1]  START bits of half 0 / 1: first row of a sentence's run
//   flags[tile][2..3]  LAST  bits: last row of a run
//   flags[tile][4..5]  OWNER bits (subset of START): the run that holds the sentence's first word (that lane half
//                      stores the sentence's log-sum-exp row for the backward pass)
//   flags[tile][6..7]  reserved (0)
extern "C" int glr_plan_rowflags(const int32_t* cap_lens, const int32_t* sent_slot0, const int32_t* tile_first,
                                 const int32_t* order, const int32_t* tile_nsub, int n_tiles, int capacity,
                                 uint32_t* flags) {
  if (!cap_lens || !sent_slot0 || !tile_first || !order || !tile_nsub || !flags || n_tiles <= 0) return GLR_EINVAL;
  if (capacity != GLR_TILE_WORDS) return GLR_EINVAL;          // the pair kernel runs full-width (bf16) tiles only
  memset(flags, 0, sizeof(uint32_t) * 8 * (size_t)n_tiles);
  auto row_of = [](int w) { return 16 * (w >> 5) + 4 * ((w & 31) >> 3) + (w & 3); };
  int head = 0;                                               // head tile of the current multi-tile sentence
  for (int t = 0; t < n_tiles; ++t) {
    if (tile_nsub[t] >= 0) head = t;
    uint32_t* f = flags + 8 * (size_t)t;
    for (int k = tile_first[t]; k < tile_first[t + 1]; ++k) {
      const int sent = order[k];
      int a, e;
      bool owner = true;
      if (tile_nsub[t] == 0) {
        a = sent_slot0[sent] - t * GLR_TILE_WORDS;
        e = a + cap_lens[sent];
      } else {                                                // tile (t - head) of a sentence that owns whole tiles
        const int sub = t - head;
        a = 0;
        e = cap_lens[sent] - sub * capacity;
        if (e > capacity) e = capacity;
        owner = sub == 0;
      }
      if (a < 0 || e > GLR_TILE_WORDS || e <= a) return GLR_EINVAL;
      int first[2] = {-1, -1}, last[2] = {-1, -1};
      for (int w = a; w < e; ++w) {
        const int hh = (w >> 2) & 1;
        if (first[hh] < 0) first[hh] = w;
        last[hh] = w;
      }
      for (int hh = 0; hh < 2; ++hh) {
        if (first[hh] < 0) continue;
        f[hh] |= 1u << row_of(first[hh]);
        f[2 + hh] |= 1u << row_of(last[hh]);
        if (owner && first[hh] == a) f[4 + hh] |= 1u << row_of(first[hh]);
      }
    }
  }
  return GLR_OK;
}

// Pair descriptors of the forward pair kernel: everything a workgroup needs to know about its pair of tiles in ONE
// coalesced 256-byte read (instead of three dependent global round trips through tile_first / order / sent_slot0 /
// cap_lens plus the row flags):  desc[pair][64] int32 =
//   [0] sentences in the pair (<= 8), [1] 1 = ONE sentence of 65..128 words owning both tiles, [2..7] 0
//   [8..15] sentence ids, [16..23] first slot in the pair (0..127), [24..31] words, [32..39] / [40..47] row flags of
//   tile A / B (glr_plan_rowflags layout), [48..63] 0
extern "C" int glr_plan_pair_desc(const int32_t* cap_lens, const int32_t* sent_slot0, const int32_t* tile_first,
                                  const int32_t* order, const int32_t* tile_nsub, int n_tiles, int capacity,
                                  const int32_t* pair_tile, int n_pair, int32_t* desc) {
  if (!pair_tile || !desc || n_pair <= 0) return GLR_EINVAL;
  std::vector<uint32_t> flags(8 * (size_t)n_tiles);
  const int rc = glr_plan_rowflags(cap_lens, sent_slot0, tile_first, order, tile_nsub, n_tiles, capacity, flags.data());
  if (rc != GLR_OK) return rc;
  memset(desc, 0, sizeof(int32_t) * 64 * (size_t)n_pair);
  for (int k = 0; k < n_pair; ++k) {
    const int t0 = pair_tile[k];
    if (t0 < 0 || t0 + 1 >= n_tiles) return GLR_EINVAL;
    int32_t* d = desc + 64 * (size_t)k;
    const bool lp = tile_nsub[t0] == 2;
    const int ns = lp ? 1 : tile_first[t0 + 2] - tile_first[t0];
    if (ns < 1 || ns > 8) return GLR_EINVAL;
    d[0] = ns;
    d[1] = lp ? 1 : 0;
    for (int s = 0; s < ns; ++s) {
      const int sent = order[tile_first[t0] + s];
      d[8 + s] = sent;
      d[16 + s] = sent_slot0[sent] - t0 * GLR_TILE_WORDS;
      d[24 + s] = cap_lens[sent];
    }
    for (int q = 0; q < 8; ++q) {
      d[32 + q] = (int32_t)flags[8 * (size_t)t0 + q];
      d[40 + q] = (int32_t)flags[8 * (size_t)(t0 + 1) + q];
    }
  }
  return GLR_OK;
}

namespace {

// grid (S_pad/64, D/64, B), 256 threads: one 64(feature) x 64(region) tile.
// in_layout 0: img [B, D, S] (NCHW, the reference layout) -> transposed through LDS
// in_layout 1: img [B, S, D] (channels-last memory of the same tensor) -> straight copy
__global__ void __launch_bounds__(256) k_pack_regions(const void* __restrict__ img, int in_dtype, int in_layout,
                                                      const void* __restrict__ no_attn, void* __restrict__ vt,
                                                      int D, int S, int S_pad, int shift, int op_dtype) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, d0 = blockIdx.y * 64, r0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int S_eff = S + shift;
  if (in_layout == 1) {
    for (int rl = ty; rl < 64; rl += 4) {
      const int rp = r0 + rl, d = d0 + tx;            // rp = packed region index
      float v = 0.f;
      if (rp < S_eff) {
        if (shift && rp == 0) v = ld_any(no_attn, d, in_dtype);
        else v = ld_any(img, ((size_t)b * S + (rp - shift)) * D + d, in_dtype);
      }
      st_any(vt, ((size_t)b * S_pad + rp) * D + d, op_dtype, v);
    }
    return;
  }
  for (int dl = ty; dl < 64; dl += 4) {
    const int d = d0 + dl, rp = r0 + tx;
    float v = 0.f;
    if (rp < S_eff) {
      if (shift && rp == 0) v = ld_any(no_attn, d, in_dtype);
      else v = ld_any(img, ((size_t)b * D + d) * S + (rp - shift), in_dtype);
    }
    tile[dl][tx] = v;
  }
  __syncthreads();
  for (int rl = ty; rl < 64; rl += 4)
    st_any(vt, ((size_t)b * S_pad + r0 + rl) * D + d0 + tx, op_dtype, tile[tx][rl]);
}

// Fast path of the training step: channels-last bf16 features -> bf16 operands.  One workgroup moves a block of 32
// regions x 512 bytes (256 features) and writes it TWICE: row-major vt (the gradient GEMMs' operand) and the K-tiled,
// fragment-major copy the K1 streams read (glr_k1.h: [D*2 / 64][S_pad / 32][4 slots][32 rows][16 bytes] per image) -
// one read of the features instead of a pack pass plus a tiling pass.  Both writes are whole 512-byte runs: the
// tiled order is a transposition of (row, piece) inside the block, done through 16.5 KB of LDS (33-piece pitch: the
// transposed read is conflict-free).  Rows [S_eff, S_pad) are zero, the optional no-attention vector is row 0.
// grid (D*2 / 512, S_pad / 32, B), 256 threads.
__global__ void __launch_bounds__(256) k_pack_regions_cl16(const uint4* __restrict__ img, const uint4* __restrict__ no_attn,
                                                           uint4* __restrict__ vt, uint4* __restrict__ vt_t, int D16,
                                                           int S, int S_pad, int shift) {
  __shared__ uint4 tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const size_t b = blockIdx.z;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int q = t + 256 * k, rl = q >> 5, c = q & 31;      // 32 consecutive threads = 512 contiguous bytes of one row
    const int rp = r0 + rl;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (rp < S + shift) v = (shift && rp == 0) ? no_attn[c0 + c] : img[(b * S + (rp - shift)) * D16 + c0 + c];
    vt[(b * S_pad + rp) * D16 + c0 + c] = v;
    tile[rl][c] = v;
  }
  if (vt_t == nullptr) return;
  __syncthreads();
  const size_t nch = D16 / 4;                                // 64-byte chunks per row
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int q = t + 256 * k, ch = q >> 7, slot = (q >> 5) & 3, rl = q & 31;   // 128 consecutive threads = one 2-KiB block
    vt_t[((b * nch + (c0 >> 2) + ch) * S_pad + r0) * 4 + slot * 32 + rl] = tile[rl][ch * 4 + slot];
  }
}

// grid (D/64, B_txt), 256 threads
__global__ void __launch_bounds__(256) k_pack_words(const void* __restrict__ words, int in_dtype,
                                                    const int* __restrict__ sent_slot0,
                                                    const int* __restrict__ cap_lens, void* __restrict__ tp, int D,
                                                    int L, int word_start, int capacity, int op_dtype) {
  __shared__ float tile[64][65];
  const int i = blockIdx.y, d0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = cap_lens[i], slot0 = sent_slot0[i];
  for (int w0 = 0; w0 < n; w0 += 64) {
    for (int dl = ty; dl < 64; dl += 4) {
      const int w = w0 + tx;
      float v = 0.f;
      if (w < n && word_start + w < L) v = ld_any(words, ((size_t)i * D + d0 + dl) * L + word_start + w, in_dtype);
      tile[dl][tx] = v;
    }
    __syncthreads();
    for (int wl = ty; wl < 64; wl += 4)
      if (w0 + wl < n) {
        const int w = w0 + wl;          // word w of a multi-tile sentence: tile w / capacity, position w % capacity
        const size_t slot = (size_t)slot0 + (w / capacity) * GLR_TILE_WORDS + (w % capacity);
        st_any(tp, slot * D + d0 + tx, op_dtype, tile[tx][wl]);
      }
    __syncthreads();
  }
}

// one wave per slot: L2 norm of the packed word as stored
__global__ void __launch_bounds__(256) k_word_norms(const void* __restrict__ tp, float* __restrict__ tnorm,
                                                    int n_slots, int D, int op_dtype) {
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= n_slots) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float v = ld_any(tp, (size_t)slot * D + d, op_dtype);
    s += v * v;
  }
  s = wave_sum(s);
  if (lane == 0) tnorm[slot] = sqrtf(s);
}

// K-tiling copy: every block of `rows` rows x `row_bytes` bytes is rewritten as [row_bytes / 64] chunks of `rows` * 64
// bytes, fragment-major inside a chunk (glr_k1.h, glr_ktile_off).  One thread moves 16 bytes; writes are linear.
// ones_row >= 0: row `ones_row` of every block is written as ones_cols elements of 1.0 (element size esz) followed
// by zeros instead of being copied - the Gram operand's ones row (glr_local_attn_fwd, tile_rowflags).
__global__ void k_tile_k(const uint4* __restrict__ src, uint4* __restrict__ dst, int rows, int nch, size_t total,
                         int ones_row, int ones_cols, int esz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int rl = (int)(i & 31);
  const int s16 = (int)((i >> 5) & 3);
  const size_t t = i >> 7;                        // 32-row block index over (block, chunk, row block)
  const int nrb = rows >> 5;
  const int r = (int)(t % nrb) * 32 + rl;
  const size_t u = t / nrb;
  const int c = (int)(u % nch);
  const size_t blk = u / nch;
  uint4 v = src[((blk * rows + r) * nch + c) * 4 + s16];
  if (r == ones_row) {
    const int per = 16 / esz;                     // elements per 16-byte piece
    const int e0 = (c * 4 + s16) * per;           // first element (column) of this piece
    unsigned w[4] = {0u, 0u, 0u, 0u};
    for (int k = 0; k < per; ++k) {
      if (e0 + k >= ones_cols) break;
      if (esz == 2) w[k >> 1] |= 0x3f80u << (16 * (k & 1));       // bf16 1.0
      else w[k] = 0x3f800000u;                                     // fp32 1.0
    }
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  dst[i] = v;
}

}  // namespace

static int tile_k_impl(const void* src, void* dst, int rows, long long n_blocks, int row_bytes, int ones_row, int ones_cols,
                       int esz, void* stream) {
  if (!src || !dst || rows <= 0 || rows % 32 != 0 || n_blocks <= 0 || row_bytes <= 0 || row_bytes % 64 != 0) return GLR_EINVAL;
  const size_t total = (size_t)n_blocks * rows * (row_bytes / 16);
  const int nch = row_bytes / 64;
  hipLaunchKernelGGL(k_tile_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, (uint4*)dst, rows, nch, total, ones_row, ones_cols, esz);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}

extern "C" int glr_tile_k(const void* src, void* dst, int rows, long long n_blocks, int row_bytes, void* stream) {
  return tile_k_impl(src, dst, rows, n_blocks, row_bytes, -1, 0, 2, stream);
}

extern "C" int glr_tile_gram(const void* gram, void* gram_t, int S_pad, long long B, int S_eff, int op_dtype, void* stream) {
  if (op_dtype != GLR_F32 && op_dtype != GLR_BF16) return GLR_EDTYPE;
  if (S_eff <= 0 || S_eff > S_pad) return GLR_EINVAL;
  const int esz = op_dtype == GLR_F32 ? 4 : 2;
  return tile_k_impl(gram, gram_t, S_pad, B, S_pad * esz, S_eff < S_pad ? S_pad - 1 : -1, S_eff, esz, stream);
}

static int pack_regions_impl(const void* img_features, int in_dtype, int in_layout, const void* no_attn_vec, void* vt,
                             void* vt_t, int B, int D, int S, int op_dtype, void* stream) {
  if (!img_features || !vt || B <= 0 || D <= 0 || S <= 0 || D % 64 != 0) return GLR_EINVAL;
  if (in_layout != 0 && in_layout != 1) return GLR_EINVAL;
  if ((in_dtype != GLR_F32 && in_dtype != GLR_BF16) || (op_dtype != GLR_F32 && op_dtype != GLR_BF16)) return GLR_EDTYPE;
  const int shift = no_attn_vec ? 1 : 0;
  const int S_pad = glr_region_pad(S + shift);
  if (S_pad > GLR_MAX_SPAD) return GLR_EINVAL;
  const bool aligned = ((uintptr_t)img_features % 16 == 0) && (!no_attn_vec || (uintptr_t)no_attn_vec % 16 == 0);
  if (in_layout == 1 && in_dtype == GLR_BF16 && op_dtype == GLR_BF16 && aligned && D % 256 == 0) {
    // the training step's case: one pass that also emits the K-tiled copy
    const int D16 = D * 2 / 16;
    hipLaunchKernelGGL(k_pack_regions_cl16, dim3(D16 / 32, S_pad / 32, B), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)img_features, (const uint4*)no_attn_vec, (uint4*)vt, (uint4*)vt_t, D16, S, S_pad, shift);
    GLR_CHECK_LAUNCH();
    return GLR_OK;
  }
  hipLaunchKernelGGL(k_pack_regions, dim3(S_pad / 64, D / 64, B), dim3(256), 0, (hipStream_t)stream, img_features,
                     in_dtype, in_layout, no_attn_vec, vt, D, S, S_pad, shift, op_dtype);
  GLR_CHECK_LAUNCH();
  if (vt_t) return glr_tile_k(vt, vt_t, S_pad, B, D * (op_dtype == GLR_F32 ? 4 : 2), stream);
  return GLR_OK;
}

extern "C" int glr_pack_regions(const void* img_features, int in_dtype, int in_layout, const void* no_attn_vec,
                                void* vt, int B, int D, int S, int op_dtype, void* stream) {
  return pack_regions_impl(img_features, in_dtype, in_layout, no_attn_vec, vt, nullptr, B, D, S, op_dtype, stream);
}

extern "C" int glr_pack_regions_tiled(const void* img_features, int in_dtype, int in_layout, const void* no_attn_vec,
                                      void* vt, void* vt_t, int B, int D, int S, int op_dtype, void* stream) {
  if (!vt_t) return GLR_EINVAL;
  return pack_regions_impl(img_features, in_dtype, in_layout, no_attn_vec, vt, vt_t, B, D, S, op_dtype, stream);
}

extern "C" int glr_pack_words(const void* words_emb, int in_dtype, const int32_t* sent_slot0_dev,
                              const int32_t* cap_lens_dev, void* tp, float* tnorm, int B_txt, int D, int L,
                              int word_start, int n_slots, int capacity, int op_dtype, void* stream) {
  if (!words_emb || !sent_slot0_dev || !cap_lens_dev || !tp || !tnorm) return GLR_EINVAL;
  if (B_txt <= 0 || D <= 0 || L <= 0 || n_slots <= 0 || D % 64 != 0 || word_start < 0) return GLR_EINVAL;
  if (capacity != 32 && capacity != GLR_TILE_WORDS) return GLR_EINVAL;
  if ((in_dtype != GLR_F32 && in_dtype != GLR_BF16) || (op_dtype != GLR_F32 && op_dtype != GLR_BF16)) return GLR_EDTYPE;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = op_dtype == GLR_F32 ? 4 : 2;
  if (hipMemsetAsync(tp, 0, (size_t)n_slots * D * esz, st) != hipSuccess) return GLR_ELAUNCH;
  hipLaunchKernelGGL(k_pack_words, dim3(D / 64, B_txt), dim3(256), 0, st, words_emb, in_dtype, sent_slot0_dev,
                     cap_lens_dev, tp, D, L, word_start, capacity, op_dtype);
  GLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_word_norms, dim3((n_slots + 3) / 4), dim3(256), 0, st, tp, tnorm, n_slots, D, op_dtype);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
