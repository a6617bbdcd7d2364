// K1 launch parameters and constants shared by the local-attention kernels (glr_local_attn.hip: single-tile and pair
// kernels, forward and backward; glr_local_attn_t1.hip: the 4-wave single-tile forward).
#pragma once
#include "glr_common.h"

constexpr int TW = GLR_TILE_WORDS;  // 64 word slots per tile
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
  const int* item_tile;         // first tiles of the work items of this launch (single tiles or pairs)
  int n_items;
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
  float* amean;                 // fwd optional out: [B_img][n_sent][S_pad] word-mean attention row of every pair
  const float* damean;          // bwd optional in:  gradient w.r.t. amean
  const float* dattn;           // bwd optional in:  gradient w.r.t. the diagonal attention maps (layout of attn)
  const long long* attn_off;
  int strip;
  int pair_only, img_offset;
  int img_block;                // pair kernel: images per L2 group (block -> (image, item) mapping)
  const int* pair_desc;         // [n_pair][64] sentences + row flags of every forward pair (glr_plan_pair_desc)
  unsigned* a1buf;              // optional [B_img][n_pair][8 waves][2][3][8][64] fp16 pairs of a1 in the pair kernels' own
                                // register order: written by the forward, read by the backward instead of its score stream
  int a1_items, a1_base;        // pairs per image in a1buf (all pairs of the plan) and the index of this launch's first pair
#ifdef GLR_ABLATE
  int dbg;                      // diagnostic build only (libglr_ablate.so): phases to SKIP, GLR_K1_DBG bit mask
#endif
  // backward only
  const float* dsim;            // [B_img][ld_sim]
  unsigned char* xout;          // [n_slots][B_img][S_pad] op dtype
  unsigned char* aout;          // [B_img][n_slots][S_pad] a2
  float* gamma;                 // [B_img][n_slots]
  float* beta;                  // [B_img][n_slots]
  unsigned char* baout;         // optional [B_img][n_slots][S_pad] beta * a2 (the P3 operand image)
  // LDS carve (bytes)
  int off_img, off_small;
#ifdef GLR_STAMPS
  unsigned long long* stamps;   // diagnostic build only: [grid][12] s_memtime at phase boundaries
  unsigned long long* stamps2;  // same, for the pair kernel's grid
#endif
};


// K-tiled operand layout of the K1 streams (tp_t, vt_t, gram_t).  A block of R rows (R % 32 == 0: one image's regions,
// one tile's 64 word slots) x K bytes is stored as K / 64 chunks of R * 64 bytes; inside a chunk the 16-byte pieces
// are FRAGMENT-MAJOR: [32-row block][slot = 16-byte piece of the 64-byte row][row in block].  One MFMA fragment load of
// a wave (32 rows x the two pieces of a k-step) is then 1 KiB of contiguous memory, so an operand that only ONE wave
// needs goes straight from global memory to registers in full lines; the LDS rings of the DMA streams hold a chunk in
// this very order (a piece = a linear 1-KiB copy, fragment reads conflict-free without a swizzle).
__host__ __device__ __forceinline__ constexpr int glr_ktile_off(int row, int slot) {
  return (row >> 5) * 2048 + slot * 512 + (row & 31) * 16;
}

constexpr int PW_MAXSEG = 8;    // sentences per pair (planner: max_pair_seg); the spanning sentence of a long pair uses rows 0 / 1

// forward of the ordinary (two whole 64-slot tiles, <= 8 sentences) pairs, one 4-wave workgroup per tile
// (glr_local_attn_t1.hip); p.item_tile / p.pair_desc / p.n_items describe those pairs, p.a1_base their position in a1buf
int glr_k1_launch_tiles(LaParams& p, int op_dtype, void* stream);
