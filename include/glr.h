/*
 * glr.h - C ABI of libglr.so: the GLoRIA global+local contrastive hot path on MI355X (gfx950).
 *
 * The reference (strongbeamsprout/gloria-nlp-project) has NO FFI/plugin layer: its operator
 * boundary is the Python functions of gloria/loss/gloria_loss.py.  This header is
 * the native boundary those functions bind to in the MI355X build; every entry point names the
 * reference interface it replaces.  The Python side (gloria-nlp-project_amd/gloria/_native.py)
 * binds it with ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions (all entry points)
 *   - plain C types only: raw DEVICE pointers, sizes, floats.  No torch types.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Nothing is allocated,
 *     freed or synchronised inside; work is enqueued on `stream` and the call returns.
 *     Workspace is caller-provided; sizes come from the *_bytes queries.
 *   - return 0 on success, a negative GLR_E* code on error (bad shape / unsupported dtype /
 *     launch failure).  No global mutable state: calls are thread safe (every operand travels as an argument).
 *   - dtype codes: GLR_F32 = 0 (fp32 operands, fp32 MFMA, the 1e-4 parity mode),
 *                  GLR_BF16 = 1 (bf16 operands, fp32 accumulate / softmax / log / exp).
 *   - region features are `[B, D, S]` (NCHW with H*W = S flattened, region contiguous), word
 *     embeddings `[B, D, L]`, exactly the layouts the reference passes to local_loss.
 */
#ifndef GLR_H_
#define GLR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLR_F32 0
#define GLR_BF16 1

#define GLR_OK 0
#define GLR_EINVAL (-1)   /* bad shape / null pointer / unsupported size   */
#define GLR_EDTYPE (-2)   /* unsupported dtype code                        */
#define GLR_ELAUNCH (-3)  /* hip launch / attribute call failed            */

#define GLR_AGG_SUM 0     /* local_loss(agg="sum")  gloria_loss.py:154-155 */
#define GLR_AGG_MEAN 1    /* local_loss(agg!="sum") gloria_loss.py:156-157 */
#define GLR_AGG_MAX 2     /* get_local_similarities  gloria_model.py:199   */

#define GLR_TILE_WORDS 64 /* word slots per tile of the local-attention kernel */
#define GLR_MAX_SPAD 384  /* max padded region count (multiple of 64)          */
#define GLR_MAX_WORDS 512 /* longest sentence (words) the planner accepts          */

#define GLR_SRC_U8 0      /* element types of the raw images handed to glr_collate_images */
#define GLR_SRC_I16 1
#define GLR_SRC_F32 2

/* ABI version; bumped on any signature change. */
int glr_version(void);

/* Padded region count used by the packed operand layouts: S_eff rounded up to a multiple of 64. */
int glr_region_pad(int s_eff);

/* Populated word slots per 64-slot tile for an operand dtype: 64 (GLR_BF16) or 32 (GLR_F32: the
 * fp32 score tile and fp32 attention image of 64 words do not fit the 160 KB LDS together). */
int glr_tile_capacity(int op_dtype);

/* ------------------------------------------------------------------------------------------
 * Host-side planning: pack sentences into tiles of GLR_TILE_WORDS word slots.
 * Replaces the per-sentence slicing `words_emb[i, :, :cap_lens[i]]` of the reference loop
 * (gloria_loss.py:116-123): sentence i occupies cap_lens[i] consecutive slots.  A sentence of at
 * most `capacity` words lies inside one tile (first fit, in the given order, at most `capacity`
 * populated slots per tile); a longer one owns ceil(n / capacity) consecutive tiles of its own.
 *
 *   cap_lens[n_sent]    words per sentence (1..GLR_MAX_WORDS)
 *   capacity            glr_tile_capacity(op_dtype)
 *   max_pair_seg        0: plain first fit.  > 1 (the value later given to glr_plan_items): tiles will be paired, a pair
 *                       holding at most that many sentences - the planner then also caps the sentences per tile
 *                       where that lowers the number of work items (a few more, emptier tiles instead of crowded
 *                       tiles that cannot pair), and orders ordinary tiles fewest-sentences-next-to-most
 *   sent_slot0[n_sent]  out: global slot of the sentence's first word (tile * GLR_TILE_WORDS + pos);
 *                       word w of a multi-tile sentence sits at slot0 + (w / capacity) * GLR_TILE_WORDS
 *                       + w % capacity
 *   tile_first[cap]     out: for tile t, index of its first sentence in `order`; [n_tiles] = end.
 *                       Must hold (upper bound on tiles) + 1 ints; glr_plan_tiles_bound gives it.
 *   order[cap]          out: sentence ids in tile order (a multi-tile sentence appears once per tile)
 *   tile_nsub[cap]      out: 0 = ordinary tile, k > 1 = first tile of a k-tile sentence, -1 = its
 *                       continuation tiles
 * returns the number of tiles (>0) or a negative error.
 */
int glr_plan_tiles_bound(const int32_t* cap_lens, int n_sent, int capacity);
int glr_plan_tiles(const int32_t* cap_lens, int n_sent, int capacity, int max_pair_seg, int32_t* sent_slot0,
                   int32_t* tile_first, int32_t* order, int32_t* tile_nsub);

/* Work items of the local-attention kernels.  With allow_pairs, two consecutive ordinary tiles that hold
 * at most max_pair_seg (<= 8) sentences IN TOTAL become ONE forward work item: a workgroup then streams vt[b] and
 * gram[b] once for 128 words (the streams are the bound).  Outputs (each must hold n_tiles ints):
 *   single_tile  first tile of every un-paired item (ordinary tile or head of a multi-tile sentence)
 *   pair_tile    first tile of every pair: two ordinary tiles, or the two tiles owned by ONE sentence of 65..128 words
 *   all_tile     every item-head tile with pairs expanded (the backward kernel works tile by tile)
 *   counts[3]    number of entries of the three lists
 */
int glr_plan_items(const int32_t* tile_nsub, const int32_t* tile_first, int n_tiles, int allow_pairs,
                   int max_pair_seg, int32_t* single_tile, int32_t* pair_tile, int32_t* all_tile, int32_t* counts);

/* Row flags of the forward pair kernel (host).  In that kernel one wave holds all 64 word slots of a tile for its
 * region columns, 32 rows per lane half in word order; a sentence is a run of rows, and the kernel acts only where
 * a run starts or ends.  flags[n_tiles][8] (uint32, bit k = row k of the lane half): [0..1] run starts of half
 * 0 / 1, [2..3] run ends, [4..5] the run holding the sentence's first word, [6..7] reserved.  Replaces the
 * per-sentence slice boundaries of the reference loop (gloria_loss.py:122) inside a packed tile.  capacity must be
 * GLR_TILE_WORDS. */
int glr_plan_rowflags(const int32_t* cap_lens, const int32_t* sent_slot0, const int32_t* tile_first,
                      const int32_t* order, const int32_t* tile_nsub, int n_tiles, int capacity, uint32_t* flags);

/* Pair descriptors (host): per forward pair ONE 256-byte record with its sentences (ids, first slots, lengths) and the
 * row flags of both tiles - what a workgroup of the pair kernel reads (coalesced, once) instead of walking tile_first /
 * order / sent_slot0 / cap_lens.  desc[n_pair][64] int32: [0] sentences (<= 8), [1] long-pair flag, [8..15] sentence
 * ids, [16..23] first slot in the pair, [24..31] words, [32..39] / [40..47] glr_plan_rowflags of tile A / B. */
int glr_plan_pair_desc(const int32_t* cap_lens, const int32_t* sent_slot0, const int32_t* tile_first,
                       const int32_t* order, const int32_t* tile_nsub, int n_tiles, int capacity,
                       const int32_t* pair_tile, int n_pair, int32_t* desc);

/* ------------------------------------------------------------------------------------------
 * Operand packing (device).  HBM-bound layout/convert kernels.
 *
 * glr_pack_regions: img_features -> vt [B, S_pad, D] in op_dtype (region-major, feature
 *   contiguous: the B^T operand of the score contraction).  in_layout 0: img_features is
 *   [B, D, S] (NCHW, what the reference passes); in_layout 1: [B, S, D] (the same tensor in
 *   channels-last memory).  With no_attn_vec != NULL ([D], in_dtype) a learned column is PREPENDED
 *   (region 0), i.e. S_eff = S + 1 (gloria_loss.py:31-34).  Regions [S_eff, S_pad) are zero.
 *   This replaces context.view + cat + transpose(1,2).contiguous() of attention_fn (:30-35), once
 *   per step instead of once per sentence.
 *   The Gram matrices gram[b] = vt[b] . vt[b]^T ([B, S_pad, S_pad], op_dtype) the kernels also need
 *   are a plain batched GEMM left to the caller's BLAS (hipBLASLt via torch.bmm in the Python host).
 *
 * glr_pack_words: words_emb [B_txt, D, L] (in_dtype) -> tp [n_slots, D] (op_dtype), word-major,
 *   word w of sentence i <- words_emb[i, :, word_start + w] for w < cap_lens[i], at the slot given
 *   by the plan (gloria_loss.py:122; word_start = 1 is the inference slice of gloria_model.py:179).
 *   Unused slots are zero.  tnorm[n_slots] (fp32) = L2 norm of each packed word as rounded to
 *   op_dtype (gloria_loss.py:14).
 */
int glr_pack_regions(const void* img_features, int in_dtype, int in_layout, const void* no_attn_vec, void* vt,
                     int B, int D, int S, int op_dtype, void* stream);

/* glr_pack_regions_tiled: the same packing that ALSO writes vt_t, the K-tiled copy the K1 streams read (layout of
 * glr_tile_k with rows = S_pad) - for channels-last bf16 features in ONE pass over the input (16 bytes per thread). */
int glr_pack_regions_tiled(const void* img_features, int in_dtype, int in_layout, const void* no_attn_vec, void* vt,
                           void* vt_t, int B, int D, int S, int op_dtype, void* stream);

int glr_pack_words(const void* words_emb, int in_dtype, const int32_t* sent_slot0_dev,
                   const int32_t* cap_lens_dev, void* tp, float* tnorm, int B_txt, int D, int L,
                   int word_start, int n_slots, int capacity, int op_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K1  glr_local_attn_fwd / glr_local_attn_bwd - fused region x word attention + cosine +
 * log-sum-exp and its gradient.  Replaces, for ALL image x sentence pairs at once, the body of the
 * reference sentence loop: attention_fn (gloria_loss.py:19-63), cosine_similarity (:11-16) and the
 * exp/sum/log of local_loss (:150-158, :164), and torch autograd through them.
 *
 *   sim[b, i] = temp3 * log( agg_w exp(temp2 * cos(T[i,:,w], c[b,i,w,:])) )
 *
 * One workgroup per (image b, word tile); kernel anatomy in DESIGN.md.
 *
 *   vt, gram     packed regions of the B_img LOCAL images and their Gram matrices, K-TILED (glr_tile_k, rows = S_pad)
 *   tp, tnorm    packed words of ALL sentences (glr_pack_words); tp K-TILED (glr_tile_k, rows = 64)
 *   sent_slot0, cap_lens   [n_sent] device int32 (same arrays as given to glr_pack_words)
 *   tile_first, order, tile_nsub   device int32 copies of the glr_plan_tiles outputs
 *   single_tile/n_single, pair_tile/n_pair (fwd), item_tile/n_items (bwd: the all_tile list)
 *                device int32 copies of the glr_plan_items outputs
 *   n_long_pair  (fwd) how many LEADING entries of pair_tile are the two tiles of ONE 65..128-word sentence (the planner
 *                lists multi-tile sentences first).  Those run the 8-wave pair kernel; the ordinary pairs (two whole
 *                tiles) run one 4-wave workgroup per tile, two workgroups per CU (csrc/glr_local_attn_t1.hip).
 *   pair_desc    (fwd) device copy of the glr_plan_pair_desc output [n_pair][64]; required when n_pair > 0.  The
 *                pair kernels also need S_eff < S_pad and expects gram[b] to carry ONES in row S_pad - 1, columns
 *                r < S_eff (a padded region; glr_tile_gram writes it): the second contraction then delivers
 *                Z_w = sum_r e2[w, r] in output column S_pad - 1.  Every other kernel masks padded regions, so the
 *                same gram serves them and the backward.
 *   sim          fp32 [B_img, ld_sim]; column = sentence id (fwd: out, bwd: in)
 *   lse          fp32 [B_img, n_sent, S_pad]: log-sum-exp over the words of sentence i of the
 *                scores of region r (fwd: optional out, needed by bwd)
 *   wstat        fp32 [B_img, n_slots, 4]: per word Z, cosine, |c|^2, 0 (fwd: optional out, bwd: in)
 *   attn         optional out fp32: attention maps of the DIAGONAL pairs only
 *                (image b with sentence img_offset + b; gloria_loss.py:141-143), packed:
 *                sentence i at attn_off[i] floats, [cap_lens[i], S_eff - strip] row-major where
 *                strip = 1 drops the no-attention column (:60-61).  NULL = not wanted.
 *   pair_only    1: launch only the B_img diagonal (image b, tile of sentence img_offset+b) pairs;
 *                sim then only receives the diagonal entries (attention_fn, attention-finetune).
 *   img_offset   global index of local image 0 (data-parallel shard offset).
 *   amean        optional out fp32 [B_img, n_sent, S_pad]: the word-mean attention row of EVERY pair,
 *                A[b, i, r] = mean_w a2[b, i, w, r] (no-attention column at r = 0 when present, zeros for
 *                r >= S_eff) - the input of the attention regularisers (gloria_loss.py:129-139; K6
 *                glr_attn_reg_fwd).  NULL = not wanted.  bwd: damean = gradient w.r.t. amean (optional in).
 *   dattn        bwd optional in: gradient w.r.t. the diagonal attention maps, same packed layout as attn
 *                (attn_off / strip / img_offset as in the forward) - the attention-supervision loss (K4).
 *
 * bwd outputs, consumed by three plain GEMMs on the caller's BLAS:
 *   xout  [n_slots, B_img, S_pad] op dtype   X = ds + alpha*a2 :
 *                dT_packed = X2d . vt2d - gamma_sum * T,   dvt2d = X2d^T . tp - P . vt
 *   aout  [B_img, n_slots, S_pad] op dtype   a2;   beta [B_img, n_slots] fp32 :
 *                P[b] = (beta[b] * aout[b])^T . aout[b]
 *   baout [B_img, n_slots, S_pad] op dtype   optional (NULL = not wanted): beta * a2, the first factor of P, as the
 *                kernel's own second-contraction operand holds it
 *   gamma [B_img, n_slots] fp32              coefficient of T_w in dT (from the word norm)
 * with X2d = xout viewed [n_slots, B_img*S_pad] and vt2d = vt viewed [B_img*S_pad, D].
 *   a1buf  optional, both directions (NULL = none): n_pair * B_img * 98304 bytes.  The forward pair kernel leaves the
 *          word-softmax values a1 of every pair there (fp16, in the pair kernels' own register order - opaque to the
 *          caller); given the same buffer, the backward pair kernel reads them instead of re-streaming vt[b] and the
 *          word tiles for the scores (s = lse + log a1).
 * The backward takes the forward's work items: single tiles and (bf16, 384 regions, no damean / dattn) pairs with
 * their descriptors; with damean / dattn every tile must be passed as a single tile.
 */
int glr_local_attn_fwd(const void* vt, const void* gram, const void* tp, const float* tnorm,
                       const int32_t* sent_slot0, const int32_t* cap_lens, const int32_t* tile_first,
                       const int32_t* order, const int32_t* tile_nsub, const int32_t* single_tile, int n_single,
                       const int32_t* pair_tile, int n_pair, int n_long_pair, const int32_t* pair_desc, int n_tiles,
                       int n_sent, int B_img, int D, int S_eff, float temp1, float temp2, float temp3, int agg, float eps,
                       float* sim, int ld_sim, float* lse, float* wstat, float* attn, const int64_t* attn_off,
                       int strip, int pair_only, int img_offset, float* amean, void* a1buf, int op_dtype, void* stream);

int glr_local_attn_bwd(const void* vt, const void* gram, const void* tp, const float* tnorm,
                       const int32_t* sent_slot0, const int32_t* cap_lens, const int32_t* tile_first,
                       const int32_t* order, const int32_t* tile_nsub, const int32_t* single_tile, int n_single,
                       const int32_t* pair_tile, int n_pair, const int32_t* pair_desc,
                       int n_tiles, int n_sent, int B_img, int D, int S_eff, float temp1, float temp2, float temp3,
                       int agg, float eps, const float* sim, const float* dsim, int ld_sim, const float* lse,
                       const float* wstat, const float* damean, const float* dattn, const int64_t* attn_off,
                       int strip, int img_offset, void* xout, void* aout, void* baout, float* gamma, float* beta,
                       const void* a1buf, int op_dtype, void* stream);

/* K-tiling of the K1 operands (device, HBM-bound copy).  glr_local_attn_fwd / _bwd take vt, gram and tp in
 * the K-TILED layout: every block of `rows` rows (vt, gram: the S_pad rows of one image; tp: the 64 slots of one
 * tile) is stored as [row_bytes / 64] chunks of rows * 64 bytes, and inside a chunk FRAGMENT-MAJOR:
 * [rows / 32][4 x 16-byte slot of the row's 64 bytes][32 rows][16 bytes].  The MFMA fragment one wave loads per k-step
 * (32 rows x 2 slots) is then 1 KiB of contiguous memory - operands only one wave needs go straight to registers in
 * whole lines - and an LDS-DMA piece of the K1 streams (half a block) is a linear 1-KiB copy.
 *   src   row-major [n_blocks * rows][row_bytes]     dst  same size, tiled     row_bytes % 64 == 0, rows % 32 == 0
 */
int glr_tile_k(const void* src, void* dst, int rows, long long n_blocks, int row_bytes, void* stream);
/* glr_tile_gram: K-tiling of the Gram matrices gram [B, S_pad, S_pad] (op_dtype) that also writes the ONES ROW the
 * forward pair kernel expects when S_eff < S_pad (row S_pad - 1: ones in columns r < S_eff, zeros after; see
 * tile_rowflags of glr_local_attn_fwd) - the row-major gram stays untouched. */
int glr_tile_gram(const void* gram, void* gram_t, int S_pad, long long B, int S_eff, int op_dtype, void* stream);


/* ------------------------------------------------------------------------------------------
 * K2  dual cross-entropy on a square similarity matrix (labels = arange).
 * Replaces the two nn.CrossEntropyLoss calls of local_loss (gloria_loss.py:167-170) and
 * global_loss (:86-87).
 *
 * fwd: sim [B, B] fp32 -> lse_row[B], lse_col[B], losses[2] = (mean_b lse_row[b]-sim[b,b],
 *      mean_i lse_col[i]-sim[i,i]).
 * bwd: dsim[r - row0, i] = g0/B * (exp(sim[r,i]-lse_row[r]) - [r==i])
 *                        + g1/B * (exp(sim[r,i]-lse_col[i]) - [r==i])   for r in [row0,row0+n_rows)
 *      g points to two DEVICE floats (upstream gradients of loss0 / loss1).
 */
int glr_dual_ce_fwd(const float* sim, int B, float* lse_row, float* lse_col, float* losses, void* stream);
int glr_dual_ce_bwd(const float* sim, int B, const float* lse_row, const float* lse_col, const float* g,
                    int row0, int n_rows, float* dsim, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3  global similarity matrix  sim[b,i] = temp3 * <I_b,T_i> / max(|I_b|*|T_i|, eps).
 * Replaces the norm / bmm / clamp / scale of global_loss (gloria_loss.py:75-80): ONE launch per direction on the
 * matrix cores (exact-fp32 MFMA), row norms fused into the contraction loop, clamp / scale in the epilogue.
 *   img [B_img, D], txt [B_txt, D] fp32;  sim [B_img, ld_sim] fp32; ni[B_img], nt[B_txt] norms (out).
 * bwd: given the forward's sim and norms and dsim [B_img, ld_sim] -> dimg [B_img, D] and dtxt [B_txt, D], both
 *      as MFMA GEMMs whose coefficient operand is formed in registers, incl. torch's sub-gradient of the clamp
 *      (dtxt holds only the contribution of these B_img images: sum across ranks in the data-parallel case).
 */
int glr_global_sim_fwd(const float* img, const float* txt, int B_img, int B_txt, int D, float temp3,
                       float eps, float* sim, int ld_sim, float* ni, float* nt, void* stream);
int glr_global_sim_bwd(const float* img, const float* txt, const float* ni, const float* nt, const float* sim,
                       const float* dsim, int ld_sim, int B_img, int B_txt, int D, float temp3, float eps, float* dimg,
                       float* dtxt, void* stream);

/* ------------------------------------------------------------------------------------------
 * K5  word-piece -> word aggregation (segment sum) fused with the reduction over the last BERT layers
 * and the mean over the L word slots.  Replaces BertEncoder.aggregate_tokens + the post-processing of
 * BertEncoder.forward (gloria/models/text_model.py:32-90, 96-131).
 *   hidden[n_layers]  device pointers to [B, L, D] hidden states (in_dtype), n_layers <= 16, any D
 *   dst [B, L] int32  word slot of every token (-1 = dropped), computed on the host from the ids
 *   word_emb [B, D, L] fp32 (the layout local_loss takes), sent_emb [B, D] fp32
 *   mean_layers       0: sum over layers (aggregate_method 'sum'), 1: mean
 * bwd: d_hidden [B, L, D] (out_dtype) is the gradient for EVERY one of the n_layers inputs.
 */
int glr_wordpiece_segsum_fwd(const void* const* hidden, int n_layers, int in_dtype, const int32_t* dst,
                             float* word_emb, float* sent_emb, int B, int L, int D, int mean_layers, void* stream);
int glr_wordpiece_segsum_bwd(const float* d_word, const float* d_sent, const int32_t* dst, void* d_hidden,
                             int out_dtype, int B, int L, int D, int n_layers, int mean_layers, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4  attention-supervision loss on the diagonal attention maps (gloria/models/gloria_model.py:143-147):
 *   loss_b[b] = -log( sum(label_b * U_b) / sum(U_b) ),  U_b = nearest-upsample(mean_w attn_b[w]) to Hl x Wl,
 * computed from per-region label pixel counts, never materialising U.  attn / attn_off / cap_lens are
 * the packed diagonal maps as written by glr_local_attn_fwd (strip applied, S = ih*iw per row).
 * dmap (optional, same packing): d loss_b / d attn element (multiply by weight / B upstream).
 * labels: [B, Hl, Wl] bytes (0 / non-zero).
 */
int glr_attn_sup_fwd(const float* attn, const int64_t* attn_off, const int32_t* cap_lens, int img_offset,
                     const uint8_t* labels, int B, int Hl, int Wl, int ih, int iw, float* loss_b, float* dmap,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-wise cosine similarity  <x1,x2> / max(|x1|*|x2|, eps)  (gloria/loss/gloria_loss.py:11-16), fp32
 * [rows, D] inputs.  stats [rows, 3] (dot, |x1|, |x2|) feeds the backward.
 */
int glr_cosine_fwd(const float* x1, const float* x2, int rows, int D, float eps, float* out, float* stats,
                   void* stream);
int glr_cosine_bwd(const float* x1, const float* x2, const float* stats, const float* g, int rows, int D, float eps,
                   float* dx1, float* dx2, void* stream);

/* ------------------------------------------------------------------------------------------
 * K6  attention regularisers of local_loss (gloria_loss.py:108-114, 129-139, 172-199) on the word-mean
 * attention rows of ALL pairs (amean of glr_local_attn_fwd).  With shift = 1 (no-attention column present)
 * the rows are P = [1 - sum_{r>=1} A[r], A[1:]] (:133-135), else P = A.
 *   out[b, 0] = sum_i entropy(P[b, i])                               entropy (:95-96)
 *   out[b, 1] = sum_{i != d} 1/2 [KL(P[b,d] || P[b,i]) + KL(P[b,i] || P[b,d])],  d = img_offset + b   (:180-190)
 *   out[b, 2] = log(1 - sum_{r >= shift} A[b, d, r])                 no-attention score of the diagonal pair (:130)
 * The caller forms the reference's means / weights.  bwd: damean = gradient of
 * coef[0] * sum_b out[b,0] + coef[1] * sum_b out[b,1] + coef[2] * sum_b out[b,2]  (coef: 3 device floats).
 * HBM-bound (B_img * n_sent * S_pad * 4 bytes read); fixed summation order (bitwise reproducible).
 */
int glr_attn_reg_fwd(const float* amean, int B_img, int n_sent, int S_pad, int S_eff, int shift, int img_offset,
                     float* out, void* stream);
int glr_attn_reg_bwd(const float* amean, int B_img, int n_sent, int S_pad, int S_eff, int shift, int img_offset,
                     const float* coef, float* damean, void* stream);

/* ------------------------------------------------------------------------------------------
 * Exact selection on fp32 scores (MSD radix select on the order-preserving bit pattern: the SAME element a
 * CPU sort picks, bit exact).  Ties rank by DESCENDING index (a reversed stable ascending argsort); -0.0 == +0.0;
 * NaN ranks above +inf.  rows independent problems of n contiguous floats each, n < 2^32.
 *   glr_kth_value        out[row] = k-th SMALLEST value (k 1-indexed)
 *                        == torch.topk(preds, k, largest=False).values.max()   (gloria/lightning/callbacks.py:56)
 *   glr_topk_desc        idx[row, :k] = np.argsort(x)[::-1][:k]  (gloria/models/retrival_model.py:118), k <= 1024;
 *                        val (optional) the scores in that order
 *   glr_threshold_counts out[row] = { #(pred > thr & target), #(pred > thr), #(target), #(pred > thr | target) }:
 *                        the counts behind precision / recall / F1 / IoU at a percentile threshold (callbacks.py:57-61)
 */
int glr_kth_value(const float* x, int rows, long long n, long long k, float* out, void* stream);
/* Footprint counts of nearest upsampling (gloria/lightning/callbacks.py:319 nn.Upsample(size=image_shape), torch's
 * fp32 floor(dst * in / out) rule): for every cell of an ih x iw map, npix = pixels of the Hl x Wl overlay that copy
 * it and cnt = those of them set in labels [B, Hl, Wl] (uint8).  cnt, npix: int32 [B, ih, iw].  With them the
 * localization metrics (callbacks.py:38-70) are exact statistics of ih * iw weighted cell values: the overlay is
 * never materialised. */
int glr_cell_counts(const uint8_t* labels, int B, int Hl, int Wl, int ih, int iw, int32_t* cnt, int32_t* npix,
                    void* stream);
int glr_topk_desc(const float* x, int rows, long long n, int k, int64_t* idx, float* val, void* stream);
int glr_threshold_counts(const float* pred, const uint8_t* target, const float* thr, int rows, long long n,
                         uint64_t* out, void* stream);

/* MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem (torchvision resnet50 through cnn_backbones.py:31-35) on
 * channels-last bf16: x [B, H, W, C] -> y [B, Ho, Wo, C], Ho = (H - 1) / 2 + 1; idx uint8 [B, Ho, Wo, C] = position 0..8 of
 * the first maximum inside its window (instead of torch's int64 argmax); the backward gathers (no zero fill, no scatter). */
int glr_maxpool3s2_fwd(const void* x, int B, int H, int W, int C, void* y, uint8_t* idx, void* stream);
int glr_maxpool3s2_bwd(const void* dy, const uint8_t* idx, int B, int H, int W, int C, void* dx, void* stream);

/* Input resize of the image encoder: F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=True)
 * (gloria/models/vision_model.py:68) fused with the layout copy and the bf16 cast autocast puts in front of conv1.
 * x fp32 [B, C, Hi, Wi] with element strides (sn, sc, sh, sw) (NCHW or channels-last); y bf16 [B, Ho, Wo, C]. */
int glr_upsample_bilinear_cl(const float* x, long long sn, long long sc, long long sh, long long sw, int B, int C, int Hi,
                             int Wi, int Ho, int Wo, void* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused training-mode BatchNorm2d (+ residual add) (+ ReLU) on channels-last bf16 activations: the 53 normalisation
 * sites of the ResNet-50 image encoder (reference: torchvision resnet50 through
 * gloria/models/cnn_backbones.py:31-35, vision_model.py:67-86).  x / y / dy / dx / residual / dres: bf16
 * [R = N*H*W, C] (NHWC memory), C a power of two in [8, 2048]; `dtype` GLR_BF16 (the training configuration) or GLR_F32 (the
 * fp32 parity configuration) is the element type of x / residual / y / dy / dy2 / dx / dres.
 *   fwd   mean, invstd [C] out (batch statistics, biased variance + eps); run_mean / run_var updated with
 *         `momentum` and the unbiased variance like nn.BatchNorm2d (NULL = no running statistics); num_batches_tracked
 *         (int64 scalar, NULL = none) is incremented by the same launch;
 *         y = relu?( (x - mean) invstd gamma + beta (+ residual) )
 *   bwd   dx; out4c = [dgamma | dbeta | 2C floats of scratch]; with a residual also dres = dy * [y > 0] (the
 *         gradient of the skip branch), y = the forward's output; dy2 (optional, with a residual only): a second
 *         gradient tensor of y, added to dy while it is read (the next block's main and skip consumers)
 *   workspace: glr_bn_workspace_floats(R, C) floats (0 = shape not supported).
 * HBM-bound: 3 (4) tensor passes forward, 5 (7) backward; fixed-order two-level reductions (bitwise reproducible).
 */
int glr_bn_workspace_floats(long long R, int C);
int glr_bn_act_fwd(const void* x, const void* residual, const float* gamma, const float* beta, long long R, int C,
                   float eps, float momentum, int relu, float* run_mean, float* run_var, long long* num_batches_tracked,
                   float* mean, float* invstd, float* workspace, void* y, int dtype, void* stream);
int glr_bn_act_bwd(const void* x, const void* dy, const void* dy2, const void* y, const float* gamma, const float* beta, const float* mean,
                   const float* invstd, long long R, int C, int relu, int has_residual, float* workspace, float* out4c,
                   void* dx, void* dres, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused  y = LayerNorm(dropout(h) + inp)  of the BERT sub-layer outputs (reference: transformers'
 * BertSelfOutput / BertOutput inside the BertModel of gloria/models/text_model.py:18-20, run under native AMP).
 *   h16    bf16 [R, H]  output of the sub-layer's dense Linear        inp32  fp32 [R, H]  the residual stream
 *   out32  fp32 [R, H]  y (the next residual stream)                  out16  bf16 [R, H]  y rounded (next GEMM operand)
 *   stats  fp32 [R, 2]  mean, 1/std per row (for the backward)        mask   u64 [R, H/64] dropout keep bits: element
 *                       4 (l + 64 i) + c of a row is bit l of word 4 i + c  (NULL when p_drop == 0)
 *   seed / offset       Philox4x32-10 key / counter prefix: the mask is a pure function of (seed, offset, row, column)
 *   rng_cell            NULL, or device memory {seed, offset base}: key = (cell[0], cell[1] + offset) (hipGraph replays)
 * bwd: dy32 / dy16 = gradients w.r.t. out32 / out16 (either may be NULL); d_inp32 fp32, d_h16 bf16, dgamma / dbeta fp32 [H];
 *      dhsum (NULL = not wanted): column sums of d_h, fp32 or bf16 [H] (dhsum_bf16) - the bias gradient of the dense
 *      Linear that produced h;  workspace: glr_ln_workspace_floats(R, H) floats (0 = shape not supported).
 *      H in {256, 512, 768, 1024}.
 * HBM-bound: 12 bytes per element forward, 18 backward; fixed-order reductions (bitwise reproducible).
 * glr_colsum_bf16: out[c] = sum_r x16[r, c] of a contiguous bf16 [R, C] matrix, C % 256 == 0, fp32 accumulation in a fixed
 *      order, out fp32 or bf16 (out_bf16) - the bias gradient of the other Linears (sum over tokens of dy: torch's
 *      linear backward does it with aten's generic reduce_kernel); workspace: glr_colsum_workspace_floats(R, C) floats.
 */
int glr_ln_workspace_floats(long long R, int H);
int glr_drop_add_ln_fwd(const void* h16, const float* inp32, const float* gamma, const float* beta, long long R, int H,
                        float eps, float p_drop, unsigned long long seed, unsigned long long offset,
                        const unsigned long long* rng_cell, float* out32, void* out16, float* stats, unsigned long long* mask,
                        void* stream);
int glr_drop_add_ln_bwd(const float* dy32, const void* dy16, const void* h16, const float* inp32, const float* gamma,
                        const float* stats, const unsigned long long* mask, long long R, int H, float p_drop,
                        float* d_inp32, void* d_h16, float* workspace, float* dgamma, float* dbeta, void* dhsum, int dhsum_bf16,
                        void* stream);
/* Gradients of BertEmbeddings' lookup tables (torch: embedding_dense_backward, a device sort + ~17 launches per table).
 *   glr_embedding_bwd       dW[seg_tok[u], :] = sum over k in [seg_lo[u], seg_hi[u]) of dy[order[k], :], rows added in that
 *                           order (fp32 [tokens, D] in, fp32 table out; rows of no segment keep what dW holds: pass zeros).
 *                           order / seg_*: the HOST's stable argsort of the token ids and its runs (padding run left out).
 *   glr_type_embedding_bwd  two-row table: dW2[1] = sum of dy rows with token_type != 0, dW2[0] = the others (two-level
 *                           fixed-order sums); workspace glr_type_embedding_workspace_floats(R, D) floats, D % 256 == 0.
 */
int glr_embedding_bwd(const float* dy, const int32_t* order, const int32_t* seg_lo, const int32_t* seg_hi, const int32_t* seg_tok,
                      int n_seg, int D, float* dW, void* stream);
int glr_type_embedding_workspace_floats(long long R, int D);
int glr_type_embedding_bwd(const float* dy, const int64_t* token_type, long long R, int D, float* workspace, float* dW2,
                           void* stream);
int glr_colsum_workspace_floats(long long R, int C);
int glr_colsum_bf16(const void* x16, long long R, int C, float* workspace, void* out, int out_bf16, void* stream);

/* ------------------------------------------------------------------------------------------
 * Self-attention of the BERT text encoder for short captions (reference: BertSelfAttention of
 * transformers' BertModel, gloria/models/text_model.py:18-20; 97 tokens in imagenome_pretrain_config.yaml):
 *   O = dropout(softmax(Q K^T * scale + key mask)) V      per (sentence, head), head size 64, L <= 128 tokens.
 * q, k, v, dq, dk, dv: bf16 rows of ld elements, o, d_o: rows of ld_o elements, head h in columns [64 h, 64 h + 64) (the
 * Linear outputs read in place: ld = hidden size for three Linears, 3 x hidden for one fused query|key|value Linear whose
 * three column blocks are passed as q, k, v); key_mask uint8 [B, L] (nonzero = attend) or NULL; lse fp32 [B * n_heads, 128] (row
 * log-sum-exp, saved for the backward); keep uint32 [B * n_heads, 128, 4]: dropout keep bits, key 32 j + i of query
 * row r = bit i of word (r, j) (NULL when p_drop == 0).  seed / offset: key of the counter hash that draws the bits;
 * rng_cell (NULL = not used): device memory {seed, offset base} read by the kernel, key = (cell[0], cell[1] + offset) -
 * launches captured in a hipGraph get fresh bits per replay by rewriting the cell (gloria/models/rng.py).
 * One workgroup per (sentence, head) and pass (the backward runs a query pass and a key pass); glr_attn_max_tokens(
 * backward) = largest supported L (128).  Replaces a library flash-attention call that spends 95 + 390 us per layer here.
 */
int glr_attn_max_tokens(int backward);
int glr_attn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, int B, int n_heads, int L, int ld,
                 int ld_o, float scale, float p_drop, unsigned long long seed, unsigned long long offset,
                 const unsigned long long* rng_cell, void* o, float* lse, uint32_t* keep, void* stream);
int glr_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const uint8_t* key_mask,
                 const float* lse, const uint32_t* keep, int B, int n_heads, int L, int ld, int ld_o, float scale, float p_drop,
                 void* dq, void* dk, void* dv, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image half of the collate function: replaces, for a ragged batch of single-channel images,
 *   original_tensor_to_numpy_image   gloria/datasets/mimic_for_gloria.py:36-42  (min-max -> uint8, truncating)
 *   GloriaCollateFn._resize_img      mimic_for_gloria.py:136-181   (cv2.INTER_AREA long side -> scale, zero pad)
 *   GloriaCollateFn.process_img      mimic_for_gloria.py:120-133   ("L"->"RGB", crop, ToTensor, Normalize(0.5,0.5);
 *                                    transform of gloria/builder.py:159-201 for configs/imagenome_pretrain_config.yaml)
 * src     one device buffer holding the B images back to back (row-major H x W, element type src_dtype);
 * offset  [B] device int64: byte offset of each image in src (16-byte aligned offsets use vector loads);
 * desc    [B][8] device int32 per image: { H, W, dst_h, dst_w, pad_top, pad_left, crop_top, crop_left } where
 *         dst_h x dst_w is the resized size (smaller OR larger than H x W), pad_* its position in the scale x scale
 *         frame and crop_* the crop window's corner in that frame (host planning: gloria/datasets/collate.py);
 * state   [B][2] device uint32: per-image min / max as order-preserving keys, written by glr_image_minmax and read
 *         by glr_collate_images; NULL = src already holds the 8-bit image (src_dtype GLR_SRC_U8);
 * out     float32 [B, 3, crop, crop] (crop <= 256), values in [-1, 1], three equal channels; or
 * out_u8  uint8 [B, crop, crop]: the cropped 8-bit image, input of the transform passes below (exactly one of the two).
 * One thread per output pixel evaluates OpenCV 4.5's INTER_AREA cell (integer-scale fast path incl. the 2x2 8-bit
 * rounding, general fp32 tap path in OpenCV's summation order; when the image is enlarged: cv::resize's fixed-point
 * bilinear emulation with area coordinates) from the source directly: no intermediate image is materialised and only
 * pixels under the crop window are read.  Bit-exact against oracle/collate_oracle.py;
 * HBM-bound (algorithmic bytes: H*W*esz read once for min-max, <= H*W*esz read + 12*crop^2 written by the collate).
 *
 * Random transforms of gloria/builder.py:167-186 (torchvision 0.8.2 RandomHorizontalFlip / RandomAffine / ColorJitter
 * on the PIL image; parameters drawn on the HOST - gloria/datasets/collate.py draw_augmentation) on uint8 [B, size, size]:
 *   glr_aug_geom     dst = affine(flip(src)): flip [B] int32 (non-zero = mirrored), matrix [B][6] float64 = the
 *                    coefficients PIL's Image.transform(AFFINE) takes (NaN in [0] = no affine; NULL = none at all),
 *                    nearest neighbour, zero fill, PIL's arithmetic (Geometry.c ImagingScaleAffine / affine_fixed)
 *   glr_aug_jitter   in place, ONE ImageEnhance step per call: kind [B] int32 (0 none, 1 brightness, 2 contrast),
 *                    alpha [B] float (the factor); sums_ws [B] uint64 scratch (per-image grey sums for the contrast mean)
 *   glr_u8_to_tensor ToTensor + Normalize(0.5, 0.5): float32 [B, 3, size, size]
 * Bit-exact against oracle/collate_oracle.py, whose Pillow half is pinned against Pillow itself in the CPU tests.
 */
int glr_image_minmax(const void* src, const int64_t* offset, const int32_t* desc, int B, int src_dtype,
                     uint32_t* state, void* stream);
int glr_collate_images(const void* src, const int64_t* offset, const int32_t* desc, const uint32_t* state, int B,
                       int src_dtype, int crop, float* out, uint8_t* out_u8, void* stream);
int glr_aug_geom(const uint8_t* src, uint8_t* dst, int B, int size, const int32_t* flip, const double* matrix, void* stream);
int glr_aug_jitter(uint8_t* img, int B, int size, const int32_t* kind, const float* alpha, uint64_t* sums_ws, void* stream);
int glr_u8_to_tensor(const uint8_t* img, int B, int size, float* out, void* stream);
/* diagnostic: *bad += number of integer pairs 0 <= n <= d, d in [d_lo, d_hi), d_hi <= 2^17 + 1, for which the collate kernel's
 * shortened fp32 division differs from the `/` operator (*bad must be zeroed by the caller; expected to stay 0) */
int glr_selftest_quotient(int d_lo, int d_hi, uint64_t* bad, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimisation step over FLAT parameter buffers (the reference delegates it to Lightning: Adam(betas=(0.5, 0.999)),
 * gradient_clip_val 0.25, native AMP - gloria/builder.py:84-87, run.py:172-207).  n % 8 == 0 (pad the buffers).
 *   glr_sumsq_blocks    number of partial sums glr_sumsq_partial writes for n elements
 *   glr_sumsq_partial   partial[blocks] = fixed-order partial sums of squares of x (GLR_F32 / GLR_BF16)
 *   glr_clip_coef       out[0] = sqrt(sum of all partials) (the global gradient norm), out[1] = the coefficient of
 *                       torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6)); stays on the device
 *   glr_adam_step       torch.optim.Adam on fp32 master weights: g = clip[1] * grad + weight_decay * p,
 *                       m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2, p -= lr / (1 - b1^step) * m /
 *                       (sqrt(v) / sqrt(1 - b2^step) + eps); writes the bf16 shadow of p when shadow_bf16 != NULL.
 *                       clip may be NULL (no clipping).
 */
/* Pointer-table forms (single process: gradients stay one tensor per parameter, nothing is accumulated into a flat
 * buffer).  chunk_table: device array of { int32 param, int32 count, int64 offset in the parameter, int64 offset in
 * the flat buffers } (24 bytes, count <= 16384, both offsets multiples of 8), one workgroup per entry;
 * grad_ptrs[param]: device address of that parameter's gradient this step, 0 = no gradient (parameter skipped, as
 * torch.optim.Adam skips None).  glr_sumsq_mt writes one partial per chunk. */
/* glr_gather_mt: copy the gradients named by (a range of) the chunk table into their slots of the flat gradient buffer
 * `flat` (element offsets = the table's flat offsets), one launch: how a data-parallel rank fills an all-reduce bucket
 * when the bucket's last gradient has arrived. */
int glr_gather_mt(const void* chunk_table, int n_chunks, const uint64_t* grad_ptrs, int dtype, void* flat, void* stream);
int glr_sumsq_mt(const void* chunk_table, int n_chunks, const uint64_t* grad_ptrs, int dtype, float* partial,
                 void* stream);
int glr_adam_step_mt(const void* chunk_table, int n_chunks, const uint64_t* grad_ptrs, int grad_dtype, float* master,
                     float* exp_avg, float* exp_avg_sq, void* shadow_bf16, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, const float* clip, void* stream);
int glr_sumsq_blocks(long long n);
int glr_sumsq_partial(const void* x, int dtype, long long n, float* partial, void* stream);
int glr_clip_coef(const float* partial, int n_partial, float max_norm, float* out, void* stream);
int glr_adam_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_dtype, void* shadow_bf16,
                  long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  const float* clip, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLR_H_ */
