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
 *     launch failure).  No global mutable state: calls are thread safe.
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

/* ABI version; bumped on any signature change. */
int glr_version(void);

/* Padded region count used by the packed operand layouts: S_eff rounded up to a multiple of 64. */
int glr_region_pad(int s_eff);

/* ------------------------------------------------------------------------------------------
 * Host-side planning: pack sentences into tiles of GLR_TILE_WORDS word slots.
 * Replaces the per-sentence slicing `words_emb[i, :, :cap_lens[i]]` of the reference loop
 * (gloria_loss.py:116-123): sentence i occupies cap_lens[i] consecutive slots.  A sentence of at
 * most GLR_TILE_WORDS words lies inside one tile (first fit, in the given order); a longer one
 * owns ceil(n / GLR_TILE_WORDS) consecutive tiles of its own.
 *
 *   cap_lens[n_sent]    words per sentence (1..GLR_MAX_WORDS)
 *   sent_slot0[n_sent]  out: global slot of the sentence's first word (tile * GLR_TILE_WORDS + pos)
 *   tile_first[cap]     out: for tile t, index of its first sentence in `order`; [n_tiles] = end.
 *                       Must hold (upper bound on tiles) + 1 ints; glr_plan_tiles_bound gives it.
 *   order[cap]          out: sentence ids in tile order (a multi-tile sentence appears once per tile)
 *   tile_nsub[cap]      out: 0 = ordinary tile, k > 1 = first tile of a k-tile sentence, -1 = its
 *                       continuation tiles
 * returns the number of tiles (>0) or a negative error.
 */
int glr_plan_tiles_bound(const int32_t* cap_lens, int n_sent);
int glr_plan_tiles(const int32_t* cap_lens, int n_sent, int32_t* sent_slot0, int32_t* tile_first,
                   int32_t* order, int32_t* tile_nsub);

/* ------------------------------------------------------------------------------------------
 * Operand packing (device).  HBM-bound layout/convert kernels.
 *
 * glr_pack_regions: img_features [B, D, S] (in_dtype) -> two operand copies in op_dtype
 *     vt [B, S_pad, D]  (region-major, feature contiguous: B^T operand of the score GEMM)
 *     vd [B, D, S_pad]  (feature-major, region contiguous: A operand of the context GEMM)
 *   With no_attn_vec != NULL ([D], in_dtype) a learned column is PREPENDED (region 0), i.e.
 *   S_eff = S + 1 (gloria_loss.py:31-34).  Regions [S_eff, S_pad) are zero.
 *
 * glr_pack_words: words_emb [B_txt, D, L] (in_dtype) -> tp [n_slots, D] (op_dtype), word-major,
 *   slot sent_slot0[i] + w  <-  words_emb[i, :, word_start + w] for w < cap_lens[i]
 *   (gloria_loss.py:122; word_start = 1 is the inference slice of gloria_model.py:179).
 *   Unused slots are zero.  tnorm[n_slots] (fp32) = L2 norm of each packed word as rounded to
 *   op_dtype (gloria_loss.py:14).
 */
int glr_pack_regions(const void* img_features, int in_dtype, const void* no_attn_vec, void* vt, void* vd,
                     int B, int D, int S, int op_dtype, void* stream);

int glr_pack_words(const void* words_emb, int in_dtype, const int32_t* sent_slot0_dev,
                   const int32_t* cap_lens_dev, void* tp, float* tnorm, int B_txt, int D, int L,
                   int word_start, int n_slots, int op_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K1  glr_local_attn_fwd - fused region x word attention + cosine + log-sum-exp.
 * Replaces, for ALL image x sentence pairs at once, the body of the reference sentence loop:
 * attention_fn (gloria_loss.py:19-63), cosine_similarity (:11-16) and the exp/sum/log of
 * local_loss (:150-158, :164).
 *
 *   sim[b, i] = temp3 * log( agg_w exp(temp2 * cos(T[i,:,w], c[b,i,w,:])) )
 *
 * One workgroup per (image b, word tile); see DESIGN.md for the kernel anatomy.
 *
 *   vt, vd       packed regions of the B_img LOCAL images (glr_pack_regions)
 *   tp, tnorm    packed words of ALL sentences (glr_pack_words)
 *   sent_slot0, cap_lens   [n_sent] device int32 (same arrays as given to glr_pack_words)
 *   tile_first, order, tile_nsub   device int32 copies of the glr_plan_tiles outputs
 *   sim          out fp32 [B_img, ld_sim]; column = sentence id
 *   attn         optional out fp32: attention maps of the DIAGONAL pairs only
 *                (image b with sentence img_offset + b; gloria_loss.py:141-143), packed:
 *                sentence i at attn_off[i] floats, [cap_lens[i], S_eff - strip] row-major where
 *                strip = 1 drops the no-attention column (:60-61).  NULL = not wanted.
 *   wctx         optional out fp32 [B_img, D, ld_wctx]: weighted context of the diagonal pairs
 *                (attention_fn's first return value, :59).  NULL = not wanted.
 *   pair_only    1: launch only the B_img diagonal (image b, tile of sentence img_offset+b) pairs;
 *                sim then only receives the diagonal entries.  Used by attention_fn and by the
 *                attention-finetune configuration, which needs only the maps.
 *   img_offset   global index of local image 0 (data-parallel shard offset).
 */
int glr_local_attn_fwd(const void* vt, const void* vd, const void* tp, const float* tnorm,
                       const int32_t* sent_slot0, const int32_t* cap_lens, const int32_t* tile_first,
                       const int32_t* order, const int32_t* tile_nsub, int n_tiles, int n_sent, int B_img,
                       int D, int S_eff,
                       float temp1, float temp2, float temp3, int agg, float eps, float* sim, int ld_sim,
                       float* attn, const int64_t* attn_off, int strip, float* wctx, int ld_wctx,
                       int pair_only, int img_offset, int op_dtype, void* stream);

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
 * Replaces the norm / bmm / clamp / scale of global_loss (gloria_loss.py:75-80).
 *   img [B_img, D], txt [B_txt, D] fp32;  sim [B_img, ld_sim] fp32; ni[B_img], nt[B_txt] norms (out).
 * bwd: given dsim [B_img, ld_sim] -> dimg [B_img, D] and dtxt [B_txt, D] (dtxt holds only the
 *      contribution of these B_img images: sum across ranks in the data-parallel case).
 */
int glr_global_sim_fwd(const float* img, const float* txt, int B_img, int B_txt, int D, float temp3,
                       float eps, float* sim, int ld_sim, float* ni, float* nt, void* stream);
int glr_global_sim_bwd(const float* img, const float* txt, const float* ni, const float* nt,
                       const float* dsim, int ld_sim, int B_img, int B_txt, int D, float temp3, float eps,
                       float* dimg, float* dtxt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLR_H_ */
