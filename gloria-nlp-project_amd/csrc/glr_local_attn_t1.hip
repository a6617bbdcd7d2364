// K1 forward, production shape (384 padded regions, bf16): ONE 64-slot word tile per 4-wave workgroup, TWO workgroups
// per CU.  Replaces, for every (image b, sentence i) pair, attention_fn + cosine_similarity + the exp / sum / log of
// local_loss (/root/reference/gloria/loss/gloria_loss.py:19-63, :11-16, :150-164); maths as in glr_local_attn.hip.
//
// Why this shape (DESIGN.md section 4, K1 forward).  The 8-wave pair kernel (k_local_attn_pw) owns a whole CU: its vector
// phases (word softmax statistics, e2, cosine sums: ~0.6 of its 2.0 ms) run with the matrix pipe idle, and nothing else
// can be resident beside 8 waves x 256 registers.  Here a workgroup is 4 waves x 256 registers and <= 80 KiB of LDS, so
// a CU holds two of them in DIFFERENT phases: one workgroup's vector phases overlap the other's MFMA streams
// (separate pipes; the two waves of a SIMD belong to different workgroups).  The price is one pass over vt[b] / gram[b]
// per 64 words instead of per 128, i.e. the kernel is bound by the L2 -> CU path (~47 B/clk/CU measured,
// tools/ubench/l2_to_lds.hip) instead of by un-overlapped phases.
//
// Wave w (0..3) holds ALL 64 word slots of the tile for region blocks {w, w + 4, w + 8} (32 regions each): 6
// accumulator tiles = 96 registers.  Consequences:
//   * the B operands (vt rows in P1, Gram rows in P3) of a wave are needed by NO other wave: they go from global memory
//     straight to registers, never through LDS (fragment-major operand tiling, glr_k1.h: one fragment load of a wave is
//     1 KiB of contiguous memory), two K chunks in flight per wave, no barrier, no LDS traffic;
//   * only the A operand of P1 (the word tile, 4 KiB per 32-wide K chunk, shared by the 4 waves) is staged through a
//     4-deep LDS ring by plain loads + ds_write (a linear copy: the HBM image already is the fragment order), one
//     workgroup barrier per TWO chunks; the A operand of P3 is the e2 image the vector phases leave in LDS;
//   * every load is an ordinary load hipcc counts: its vmcnt waits are exact, nothing drains.
// The vector phases are those of the pair kernel (same lane -> (word row, region) map, same scalar-bit run boundaries,
// same one-writer statistics tables, same fixed combination orders: bitwise reproducible, and the a1 hand-over to the
// backward pair kernel keeps its layout).
//
// LDS (79.3 KiB): [0, 49 KiB) e2 image [64 words][384 regions] bf16, pitch 784 B - earlier the P1 ring (16 KiB), then the
// run-sum half tables | [49, 73 KiB) run-max half tables / lse table | small.
#include <type_traits>

#include "glr_k1.h"

// timing-only diagnostic build (make ablate): a phase is skipped when its bit is set in GLR_K1_DBG - results are garbage,
// only the run time matters (tools/ablate_k1_t1.py); never compiled into libglr.so
//   1 score stream (P1)   2 statistics passes   4 P2   8 Gram stream (P3)   16 P4
//   32 P1 / P3 without their B loads (the registers keep the prologue's chunks): what the L2 -> register path costs
//   64 P1 without its A staging
#ifdef GLR_ABLATE
#define GLR_SKIP(bit) (p.dbg & (bit))
#else
// Not `false`: an always-false test the compiler cannot fold.  The scalar branches it leaves around the load blocks of
// the two streams do two things, both measured (gpurun_out r03e / r03f, same source otherwise):
//   * hipcc's waitcnt pass merges the "loads issued" and "loads skipped" paths conservatively, so a chunk waits for
//     (nearly) everything in flight instead of leaving three chunks of loads outstanding - and that is FASTER here:
//     1.70 ms against 1.93 ms with exact counted waits around the same pinned load clusters (sched_barrier) and 1.99 ms
//     with the loads free to float between the MFMAs.  A CU in this kernel has 8 waves x up to 18 KiB of loads in
//     flight against a 32-KiB L1: deeper queues only add misses-under-miss; the second workgroup, not the depth of a
//     wave's own queue, is what covers the latency;
//   * the kernel needs 242 registers instead of 256 + scratch.
// S_pad is 384 whenever this kernel runs.
#define GLR_SKIP(bit) (p.S_pad == (int)(0x40000000u | (bit)))
#endif


namespace {

constexpr int NT1 = 256;
constexpr int SP = GLR_MAX_SPAD;            // 384
constexpr int ESZ = 2;
constexpr int CB = 64;                      // bytes of K per chunk row
constexpr int IMP = SP * ESZ + 16;          // image pitch (bytes): 196 dwords = 4 banks per row -> conflict-free b128 reads
constexpr int IMG = TW * IMP;               // 50176
constexpr int HT = PW_MAXSEG * SP;          // floats from one lane half's table to the other's
constexpr int TAB = 2 * HT * 4;             // 24576 bytes: [2 halves][8 sentences][384 regions] fp32
constexpr int NBA = 4;                      // A ring depth (chunks)
constexpr int OFF_MX = IMG;
constexpr int OFF_SMALL = IMG + TAB;
constexpr int SMALL = 64 + 3 * PW_MAXSEG * 4 + 80 * 4 + 3 * TW * 4 + 8 * TW * 4;     // 3296
constexpr int LDS_T1 = OFF_SMALL + ((SMALL + 255) / 256) * 256;
static_assert(LDS_T1 <= 80 * 1024, "two workgroups per CU");
static_assert(NBA * TW * CB <= IMG && TAB <= IMG, "ring and run-sum tables alias the image");

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

template <int N>
__device__ __forceinline__ float dpp_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}
// lane 15 of every 16-lane row ends up with the row's total
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_shr<1>(v);
  v += dpp_shr<2>(v);
  v += dpp_shr<4>(v);
  v += dpp_shr<8>(v);
  return v;
}

typedef OpBF16 O;
typedef O::frag frag;

// plain global loads the compiler counts (explicit global address space: never a flat_load, whose completion is unordered)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 g_u32x4;
__device__ __forceinline__ u32x4 ldg16(const unsigned char* p) { return *(g_u32x4*)(uintptr_t)p; }
__device__ __forceinline__ frag ldg(const unsigned char* p) { return __builtin_bit_cast(frag, ldg16(p)); }

__global__ void __launch_bounds__(NT1, 2) k_local_attn_t1(LaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wg = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = region group
  const int l31 = lane & 31, h = lane >> 5;

  // Block -> (image, tile).  Blocks with equal blockIdx % 8 share an XCD (speed only).  Per XCD the blocks walk groups
  // of `img_block` images x all tiles, images innermost: the XCD's resident workgroups (64) then share img_block
  // images (vt + gram: 885 KB each) and a few word tiles (98 KB each) in the 4 MiB L2.
  const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
  const int ib = p.img_block;
  const int n_units = 2 * p.n_items;
  const int grp = qq / (ib * n_units), rem = qq - grp * (ib * n_units);
  const int b = (grp * ib + rem % ib) * 8 + xcd;
  if (b >= p.B_img) return;
#ifdef GLR_ABLATE
  if (p.dbg & 128) return;                      // what launching the grid alone costs
#endif
  const int unit = rem / ib;
  const int item = unit >> 1, t = unit & 1;
  const int tile = p.item_tile[item] + t;
  const int D = p.D;

  unsigned char* ring = smem;
  unsigned char* img0 = smem;
  float* sm = reinterpret_cast<float*>(smem);                       // [2][PW_MAXSEG][SP] run sums (dead before the image is written)
  float* mx = reinterpret_cast<float*>(smem + OFF_MX);              // [2][PW_MAXSEG][SP] run maxima, log2 units
  float* lt = mx + HT;                                              // [PW_MAXSEG][SP] lse, log2 units (half-1 maxima are dead after fin1)
  signed char* wsegb = reinterpret_cast<signed char*>(smem + OFF_SMALL);   // [TW] sentence index in the pair, -1 = empty
  int* seg_w0 = reinterpret_cast<int*>(wsegb + TW);                 // slot of the first word INSIDE THIS TILE
  int* seg_n = seg_w0 + PW_MAXSEG;
  int* seg_sent = seg_n + PW_MAXSEG;
  int* misc = seg_sent + PW_MAXSEG;                                 // [1..2] diagonal w0, n; [16..79] descriptor
  float* tnl = reinterpret_cast<float*>(misc + 80);                 // [TW] word norms
  float* zsum = tnl + TW;
  float* dsum = zsum + TW;
  float* red = dsum + TW;                                           // [8][TW]

  const size_t rowbytes1 = (size_t)D * ESZ, rowbytes2 = (size_t)SP * ESZ;
  const unsigned char* vt_b = p.vt + (size_t)b * SP * rowbytes1;
  const unsigned char* gram_b = p.gram + (size_t)b * SP * rowbytes2;
  const unsigned char* tp_t = p.tp + (size_t)tile * TW * rowbytes1;
  const int nch1 = (int)(rowbytes1 / CB);

  // everything about the tile comes from the ONE 256-byte descriptor of its pair; its loads go out first so that the
  // set-up below waits for them only, not for the stream's first chunks issued right behind them
  int* dsc = misc + 16;
  int dsc_v = 0;
  float tn_v = 0.f;
  if (tid < 64) {
    dsc_v = p.pair_desc[(size_t)item * 64 + tid];
    tn_v = p.tnorm[(size_t)tile * TW + tid];
  }
  // ---- the stream's first loads go out before anything else: B chunks 0 / 1 (registers), A chunks 0..3 (staging)
  // lane's byte offset inside a 32-row fragment block: slot (kk * 2 + h), row l31
  const int foff = h * 512 + l31 * 16;
  const unsigned char* bp = vt_b + (size_t)wg * 2048 + foff;        // + c * SP * CB + j * 8192 + kk * 1024
  const size_t bstep = (size_t)SP * CB;
  // Two chunks of B fragments per wave (48 registers).  Deeper queues were measured SLOWER (three sets with exact
  // counted waits: 1.93 ms against 1.70 ms, gpurun_out/r03f): what covers a wave's load latency here is the CU's second
  // workgroup, not the depth of its own queue.
  frag bq[2][3][2];                                                 // [set = chunk parity][region block j][k-step]
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) bq[u][j][kk] = ldg(bp + u * bstep + j * 8192 + kk * 1024);
  const unsigned char* ap = tp_t + tid * 16;                        // + c * TW * CB: linear copy of the 4-KiB chunk
  u32x4 ast0 = ldg16(ap), ast1 = ldg16(ap + TW * CB), ast2 = ldg16(ap + 2 * TW * CB), ast3 = ldg16(ap + 3 * TW * CB);

  if (tid < 64) {
    dsc[tid] = dsc_v;
    wsegb[tid] = -1;
    tnl[tid] = tn_v;
  }
  if (tid < 3) misc[tid] = 0;
  __syncthreads();
  const int NS = dsc[0];                                            // sentences of the PAIR; table rows are pair-level indices
  if (tid < NS) {
    const int sent = dsc[8 + tid], w0 = dsc[16 + tid], n = dsc[24 + tid];
    seg_sent[tid] = sent;
    seg_n[tid] = n;
    seg_w0[tid] = (w0 >> 6) == t ? (w0 & 63) : -1;                  // -1: the sentence lives in the pair's other tile
    if ((w0 >> 6) == t) {
      for (int w = 0; w < n; ++w) wsegb[(w0 & 63) + w] = (signed char)tid;
      if (sent == p.img_offset + b) { misc[1] = w0 & 63; misc[2] = n; }
    }
  }

  using T = std::true_type; using F = std::false_type;
  // ================= P1: acc[w, r] = T . V^T (K = D) =================
  f32x16 acc0[3], acc1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[j][q] = 0.f; acc1[j][q] = 0.f; }
  {
    u32x4* rst = reinterpret_cast<u32x4*>(ring) + tid;              // + buf * 256
    rst[0] = ast0;
    rst[256] = ast1;
    ast0 = ldg16(ap + 4 * TW * CB);                                 // chunks 4 / 5 (nch1 >= 8 is checked by the host)
    ast1 = ldg16(ap + 5 * TW * CB);
    const unsigned char* ard = ring + foff;                         // + buf * 4096 + block * 2048 + kk * 1024
    __syncthreads();
    // Half step = two K chunks: the A fragments of BOTH chunks are read up front (ring buffers B0, B0 + 1: one exposed
    // LDS round trip per half step instead of one per chunk), chunk c runs against register set 0, chunk c + 1 against
    // set 1; right behind the MFMAs that read a set it is refilled (chunks c + 2 / c + 3), the staged chunks c + 2 /
    // c + 3 (registers w0 / w1, loaded three half steps ago) go into ring buffers B0 + 2 / B0 + 3 and chunks c + 6 /
    // c + 7 into those registers.  One barrier per half step publishes the two ring writes and proves every wave has
    // finished with the two buffers the NEXT half step overwrites.  LOADB / WR / LD are compile-time: a load under a
    // run-time condition that differs between the paths of a join makes hipcc's waitcnt pass assume the shorter queue.
    auto half_step = [&](int c, auto B0c, u32x4& w0, u32x4& w1, auto loadb, auto stage_wr, auto stage_ld) {
      constexpr int B0 = decltype(B0c)::value;
      constexpr bool LOADB = decltype(loadb)::value, WR = decltype(stage_wr)::value, LD = decltype(stage_ld)::value;
      frag fa[2][2][2];                                             // [chunk of the half step][word block][k-step]
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int wb = 0; wb < 2; ++wb)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) fa[u][wb][kk] = O::ld(ard + (B0 + u) * 4096 + wb * 2048 + kk * 1024);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            O::mma(fa[u][0][kk], bq[u][j][kk], acc0[j]);
            O::mma(fa[u][1][kk], bq[u][j][kk], acc1[j]);
          }
        if (LOADB && !GLR_SKIP(32)) {
          const unsigned char* bn = bp + (size_t)(c + 2 + u) * bstep;
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bq[u][j][kk] = ldg(bn + j * 8192 + kk * 1024);
        }
        if (WR && !GLR_SKIP(64)) {
          u32x4& w = u == 0 ? w0 : w1;
          rst[((B0 + 2 + u) & 3) * 256] = w;
          if (LD) w = ldg16(ap + (size_t)(c + 6 + u) * TW * CB);
        }
      }
      __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>;
    int c = 0;
    if (!GLR_SKIP(1)) {
    for (; c + 12 <= nch1; c += 4) {                 // nch1 % 4 == 0 (host check): ends at c = nch1 - 8
      half_step(c, I0{}, ast2, ast3, T{}, T{}, T{});
      half_step(c + 2, I2{}, ast0, ast1, T{}, T{}, T{});
    }
    half_step(c, I0{}, ast2, ast3, T{}, T{}, T{});        // chunks nch1 - 8 / - 7: the last staging loads (chunks nch1 - 2 / - 1)
    half_step(c + 2, I2{}, ast0, ast1, T{}, T{}, F{});
    half_step(c + 4, I0{}, ast2, ast3, T{}, T{}, F{});
    half_step(c + 6, I2{}, ast0, ast1, F{}, F{}, F{});    // chunks nch1 - 2 / - 1: nothing left to fetch
    }
  }

  // run boundaries of this tile (scalar: same for every lane of a half)
  const int* fl = dsc + 32 + 8 * t;
  const unsigned ST0 = __builtin_amdgcn_readfirstlane(fl[0]), ST1 = __builtin_amdgcn_readfirstlane(fl[1]),
                 LA0 = __builtin_amdgcn_readfirstlane(fl[2]), LA1 = __builtin_amdgcn_readfirstlane(fl[3]);
  const unsigned STANY = ST0 | ST1, LAANY = LA0 | LA1;
  const unsigned STh = h ? ST1 : ST0, LAh = h ? LA1 : LA0;
  // a wave-uniform bit test the compiler must keep as a SCALAR branch (expected false: run boundaries are rare, the
  // boundary blocks are laid out of line)
#define GLR_SBIT(mask, k) __builtin_expect(([&] { unsigned b_ = ((mask) >> (k)) & 1u; asm volatile("" : "+s"(b_)); return b_ != 0; }()), 0)

  // sentence ids of this lane's 32 rows: row k -> slot (k >> 4) * 32 + 8 * ((k & 15) >> 2) + 4 h + (k & 3)
  int sgp[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) sgp[g] = *reinterpret_cast<const int*>(wsegb + (g >> 2) * 32 + 4 * h + 8 * (g & 3));
#define GLR_SGK(k) ((sgp[(k) >> 2] << (24 - 8 * ((k) & 3))) >> 24)
  const int rbase = wg * 32 + l31;              // this lane's region in block j: rbase + 128 * j
  const float t1l = p.temp1 * LOG2E;
  const int rslot = wg * 2 + ((lane >> 4) & 1);
  unsigned e2k0[3][8], e2k1[3][8];              // bf16-rounded e2 of both word blocks, two per dword (for |c|^2 in P4)

  {
    // this wave's table entries (its columns, the rows of this tile's sentences): every lane half clears its own half
    {
      float* tm = mx + h * HT + rbase;
      float* ts = sm + h * HT + rbase;
      for (int s2 = 0; s2 < NS; ++s2) {
        if (seg_w0[s2] < 0) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) { tm[s2 * SP + 128 * j] = -INFINITY; ts[s2 * SP + 128 * j] = 0.f; }
      }
    }
    if (!GLR_SKIP(2)) {
    // ---- pass 1: run maxima
    {
      float rm[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const f32x16(&acc)[3] = k < 16 ? acc0 : acc1;
        const int q = k & 15;
#pragma unroll
        for (int j = 0; j < 3; ++j) asm("v_max_f32 %0, %1, %2" : "=v"(rm[j]) : "v"(rm[j]), "v"(acc[j][q]));
        if (GLR_SBIT(LAANY, k)) {
          const bool mine = (LAh >> k) & 1;
          if (mine) {
            float* dst = mx + h * HT + GLR_SGK(k) * SP + rbase;
#pragma unroll
            for (int j = 0; j < 3; ++j) dst[128 * j] = rm[j] * LOG2E;
          }
#pragma unroll
          for (int j = 0; j < 3; ++j) rm[j] = mine ? -INFINITY : rm[j];
        }
      }
    }
    // fin1: maxima of both lane halves -> half-0 table, sentences split between the halves
    for (int s2 = h; s2 < NS; s2 += 2) {
      if (seg_w0[s2] < 0) continue;
      float* e = mx + s2 * SP + rbase;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float m = e[128 * j];
        const float mb = e[HT + 128 * j];
        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(mb));
        e[128 * j] = m;
      }
    }
    // ---- pass 2: run sums of exp2(s log2e - max)
    {
      float rs[3] = {0.f, 0.f, 0.f}, mc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const f32x16(&acc)[3] = k < 16 ? acc0 : acc1;
        const int q = k & 15;
        if (GLR_SBIT(STANY, k)) {
          const bool mine = (STh >> k) & 1;
          const float* src = mx + max(GLR_SGK(k), 0) * SP + rbase;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float m = src[128 * j];
            mc[j] = mine ? m : mc[j];
            rs[j] = mine ? 0.f : rs[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) rs[j] += __builtin_amdgcn_exp2f(__builtin_fmaf(acc[j][q], LOG2E, -mc[j]));
        if (GLR_SBIT(LAANY, k)) {
          if ((LAh >> k) & 1) {
            float* dst = sm + h * HT + GLR_SGK(k) * SP + rbase;
#pragma unroll
            for (int j = 0; j < 3; ++j) dst[128 * j] = rs[j];
          }
        }
      }
    }
    // fin2: lse = max + log2(sum of the halves), fixed order; stored for the backward pass
    for (int s2 = h; s2 < NS; s2 += 2) {
      if (seg_w0[s2] < 0) continue;
      const float* em = mx + s2 * SP + rbase;
      const float* es = sm + s2 * SP + rbase;
      float* dst = p.lse != nullptr ? p.lse + ((size_t)b * p.n_sent + seg_sent[s2]) * SP + rbase : nullptr;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float su = es[128 * j] + es[HT + 128 * j];
        const float l2 = em[128 * j] + __builtin_amdgcn_logf(su);     // v_log_f32 = log2
        lt[s2 * SP + rbase + 128 * j] = l2;
        if (dst != nullptr) dst[128 * j] = l2 * LN2;
      }
    }
    }
    // the run-sum tables share their bytes with the image: every wave must be done reading them
    __syncthreads();

    // ---- P2: a1, e2 from the scores in registers; LDS image; per-word dot~
    if (!GLR_SKIP(4)) {
      float lc[3] = {0.f, 0.f, 0.f};
      unsigned* a1out = p.a1buf == nullptr ? nullptr
                        : p.a1buf + (((size_t)b * p.a1_items + p.a1_base + item) * 8 + (t + 2 * wg)) * (2 * 3 * 8 * 64) + lane;
      unsigned char* imgw = img0 + (4 * h) * IMP + rbase * ESZ;     // + (blk * 32 + row(q)) * IMP + 128 * j * ESZ
      float* redt = red + rslot * TW + 4 * h;                      // + blk * 32 + row(q)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        f32x16(&acc)[3] = blk == 0 ? acc0 : acc1;
        unsigned(&e2k)[3][8] = blk == 0 ? e2k0 : e2k1;
        float dq[16];
        float a1e[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int k = blk * 16 + q;
          const int row = blk * 32 + (q & 3) + 8 * (q >> 2);
          if (GLR_SBIT(STANY, k)) {
            const bool mine = (STh >> k) & 1;
            const float* src = lt + max(GLR_SGK(k), 0) * SP + rbase;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const float l2 = src[128 * j];
              lc[j] = mine ? l2 : lc[j];
            }
          }
          float dacc = 0.f;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float a1 = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[j][q], LOG2E, -lc[j]));
            if (a1out != nullptr) {
              // hand a1 to the backward (fp16 pairs of rows q, q + 1; one coalesced 256-byte store per wave instruction;
              // clamped: an empty word slot sees a stale lse and may give inf, which the backward must never meet)
              float a1c;
              asm("v_min_f32 %0, 1.0, %1" : "=v"(a1c) : "v"(a1));
              if (q & 1) {
                unsigned pk;
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a1e[j]), "v"(a1c));
                a1out[((blk * 3 + j) * 8 + (q >> 1)) * 64] = pk;
              } else {
                a1e[j] = a1c;
              }
            }
            const float e2 = __builtin_amdgcn_exp2f(t1l * a1);
            O::from_f32(imgw + row * IMP + 128 * j * ESZ, e2);
            const float e2r = bf2f(f2bf(e2));
            const unsigned eb = __float_as_uint(e2r);                      // low 16 bits are zero
            e2k[j][q >> 1] = (q & 1) ? (e2k[j][q >> 1] | eb) : (eb >> 16);
            dacc = __builtin_fmaf(e2r, acc[j][q], dacc);
          }
          dq[q] = dacc;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float d = row_sum16(dq[q]);
          if ((lane & 15) == 15) redt[blk * 32 + (q & 3) + 8 * (q >> 2)] = d;
        }
      }
    }
  }
#undef GLR_SBIT

  // ================= P3: acc[w, r'] = E . G^T (K = S_pad); the A operand is the image, B rows go straight to registers ====
  {
    const unsigned char* gp = gram_b + (size_t)wg * 2048 + foff;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) bq[u][j][kk] = ldg(gp + u * bstep + j * 8192 + kk * 1024);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc0[j][q] = 0.f; acc1[j][q] = 0.f; }
    __syncthreads();                              // image complete
    if (tid < TW) {
      const float* redw = red + tid;
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) d += redw[k * TW];
      dsum[tid] = d;
    }
    const unsigned char* aimg = img0 + l31 * IMP + h * 16;          // + wb * 32 * IMP + c * 64 + kk * 32
    auto gram_step = [&](int c, auto loadb) {
      constexpr bool LOADB = decltype(loadb)::value;
      frag fa[2][2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int wb = 0; wb < 2; ++wb)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) fa[u][wb][kk] = O::ld(aimg + wb * 32 * IMP + (c + u) * CB + kk * 32);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            O::mma(fa[u][0][kk], bq[u][j][kk], acc0[j]);
            O::mma(fa[u][1][kk], bq[u][j][kk], acc1[j]);
          }
        if (LOADB && !GLR_SKIP(32)) {
          const unsigned char* bn = gp + (size_t)(c + 2 + u) * bstep;
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bq[u][j][kk] = ldg(bn + j * 8192 + kk * 1024);
        }
      }
    };
    constexpr int nch2 = SP * ESZ / CB;
    if (!GLR_SKIP(8)) {
      int c = 0;
      for (; c + 4 <= nch2; c += 2) gram_step(c, T{});
      gram_step(c, F{});                          // the last two chunks: nothing left to fetch
    }
  }

  // ================= P4: Z from the ones row, |c|^2, cosine, per-sentence aggregate, maps =================
  if (wg == 3 && l31 == 31) {                   // output column SP - 1 = sum_{r < S_eff} e2[w, r]
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const f32x16(&acc)[3] = k < 16 ? acc0 : acc1;
      const int q = k & 15;
      zsum[(k >> 4) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] = acc[2][q];
    }
  }
  __syncthreads();                              // dsum's readers of `red` are done before it is rewritten
  if (!GLR_SKIP(16)) {
    const float ok2 = (rbase + 256 < p.S_eff) ? 1.f : 0.f;      // padded columns (incl. the Z column) live in block 2 only
    float* redt = red + rslot * TW + 4 * h;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const f32x16(&acc)[3] = k < 16 ? acc0 : acc1;
      const unsigned(&e2k)[3][8] = k < 16 ? e2k0 : e2k1;
      const int q = k & 15;
      const int row = (k >> 4) * 32 + (q & 3) + 8 * (q >> 2);
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned u = e2k[j][q >> 1];
        const float e = __uint_as_float((q & 1) ? (u & 0xffff0000u) : (u << 16));
        v += (j == 2 ? ok2 * e : e) * acc[j][q];
      }
      v = row_sum16(v);
      if ((lane & 15) == 15) redt[row] = v;
    }
  }
  __syncthreads();
  if (tid < TW) {                // wave 0, lane = word slot
    const float* redw = red + lane;
    float nn = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) nn += redw[k * TW];
    const float z = zsum[tid], dd = dsum[tid];
    float cosv = 0.f, nc2 = 0.f;
    if (z > 0.f) {
      const float iz = 1.f / z;
      nc2 = fmaxf(nn, 0.f) * iz * iz;
      const float den = fmaxf(tnl[tid] * sqrtf(nc2), p.eps);
      cosv = dd * iz / den;
    }
    if (p.wstat) {
      float* ws = p.wstat + ((size_t)b * p.n_slots + (size_t)tile * TW + tid) * WSTAT;
      ws[0] = z; ws[1] = cosv; ws[2] = nc2; ws[3] = 0.f;
    }
    // per-sentence aggregate: segmented inclusive scan along the lanes (sentences are lane runs), fixed order
    const int sg = wsegb[tid];
    float v = __expf(p.temp2 * cosv);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(v, off, 64);
      const int so = __shfl_up(sg, off, 64);
      if (lane >= off && so == sg) v = (p.agg == GLR_AGG_MAX) ? fmaxf(v, o) : v + o;
    }
    const int snext = __shfl_down(sg, 1, 64);
    if (sg >= 0 && (lane == 63 || snext != sg)) {
      if (p.agg == GLR_AGG_MEAN) v /= (float)seg_n[sg];
      p.sim[(size_t)b * p.ld_sim + seg_sent[sg]] = p.temp3 * __logf(v);
    }
    dsum[tid] = z > 0.f ? 1.f / z : 0.f;        // 1 / Z_w for the optional outputs below (dsum is dead by now)
  }
  const int dw0 = misc[1], dn = misc[2];
  const float* zinv = dsum;
  if (p.amean != nullptr || (p.attn != nullptr && dn > 0)) __syncthreads();      // (workgroup-uniform condition)
  if (p.attn != nullptr && dn > 0) {            // the diagonal sentence lies inside this tile
    const int sout = p.S_eff - p.strip;
    float* out = p.attn + p.attn_off[p.img_offset + b];
    for (int idx = tid; idx < dn * sout; idx += NT1) {
      const int w = dw0 + idx / sout, r = idx % sout + p.strip;
      out[idx] = O::to_f32(img0 + w * IMP + r * ESZ) * zinv[w];
    }
  }
  if (p.amean != nullptr) {
    // word-mean attention row A[r] = mean_w a2[w, r] of every sentence of the tile (aux regularisers,
    // gloria_loss.py:131-139), from the e2 image and the per-word 1 / Z
    for (int r = tid; r < SP; r += NT1)
      for (int s2 = 0; s2 < NS; ++s2) {
        const int w0 = seg_w0[s2], n = seg_n[s2];
        if (w0 < 0) continue;
        float a = 0.f;
        for (int w = w0; w < w0 + n; ++w) a = __builtin_fmaf(O::to_f32(img0 + w * IMP + r * ESZ), zinv[w], a);
        p.amean[((size_t)b * p.n_sent + seg_sent[s2]) * SP + r] = r < p.S_eff ? a / (float)n : 0.f;
      }
  }
#undef GLR_SGK
}

}  // namespace

int glr_k1_launch_tiles(LaParams& p, int op_dtype, void* stream) {
  if (op_dtype != GLR_BF16 || p.S_pad != SP) return GLR_EINVAL;
  if (p.pair_desc == nullptr || p.S_eff >= p.S_pad || p.n_items <= 0) return GLR_EINVAL;
  if (p.D % 128 != 0 || p.D < 256) return GLR_EINVAL;   // the score stream walks four K chunks per iteration (caller: pair kernel otherwise)
  static const int env_ib = [] { const char* e = getenv("GLR_K1_IMG_BLOCK"); return e ? atoi(e) : 0; }();
  p.img_block = env_ib > 0 ? env_ib : 4;
  const int per_xcd = ((p.B_img + 7) / 8 + p.img_block - 1) / p.img_block * p.img_block;
  const int grid = per_xcd * 8 * p.n_items * 2;
#ifdef GLR_ABLATE
  { const char* e = getenv("GLR_K1_DBG"); p.dbg = e ? atoi(e) : 0; }
#endif
  static GlrLdsAttr la;
  if (glr_ensure_lds(la, (const void*)k_local_attn_t1, LDS_T1) != GLR_OK) return GLR_ELAUNCH;
  hipLaunchKernelGGL(k_local_attn_t1, dim3(grid), dim3(NT1), LDS_T1, (hipStream_t)stream, p);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
