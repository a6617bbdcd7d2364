// Gram matrices of the packed regions, written straight into the K-tiled operand layout of the K1 streams.
//
// K1 contracts its second product against G[b] = V_b V_b^T (S x S, glr_local_attn.hip: |c_w|^2 = sum a2 (a2 G)), the
// transposed-copy-per-sentence of the reference's attention_fn (/root/reference/gloria/loss/gloria_loss.py:30-35, :59)
// never being formed.  Rounds 1-2 took G from a library batched GEMM (row-major) and re-tiled it in a second pass
// (glr_tile_gram: 75 MB read + 75 MB written at 256 images); this kernel reads the K-tiled vt once and emits the K-tiled
// Gram operand, the ones row of the forward kernels included (include/glr.h, glr_local_attn_fwd: row S_pad - 1 holds
// ones in columns r < S_eff, so the second contraction delivers Z_w).
//
// One 4-wave workgroup = one image x 64 Gram rows r; wave w computes the 64 x 96 tile against columns r' of region
// blocks {w, w + 4, w + 8} (the K1 wave tile: 6 accumulators of 32 x 32, MFMA 32x32x16 bf16, K = D).  Both operands are
// rows of vt[b]: fragment-major tiling (glr_k1.h) makes every fragment load 1 KiB of contiguous memory, so they go
// straight to registers - no LDS, no barrier.  G is symmetric: the tile is stored as gram_t[r'][k = r], i.e. lane =
// row r' of the B operand K1 will load, registers = 16 of the 32 k values of a chunk; the two lane halves exchange half
// pieces (v_permlane32_swap) so that every lane stores whole 16-byte pieces, 512 contiguous bytes per lane half.
#include <type_traits>

#include "glr_k1.h"

namespace {

constexpr int GN = 256;
constexpr int CBG = 64;

typedef OpBF16 O;
typedef O::frag frag;
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4g g_u32x4g;
__device__ __forceinline__ frag ldgf(const unsigned char* p) { return __builtin_bit_cast(frag, *(g_u32x4g*)(uintptr_t)p); }

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// grid (S_pad / 64, B), 256 threads
__global__ void __launch_bounds__(GN, 2) k_gram_tiled(const unsigned char* __restrict__ vt_t, unsigned char* __restrict__ gram_t,
                                                      int D, int S_pad, int S_eff) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int r0 = blockIdx.x * 64;
  const size_t b = blockIdx.y;
  const int nrb = S_pad >> 5;                      // 32-row blocks per image
  const int nch = D * 2 / CBG;                     // K chunks of vt
  const size_t cstep = (size_t)S_pad * CBG;        // bytes from one K chunk of an image to the next
  const unsigned char* vb = vt_t + b * (size_t)S_pad * D * 2 + h * 512 + l31 * 16;     // + c * cstep + block * 2048 + kk * 1024
  const int ablk = r0 >> 5;                        // A blocks ablk, ablk + 1
  int bblk[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) bblk[j] = min(wg + 4 * j, nrb - 1);                       // (clamped: S_pad < 384 leaves blocks unowned)

  f32x16 acc0[3], acc1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[j][q] = 0.f; acc1[j][q] = 0.f; }

  frag fa[2][2][2], fb[2][3][2];                   // [set][block][k-step]
  auto load = [&](int c, auto setc) {
    constexpr int SET = decltype(setc)::value;
    const unsigned char* pc = vb + (size_t)c * cstep;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int wb = 0; wb < 2; ++wb) fa[SET][wb][kk] = ldgf(pc + (size_t)(ablk + wb) * 2048 + kk * 1024);
#pragma unroll
      for (int j = 0; j < 3; ++j) fb[SET][j][kk] = ldgf(pc + (size_t)bblk[j] * 2048 + kk * 1024);
    }
  };
  auto mma = [&](auto setc) {
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        O::mma(fa[SET][0][kk], fb[SET][j][kk], acc0[j]);
        O::mma(fa[SET][1][kk], fb[SET][j][kk], acc1[j]);
      }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  load(0, S0{});
  load(1, S1{});
  int c = 0;
  // (`keep`: an always-true test hipcc cannot fold - the branch keeps each load block behind the MFMAs that free its
  // registers; floating freely the loads cost 256 registers + 236 bytes of scratch, cf. glr_local_attn_t1.hip)
  const bool keep = S_pad != 0x40000001;
  for (; c + 2 < nch; c += 2) {                    // nch is even (D % 64 == 0)
    mma(S0{});
    if (keep) load(c + 2, S0{});
    mma(S1{});
    if (keep) load(c + 3, S1{});
  }
  mma(S0{});
  mma(S1{});

  // ---- epilogue: gram_t[b][chunk = r / 32][block = r' / 32][slot = (r % 32) / 8][r' % 32][8 x bf16 of r]
  const bool ones = S_eff < S_pad;                 // the forward kernels' ones row: B-operand row r' = S_pad - 1
  unsigned char* gb = gram_t + b * (size_t)S_pad * S_pad * 2;
#pragma unroll
  for (int wb = 0; wb < 2; ++wb) {
    const f32x16(&acc)[3] = wb == 0 ? acc0 : acc1;
    const int rbase = r0 + wb * 32;                // Gram row (= k index of the operand) of accumulator register q:
                                                   // rbase + (q & 3) + 8 * (q >> 2) + 4 * h
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (wg + 4 * j >= nrb) continue;
      const int rp = (wg + 4 * j) * 32 + l31;      // operand row r'
      const bool one_row = ones && rp == S_pad - 1;
#pragma unroll
      for (int i = 0; i < 4; i += 2) {             // slots i (stored by lane half 0) and i + 1 (lane half 1)
        float v[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = 4 * (i + s) + e;
            float x = acc[j][q];
            if (one_row) x = (rbase + 8 * (i + s) + 4 * h + e < S_eff) ? 1.f : 0.f;
            v[s][e] = x;
          }
        // X = this lane's half piece of slot i, Y = of slot i + 1 (two dwords each)
        unsigned x0 = pack2(v[0][0], v[0][1]), x1 = pack2(v[0][2], v[0][3]);
        unsigned y0 = pack2(v[1][0], v[1][1]), y1 = pack2(v[1][2], v[1][3]);
        // lanes 32..63 of X swap with lanes 0..31 of Y: afterwards the low half holds [own P0(i) | partner's P1(i)] and
        // the high half [partner's P0(i + 1) | own P1(i + 1)] - whole 16-byte pieces
        auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
        const u32x4g piece = {s0[0], s1[0], s0[1], s1[1]};
        const int slot = i + h;
        unsigned char* dst = gb + (size_t)(ablk + wb) * cstep + (size_t)(wg + 4 * j) * 2048 + slot * 512 + l31 * 16;
        *reinterpret_cast<u32x4g*>(dst) = piece;
      }
    }
  }
}

}  // namespace

extern "C" int glr_gram_tiled(const void* vt_t, void* gram_t, int B, int D, int S_pad, int S_eff, void* stream) {
  if (!vt_t || !gram_t || B <= 0 || D <= 0 || D % 64 != 0 || S_pad <= 0 || S_pad % 64 != 0 || S_pad > GLR_MAX_SPAD) return GLR_EINVAL;
  if (S_eff <= 0 || S_eff > S_pad) return GLR_EINVAL;
  hipLaunchKernelGGL(k_gram_tiled, dim3(S_pad / 64, B), dim3(GN), 0, (hipStream_t)stream, (const unsigned char*)vt_t,
                     (unsigned char*)gram_t, D, S_pad, S_eff);
  GLR_CHECK_LAUNCH();
  return GLR_OK;
}
