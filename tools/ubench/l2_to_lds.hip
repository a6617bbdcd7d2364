// Micro-benchmark: how fast can ONE workgroup per CU pull an L2-resident matrix into LDS?
//   mode 0: LDS-DMA   (global_load_lds_dwordx4, 1 KiB per wave instruction)
//   mode 1: registers (global_load_dwordx4 -> ds_write_b128)
//   mode 2: registers only (global_load_dwordx4, no LDS write; values xor-folded)
// Every workgroup streams the same per-XCD-shared buffer of `bytes` (vt[b]-like, 590 KB) `reps` times.
// Build: hipcc -O3 --offload-arch=gfx950 -o l2_to_lds l2_to_lds.hip   (the binary is not kept in the repository)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE, int INFLIGHT>
__global__ void __launch_bounds__(512) k(const unsigned char* src, size_t bytes_per_img, int nimg, int reps,
                                         unsigned long long* cycles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x; const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = src + (size_t)(blockIdx.x % nimg) * bytes_per_img;
  const size_t chunk = 512 * 16;                 // bytes per workgroup-wide load instruction
  const int nchunk = (int)(bytes_per_img / chunk);
  unsigned acc = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
      const unsigned dst0 = lds_addr(smem) + wave * 1024;
      for (int c = 0; c < nchunk; c += INFLIGHT) {
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i)
          glds16(base + (size_t)(c + i) * chunk + tid * 16, dst0 + ((c + i) % 16) * 8192);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT / 2) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      for (int c = 0; c < nchunk; c += INFLIGHT) {
        uint4 v[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) v[i] = *reinterpret_cast<const uint4*>(base + (size_t)(c + i) * chunk + tid * 16);
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
          if (MODE == 1) *reinterpret_cast<uint4*>(smem + ((c + i) % 16) * 8192 + tid * 16) = v[i];
          else acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        }
      }
    }
  }
  __syncthreads();
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (MODE == 1) acc = *reinterpret_cast<unsigned*>(smem + tid * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int INFLIGHT>
void run(const char* name, const unsigned char* d, size_t bytes_per_img, int nimg, unsigned long long* dc, unsigned* ds) {
  const int grid = 256, reps = 20;
  hipFuncSetAttribute((const void*)k<MODE, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int it = 0; it < 2; ++it) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, INFLIGHT>), dim3(grid), dim3(512), 131072, 0, d, bytes_per_img, nimg, reps, dc, ds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it == 1) {
      double tot = (double)bytes_per_img * reps * grid;
      printf("%-34s inflight %2d: %.3f ms  %.2f TB/s chip  = %.1f GB/s per CU = %.1f B/clk/CU @2.4GHz\n", name, INFLIGHT, ms,
             tot / ms / 1e9, tot / ms / 1e6 / grid, tot / (ms * 1e-3) / grid / 2.4e9);
    }
  }
}

int main() {
  const size_t bytes_per_img = 384 * 768 * 2;     // vt[b], bf16
  const int nimg = 32;                            // 32 images x 590 KB = 18.9 MB: L2 (4 MiB/XCD) holds the 4 images of its XCD
  unsigned char* d; unsigned long long* dc; unsigned* ds;
  hipMalloc(&d, bytes_per_img * nimg); hipMemset(d, 1, bytes_per_img * nimg);
  hipMalloc(&dc, 256 * 8); hipMalloc(&ds, 4);
  run<0, 4>("LDS-DMA global_load_lds_dwordx4", d, bytes_per_img, nimg, dc, ds);
  run<0, 8>("LDS-DMA global_load_lds_dwordx4", d, bytes_per_img, nimg, dc, ds);
  run<0, 12>("LDS-DMA global_load_lds_dwordx4", d, bytes_per_img, nimg, dc, ds);
  run<1, 4>("global_load_dwordx4 + ds_write_b128", d, bytes_per_img, nimg, dc, ds);
  run<1, 8>("global_load_dwordx4 + ds_write_b128", d, bytes_per_img, nimg, dc, ds);
  run<2, 4>("global_load_dwordx4 only", d, bytes_per_img, nimg, dc, ds);
  run<2, 8>("global_load_dwordx4 only", d, bytes_per_img, nimg, dc, ds);
  run<2, 12>("global_load_dwordx4 only", d, bytes_per_img, nimg, dc, ds);
  return 0;
}
