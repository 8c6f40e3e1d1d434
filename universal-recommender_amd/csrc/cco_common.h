// Hand-written gfx950 (MI355X / CDNA4) kernels of the Correlated Cross-Occurrence model build.
//
// Replaces, stage for stage, what Mahout 0.13.0 SimilarityAnalysis does on Spark when called from
// URAlgorithm.calcAll (reference src/main/scala/URAlgorithm.scala:323-329, :343-346):
//   column_counts_kernel            numNonZeroElementsPerColumn
//   downsample_flags_kernel  +      sampleDownAndBinarize  (the "CSR row scan": flat, 16 B/lane coalesced reads,
//   downsample_compact_kernel         wave-assembled keep bitmask, prefix-sum compaction)
//   transpose_kernel                the `A.t` of `A.t %*% B`
//   row_work / binning kernels      row-tile partitioning of the SpGEMM by upper-bound work
//   cco_rows_kernel<T,E>            `A.t %*% B` (Gustavson over rows of A', LDS hash accumulators) fused with
//                                   computeSimilarities (fp64 LLR + top-k) -- counts never touch HBM
//   cco_rows_global_kernel          same, dense global accumulator for rows too heavy for LDS
// All of it is irregular integer/byte work bounded by HBM / L2 / LDS-atomic throughput: no MFMA.
// Wave = 64 lanes everywhere.  Wave-level primitives (__shfl*, __ballot) are only ever executed under
// wave-uniform control flow.
// This header: what more than one stage file uses -- wave shuffles, the tiled exclusive scan (templates + launch_scan), a bound search.
// Stage files:  cco_counts.hip (column counts, public scans, PopModel histograms)   cco_rowscan.hip (sampleDownAndBinarize)   cco_transpose.hip (A.t)
//               cco_expand.hip (entropies, counts aboard, expand preparation)     cco_rows.hip (binning + A.t %*% B + LLR + top-k)   cco_misc.hip (indicator
//               compaction, item ranges, exchange helpers, boundary checks, test hooks)
#pragma once
#include "cco_kernels.h"

namespace urcco {


constexpr int WAVE = 64;

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ long long shfl_i64(long long v, int src) {
  unsigned lo = (unsigned)v, hi = (unsigned)((unsigned long long)v >> 32);
  lo = __shfl(lo, src);
  hi = __shfl(hi, src);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long shfl_up_i64(long long v, unsigned d) {
  unsigned lo = (unsigned)v, hi = (unsigned)((unsigned long long)v >> 32);
  lo = __shfl_up(lo, d);
  hi = __shfl_up(hi, d);
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// first idx in [lo, hi] with rp[idx] > e   (rp[hi] > e guaranteed by the caller)
__device__ __forceinline__ int64_t upper_bound_i64(const int64_t* __restrict__ rp, int64_t lo, int64_t hi, int64_t e) {
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (rp[mid] > e) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// ============================================================================================
// Exclusive scan (three-kernel tile scan): out[i] = sum_{t<i} f(in[t]), out[n] = total
// ============================================================================================
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = SCAN_TILE / SCAN_THREADS;  // 8

// load8: 8 consecutive elements starting at a multiple of 8 -- 16-byte vector loads when the array is 16-byte aligned,
// so that a wave's reads cover one contiguous span (a scalar loop would touch every cache line 8 times).
struct LoadI32 {
  const int32_t* p;
  __device__ __forceinline__ long long operator()(int64_t i) const { return p[i]; }
  __device__ __forceinline__ void load8(int64_t i, long long* x) const {
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      const int4 a = *reinterpret_cast<const int4*>(p + i), b = *reinterpret_cast<const int4*>(p + i + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = p[i + q];
    }
  }
};
struct LoadI64 {
  const int64_t* p;
  __device__ __forceinline__ long long operator()(int64_t i) const { return p[i]; }
  __device__ __forceinline__ void load8(int64_t i, long long* x) const {
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int4 a = *reinterpret_cast<const int4*>(p + i + 2 * q);
        x[2 * q] = (long long)(((unsigned long long)(unsigned)a.y << 32) | (unsigned)a.x);
        x[2 * q + 1] = (long long)(((unsigned long long)(unsigned)a.w << 32) | (unsigned)a.z);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = p[i + q];
    }
  }
};
// inclusive scan of one value per thread over a block of NT threads; returns the exclusive prefix, *total = block sum.
// All NT threads must call it.
template <int NT = SCAN_THREADS>
__device__ __forceinline__ long long block_exclusive_scan(long long v, long long* s_wave /*[NT / WAVE]*/, long long* total) {
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  long long inc = v;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const long long o = shfl_up_i64(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == WAVE - 1) s_wave[wave] = inc;
  __syncthreads();
  long long base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / WAVE; ++w) {
    const long long sw = s_wave[w];
    if (w < wave) base += sw;
    tot += sw;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// n_live (nullable, device): elements at index >= *n_live are known to be zero -- their tiles are skipped (the caller
// sized the launch for an upper bound of a device-side length)
template <typename Load>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(Load ld, int64_t n, int64_t* __restrict__ tile_sums,
                                                                   const int64_t* __restrict__ n_live) {
  __shared__ long long s_wave[SCAN_THREADS / WAVE];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  if (n_live && base >= *n_live) {  // block-uniform
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = 0;
    return;
  }
  long long v = 0;
#pragma unroll
  for (int q = 0; q < SCAN_ITEMS; ++q) {
    const int64_t i = base + (int64_t)q * SCAN_THREADS + threadIdx.x;
    if (i < n) v += ld(i);
  }
  long long tot;
  block_exclusive_scan(v, s_wave, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of tile_sums[0..n_tiles), tile_sums[n_tiles] = total.  1024 threads x 8 consecutive values per
// round (a 97M-element scan has 47K tile sums: with 256 values per round this one block ran 185 rounds of two barriers each,
// 35 us -- 0.7 ms per build of config 4 over its twenty scans); tiles at or beyond *n_live hold zeros and are not visited.
constexpr int ST_THREADS = 1024;
constexpr int ST_ITEMS = 8;
static __global__ __launch_bounds__(ST_THREADS) void scan_tiles_kernel(int64_t* __restrict__ tile_sums, int64_t n_tiles, const int64_t* __restrict__ n_live) {
  __shared__ long long s_wave[ST_THREADS / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  int64_t live_tiles = n_tiles;
  if (n_live) {
    const int64_t lt = *n_live / SCAN_TILE + 1;  // tiles that can hold a non-zero sum
    if (lt < live_tiles) live_tiles = lt;
  }
  long long carry = 0;
  for (int64_t base = 0; base < live_tiles; base += ST_THREADS * ST_ITEMS) {  // block-uniform trip count
    const int64_t first = base + (int64_t)threadIdx.x * ST_ITEMS;
    long long x[ST_ITEMS];
    long long sum = 0;
#pragma unroll
    for (int q = 0; q < ST_ITEMS; ++q) {
      x[q] = first + q < live_tiles ? tile_sums[first + q] : 0;
      sum += x[q];
    }
    long long inc = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const long long o = shfl_up_i64(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == WAVE - 1) s_wave[wave] = inc;
    __syncthreads();
    long long before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ST_THREADS / WAVE; ++w) {
      const long long sw = s_wave[w];
      if (w < wave) before += sw;
      tot += sw;
    }
    __syncthreads();
    long long run = carry + before + inc - sum;
#pragma unroll
    for (int q = 0; q < ST_ITEMS; ++q) {
      if (first + q < live_tiles) tile_sums[first + q] = run;
      run += x[q];
    }
    carry += tot;
  }
  // the prefix of a tile beyond the live ones is the total (the downsweep never reads them, but keep the table well-defined)
  for (int64_t t = live_tiles + threadIdx.x; t < n_tiles; t += ST_THREADS) tile_sums[t] = carry;
  if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}

template <typename Load>
__global__ __launch_bounds__(SCAN_THREADS) void scan_downsweep_kernel(Load ld, int64_t n, const int64_t* __restrict__ tile_sums,
                                                                      int64_t n_tiles, int64_t* __restrict__ out,
                                                                      const int64_t* __restrict__ n_live) {
  __shared__ long long s_wave[SCAN_THREADS / WAVE];
  if (n_live && (int64_t)blockIdx.x * SCAN_TILE > *n_live) return;  // block-uniform; out[] beyond *n_live is never read
  // thread t owns SCAN_ITEMS consecutive elements so that the scan order is the element order
  const int64_t first = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  static_assert(SCAN_ITEMS == 8, "load8");
  long long x[SCAN_ITEMS];
  long long v = 0;
  const bool interior = first + SCAN_ITEMS <= n;
  if (interior) {
    ld.load8(first, x);
  } else {
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) x[q] = first + q < n ? ld(first + q) : 0;
  }
#pragma unroll
  for (int q = 0; q < SCAN_ITEMS; ++q) v += x[q];
  long long tot;
  long long run = block_exclusive_scan(v, s_wave, &tot) + tile_sums[blockIdx.x];
  if (interior && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {  // four 16-byte stores per thread
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q += 2) {
      const long long e0 = run, e1 = run + x[q];
      run = e1 + x[q + 1];
      int4 w;
      w.x = (int)(unsigned)e0; w.y = (int)(unsigned)((unsigned long long)e0 >> 32);
      w.z = (int)(unsigned)e1; w.w = (int)(unsigned)((unsigned long long)e1 >> 32);
      *reinterpret_cast<int4*>(out + first + q) = w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) {
      const int64_t i = first + q;
      if (i < n) out[i] = run;
      run += x[q];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_sums[n_tiles];
}

// One 1024-thread block scans the whole input, 8 consecutive values per thread and pass: for inputs of a few tens of
// thousands of values one launch instead of the three of the tiled scan (each of which is a ~5 us kernel plus a boundary).
constexpr int SB_THREADS = 1024;
constexpr int SB_ITEMS = 8;
constexpr int64_t SB_MAX = 32768;
template <typename Load>
__global__ __launch_bounds__(SB_THREADS) void scan_block_kernel(Load ld, int64_t n, int64_t* __restrict__ out) {
  __shared__ long long s_wave[SB_THREADS / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  long long carry = 0;
  for (int64_t base = 0; base < n; base += SB_THREADS * SB_ITEMS) {  // block-uniform trip count
    const int64_t first = base + (int64_t)threadIdx.x * SB_ITEMS;
    long long x[SB_ITEMS];
    long long sum = 0;
    if (first + SB_ITEMS <= n) {
      ld.load8(first, x);
    } else {
#pragma unroll
      for (int q = 0; q < SB_ITEMS; ++q) x[q] = first + q < n ? ld(first + q) : 0;
    }
#pragma unroll
    for (int q = 0; q < SB_ITEMS; ++q) sum += x[q];
    long long inc = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const long long o = shfl_up_i64(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == WAVE - 1) s_wave[wave] = inc;
    __syncthreads();
    long long before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SB_THREADS / WAVE; ++w) {
      const long long sw = s_wave[w];
      if (w < wave) before += sw;
      tot += sw;
    }
    __syncthreads();
    long long run = carry + before + inc - sum;
#pragma unroll
    for (int q = 0; q < SB_ITEMS; ++q) {
      if (first + q < n) out[first + q] = run;
      run += x[q];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) out[n] = carry;
}

template <typename Load>
static hipError_t launch_scan(hipStream_t st, Load ld, int64_t n, int64_t* out, int64_t* tile_sums, const int64_t* n_live = nullptr, bool tile_sums_ready = false) {
  if (n <= 0) return hipMemsetAsync(out, 0, sizeof(int64_t), st);
  if (n <= SB_MAX) {
    hipLaunchKernelGGL((scan_block_kernel<Load>), dim3(1), dim3(SB_THREADS), 0, st, ld, n, out);
    return hipGetLastError();
  }
  const int64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (!tile_sums_ready) hipLaunchKernelGGL((scan_reduce_kernel<Load>), dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, st, ld, n, tile_sums, n_live);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(ST_THREADS), 0, st, tile_sums, n_tiles, n_live);
  hipLaunchKernelGGL((scan_downsweep_kernel<Load>), dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, st, ld, n, tile_sums, n_tiles, out, n_live);
  return hipGetLastError();
}
}  // namespace urcco
