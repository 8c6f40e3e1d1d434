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
#include "cco_kernels.h"
#include <atomic>

#include <cstdio>
#include <cstdlib>

#include "cco_device.h"

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

// ============================================================================================
// K1  column counts (numNonZeroElementsPerColumn)
// Zipf-headed data puts millions of increments on a handful of addresses and a device-scope atomic on one
// address retires at ~11 ns, so every block keeps a small open-addressing LDS cache of (column,count):
// hot columns claim a slot early and cost one global atomic per block; cold ones fall through to L2 atomics.
// ============================================================================================
constexpr int CC_THREADS = 256;
constexpr int CC_SLOTS = 4096;

__device__ __forceinline__ void cc_insert(int* s_key, int* s_cnt, int32_t* __restrict__ counts, int col) {
  const int key = col + 1;
  unsigned h = ((unsigned)key * 0x9E3779B1u) >> 20;  // 12 bits
#pragma unroll
  for (int probe = 0; probe < 2; ++probe) {
    int kk = __hip_atomic_load(&s_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kk == 0) kk = atomicCAS(&s_key[h], 0, key), kk = (kk == 0) ? key : kk;
    if (kk == key) {
      atomicAdd(&s_cnt[h], 1);
      return;
    }
    h = (h + 1) & (CC_SLOTS - 1);
  }
  atomicAdd(&counts[col], 1);
}

template <bool VEC>
__global__ __launch_bounds__(CC_THREADS) void column_counts_kernel(const int32_t* __restrict__ ci, int64_t nnz,
                                                                   int32_t* __restrict__ counts) {
  __shared__ int s_key[CC_SLOTS];
  __shared__ int s_cnt[CC_SLOTS];
  for (int s = threadIdx.x; s < CC_SLOTS; s += CC_THREADS) {
    s_key[s] = 0;
    s_cnt[s] = 0;
  }
  __syncthreads();
  const int64_t gtid = (int64_t)blockIdx.x * CC_THREADS + threadIdx.x;
  const int64_t gstride = (int64_t)gridDim.x * CC_THREADS;
  if (VEC) {
    const int64_t nvec = nnz >> 2;
    const int4* ci4 = reinterpret_cast<const int4*>(ci);
    for (int64_t v = gtid; v < nvec; v += gstride) {
      const int4 x = ci4[v];
      cc_insert(s_key, s_cnt, counts, x.x);
      cc_insert(s_key, s_cnt, counts, x.y);
      cc_insert(s_key, s_cnt, counts, x.z);
      cc_insert(s_key, s_cnt, counts, x.w);
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < (nnz & 3)) cc_insert(s_key, s_cnt, counts, ci[(nvec << 2) + threadIdx.x]);
  } else {
    for (int64_t e = gtid; e < nnz; e += gstride) cc_insert(s_key, s_cnt, counts, ci[e]);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < CC_SLOTS; s += CC_THREADS)
    if (s_key[s] != 0) atomicAdd(&counts[s_key[s] - 1], s_cnt[s]);
}

hipError_t launch_column_counts(hipStream_t st, int n_cu, const int32_t* col_idx, int64_t nnz, int32_t n_cols, int32_t* counts) {
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)n_cols, st);
  if (e != hipSuccess || nnz == 0) return e;
  const bool vec = (reinterpret_cast<uintptr_t>(col_idx) & 15) == 0;
  const int64_t work_items = vec ? (nnz + 3) / 4 : nnz;
  int64_t blocks = (work_items + (int64_t)CC_THREADS * 8 - 1) / ((int64_t)CC_THREADS * 8);  // >= 8 vectors per thread
  const int64_t cap = (int64_t)n_cu * 4;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (vec)
    hipLaunchKernelGGL((column_counts_kernel<true>), dim3((unsigned)blocks), dim3(CC_THREADS), 0, st, col_idx, nnz, counts);
  else
    hipLaunchKernelGGL((column_counts_kernel<false>), dim3((unsigned)blocks), dim3(CC_THREADS), 0, st, col_idx, nnz, counts);
  return hipGetLastError();
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
__global__ __launch_bounds__(ST_THREADS) void scan_tiles_kernel(int64_t* __restrict__ tile_sums, int64_t n_tiles, const int64_t* __restrict__ n_live) {
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
hipError_t launch_scan_i32(hipStream_t st, const int32_t* in, int64_t n, int64_t* out, int64_t* tile_sums) {
  return launch_scan(st, LoadI32{in}, n, out, tile_sums);
}
hipError_t launch_scan_i64(hipStream_t st, const int64_t* in, int64_t n, int64_t* out, int64_t* tile_sums) {
  return launch_scan(st, LoadI64{in}, n, out, tile_sums);
}

// ============================================================================================
// K1b  column counts without global atomics (matrices large enough to repay six launches).
// A global atomic per interaction caps the histogram at ~20-40 G updates/s; here the interactions are first
// partitioned by column range (PH_BUCKET = 8192 columns, so one bucket's counters fit 32 KiB of LDS) and then counted
// densely in LDS:
//   count    per part of 16384 interactions: how many fall in each bucket (LDS atomics)
//   scan     bucket-major exclusive prefix -> where every (bucket, part) slice starts
//   scatter  interactions -> 16-bit in-bucket column ids, grouped by bucket
//   blockmap buckets -> histogram blocks of 32768 interactions each
//   hist     one block per slice: dense LDS counters, written out as a partial histogram (plain coalesced stores)
//   reduce   counts[col] = sum of its bucket's partials
// Every pass streams; traffic is ~3.5x the column-index array however many columns there are.
// `nnz_dev` (nullable) overrides nnz with a device-side value <= nnz (no host sync after compaction).
// ============================================================================================
constexpr int PH_BITS = 13;
constexpr int PH_BUCKET = 1 << PH_BITS;
constexpr int PH_PART = 16384;
// Interactions per histogram block.  A block writes one partial histogram of its bucket (PH_BUCKET 16-bit counters: a chunk holds
// fewer than 65536 ids) which the reduce pass reads back, so the partials cost 2 * 16 KB / chunk bytes per interaction: 1.0 B at
// 32768, 0.53 B at 61440.  The larger chunk only where it still leaves a few thousand blocks (measured with 32-bit partials: 131072
// was -6 % on config 4's column counts and +9 % on config 3's, whose largest matrix then had 305 blocks for 512 slots).
// URCCO_PH_CHUNK_BIG_NNZ (environment, read per call): the entry count from which the larger chunk is used (tests lower it).
constexpr int PH_CHUNK_SMALL = 32768, PH_CHUNK_BIG = 61440;
static inline int ph_chunk(int64_t nnz) {
  const char* e = getenv("URCCO_PH_CHUNK_BIG_NNZ");
  const long long big = e && *e ? atoll(e) : 120000000ll;
  return nnz >= big ? PH_CHUNK_BIG : PH_CHUNK_SMALL;
}
static_assert(PH_CHUNK_BIG < 65536 && PH_CHUNK_BIG % 8 == 0 && PH_CHUNK_SMALL % 8 == 0, "16-bit partial counters; 16-byte loads");
constexpr int PH_MAX_BUCKETS = 1024;

// Eight private copies of the bucket counters, chosen by lane: with a few dozen buckets (25 for a 200K-column matrix) the 64
// lanes of a wave would otherwise queue on a handful of LDS addresses.
constexpr int PH_COPIES = 8;
__global__ __launch_bounds__(256) void ph_count_kernel(const int32_t* __restrict__ ci, int64_t nnz_host, const int64_t* __restrict__ nnz_dev,
                                                       int n_buckets, int64_t n_parts, int32_t* __restrict__ part_counts, int vec_ok) {
  __shared__ int s_cnt[PH_COPIES * PH_MAX_BUCKETS];
  const int64_t nnz = nnz_dev ? *nnz_dev : nnz_host;
  for (int b = threadIdx.x; b < PH_COPIES * n_buckets; b += 256) s_cnt[b] = 0;
  __syncthreads();
  int* mine = s_cnt + (threadIdx.x & (PH_COPIES - 1)) * n_buckets;
  const int64_t e0 = (int64_t)blockIdx.x * PH_PART;
  const int64_t e1 = e0 + PH_PART < nnz ? e0 + PH_PART : nnz;
  for (int64_t e = e0 + (int64_t)threadIdx.x * 4; e < e1; e += 256 * 4) {
    if (vec_ok && e + 3 < e1) {
      const int4 x = *reinterpret_cast<const int4*>(ci + e);
      atomicAdd(&mine[x.x >> PH_BITS], 1);
      atomicAdd(&mine[x.y >> PH_BITS], 1);
      atomicAdd(&mine[x.z >> PH_BITS], 1);
      atomicAdd(&mine[x.w >> PH_BITS], 1);
    } else {
      for (int q = 0; q < 4 && e + q < e1; ++q) atomicAdd(&mine[ci[e + q] >> PH_BITS], 1);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < n_buckets; b += 256) {
    int tot = 0;
#pragma unroll
    for (int c = 0; c < PH_COPIES; ++c) tot += s_cnt[c * n_buckets + b];
    part_counts[(int64_t)b * n_parts + blockIdx.x] = tot;
  }
}

// The part's ids are first grouped by bucket in LDS (the (bucket, part) slice lengths are already known from the offsets),
// then every slice leaves as one run of consecutive 2-byte stores -- whole lines instead of 16384 isolated 2-byte writes
// (measured on config 3: -4 % on the column-count stage against scattering straight to global memory).
constexpr int PHS_THREADS = 512;  // 48 KB of LDS per block: three blocks per CU, so 512 threads keep 24 waves per CU in flight
__global__ __launch_bounds__(PHS_THREADS) void ph_scatter_kernel(const int32_t* __restrict__ ci, int64_t nnz_host, const int64_t* __restrict__ nnz_dev,
                                                                int n_buckets, int64_t n_parts, const int64_t* __restrict__ offsets,
                                                                unsigned short* __restrict__ bucketed, int vec_ok) {
  __shared__ long long s_base[PH_MAX_BUCKETS];
  __shared__ int s_loc[PH_MAX_BUCKETS + 1];  // where the bucket's run starts inside the staging array
  __shared__ int s_cur[PH_MAX_BUCKETS];
  __shared__ unsigned short s_stage[PH_PART];
  __shared__ long long s_wave[PHS_THREADS / WAVE];
  const int64_t nnz = nnz_dev ? *nnz_dev : nnz_host;
  int carry = 0;
  for (int base = 0; base < n_buckets; base += PHS_THREADS) {  // block-uniform: exclusive prefix of this part's slice lengths
    const int b = base + threadIdx.x;
    long long len = 0;
    if (b < n_buckets) {
      const int64_t idx = (int64_t)b * n_parts + blockIdx.x;
      const long long o = offsets[idx];
      s_base[b] = o;
      len = offsets[idx + 1] - o;
      s_cur[b] = 0;
    }
    long long tot;
    const long long ex = block_exclusive_scan<PHS_THREADS>(len, s_wave, &tot);
    if (b < n_buckets) s_loc[b] = carry + (int)ex;
    carry += (int)tot;
  }
  if (threadIdx.x == 0) s_loc[n_buckets] = carry;
  __syncthreads();
  const int64_t e0 = (int64_t)blockIdx.x * PH_PART;
  const int64_t e1 = e0 + PH_PART < nnz ? e0 + PH_PART : nnz;
  for (int64_t e = e0 + (int64_t)threadIdx.x * 4; e < e1; e += PHS_THREADS * 4) {
    int cols[4];
    int n = 4;
    if (vec_ok && e + 3 < e1) {
      const int4 x = *reinterpret_cast<const int4*>(ci + e);
      cols[0] = x.x; cols[1] = x.y; cols[2] = x.z; cols[3] = x.w;
    } else {
      n = (int)(e1 - e < 4 ? e1 - e : 4);
      for (int q = 0; q < n; ++q) cols[q] = ci[e + q];
    }
    for (int q = 0; q < n; ++q) {
      const int b = cols[q] >> PH_BITS;
      s_stage[s_loc[b] + atomicAdd(&s_cur[b], 1)] = (unsigned short)(cols[q] & (PH_BUCKET - 1));
    }
  }
  __syncthreads();
  // one wave per bucket run, lanes on consecutive ids
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  for (int b = wave; b < n_buckets; b += PHS_THREADS / WAVE) {
    const int l0 = s_loc[b], len = s_loc[b + 1] - l0;
    const long long dst = s_base[b];
    for (int t = lane; t < len; t += WAVE) bucketed[dst + t] = s_stage[l0 + t];
  }
}

// single block: blk_prefix[b] = first histogram block of bucket b, blk_prefix[n_buckets] = number of blocks
__global__ __launch_bounds__(SCAN_THREADS) void ph_blockmap_kernel(const int64_t* __restrict__ offsets, int n_buckets, int64_t n_parts,
                                                                  int32_t* __restrict__ blk_prefix, int PH_CHUNK) {
  __shared__ long long s_wave[SCAN_THREADS / WAVE];
  long long carry = 0;
  for (int base = 0; base < n_buckets; base += SCAN_THREADS) {  // block-uniform
    const int b = base + threadIdx.x;
    long long v = 0;
    if (b < n_buckets) {
      const long long size = offsets[(int64_t)(b + 1) * n_parts] - offsets[(int64_t)b * n_parts];
      v = (size + PH_CHUNK - 1) / PH_CHUNK;
    }
    long long tot;
    const long long ex = block_exclusive_scan(v, s_wave, &tot);
    if (b < n_buckets) blk_prefix[b] = (int32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0) blk_prefix[n_buckets] = (int32_t)carry;
}

constexpr int PHH_THREADS = 1024;  // a few-million-entry matrix has only ~150 chunks: four times the waves per chunk
__global__ __launch_bounds__(PHH_THREADS) void ph_hist_kernel(const unsigned short* __restrict__ bucketed, const int64_t* __restrict__ offsets,
                                                      int n_buckets, int64_t n_parts, const int32_t* __restrict__ blk_prefix,
                                                      unsigned short* __restrict__ partial, int PH_CHUNK) {
  __shared__ unsigned s_cnt[PH_BUCKET];
  const int blk = blockIdx.x;
  if (blk >= blk_prefix[n_buckets]) return;  // block-uniform
  int lo = 0, hi = n_buckets;  // last b with blk_prefix[b] <= blk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk_prefix[mid] <= blk) lo = mid; else hi = mid;
  }
  const int b = lo;
  for (int c = threadIdx.x; c < PH_BUCKET; c += PHH_THREADS) s_cnt[c] = 0u;
  __syncthreads();
  const int64_t bs = offsets[(int64_t)b * n_parts], be = offsets[(int64_t)(b + 1) * n_parts];
  const int64_t e0 = bs + (int64_t)(blk - blk_prefix[b]) * PH_CHUNK;
  const int64_t e1 = e0 + PH_CHUNK < be ? e0 + PH_CHUNK : be;
  // 16-byte loads of eight 16-bit ids where aligned
  int64_t e = e0 + threadIdx.x;
  const int64_t a0 = (e0 + 7) & ~(int64_t)7;
  for (; e < e1 && e < a0; e += PHH_THREADS) atomicAdd(&s_cnt[bucketed[e]], 1u);  // unaligned head (< 8 entries: first iteration only)
  for (int64_t v = a0 + (int64_t)threadIdx.x * 8; v + 7 < e1; v += PHH_THREADS * 8) {
    const uint4 x = *reinterpret_cast<const uint4*>(bucketed + v);
    atomicAdd(&s_cnt[x.x & 0xffffu], 1u); atomicAdd(&s_cnt[x.x >> 16], 1u);
    atomicAdd(&s_cnt[x.y & 0xffffu], 1u); atomicAdd(&s_cnt[x.y >> 16], 1u);
    atomicAdd(&s_cnt[x.z & 0xffffu], 1u); atomicAdd(&s_cnt[x.z >> 16], 1u);
    atomicAdd(&s_cnt[x.w & 0xffffu], 1u); atomicAdd(&s_cnt[x.w >> 16], 1u);
  }
  {
    const int64_t n_vec = e1 > a0 ? (e1 - a0) / 8 : 0;
    for (int64_t t = a0 + n_vec * 8 + threadIdx.x; t < e1; t += PHH_THREADS) atomicAdd(&s_cnt[bucketed[t]], 1u);  // tail
  }
  __syncthreads();
  // two counters per 4-byte store
  unsigned* out = reinterpret_cast<unsigned*>(partial + (int64_t)blk * PH_BUCKET);
  for (int c = threadIdx.x; c < PH_BUCKET / 2; c += PHH_THREADS) out[c] = s_cnt[2 * c] | (s_cnt[2 * c + 1] << 16);
}

__global__ __launch_bounds__(256) void ph_reduce_kernel(const unsigned short* __restrict__ partial, const int32_t* __restrict__ blk_prefix, int32_t n_cols,
                                                        int32_t* __restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_cols) return;
  const int b = (int)(j >> PH_BITS);
  const int c = (int)(j & (PH_BUCKET - 1));
  unsigned sum = 0;
  for (int blk = blk_prefix[b]; blk < blk_prefix[b + 1]; ++blk) sum += partial[(int64_t)blk * PH_BUCKET + c];
  counts[j] = (int32_t)sum;
}

// --------------------------------------------------------------------------------------------
// K1c  part-local form of the partitioned histogram (round 5; catalogues of up to PL_MAX_BUCKETS * 16384 columns).
// The form above reads the column indices twice (count, then scatter) and scans a (bucket x part) table in between, because a
// bucket's ids are to lie contiguously in memory: ~13.5 bytes moved per 4-byte interaction.  Here a part of PL_PART
// interactions is read ONCE into registers, ranked inside its bucket while it is counted (one returning LDS atomic per id on a
// lane-private copy of the bucket counters), grouped by bucket in LDS and written where it lies -- part p's ids at
// bucketed[p * PL_PART ...), whole lines -- together with the (transposed) table of where each bucket's slice starts inside the
// part.  The histogram block of (bucket, part range) then walks that range's slices: 4 B read + 2 B written + 2 B read per
// interaction, no count pass, no scan, no block map.
//   partition  one block per part          ids -> 14-bit in-bucket ids grouped by bucket, loc_t[b][p] = start of bucket b in part p
//   hist       one block per (bucket, s)   dense 64 KiB LDS counters over the slices of parts [s, s + 1) * pp, 32-bit partials
//   reduce     counts[col] = sum over the bucket's S partials
// --------------------------------------------------------------------------------------------
constexpr int PL_BITS = 14;
constexpr int PL_BUCKET = 1 << PL_BITS;
constexpr int PL_PART = 16384;
constexpr int PL_THREADS = 512;
constexpr int PL_PER_THREAD = PL_PART / PL_THREADS;  // 32 ids in registers
constexpr int PL_MAX_BUCKETS = 256;                  // the packed (bucket, id, rank) word has 8 bits for the bucket
constexpr int PL_COPIES = 16;                        // lane-private counter copies: a copy sees PL_PART / 16 = 1024 ids -> ranks fit 10 bits
static_assert(PL_PER_THREAD % 4 == 0 && PL_PART / PL_COPIES <= 1024 && PL_THREADS % PL_COPIES == 0, "packed word: 8 + 14 + 10 bits");

__global__ __launch_bounds__(PL_THREADS, 6) void pl_partition_kernel(const int32_t* __restrict__ ci, int64_t nnz_host, const int64_t* __restrict__ nnz_dev,
                                                                  int n_buckets, int64_t n_parts, unsigned short* __restrict__ bucketed,
                                                                  unsigned short* __restrict__ loc_t, int vec_ok) {
  __shared__ int s_cnt[PL_COPIES * PL_MAX_BUCKETS];  // counts, then the start of every (copy, bucket) run inside the staging array
  __shared__ uint4 s_stage4[PL_PART / 8];  // (16-byte aligned: the part leaves in 16-byte stores)
  __shared__ long long s_wave[PL_THREADS / WAVE];
  unsigned short* s_stage = reinterpret_cast<unsigned short*>(s_stage4);
  const int64_t nnz = nnz_dev ? *nnz_dev : nnz_host;
  const int64_t e0 = (int64_t)blockIdx.x * PL_PART;
  const int64_t e1 = e0 + PL_PART < nnz ? e0 + PL_PART : nnz;
  if (e0 >= e1) {  // a part beyond the device-side length: every slice is empty (block-uniform)
    for (int b = threadIdx.x; b <= n_buckets; b += PL_THREADS) loc_t[(int64_t)b * n_parts + blockIdx.x] = 0;
    return;
  }
  for (int b = threadIdx.x; b < PL_COPIES * n_buckets; b += PL_THREADS) s_cnt[b] = 0;
  __syncthreads();
  int* mine = s_cnt + (threadIdx.x & (PL_COPIES - 1)) * n_buckets;
  const int n = (int)(e1 - e0);
  auto live = [&](int q) { return (q >> 2) * (PL_THREADS * 4) + (int)threadIdx.x * 4 + (q & 3) < n; };  // register q holds an entry of the part
  unsigned w[PL_PER_THREAD];  // the column, then (bucket << 24) | (id << 10) | rank inside (copy, bucket)
#pragma unroll
  for (int r = 0; r < PL_PER_THREAD / 4; ++r) {  // all loads of the part are issued before the first atomic
    const int64_t e = e0 + (int64_t)r * (PL_THREADS * 4) + (int64_t)threadIdx.x * 4;
    int4 x = make_int4(0, 0, 0, 0);
    if (vec_ok && e + 3 < e1) {
      x = *reinterpret_cast<const int4*>(ci + e);
    } else {
      if (e < e1) x.x = ci[e];
      if (e + 1 < e1) x.y = ci[e + 1];
      if (e + 2 < e1) x.z = ci[e + 2];
      if (e + 3 < e1) x.w = ci[e + 3];
    }
    w[4 * r] = (unsigned)x.x; w[4 * r + 1] = (unsigned)x.y; w[4 * r + 2] = (unsigned)x.z; w[4 * r + 3] = (unsigned)x.w;
  }
#pragma unroll
  for (int q = 0; q < PL_PER_THREAD; ++q) {
    if (live(q)) {
      const unsigned b = w[q] >> PL_BITS;
      const unsigned rank = (unsigned)atomicAdd(&mine[b], 1);
      w[q] = (b << 24) | ((w[q] & (PL_BUCKET - 1)) << 10) | rank;
    }
  }
  __syncthreads();
  {  // exclusive prefix over (bucket, copy), bucket-major: where every run starts; the bucket starts go out as loc_t[b][part]
    const int b = threadIdx.x;  // n_buckets <= PL_MAX_BUCKETS <= PL_THREADS: one round
    long long tot = 0;
    if (b < n_buckets) {
#pragma unroll
      for (int k = 0; k < PL_COPIES; ++k) tot += s_cnt[k * n_buckets + b];
    }
    long long all;
    const long long ex = block_exclusive_scan<PL_THREADS>(tot, s_wave, &all);
    if (b < n_buckets) {
      int run = (int)ex;
      loc_t[(int64_t)b * n_parts + blockIdx.x] = (unsigned short)run;
#pragma unroll
      for (int k = 0; k < PL_COPIES; ++k) {  // (read a second time rather than held across the scan: 16 registers less)
        const int c = s_cnt[k * n_buckets + b];
        s_cnt[k * n_buckets + b] = run;
        run += c;
      }
    }
    if (b == n_buckets) loc_t[(int64_t)b * n_parts + blockIdx.x] = (unsigned short)all;  // <= PL_PART = 16384
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PL_PER_THREAD; ++q) {
    if (live(q)) s_stage[mine[w[q] >> 24] + (int)(w[q] & 1023u)] = (unsigned short)((w[q] >> 10) & (PL_BUCKET - 1));
  }
  __syncthreads();
  // the part leaves as it lies: 16-byte stores (bucketed + e0 is 32 KiB-aligned relative to the array's 256-byte-aligned base)
  uint4* dst = reinterpret_cast<uint4*>(bucketed + e0);
  for (int v = threadIdx.x; v * 8 < n; v += PL_THREADS) dst[v] = s_stage4[v];  // the last vector may carry up to 7 stale ids: inside the part's own 32 KiB, never read
}

constexpr int PLH_THREADS = 1024;
// LPS lanes walk one slice together, 16 ids (two 16-byte loads) per lane and step.  A slice of a 2M-column catalogue holds ~130 ids:
// with 16 lanes per slice a step covers 256 and half of the lanes idle through the masked atomics; fewer lanes per slice waste less
// but touch more parts (pages) per load instruction.  Measured on config 4's five raw matrices (profiles/r05_colcount_variants.log):
// 16 / 8 / 4 lanes 3.07 / 3.02 / 3.19 ms, the bucket-contiguous form 3.45.  dbg (URCCO_PL_DEBUG, profiling only): 1 = no LDS atomics,
// 2 = no loads -- which is how the same log prices the pass: without the atomics 1.76 ms, with neither 1.37: the RANDOM LDS atomics
// (~1.1 lane updates per clock and CU: ~58 cycles per wave instruction against 4.6 for conflict-free addresses) are what the histogram
// costs, not its loads -- the same bound the bucket-contiguous form sits on.
// Buckets are split by WEIGHT: a bucket gets S blocks per average bucket weight it carries (pl_blockmap_kernel), each block an equal share of
// the parts.  A catalogue's hottest item draws 6.6 % of a Zipf(1) matrix into ONE bucket -- 17x the average; with S blocks for every
// bucket the few blocks of that bucket were the kernel (0.98 ms on config 4's largest matrix for 0.45 ms of evenly spread work).
template <int LPS>
__global__ __launch_bounds__(PLH_THREADS) void pl_hist_kernel(const unsigned short* __restrict__ bucketed, const unsigned short* __restrict__ loc_t,
                                                              int n_buckets, int64_t n_parts, const int32_t* __restrict__ blk_prefix, int32_t n_cols,
                                                              unsigned* __restrict__ partial, int dbg) {
  __shared__ unsigned s_cnt[PL_BUCKET];
  const int blk = blockIdx.x;
  if (blk >= blk_prefix[n_buckets]) return;  // block-uniform
  int blo = 0, bhi = n_buckets;  // last b with blk_prefix[b] <= blk
  while (bhi - blo > 1) {
    const int mid = (blo + bhi) >> 1;
    if (blk_prefix[mid] <= blk) blo = mid; else bhi = mid;
  }
  const int b = blo, s = blk - blk_prefix[b], S = blk_prefix[b + 1] - blk_prefix[b];
  const int width = (int)((int64_t)n_cols - ((int64_t)b << PL_BITS) < PL_BUCKET ? (int64_t)n_cols - ((int64_t)b << PL_BITS) : PL_BUCKET);  // columns of this bucket
  for (int c = threadIdx.x; c < width; c += PLH_THREADS) s_cnt[c] = 0u;
  __syncthreads();
  const int64_t pp = (n_parts + S - 1) / S;
  const int64_t p0 = (int64_t)s * pp < n_parts ? (int64_t)s * pp : n_parts, p1 = p0 + pp < n_parts ? p0 + pp : n_parts;
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  const unsigned short* lo_t = loc_t + (int64_t)b * n_parts;
  const unsigned short* hi_t = loc_t + (int64_t)(b + 1) * n_parts;
  // Steps are aligned to 8 ids; ids of a step outside [lo, hi) -- the neighbouring buckets' -- are masked (the array has 64 bytes of slack
  // behind it).  (Measured: one WAVE per slice with 2-byte loads -- a chain of short waits -- took 2x the bucket-contiguous form; one
  // LANE per slice -- 64 parts, i.e. 64 pages, per load instruction -- 2.4x.)
  constexpr int SPR = WAVE / LPS;  // slices per wave and round
  constexpr int GP = 16;           // parts per group (a wave takes groups round robin; lane l < GP holds the bounds of part g + l)
  static_assert(GP % SPR == 0 && SPR <= GP, "rounds per group");
  const int sub = lane / LPS, sl = lane % LPS;
  const int64_t gstep = (int64_t)(PLH_THREADS / WAVE) * GP;
  unsigned fake = 0u;
  int64_t g = p0 + (int64_t)wave * GP;
  unsigned lo_n = 0u, hi_n = 0u;  // the NEXT group's bounds travel while this group's slices are counted
  if (g < p1 && lane < GP && g + lane < p1) {
    lo_n = lo_t[g + lane];
    hi_n = hi_t[g + lane];
  }
  for (; g < p1; g += gstep) {  // wave-uniform
    const unsigned lo = lo_n, hi = hi_n;
    lo_n = 0u;
    hi_n = 0u;
    if (g + gstep < p1 && lane < GP && g + gstep + lane < p1) {
      lo_n = lo_t[g + gstep + lane];
      hi_n = hi_t[g + gstep + lane];
    }
    const int rounds = (int)(p1 - g < GP ? (p1 - g + SPR - 1) / SPR : GP / SPR);
    for (int r = 0; r < rounds; ++r) {  // wave-uniform
      const int pj = SPR * r + sub;  // (parts past the range carry lo == hi == 0)
      const unsigned lo_j = (unsigned)__shfl((int)lo, pj);
      const unsigned hi_j = (unsigned)__shfl((int)hi, pj);
      const unsigned short* src = bucketed + (g + pj) * PL_PART;
      for (unsigned base = (lo_j & ~7u) + 16u * (unsigned)sl; base < hi_j; base += 16u * LPS) {
        uint4 x0 = make_uint4(base, base + 2u, base + 4u, base + 6u), x1 = x0;
        if (!(dbg & 2)) {
          x0 = *reinterpret_cast<const uint4*>(src + base);
          x1 = *reinterpret_cast<const uint4*>(src + base + 8);
        }
        const unsigned wds[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        if (dbg & 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) fake ^= wds[k];
          continue;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned t = base + 2u * k;
          if (t >= lo_j && t < hi_j) atomicAdd(&s_cnt[wds[k] & (PL_BUCKET - 1)], 1u);
          if (t + 1u >= lo_j && t + 1u < hi_j) atomicAdd(&s_cnt[(wds[k] >> 16) & (PL_BUCKET - 1)], 1u);
        }
      }
    }
  }
  if ((dbg & 1) && fake == 0x9e3779b9u) s_cnt[0] = 1u;
  __syncthreads();
  unsigned* out = partial + (int64_t)blk * PL_BUCKET;
  for (int c = threadIdx.x; c < width; c += PLH_THREADS) out[c] = s_cnt[c];
}

__global__ __launch_bounds__(256) void pl_reduce_kernel(const unsigned* __restrict__ partial, const int32_t* __restrict__ blk_prefix, int32_t n_cols,
                                                        int32_t* __restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_cols) return;
  const int b = (int)(j >> PL_BITS);
  const int c = (int)(j & (PL_BUCKET - 1));
  unsigned sum = 0;
  for (int blk = blk_prefix[b]; blk < blk_prefix[b + 1]; ++blk) sum += partial[(int64_t)blk * PL_BUCKET + c];
  counts[j] = (int32_t)sum;
}

// weight[b] = ids of bucket b over all parts: gridDim.y blocks per bucket sum their share of its slice lengths (one block per bucket was a
// 16 us latency chain ten times per build); weight[] is zero on entry ...
__global__ __launch_bounds__(256) void pl_weights_kernel(const unsigned short* __restrict__ loc_t, int64_t n_parts, unsigned long long* __restrict__ weight) {
  __shared__ long long s_wave[256 / WAVE];
  const unsigned short* lo_t = loc_t + (int64_t)blockIdx.x * n_parts;
  const unsigned short* hi_t = lo_t + n_parts;
  long long sum = 0;
  for (int64_t p = (int64_t)blockIdx.y * 256 + threadIdx.x; p < n_parts; p += (int64_t)gridDim.y * 256) sum += (long long)hi_t[p] - (long long)lo_t[p];
  long long tot;
  block_exclusive_scan<256>(sum, s_wave, &tot);
  if (threadIdx.x == 0 && tot != 0) atomicAdd(&weight[blockIdx.x], (unsigned long long)tot);
}
// ... and (single block) blk_prefix[b] = first histogram block of bucket b: S blocks per average bucket weight, at least one, at most one per part
__global__ __launch_bounds__(SCAN_THREADS) void pl_blockmap_kernel(const long long* __restrict__ weight, int n_buckets, int64_t n_parts, int S,
                                                                  int32_t* __restrict__ blk_prefix) {
  __shared__ long long s_wave[SCAN_THREADS / WAVE];
  __shared__ long long s_total;
  {
    const int b = threadIdx.x;  // n_buckets <= PL_MAX_BUCKETS <= SCAN_THREADS
    long long tot;
    block_exclusive_scan(b < n_buckets ? weight[b] : 0ll, s_wave, &tot);
    if (threadIdx.x == 0) s_total = tot;
  }
  __syncthreads();
  const long long total = s_total;
  const int b = threadIdx.x;
  long long v = 0;
  if (b < n_buckets) {
    // ceil(S * n_buckets * weight / total): the sum over the buckets is at most (S + 1) * n_buckets
    v = total > 0 ? (weight[b] * (long long)S * n_buckets + total - 1) / total : 1;
    if (v < 1) v = 1;
    if (v > n_parts) v = n_parts;
  }
  long long tot;
  const long long ex = block_exclusive_scan(v, s_wave, &tot);
  if (b < n_buckets) blk_prefix[b] = (int32_t)ex;
  if (b == 0) blk_prefix[n_buckets] = (int32_t)tot;
}

// Environment knobs of the column counts.  The profiling-only ones are read ONCE per process (every launch called getenv
// from several enqueueing threads); the two the test-suite toggles at run time -- URCCO_COLCOUNT_GLOBAL_LAYOUT and URCCO_PH_CHUNK_BIG_NNZ
// -- stay per call, and column_counts_scratch_bytes sizes for whichever value the launch may later see (see there).
struct PlKnobs {
  long long block_ids = 49152;  // URCCO_PL_BLOCK_IDS: ids per average histogram block (0 = no such bound)
  int debug = 0;                // URCCO_PL_DEBUG (profiling only): 1 = no LDS atomics, 2 = no loads (the counts are then meaningless)
  int lanes = 8;                // URCCO_PL_LANES: lanes per slice (16, 8 or 4)
  PlKnobs() {
    if (const char* e = getenv("URCCO_PL_BLOCK_IDS")) if (*e) block_ids = atoll(e);
    if (const char* e = getenv("URCCO_PL_DEBUG")) if (*e) debug = atoi(e);
    if (const char* e = getenv("URCCO_PL_LANES")) if (*e) lanes = atoi(e);
  }
};
static const PlKnobs& pl_knobs() {
  static const PlKnobs k;
  return k;
}
// histogram blocks per AVERAGE bucket: a few thousand blocks in all, each with at least a handful of parts; bounded = false: without the
// bound by work (an upper bound of the split for any knob value: what the scratch is sized for)
static inline int pl_splits(int n_buckets, int64_t n_parts, bool bounded = true) {
  int64_t S = (2048 + n_buckets - 1) / n_buckets;
  if (S > n_parts / 4) S = n_parts / 4;
  // ... and an average block should count a few ten thousand ids for the 64 KiB of LDS it clears and the 64 KiB of partial counters it
  // publishes (a rank's user shard at 8 ranks: 2200 blocks of ~20K ids each; the weight split keeps the hot buckets' blocks average too)
  const int64_t ids = pl_knobs().block_ids;
  if (bounded && ids > 0) {
    const int64_t by_work = n_parts * PL_PART / ((int64_t)n_buckets * ids);
    if (S > by_work) S = by_work;
  }
  if (S < 1) S = 1;
  return (int)S;
}
static inline int64_t pl_max_blocks(int n_buckets, int S) { return ((int64_t)S + 1) * n_buckets; }
static inline bool pl_fits(int32_t n_cols) { return (((int64_t)n_cols + PL_BUCKET - 1) >> PL_BITS) <= PL_MAX_BUCKETS; }
static inline bool pl_applies(int32_t n_cols) {
  const char* e = getenv("URCCO_COLCOUNT_GLOBAL_LAYOUT");  // A/B and test knob (per call): the bucket-contiguous form above
  if (e && *e == '1') return false;
  return pl_fits(n_cols);
}

// The larger of what the two layouts need (0 when neither applies): the layout and the chunk size are chosen per call from the environment,
// and a value that changed between this call and the launch must not leave the launch with a buffer sized for the other form.
int64_t column_counts_scratch_bytes(int64_t nnz, int32_t n_cols) {
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  if (nnz < PH_MIN_NNZ) return 0;
  int64_t need = 0;
  if (pl_fits(n_cols)) {
    const int64_t n_buckets = ((int64_t)n_cols + PL_BUCKET - 1) >> PL_BITS;
    const int64_t n_parts = (nnz + PL_PART - 1) / PL_PART;
    need = al(n_parts * PL_PART * 2 + 64) + al((n_buckets + 1) * n_parts * 2) + al(n_buckets * 8) + al((n_buckets + 1) * 4) +
           al(pl_max_blocks((int)n_buckets, pl_splits((int)n_buckets, n_parts, false)) * (int64_t)PL_BUCKET * 4);
  }
  if ((((int64_t)n_cols + PH_BUCKET - 1) >> PH_BITS) <= PH_MAX_BUCKETS) {
    const int64_t n_buckets = ((int64_t)n_cols + PH_BUCKET - 1) >> PH_BITS;
    const int64_t n_parts = (nnz + PH_PART - 1) / PH_PART;
    const int64_t m = n_buckets * n_parts;
    const int64_t max_blocks = n_buckets + (nnz + PH_CHUNK_SMALL - 1) / PH_CHUNK_SMALL;  // (the smaller chunk: the larger block count)
    const int64_t ph = al(m * 4) + al((m + 1) * 8) + al(((m + SCAN_TILE - 1) / SCAN_TILE + 2) * 8) + al(nnz * 2 + 16) + al((n_buckets + 1) * 4) + al(max_blocks * PH_BUCKET * 2);
    if (ph > need) need = ph;
  }
  return need;
}

hipError_t launch_column_counts_partitioned(hipStream_t st, const int32_t* col_idx, int64_t nnz, const int64_t* nnz_dev, int32_t n_cols,
                                            int32_t* counts, char* scratch) {
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  const int vec_ok = (reinterpret_cast<uintptr_t>(col_idx) & 15) == 0;
  if (pl_applies(n_cols)) {
    const int n_buckets = (int)(((int64_t)n_cols + PL_BUCKET - 1) >> PL_BITS);
    const int64_t n_parts = (nnz + PL_PART - 1) / PL_PART;
    const int S = pl_splits(n_buckets, n_parts);
    unsigned short* bucketed = reinterpret_cast<unsigned short*>(scratch); scratch += al(n_parts * PL_PART * 2 + 64);
    unsigned short* loc_t = reinterpret_cast<unsigned short*>(scratch); scratch += al(((int64_t)n_buckets + 1) * n_parts * 2);
    long long* weight = reinterpret_cast<long long*>(scratch); scratch += al((int64_t)n_buckets * 8);
    int32_t* blk_prefix = reinterpret_cast<int32_t*>(scratch); scratch += al(((int64_t)n_buckets + 1) * 4);
    unsigned* partial = reinterpret_cast<unsigned*>(scratch);
    hipLaunchKernelGGL(pl_partition_kernel, dim3((unsigned)n_parts), dim3(PL_THREADS), 0, st, col_idx, nnz, nnz_dev, n_buckets, n_parts, bucketed, loc_t, vec_ok);
    hipError_t we = hipMemsetAsync(weight, 0, sizeof(long long) * (size_t)n_buckets, st);
    if (we != hipSuccess) return we;
    const unsigned wsplit = (unsigned)(n_parts >= 8192 ? 8 : (n_parts >= 1024 ? 4 : 1));
    hipLaunchKernelGGL(pl_weights_kernel, dim3((unsigned)n_buckets, wsplit), dim3(256), 0, st, loc_t, n_parts, reinterpret_cast<unsigned long long*>(weight));
    hipLaunchKernelGGL(pl_blockmap_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, weight, n_buckets, n_parts, S, blk_prefix);
    const int dbg = pl_knobs().debug, lps = pl_knobs().lanes;
    const dim3 hg((unsigned)pl_max_blocks(n_buckets, S)), hb(PLH_THREADS);
    if (lps == 4) hipLaunchKernelGGL((pl_hist_kernel<4>), hg, hb, 0, st, bucketed, loc_t, n_buckets, n_parts, blk_prefix, n_cols, partial, dbg);
    else if (lps == 8) hipLaunchKernelGGL((pl_hist_kernel<8>), hg, hb, 0, st, bucketed, loc_t, n_buckets, n_parts, blk_prefix, n_cols, partial, dbg);
    else hipLaunchKernelGGL((pl_hist_kernel<16>), hg, hb, 0, st, bucketed, loc_t, n_buckets, n_parts, blk_prefix, n_cols, partial, dbg);
    hipLaunchKernelGGL(pl_reduce_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, partial, blk_prefix, n_cols, counts);
    return hipGetLastError();
  }
  const int n_buckets = (int)(((int64_t)n_cols + PH_BUCKET - 1) >> PH_BITS);
  const int64_t n_parts = (nnz + PH_PART - 1) / PH_PART;
  const int64_t m = (int64_t)n_buckets * n_parts;
  const int chunk = ph_chunk(nnz);
  const int64_t max_blocks = n_buckets + (nnz + chunk - 1) / chunk;
  int32_t* part_counts = reinterpret_cast<int32_t*>(scratch); scratch += al(m * 4);
  int64_t* offsets = reinterpret_cast<int64_t*>(scratch); scratch += al((m + 1) * 8);
  int64_t* tile_sums = reinterpret_cast<int64_t*>(scratch); scratch += al(((m + SCAN_TILE - 1) / SCAN_TILE + 2) * 8);
  unsigned short* bucketed = reinterpret_cast<unsigned short*>(scratch); scratch += al(nnz * 2 + 16);
  int32_t* blk_prefix = reinterpret_cast<int32_t*>(scratch); scratch += al(((int64_t)n_buckets + 1) * 4);
  unsigned short* partial = reinterpret_cast<unsigned short*>(scratch);
  hipLaunchKernelGGL(ph_count_kernel, dim3((unsigned)n_parts), dim3(256), 0, st, col_idx, nnz, nnz_dev, n_buckets, n_parts, part_counts, vec_ok);
  hipError_t e = launch_scan_i32(st, part_counts, m, offsets, tile_sums);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ph_scatter_kernel, dim3((unsigned)n_parts), dim3(PHS_THREADS), 0, st, col_idx, nnz, nnz_dev, n_buckets, n_parts, offsets, bucketed, vec_ok);
  hipLaunchKernelGGL(ph_blockmap_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, offsets, n_buckets, n_parts, blk_prefix, chunk);
  hipLaunchKernelGGL(ph_hist_kernel, dim3((unsigned)max_blocks), dim3(PHH_THREADS), 0, st, bucketed, offsets, n_buckets, n_parts, blk_prefix, partial, chunk);
  hipLaunchKernelGGL(ph_reduce_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, partial, blk_prefix, n_cols, counts);
  return hipGetLastError();
}

// ============================================================================================
// K2  sampleDownAndBinarize -- the CSR row scan.  Tiles of DS_TILE consecutive entries, one block each.
//   tile rows  g[t] = first row that starts at or after entry t*DS_TILE: one pass over row_ptr.  (A binary search per
//              tile costs ~20 dependent global loads before the tile can start and was, measured, the larger part of
//              the scan; a single-pass form with a decoupled look-back across tiles was measured slower still --
//              the resident tiles finish their keep decisions in lock step and then queue on each other.)
//   flags      16 B per lane coalesced column loads; the tile's row_ptr slice staged in LDS for the entry -> row
//              lookup; keep decision per entry = u01(seed,row,col) <= min(perRowRate, perThingRate); the 4-bit nibbles
//              of 16 neighbouring lanes OR-assembled into one 64-bit keep word; kept count per tile.  Post-sampling
//              column counts by L2 atomics (small matrices only; <= ~max per address after the cut).
//   scan       exclusive prefix of the per-tile counts (one block)
//   compact    per tile: prefix over its 64 keep words in LDS -> output position of every kept entry, and the new
//              row_ptr of the rows that start inside the tile
// ============================================================================================
constexpr int DS_THREADS = 256;
constexpr int DS_ITERS = DS_TILE / (DS_THREADS * 4);  // 4
constexpr int DS_WORDS = DS_TILE / 64;
static_assert(DS_WORDS == WAVE, "one wave scans the keep words of a tile");
static_assert((DS_TILE & (DS_TILE - 1)) == 0, "tile index by shift");

constexpr int THR8_SHIFT = 45;  // one-byte threshold prefix = bits 45..52 of the 53-bit threshold (see sample_threshold_kernel)
constexpr int THR8_SHIFT32 = 24;  // ... = bits 24..31 of the 32-bit threshold (URCCO_RNG_MIX32)
constexpr unsigned long long RATE_ONE = 1ull << 53;  // threshold of a sample rate of 1.0 (every 53-bit hash passes)

// first idx in [lo, hi] with rp[idx] > e   (rp[hi] > e guaranteed by the caller)
__device__ __forceinline__ int64_t upper_bound_i64(const int64_t* __restrict__ rp, int64_t lo, int64_t hi, int64_t e) {
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (rp[mid] > e) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// g[t] >> 1 = first row r with rp[r] >= t * DS_TILE for t < n_tiles (row r writes the tiles with rp[r-1] < t*DS_TILE <= rp[r]:
// one writer per tile); g[n_tiles] = n_rows + 1, so that [g[t], g[t+1]) partitions the rows 0..n_rows (end marker included).
__global__ __launch_bounds__(256) void tile_rows_kernel(int64_t n_rows, const int64_t* __restrict__ rp, int64_t n_tiles, int64_t* __restrict__ g) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= n_rows; r += (int64_t)gridDim.x * 256) {
    int64_t t = r == 0 ? 0 : rp[r - 1] / DS_TILE + 1;
    const int64_t e = rp[r];
    const int64_t t_hi = e / DS_TILE;
    for (; t <= t_hi && t < n_tiles; ++t) g[t] = (r << 1) | (int64_t)(e == t * DS_TILE);  // low bit: row r starts exactly at the tile start
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) g[n_tiles] = (n_rows + 1) << 1;
}

// Each thread owns two runs of EIGHT consecutive entries (two 16-byte loads each): a run's keep bits are one byte of the
// tile's keep words, written straight from the lane -- no cross-lane assembly.
//
// entry -> row without a search and without a divergent walk (round 3; the round-2 kernel spent 40 scalar and 51 vector
// instructions per entry slot -- a per-run binary search over the LDS row_ptr slice, then `while (entry >= row end)` per
// entry, each `if` a handful of exec-mask instructions -- where the keep decision itself, the 64-bit hash, is 19): the
// tile's NON-EMPTY rows mark their start position in a 4096-bit LDS mask and leave their slice index at s_row_at[start]
// (one writer per position: empty rows own no entry).  One wave turns the mask into "last row starting before word w"
// (a prefix maximum: row indices grow with the position).  A run then reads its byte of the mask, the eight s_row_at
// words behind it (two 16-byte LDS reads) and selects, entry by entry, "the row that starts here, else the row so far":
// three vector instructions per entry, no branch.  Rows longer than the interaction cap (perRowSampleRate != 1) are
// rare: the tile notes whether it holds one and only then looks the row lengths up.  The row_ptr slice is read from
// global memory (coalesced, twice: as a start and as the previous row's end), so a tile with any number of empty rows
// needs no staging.
// `debug` (profiling only, results meaningless): 32 = cheap hash, 64 = no threshold gather, 128 = no row lookup
constexpr int DS_RUN = 8;
constexpr int DS_RUNS = DS_TILE / (DS_THREADS * DS_RUN);  // 2
constexpr int URCCO_DS_WAVES = 1;  // minimum waves per SIMD the flags kernel is compiled for (A/B knob: 8 caps it at 64 VGPRs)
template <bool DEBUG, bool RNG32>
__global__ __launch_bounds__(DS_THREADS, URCCO_DS_WAVES) void downsample_flags_kernel(int64_t n_rows, const int64_t* __restrict__ rp,
                                                                      const int32_t* __restrict__ ci, int64_t nnz,
                                                                      const int64_t* __restrict__ g,
                                                                      const unsigned long long* __restrict__ thresholds,
                                                                      const unsigned char* __restrict__ thr8, uint32_t seed,
                                                                      int32_t max_n, int row_rate_mode, int64_t row_base,
                                                                      unsigned long long* __restrict__ flags,
                                                                      int64_t* __restrict__ tile_count,
                                                                      int32_t* __restrict__ post_counts, int vec_ok, int debug_flags) {
  const int debug = DEBUG ? debug_flags : 0;  // the ablation switches exist only in the profiling instantiation
  __shared__ unsigned long long s_mask[DS_WORDS];  // bit p: a non-empty row starts at entry p of the tile
  __shared__ int s_row_at[DS_TILE];                // [p] (only where the bit is set): slice index of that row
  __shared__ int s_tbefore[DS_WORDS];              // slice index of the last row starting before word w (0: the row covering the tile start)
  __shared__ int s_cnt[DS_THREADS / WAVE];
  __shared__ int s_long;                           // the tile holds a row with more than max_n entries
  const int64_t tile = blockIdx.x;
  const int64_t e0 = tile * DS_TILE;
  const int n_live = (int)((e0 + DS_TILE < nnz) ? DS_TILE : nnz - e0);  // entries of this tile
  const int lane = threadIdx.x & (WAVE - 1);
  // all column loads of the thread are requested before anything else (independent of the row lookup)
  int cols[DS_RUNS][DS_RUN];
#pragma unroll
  for (int gq = 0; gq < DS_RUNS; ++gq) {
    const int el0 = (gq * DS_THREADS + (int)threadIdx.x) * DS_RUN;
    const int64_t e = e0 + el0;
    if (vec_ok && el0 + DS_RUN <= n_live) {
      const int4 x = *reinterpret_cast<const int4*>(ci + e), y = *reinterpret_cast<const int4*>(ci + e + 4);
      cols[gq][0] = x.x; cols[gq][1] = x.y; cols[gq][2] = x.z; cols[gq][3] = x.w;
      cols[gq][4] = y.x; cols[gq][5] = y.y; cols[gq][6] = y.z; cols[gq][7] = y.w;
    } else {
#pragma unroll
      for (int q = 0; q < DS_RUN; ++q) cols[gq][q] = (el0 + q < n_live) ? ci[e + q] : 0;
    }
  }
  if (threadIdx.x < DS_WORDS) s_mask[threadIdx.x] = 0ull;
  if (threadIdx.x == 0) s_long = 0;
  // slice rp[r_s .. r_e]: r_s = the last row known to start at or before e0, r_e = the first row starting at or after e1
  const int64_t gp0 = g[tile];
  const int64_t g0 = gp0 >> 1, g1 = g[tile + 1] >> 1;
  // the "row starts exactly at the tile start" bit travels in the tile table: one dependent global load less per tile
  const int64_t r_s = (gp0 & 1) ? g0 : g0 - 1;
  const int64_t r_e = g1 < n_rows ? g1 : n_rows;
  const int64_t n_slice = r_e - r_s + 1;
  __syncthreads();
  if (!(debug & 128)) {
    int any_long = 0;
    for (int64_t t = threadIdx.x; t + 1 < n_slice; t += DS_THREADS) {  // rows r_s + t, t < n_slice - 1 (row r_e starts behind the tile)
      const int64_t a = rp[r_s + t] - e0, b = rp[r_s + t + 1] - e0;
      if (b > a) {  // non-empty: the one row that owns the entries from a on
        if (a >= 0) {  // a < DS_TILE: only r_e may start at or behind the tile end
          s_row_at[a] = (int)t;
          atomicOr(&s_mask[a >> 6], 1ull << (a & 63));
        }
        any_long |= (b - a > (int64_t)max_n) ? 1 : 0;
      }
    }
    if (any_long) s_long = 1;
  }
  __syncthreads();
  if (threadIdx.x < WAVE) {  // wave 0: slice index of the last row starting before each word (prefix maximum)
    const unsigned long long m = s_mask[lane];
    const int here = m ? s_row_at[lane * 64 + 63 - __clzll((long long)m)] : 0;
    int inc = here;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (lane >= d) inc = o > inc ? o : inc;
    }
    const int ex = __shfl_up(inc, 1);
    s_tbefore[lane] = lane == 0 ? 0 : ex;
  }
  __syncthreads();
  const bool has_long = s_long != 0;
  const double dmax = (double)max_n;
  const uint32_t row0 = (uint32_t)(row_base + r_s);
  const uint32_t key0 = mix32_row_key(seed, row0);  // RNG32: the key of row r_s + t is key0 + t * MIX32_ROW
  int kept = 0;
#pragma unroll
  for (int gq = 0; gq < DS_RUNS; ++gq) {
    const int run = gq * DS_THREADS + (int)threadIdx.x;
    const int el0 = run * DS_RUN;
    unsigned thr_col[DS_RUN];  // the eight one-byte threshold gathers of the run travel together, under its row lookup
#pragma unroll
    for (int q = 0; q < DS_RUN; ++q) thr_col[q] = (debug & 64) ? 255u : (unsigned)thr8[cols[gq][q]];
    const int w = el0 >> 6, sh = el0 & 63;
    const unsigned long long m = s_mask[w];
    const unsigned starts = (unsigned)(m >> sh) & 0xffu;             // rows starting inside the run
    const unsigned long long low = m & ((1ull << sh) - 1ull);        // ... and before it, in the same word
    int t_cur = s_tbefore[w];
    if (low) t_cur = s_row_at[w * 64 + 63 - __clzll((long long)low)];
    const int4 ra = *reinterpret_cast<const int4*>(&s_row_at[el0]), rb = *reinterpret_cast<const int4*>(&s_row_at[el0 + 4]);
    const int at[DS_RUN] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    int r_of[DS_RUN];
#pragma unroll
    for (int q = 0; q < DS_RUN; ++q) {
      t_cur = (starts >> q) & 1u ? at[q] : t_cur;
      r_of[q] = t_cur;
    }
    // keep  <=>  hash <= perRow threshold  &&  hash <= perThing threshold: u01 = m * 2^-53 with integer m < 2^53, so
    // u01 <= rate  <=>  m <= floor(rate * 2^53) (the scaling is exact); a rate of 1.0 (threshold 2^53) always passes
    unsigned keep_byte = 0;
#pragma unroll
    for (int q = 0; q < DS_RUN; ++q) {
      const unsigned long long h = RNG32 ? (unsigned long long)mix32_finish((uint32_t)cols[gq][q] ^ (key0 + (uint32_t)r_of[q] * MIX32_ROW))
                                   : ((debug & 32) ? ((unsigned long long)((unsigned)cols[gq][q] * 0x9E3779B1u) << 21) : hash53(seed, row0 + (uint32_t)r_of[q], (uint32_t)cols[gq][q]));
      const unsigned h8 = (unsigned)(h >> (RNG32 ? THR8_SHIFT32 : THR8_SHIFT)), b = thr_col[q];
      bool keep = b == 255u || h8 < b;
      if (b == 254u || (b < 254u && h8 == b)) keep = h <= thresholds[cols[gq][q]];  // 1 sampled interaction in 256: the full threshold
      keep_byte |= (keep ? 1u : 0u) << q;
    }
    if (has_long) {  // block-uniform, rare: a user with more interactions than the cap sits in this tile
#pragma unroll 1
      for (int q = 0; q < DS_RUN; ++q) {
        const int64_t r = r_s + r_of[q];
        const int64_t n_row = rp[r + 1] - rp[r];
        if (n_row > (int64_t)max_n) {  // Int / Int = 0: only a hash of exactly 0 passes; fractional: min(max, n) / n
          const unsigned long long thr_row = row_rate_mode == 0 ? 0ull : (unsigned long long)((dmax / (double)n_row) * (RNG32 ? 4294967296.0 : 9007199254740992.0));
          const unsigned long long hr = RNG32 ? (unsigned long long)mix32(seed, row0 + (uint32_t)r_of[q], (uint32_t)cols[gq][q]) : hash53(seed, row0 + (uint32_t)r_of[q], (uint32_t)cols[gq][q]);
          if (hr > thr_row) keep_byte &= ~(1u << q);
        }
      }
    }
    const int live = n_live - el0;  // entries of the run inside the matrix (the last tile is ragged)
    keep_byte &= live >= DS_RUN ? 0xffu : (live > 0 ? (1u << live) - 1u : 0u);
    if (post_counts) {  // small matrices only: post-sampling column counts by L2 atomics
#pragma unroll 1
      for (int q = 0; q < DS_RUN; ++q)
        if ((keep_byte >> q) & 1u) atomicAdd(&post_counts[cols[gq][q]], 1);
    }
    kept += __popc(keep_byte);
    // byte b of keep word w covers entries 64 w + 8 b ..: this run's byte; runs behind the last entry are written as zero
    reinterpret_cast<unsigned char*>(flags + tile * DS_WORDS)[run] = (unsigned char)keep_byte;
  }
  for (int msk = 1; msk < WAVE; msk <<= 1) kept += __shfl_xor(kept, msk);
  if (lane == 0) s_cnt[threadIdx.x / WAVE] = kept;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
#pragma unroll
    for (int w = 0; w < DS_THREADS / WAVE; ++w) tot += s_cnt[w];
    tile_count[tile] = tot;
  }
}

__global__ __launch_bounds__(DS_THREADS) void downsample_compact_kernel(int64_t n_rows, const int64_t* __restrict__ rp,
                                                                        const int32_t* __restrict__ ci, int64_t nnz,
                                                                        const int64_t* __restrict__ g,
                                                                        const unsigned long long* __restrict__ flags,
                                                                        const int64_t* __restrict__ tile_off,
                                                                        int64_t* __restrict__ out_rp, int32_t* __restrict__ out_ci,
                                                                        int vec_ok) {
  __shared__ unsigned long long s_keep[DS_WORDS];
  __shared__ int s_wpre[DS_WORDS + 1];
  const int64_t tile = blockIdx.x;
  const int64_t e0 = tile * DS_TILE;
  const int lane = threadIdx.x & (WAVE - 1);
  int cols[DS_ITERS][4];
#pragma unroll
  for (int it = 0; it < DS_ITERS; ++it) {
    const int64_t e = e0 + ((int64_t)it * DS_THREADS + threadIdx.x) * 4;
    if (vec_ok && e + 3 < nnz) {
      const int4 x = *reinterpret_cast<const int4*>(ci + e);
      cols[it][0] = x.x; cols[it][1] = x.y; cols[it][2] = x.z; cols[it][3] = x.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) cols[it][q] = (e + q < nnz) ? ci[e + q] : 0;
    }
  }
  if (threadIdx.x < WAVE) {  // wave 0: prefix over the tile's keep words
    const unsigned long long word = flags[tile * DS_WORDS + lane];
    s_keep[lane] = word;
    const int c = __popcll(word);
    int inc = c;
    for (int d = 1; d < WAVE; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    s_wpre[lane] = inc - c;
    if (lane == WAVE - 1) s_wpre[DS_WORDS] = inc;
  }
  __syncthreads();
  const int64_t off = tile_off[tile];
#pragma unroll
  for (int it = 0; it < DS_ITERS; ++it) {  // kept column ids in entry order
    const int w = (it * DS_THREADS + (int)threadIdx.x) >> 4;
    const int b = (lane & 15) * 4;
    const unsigned long long word = s_keep[w];
    const unsigned nib = (unsigned)(word >> b) & 0xFu;
    if (nib) {
      int64_t pos = off + s_wpre[w] + __popcll(b == 0 ? 0ull : (word & ((1ull << b) - 1ull)));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (nib & (1u << q)) out_ci[pos++] = cols[it][q];
    }
  }
  // new row_ptr of the rows that start inside this tile (the last tile also takes the rows behind the last entry)
  for (int64_t r = (g[tile] >> 1) + threadIdx.x; r < (g[tile + 1] >> 1); r += DS_THREADS) {
    const int rel = (int)(rp[r] - e0);
    const int w = rel >> 6, b = rel & 63;
    out_rp[r] = off + s_wpre[w] + (b == 0 ? 0 : __popcll(s_keep[w] & ((1ull << b) - 1ull)));
  }
}

// perThingSampleRate = min(max, n) / n of sampleDownAndBinarize as the integer threshold floor(rate * 2^53), and its ONE-BYTE
// prefix.  The scan gathers one threshold per interaction; an 8-byte table of a 2M-item catalogue is 16 MB -- four times an
// XCD's L2 -- and the gather (its L2 misses, and the address processing of 64 scattered lines per wave instruction) was 70-80 %
// of the flags kernel on the 10M x 2M configurations (profiles/r03_rowscan_ablation.log: 2.13 ms with, 0.42 ms without it).  The
// byte table of the same catalogue is 2 MB; the top 8 bits of the 53-bit hash against the top 8 bits of the threshold decide all
// but 1 in 256 sampled interactions, the rest compare in full:
//   255   perThingSampleRate = 1.0: keep                     254   always compare in full (threshold prefix >= 254)
//   b     hash >> 45 < b: keep   > b: drop   == b: compare in full
__global__ __launch_bounds__(256) void sample_threshold_kernel(const int32_t* __restrict__ raw_counts, int32_t n_cols, int32_t max_n,
                                                               unsigned long long* __restrict__ thresholds, unsigned char* __restrict__ thr8, int rng32) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_cols) return;
  const double n_thing = (double)raw_counts[j];
  const double dmax = (double)max_n;
  const bool one = n_thing <= dmax;
  // u01 = h * 2^-bits with an integer h < 2^bits, so u01 <= rate  <=>  h <= floor(rate * 2^bits) (the scaling is exact); RATE_ONE passes every h
  const unsigned long long thr = one ? RATE_ONE : (unsigned long long)((dmax / n_thing) * (rng32 ? 4294967296.0 : 9007199254740992.0));
  thresholds[j] = thr;
  const unsigned t8 = (unsigned)(thr >> (rng32 ? THR8_SHIFT32 : THR8_SHIFT));
  thr8[j] = (unsigned char)(one ? 255u : (t8 >= 254u ? 254u : t8));
}

hipError_t launch_downsample_flags(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz,
                                   int32_t n_cols, const int32_t* raw_counts, unsigned long long* thresholds, uint32_t seed, int32_t max_n,
                                   int row_rate_mode, int64_t row_base, int64_t* tile_rows, unsigned long long* flags, int64_t* tile_count,
                                   int32_t* post_counts, int debug) {
  if (nnz == 0) return hipSuccess;
  unsigned char* thr8 = reinterpret_cast<unsigned char*>(thresholds + n_cols);  // the scratch holds n_cols u64 + n_cols bytes
  const int rng32 = (row_rate_mode & 0x100) ? 1 : 0;  // URCCO_RNG_MIX32
  row_rate_mode &= 0xff;
  hipLaunchKernelGGL(sample_threshold_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, raw_counts, n_cols, max_n, thresholds, thr8, rng32);
  const int64_t tiles = (nnz + DS_TILE - 1) / DS_TILE;
  int64_t rblocks = (n_rows + 1 + 255) / 256;
  const int64_t rcap = (int64_t)n_cu * 8;
  if (rblocks > rcap) rblocks = rcap;
  hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)rblocks), dim3(256), 0, st, n_rows, row_ptr, tiles, tile_rows);
  const int vec_ok = (reinterpret_cast<uintptr_t>(col_idx) & 15) == 0;
  if (debug & (32 | 64 | 128)) {
    if (rng32)
      hipLaunchKernelGGL((downsample_flags_kernel<true, true>), dim3((unsigned)tiles), dim3(DS_THREADS), 0, st, n_rows, row_ptr, col_idx, nnz, tile_rows, thresholds, thr8,
                         seed, max_n, row_rate_mode, row_base, flags, tile_count, post_counts, vec_ok, debug);
    else
      hipLaunchKernelGGL((downsample_flags_kernel<true, false>), dim3((unsigned)tiles), dim3(DS_THREADS), 0, st, n_rows, row_ptr, col_idx, nnz, tile_rows, thresholds, thr8,
                         seed, max_n, row_rate_mode, row_base, flags, tile_count, post_counts, vec_ok, debug);
  } else if (rng32) {
    hipLaunchKernelGGL((downsample_flags_kernel<false, true>), dim3((unsigned)tiles), dim3(DS_THREADS), 0, st, n_rows, row_ptr, col_idx, nnz, tile_rows, thresholds, thr8,
                       seed, max_n, row_rate_mode, row_base, flags, tile_count, post_counts, vec_ok, 0);
  } else {
    hipLaunchKernelGGL((downsample_flags_kernel<false, false>), dim3((unsigned)tiles), dim3(DS_THREADS), 0, st, n_rows, row_ptr, col_idx, nnz, tile_rows, thresholds, thr8,
                       seed, max_n, row_rate_mode, row_base, flags, tile_count, post_counts, vec_ok, 0);
  }
  return hipGetLastError();
}

// single block of 1024 threads, 8 consecutive values each: in-place exclusive scan of v[0..n), v[n] = total.  The tile
// counts of even the largest matrix are a few passes of this loop; a multi-kernel scan would cost more in launches.
constexpr int SS_THREADS = 1024;
constexpr int SS_ITEMS = 8;
__global__ __launch_bounds__(SS_THREADS) void scan_inplace_kernel(int64_t* __restrict__ v, int64_t n) {
  __shared__ long long s_wave[SS_THREADS / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  long long carry = 0;
  for (int64_t base = 0; base < n; base += SS_THREADS * SS_ITEMS) {  // block-uniform trip count
    const int64_t first = base + (int64_t)threadIdx.x * SS_ITEMS;
    long long x[SS_ITEMS];
    long long sum = 0;
#pragma unroll
    for (int q = 0; q < SS_ITEMS; ++q) {
      x[q] = first + q < n ? v[first + q] : 0;
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
    for (int w = 0; w < SS_THREADS / WAVE; ++w) {
      const long long sw = s_wave[w];
      if (w < wave) before += sw;
      tot += sw;
    }
    __syncthreads();
    long long run = carry + before + inc - sum;
#pragma unroll
    for (int q = 0; q < SS_ITEMS; ++q) {
      if (first + q < n) v[first + q] = run;
      run += x[q];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) v[n] = carry;
}

// in place: tile_count[0..tiles) -> exclusive offsets, tile_count[tiles] = number of kept entries
hipError_t launch_downsample_scan(hipStream_t st, int64_t nnz, int64_t* tile_count) {
  if (nnz == 0) return hipSuccess;
  hipLaunchKernelGGL(scan_inplace_kernel, dim3(1), dim3(SS_THREADS), 0, st, tile_count, (nnz + DS_TILE - 1) / DS_TILE);
  return hipGetLastError();
}

hipError_t launch_downsample_compact(hipStream_t st, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz,
                                     const int64_t* tile_rows, const unsigned long long* flags, const int64_t* tile_off, int64_t* out_row_ptr,
                                     int32_t* out_col_idx) {
  if (nnz == 0) return hipSuccess;
  const int64_t tiles = (nnz + DS_TILE - 1) / DS_TILE;
  const int vec_ok = (reinterpret_cast<uintptr_t>(col_idx) & 15) == 0;
  hipLaunchKernelGGL(downsample_compact_kernel, dim3((unsigned)tiles), dim3(DS_THREADS), 0, st, n_rows, row_ptr, col_idx, nnz, tile_rows, flags,
                     tile_off, out_row_ptr, out_col_idx, vec_ok);
  return hipGetLastError();
}

// ============================================================================================
// K3  CSR -> CSC (the A.t of A.t %*% B).  2^g lanes walk one user row; destination slots come from
// per-column cursors (returning L2 atomics; after the interaction cut a column sees <= ~max of them).
// Order inside a column is whatever the atomics produce: only integer sums are formed from it.
// ============================================================================================
__global__ __launch_bounds__(256) void transpose_kernel(int64_t n_rows, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                        int g_log2, const int64_t* __restrict__ col_ptr, int32_t* __restrict__ cursor,
                                                        int32_t* __restrict__ out_rows, int32_t col_lo, int32_t col_hi) {
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t groups_per_block = 256 >> g_log2;
  for (int64_t r = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> g_log2); r < n_rows;
       r += (int64_t)gridDim.x * groups_per_block) {
    const int64_t s = rp[r], e = rp[r + 1];
    for (int64_t p = s + gl; p < e; p += G) {
      const int j = ci[p];
      if (j < col_lo || j >= col_hi) continue;  // a rank only transposes the item range it owns
      const int pos = atomicAdd(&cursor[j], 1);
      out_rows[col_ptr[j] + pos] = (int32_t)r;
    }
  }
}

struct LoadI32Range {  // counts masked to [lo, hi): columns outside the range get empty CSC columns
  const int32_t* p;
  int32_t lo, hi;
  __device__ __forceinline__ long long operator()(int64_t i) const { return (i >= lo && i < hi) ? p[i] : 0; }
  __device__ __forceinline__ void load8(int64_t i, long long* x) const {
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = (*this)(i + q);
  }
};
hipError_t launch_scan_i32_range(hipStream_t st, const int32_t* in, int64_t n, int32_t lo, int32_t hi, int64_t* out, int64_t* tile_sums) {
  return launch_scan(st, LoadI32Range{in, lo, hi}, n, out, tile_sums);
}

hipError_t launch_transpose(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int g_log2,
                            const int64_t* col_ptr, int32_t* cursor, int32_t* out_row_idx, int32_t col_lo, int32_t col_hi) {
  if (n_rows == 0) return hipSuccess;
  const int64_t gpb = 256 >> g_log2;
  int64_t blocks = (n_rows + gpb - 1) / gpb;
  const int64_t cap = (int64_t)n_cu * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, row_ptr, col_idx, g_log2, col_ptr, cursor, out_row_idx, col_lo,
                     col_hi);
  return hipGetLastError();
}

constexpr int TR_PART = 16384;        // entries per part
constexpr int64_t TR_CHUNK = 1 << 18;  // a bucket heavier than this is placed by several blocks

// R[p] = first row r with rp[r] >= p * TR_PART (p < n_parts), R[n_parts] = n_rows
__global__ __launch_bounds__(256) void tr_parts_kernel(int64_t n_rows, const int64_t* __restrict__ rp, int64_t n_parts, int64_t* __restrict__ R) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p > n_parts) return;
  if (p == n_parts) { R[p] = n_rows; return; }
  const int64_t target = p * TR_PART;
  int64_t lo = 0, hi = n_rows;  // first r in [0, n_rows] with rp[r] >= target
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (rp[mid] >= target) hi = mid; else lo = mid + 1;
  }
  R[p] = lo;
}

// --------------------------------------------------------------------------------------------
// K3b  CSR -> CSC as a two-level counting sort by column, part-local form (round 6; the column counts' round-5 layout applied to the
// transposition; matrices large enough to repay the launches -- the cursor-atomic kernel above takes one RETURNING L2 atomic per entry).
// Rounds 2-5 (git 5263357, profiles/r06_transpose_rowscan_ab.log "URCCO_TRANSPOSE_V1=1") read the column indices twice (count, scatter),
// scanned a (bucket x part) table in between, walked a part ROW BY ROW (2^g lanes per row behind a dependent row_ptr read: sixteen
// short latency chains per thread) and wrote each bucket's run where the bucket lies: ~33-entry runs of 2- and 4-byte stores
// (scatter 747 us + place 507 us for config 4's 40 M entries: 1.41 ms, 4 % of HBM; this form: 0.88 ms on the same box).  Here:
//   partition  one block per FLAT part of TP_PART consecutive entries: column indices in 16-byte loads (32 per thread, all in flight
//              before anything else), entry -> row from the part's row starts (bit mask + slice index per start + prefix maximum: the
//              CSR row scan's lookup, no search), rank inside (bucket, lane copy) by one returning LDS atomic, the 16-bit in-bucket
//              columns grouped by bucket in LDS and written as the part lies (whole lines), the rows scattered INSIDE the part's own
//              64 KB (partial stores of one block into one region: the L2 merges them), loc_t[b][p] = start of bucket b in part p
//   place      one block per bucket (several per heavy bucket, sharing cursors in global memory): the bucket's column cursors in LDS,
//              four lanes per slice, eight entries (one 16-byte column load, two 16-byte row loads) per lane and step
// No count pass, no scan, no block-level row walk.  Only columns in [col_lo, col_hi) are kept; order inside a column is arbitrary.
// --------------------------------------------------------------------------------------------
constexpr int TP_PART = TR_PART;  // 16384 (tr_parts_kernel's quota)
constexpr int TP_THREADS = 512;
constexpr int TP_PER_THREAD = TP_PART / TP_THREADS;  // 32 entries in registers: eight runs of four
constexpr int TP_COPIES = 4;                         // lane-private copies of the bucket counters
constexpr int TP_MAX_BUCKETS = 512;
constexpr int TP_MAX_BITS = 14;                      // <= 16384 columns per bucket: 16-bit in-bucket columns, 64 KB of LDS cursors
constexpr int TP_WORDS = TP_PART / 64;
static_assert(TP_PER_THREAD % 4 == 0 && TP_MAX_BUCKETS <= TP_THREADS && TP_WORDS <= TP_THREADS, "one scan round; one thread per mask word");

static int tp_bucket_bits(int32_t n_cols) {
  int bits = 6;
  while (bits < TP_MAX_BITS && (((int64_t)n_cols + ((int64_t)1 << bits) - 1) >> bits) > TP_MAX_BUCKETS) ++bits;
  return bits;
}

__global__ __launch_bounds__(TP_THREADS, 4) void tp_partition_kernel(int64_t n_rows, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                                  const int64_t* __restrict__ R, int bits, int n_buckets, int64_t n_parts, int32_t col_lo,
                                                                  int32_t col_hi, unsigned short* __restrict__ bk_col, unsigned short* __restrict__ bk_row16,
                                                                  int32_t* __restrict__ bk_row, int64_t* __restrict__ part_base,
                                                                  unsigned short* __restrict__ loc_t, int vec_ok) {
  __shared__ int s_cnt[TP_COPIES * TP_MAX_BUCKETS];  // counts, then the start of every (copy, bucket) run inside the part
  __shared__ uint4 s_stage4[TP_PART / 8];            // the part's in-bucket columns grouped by bucket (leaves in 16-byte stores)
  __shared__ unsigned long long s_mask[TP_WORDS];    // bit e: a non-empty row starts at entry e of the part
  __shared__ __attribute__((aligned(16))) unsigned short s_row_at[TP_PART];  // [e] (only where the bit is set): slice index of that row
  __shared__ int s_tbefore[TP_WORDS];                // slice index of the last row starting before word w (0: the row covering the part's start)
  __shared__ long long s_wave[TP_THREADS / WAVE];
  __shared__ int s_wmax[TP_WORDS / WAVE];
  unsigned short* s_stage = reinterpret_cast<unsigned short*>(s_stage4);
  const int64_t part = blockIdx.x;
  const int64_t nnz = rp[n_rows];
  const int64_t e0 = part * TP_PART;
  if (e0 >= nnz) {  // a part beyond the device-side length (the launch is sized for the host's bound): every slice is empty (block-uniform)
    for (int b = threadIdx.x; b <= n_buckets; b += TP_THREADS) loc_t[(int64_t)b * n_parts + part] = 0;
    if (threadIdx.x == 0) part_base[part] = 0;
    return;
  }
  const int n = (int)(e0 + TP_PART < nnz ? TP_PART : nnz - e0);
  // all column loads of the thread are requested first: register 4 r + q holds entry r * (4 * TP_THREADS) + 4 * tid + q of the part
  int cols[TP_PER_THREAD];
#pragma unroll
  for (int r = 0; r < TP_PER_THREAD / 4; ++r) {
    const int el0 = r * (TP_THREADS * 4) + (int)threadIdx.x * 4;
    const int64_t e = e0 + el0;
    int4 x = make_int4(-1, -1, -1, -1);
    if (vec_ok && el0 + 3 < n) {
      x = *reinterpret_cast<const int4*>(ci + e);
    } else {
      if (el0 < n) x.x = ci[e];
      if (el0 + 1 < n) x.y = ci[e + 1];
      if (el0 + 2 < n) x.z = ci[e + 2];
      if (el0 + 3 < n) x.w = ci[e + 3];
    }
    cols[4 * r] = x.x; cols[4 * r + 1] = x.y; cols[4 * r + 2] = x.z; cols[4 * r + 3] = x.w;
  }
  for (int b = threadIdx.x; b < TP_COPIES * n_buckets; b += TP_THREADS) s_cnt[b] = 0;
  if (threadIdx.x < TP_WORDS) s_mask[threadIdx.x] = 0ull;
  // rows r_s .. r_e - 1 own the part's entries: r_s covers (or starts at) e0, r_e is the first row that starts at or behind the part's end
  const int64_t g0 = R[part];
  const int64_t r_s = (g0 < n_rows && rp[g0] == e0) ? g0 : g0 - 1;  // (rp[n_rows] = nnz > e0: g0 == n_rows means row n_rows - 1 covers e0)
  const int64_t g1 = R[part + 1];
  const int64_t r_e = g1 < n_rows ? g1 : n_rows;
  const int64_t n_slice = r_e - r_s + 1;
  const bool by_marks = n_slice <= 65536;  // block-uniform: slice indices fit the 16-bit marks (else: one binary search per run, rare)
  __syncthreads();
  if (by_marks) {
    for (int64_t t = threadIdx.x; t + 1 < n_slice; t += TP_THREADS) {
      const int64_t a = rp[r_s + t] - e0, b = rp[r_s + t + 1] - e0;
      if (b > a && a >= 0) {  // non-empty and starting inside the part (a < TP_PART: only r_e may start at or behind its end)
        s_row_at[a] = (unsigned short)t;
        atomicOr(&s_mask[a >> 6], 1ull << (a & 63));
      }
    }
  }
  __syncthreads();
  if (by_marks) {  // exclusive prefix maximum over the words: slice indices grow with the position
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    int inc = 0;
    if (threadIdx.x < TP_WORDS) {
      const unsigned long long m = s_mask[threadIdx.x];
      inc = m ? (int)s_row_at[threadIdx.x * 64 + 63 - __clzll((long long)m)] : 0;
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc = o > inc ? o : inc;
      }
      if (lane == WAVE - 1) s_wmax[wave] = inc;
    }
    __syncthreads();
    if (threadIdx.x < TP_WORDS) {
      int before = 0;
#pragma unroll
      for (int w = 0; w < TP_WORDS / WAVE; ++w)
        if (w < wave) before = s_wmax[w] > before ? s_wmax[w] : before;
      const int ex = __shfl_up(inc, 1);
      const int mine = lane == 0 ? 0 : ex;
      s_tbefore[threadIdx.x] = mine > before ? mine : before;
    }
    __syncthreads();
  }
  // rank inside (bucket, copy): one returning LDS atomic per kept entry (two 16-bit ranks per register)
  int* mine = s_cnt + (threadIdx.x & (TP_COPIES - 1)) * n_buckets;
  unsigned rank2[TP_PER_THREAD / 2];
#pragma unroll
  for (int q = 0; q < TP_PER_THREAD; ++q) {
    const int j = cols[q];  // (-1 behind the matrix's end)
    const bool keep = j >= col_lo && j < col_hi;
    unsigned rk = 0u;
    if (keep) rk = (unsigned)atomicAdd(&mine[j >> bits], 1);
    else cols[q] = -1;
    rank2[q >> 1] = (q & 1) ? (rank2[q >> 1] | (rk << 16)) : rk;
  }
  __syncthreads();
  {  // exclusive prefix over (bucket, copy), bucket-major: where every run starts; the bucket starts go out as loc_t[b][part]
    const int b = threadIdx.x;  // n_buckets <= TP_MAX_BUCKETS <= TP_THREADS: one round
    long long tot = 0;
    if (b < n_buckets) {
#pragma unroll
      for (int k = 0; k < TP_COPIES; ++k) tot += s_cnt[k * n_buckets + b];
    }
    long long all;
    const long long ex = block_exclusive_scan<TP_THREADS>(tot, s_wave, &all);
    if (b < n_buckets) {
      int run = (int)ex;
      loc_t[(int64_t)b * n_parts + part] = (unsigned short)run;
#pragma unroll
      for (int k = 0; k < TP_COPIES; ++k) {
        const int c = s_cnt[k * n_buckets + b];
        s_cnt[k * n_buckets + b] = run;
        run += c;
      }
    }
    if (b == n_buckets % TP_THREADS) loc_t[(int64_t)n_buckets * n_parts + part] = (unsigned short)all;  // <= TP_PART = 16384
  }
  __syncthreads();
  // entry -> row.  With the marks (the rule) a row is its 16-bit SLICE INDEX: it is staged through LDS like the column -- in the words of s_row_at, free once
  // every lookup has been made -- and leaves in 16-byte stores beside part_base[part] = the slice's first row (round 6, second form: the rows as 4-byte
  // stores scattered over the part's 64 KB window left the caches as partial lines: 1.7 GB of traffic per launch for 0.5 GB of entries).  A part whose slice
  // holds more than 65536 rows writes 32-bit rows the scattered way and says so with part_base[part] = -1.
  const unsigned cmask = (1u << bits) - 1u;
  if (by_marks) {
    unsigned t2[TP_PER_THREAD / 2];  // slice indices, two per register
#pragma unroll
    for (int r = 0; r < TP_PER_THREAD / 4; ++r) {
      const int el0 = r * (TP_THREADS * 4) + (int)threadIdx.x * 4;
      const int w = el0 >> 6, sh = el0 & 63;
      const unsigned long long m = s_mask[w];
      const unsigned starts = (unsigned)(m >> sh) & 0xfu;       // rows starting inside the run
      const unsigned long long low = m & ((1ull << sh) - 1ull);  // ... and before it, in the same word
      unsigned t_cur = (unsigned)s_tbefore[w];
      if (low) t_cur = (unsigned)s_row_at[w * 64 + 63 - __clzll((long long)low)];
      const uint2 at2 = *reinterpret_cast<const uint2*>(&s_row_at[el0]);
      const unsigned at[4] = {at2.x & 0xffffu, at2.x >> 16, at2.y & 0xffffu, at2.y >> 16};
      unsigned tq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        t_cur = (starts >> q) & 1u ? at[q] : t_cur;
        tq[q] = t_cur;
      }
      t2[2 * r] = tq[0] | (tq[1] << 16);
      t2[2 * r + 1] = tq[2] | (tq[3] << 16);
    }
    __syncthreads();  // every lookup has read s_row_at: its words now stage the rows
#pragma unroll
    for (int q = 0; q < TP_PER_THREAD; ++q) {
      const int j = cols[q];
      if (j >= 0) {
        const unsigned rk = (q & 1) ? rank2[q >> 1] >> 16 : rank2[q >> 1] & 0xffffu;
        const int pos = mine[j >> bits] + (int)rk;
        s_stage[pos] = (unsigned short)((unsigned)j & cmask);
        s_row_at[pos] = (unsigned short)((q & 1) ? t2[q >> 1] >> 16 : t2[q >> 1] & 0xffffu);
      }
    }
    __syncthreads();
    uint4* rdst = reinterpret_cast<uint4*>(bk_row16 + e0);
    const uint4* rsrc = reinterpret_cast<const uint4*>(s_row_at);
    for (int v = threadIdx.x; v * 8 < n; v += TP_THREADS) rdst[v] = rsrc[v];
    if (threadIdx.x == 0) part_base[part] = r_s;
  } else {
    int32_t* row_dst = bk_row + e0;
#pragma unroll 1
    for (int q = 0; q < TP_PER_THREAD; ++q) {
      const int j = cols[q];
      if (j >= 0) {
        const int el = (q >> 2) * (TP_THREADS * 4) + (int)threadIdx.x * 4 + (q & 3);
        const unsigned rk = (q & 1) ? rank2[q >> 1] >> 16 : rank2[q >> 1] & 0xffffu;
        const int pos = mine[j >> bits] + (int)rk;
        s_stage[pos] = (unsigned short)((unsigned)j & cmask);
        row_dst[pos] = (int)(upper_bound_i64(rp, r_s, r_e, e0 + el) - 1);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) part_base[part] = -1;
  }
  // the columns leave as the part lies: 16-byte stores (bk_col + e0 is 32 KiB-aligned relative to the array's 256-byte-aligned base)
  uint4* dst = reinterpret_cast<uint4*>(bk_col + e0);
  for (int v = threadIdx.x; v * 8 < n; v += TP_THREADS) dst[v] = s_stage4[v];  // entries behind the kept ones are stale: inside the part's own 32 KiB, never read
}

// weight -> placement blocks: a bucket of up to TR_CHUNK entries is placed by ONE block (cursors in LDS), a heavier one by one block per TR_CHUNK entries
__global__ __launch_bounds__(SCAN_THREADS) void tp_blockmap_kernel(const long long* __restrict__ weight, int n_buckets, int64_t n_parts, int32_t* __restrict__ blk_prefix) {
  __shared__ long long s_wave[SCAN_THREADS / WAVE];
  long long carry = 0;
  for (int base = 0; base < n_buckets; base += SCAN_THREADS) {  // block-uniform
    const int b = base + threadIdx.x;
    long long v = 0;
    if (b < n_buckets) {
      v = weight[b] > 0 ? (weight[b] + TR_CHUNK - 1) / TR_CHUNK : 1;
      if (v > n_parts) v = n_parts;
    }
    long long tot;
    const long long ex = block_exclusive_scan(v, s_wave, &tot);
    if (b < n_buckets) blk_prefix[b] = (int32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0) blk_prefix[n_buckets] = (int32_t)carry;
}

constexpr int TPP_THREADS = 1024;
constexpr int TPP_LPS = 4;  // lanes per slice: a slice of a 2M-column catalogue holds ~33 entries, a step of four lanes covers 32
__global__ __launch_bounds__(TPP_THREADS) void tp_place_kernel(const unsigned short* __restrict__ bk_col, const unsigned short* __restrict__ bk_row16,
                                                               const int32_t* __restrict__ bk_row, const int64_t* __restrict__ part_base,
                                                               const unsigned short* __restrict__ loc_t, int bits, int n_buckets, int64_t n_parts,
                                                               const int32_t* __restrict__ blk_prefix, const int64_t* __restrict__ col_ptr, int32_t n_cols,
                                                               int32_t* __restrict__ g_cursor /* [n_cols] zero */, int32_t* __restrict__ out_rows) {
  __shared__ unsigned s_cur[1 << TP_MAX_BITS];
  const int blk = blockIdx.x;
  if (blk >= blk_prefix[n_buckets]) return;  // block-uniform
  int blo = 0, bhi = n_buckets;  // last b with blk_prefix[b] <= blk
  while (bhi - blo > 1) {
    const int mid = (blo + bhi) >> 1;
    if (blk_prefix[mid] <= blk) blo = mid; else bhi = mid;
  }
  const int b = blo, s = blk - blk_prefix[b], S = blk_prefix[b + 1] - blk_prefix[b];
  const int width = 1 << bits;
  const int64_t col0 = (int64_t)b << bits;
  const bool shared_cursors = S > 1;  // block-uniform: a heavy bucket's blocks share cursors in global memory (returning L2 atomics)
  const int64_t base = col_ptr[col0];  // where the bucket's CSC segment starts (a bucket holds < 2^32 entries)
  if (!shared_cursors) {
    for (int c = threadIdx.x; c < width; c += TPP_THREADS) s_cur[c] = col0 + c < n_cols ? (unsigned)(col_ptr[col0 + c] - base) : 0u;
  }
  __syncthreads();
  const int64_t pp = (n_parts + S - 1) / S;
  const int64_t p0 = (int64_t)s * pp < n_parts ? (int64_t)s * pp : n_parts, p1 = p0 + pp < n_parts ? p0 + pp : n_parts;
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  const unsigned short* lo_t = loc_t + (int64_t)b * n_parts;
  const unsigned short* hi_t = loc_t + (int64_t)(b + 1) * n_parts;
  constexpr int GP = WAVE / TPP_LPS;  // parts per group = slices per wave and round (lane l < GP holds the bounds of part g + l)
  const int sub = lane / TPP_LPS, sl = lane % TPP_LPS;
  const int64_t gstep = (int64_t)(TPP_THREADS / WAVE) * GP;
  int64_t g = p0 + (int64_t)wave * GP;
  unsigned lo_n = 0u, hi_n = 0u;  // the NEXT group's bounds (and the parts' first rows) travel while this group's slices are placed
  int pb_n = 0;
  if (g < p1 && lane < GP && g + lane < p1) {
    lo_n = lo_t[g + lane];
    hi_n = hi_t[g + lane];
    pb_n = (int)part_base[g + lane];  // (a row index, or -1: the part's rows are 32-bit words)
  }
  for (; g < p1; g += gstep) {  // wave-uniform
    const unsigned lo = lo_n, hi = hi_n;
    const int pb = pb_n;
    lo_n = 0u;
    hi_n = 0u;
    pb_n = 0;
    if (g + gstep < p1 && lane < GP && g + gstep + lane < p1) {
      lo_n = lo_t[g + gstep + lane];
      hi_n = hi_t[g + gstep + lane];
      pb_n = (int)part_base[g + gstep + lane];
    }
    const unsigned lo_j = (unsigned)__shfl((int)lo, sub);  // (parts past the range carry lo == hi == 0)
    const unsigned hi_j = (unsigned)__shfl((int)hi, sub);
    const int pb_j = __shfl(pb, sub);
    const unsigned short* csrc = bk_col + (g + sub) * TP_PART;
    const unsigned short* r16 = bk_row16 + (g + sub) * TP_PART;
    const int32_t* rsrc = bk_row + (g + sub) * TP_PART;
    for (unsigned at = (lo_j & ~7u) + 8u * (unsigned)sl; at < hi_j; at += 8u * TPP_LPS) {
      const uint4 c4 = *reinterpret_cast<const uint4*>(csrc + at);
      int rr[8];
      if (pb_j >= 0) {  // the rule: eight 16-bit slice indices in one 16-byte load
        const uint4 t4 = *reinterpret_cast<const uint4*>(r16 + at);
        rr[0] = pb_j + (int)(t4.x & 0xffffu); rr[1] = pb_j + (int)(t4.x >> 16); rr[2] = pb_j + (int)(t4.y & 0xffffu); rr[3] = pb_j + (int)(t4.y >> 16);
        rr[4] = pb_j + (int)(t4.z & 0xffffu); rr[5] = pb_j + (int)(t4.z >> 16); rr[6] = pb_j + (int)(t4.w & 0xffffu); rr[7] = pb_j + (int)(t4.w >> 16);
      } else {
        const int4 r0 = *reinterpret_cast<const int4*>(rsrc + at), r1 = *reinterpret_cast<const int4*>(rsrc + at + 4);
        rr[0] = r0.x; rr[1] = r0.y; rr[2] = r0.z; rr[3] = r0.w; rr[4] = r1.x; rr[5] = r1.y; rr[6] = r1.z; rr[7] = r1.w;
      }
      const unsigned cw[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned t = at + (unsigned)k;
        if (t >= lo_j && t < hi_j) {
          const unsigned c = (k & 1) ? cw[k >> 1] >> 16 : cw[k >> 1] & 0xffffu;
          if (shared_cursors) out_rows[col_ptr[col0 + c] + atomicAdd(&g_cursor[col0 + c], 1)] = rr[k];
          else out_rows[base + atomicAdd(&s_cur[c], 1u)] = rr[k];
        }
      }
    }
  }
}

static void tp_geometry(int64_t nnz, int32_t n_cols, int* bits, int64_t* n_buckets, int64_t* n_parts) {
  *bits = tp_bucket_bits(n_cols);
  *n_buckets = ((int64_t)n_cols + ((int64_t)1 << *bits) - 1) >> *bits;
  *n_parts = (nnz + TP_PART - 1) / TP_PART;
}

// 0: the cursor-atomic kernel serves the matrix (small, or more than TP_MAX_BUCKETS buckets of 2^TP_MAX_BITS columns: beyond 8M columns)
int64_t transpose_scratch_bytes(int64_t n_rows, int64_t nnz, int32_t n_cols) {
  (void)n_rows;
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  if (nnz < PH_MIN_NNZ) return 0;
  int bits;
  int64_t n_buckets, n_parts;
  tp_geometry(nnz, n_cols, &bits, &n_buckets, &n_parts);
  if (n_buckets < 1 || n_buckets > TP_MAX_BUCKETS) return 0;
  return al((n_parts + 1) * 8) + al(n_parts * TP_PART * 2 + 64) + al(n_parts * TP_PART * 2 + 64) + al(n_parts * TP_PART * 4 + 64) + al((n_parts + 1) * 8) +
         al((n_buckets + 1) * n_parts * 2) + al(n_buckets * 8) + al((n_buckets + 1) * 4);
}

hipError_t launch_transpose_partitioned(hipStream_t st, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz, int32_t n_cols,
                                        const int64_t* col_ptr, int32_t* cursor, int32_t* out_row_idx, int32_t col_lo, int32_t col_hi, char* scratch) {
  int bits;
  int64_t n_buckets, n_parts;
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  tp_geometry(nnz, n_cols, &bits, &n_buckets, &n_parts);
  if (n_buckets < 1 || n_buckets > TP_MAX_BUCKETS) return hipErrorInvalidValue;
  int64_t* R = reinterpret_cast<int64_t*>(scratch); scratch += al((n_parts + 1) * 8);
  unsigned short* bk_col = reinterpret_cast<unsigned short*>(scratch); scratch += al(n_parts * TP_PART * 2 + 64);
  unsigned short* bk_row16 = reinterpret_cast<unsigned short*>(scratch); scratch += al(n_parts * TP_PART * 2 + 64);
  int32_t* bk_row = reinterpret_cast<int32_t*>(scratch); scratch += al(n_parts * TP_PART * 4 + 64);  // (only parts whose slice holds > 65536 rows touch it)
  int64_t* part_base = reinterpret_cast<int64_t*>(scratch); scratch += al((n_parts + 1) * 8);
  unsigned short* loc_t = reinterpret_cast<unsigned short*>(scratch); scratch += al((n_buckets + 1) * n_parts * 2);
  long long* weight = reinterpret_cast<long long*>(scratch); scratch += al(n_buckets * 8);
  int32_t* blk_prefix = reinterpret_cast<int32_t*>(scratch);
  const int vec_ok = (reinterpret_cast<uintptr_t>(col_idx) & 15) == 0;
  hipLaunchKernelGGL(tr_parts_kernel, dim3((unsigned)((n_parts + 256) / 256)), dim3(256), 0, st, n_rows, row_ptr, n_parts, R);
  hipLaunchKernelGGL(tp_partition_kernel, dim3((unsigned)n_parts), dim3(TP_THREADS), 0, st, n_rows, row_ptr, col_idx, R, bits, (int)n_buckets, n_parts, col_lo, col_hi, bk_col,
                     bk_row16, bk_row, part_base, loc_t, vec_ok);
  hipError_t we = hipMemsetAsync(weight, 0, sizeof(long long) * (size_t)n_buckets, st);
  if (we != hipSuccess) return we;
  const unsigned wsplit = (unsigned)(n_parts >= 8192 ? 8 : (n_parts >= 1024 ? 4 : 1));
  hipLaunchKernelGGL(pl_weights_kernel, dim3((unsigned)n_buckets, wsplit), dim3(256), 0, st, loc_t, n_parts, reinterpret_cast<unsigned long long*>(weight));
  hipLaunchKernelGGL(tp_blockmap_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, weight, (int)n_buckets, n_parts, blk_prefix);
  const int64_t max_blocks = n_buckets + nnz / TR_CHUNK;
  hipLaunchKernelGGL(tp_place_kernel, dim3((unsigned)max_blocks), dim3(TPP_THREADS), 0, st, bk_col, bk_row16, bk_row, part_base, loc_t, bits, (int)n_buckets, n_parts, blk_prefix, col_ptr, n_cols, cursor,
                     out_row_idx);
  return hipGetLastError();
}

// ============================================================================================
// Row work from a USER shard (multi-GPU input phase): work[i] += d_B(u) for every local user u holding item i.
// Summed over the ranks (all-reduce) this is the same w_i the expand prefix yields, but available before any rank
// holds the whole matrix, so the work-balanced item ranges can be fixed first and every rank transposes only its own
// range.  L2 atomics; after the interaction cut a column receives <= ~max of them.
// ============================================================================================
__global__ __launch_bounds__(256) void row_work_csr_kernel(int64_t n_rows, const int64_t* __restrict__ a_rp, const int32_t* __restrict__ a_ci,
                                                           const int64_t* __restrict__ b_rp, int g_log2, unsigned long long* __restrict__ work) {
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t groups_per_block = 256 >> g_log2;
  for (int64_t r = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> g_log2); r < n_rows;
       r += (int64_t)gridDim.x * groups_per_block) {
    const unsigned long long d = (unsigned long long)(b_rp[r + 1] - b_rp[r]);
    if (d == 0ull) continue;
    const int64_t s = a_rp[r], e = a_rp[r + 1];
    for (int64_t p = s + gl; p < e; p += G) atomicAdd(&work[a_ci[p]], d);
  }
}
hipError_t launch_row_work_csr(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* a_row_ptr, const int32_t* a_col_idx,
                               const int64_t* b_row_ptr, int g_log2, int32_t n_items_a, int64_t* work) {
  hipError_t e = hipMemsetAsync(work, 0, sizeof(int64_t) * (size_t)n_items_a, st);
  if (e != hipSuccess || n_rows == 0) return e;
  const int64_t gpb = 256 >> g_log2;
  int64_t blocks = (n_rows + gpb - 1) / gpb;
  const int64_t cap = (int64_t)n_cu * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(row_work_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, a_row_ptr, a_col_idx, b_row_ptr, g_log2,
                     reinterpret_cast<unsigned long long*>(work));
  return hipGetLastError();
}

// ============================================================================================
// Per-item entropies: rowEntropy / columnEntropy of LogLikelihood.logLikelihoodRatio are functions of the
// item's interaction count and N only, so they are evaluated once per item, not once per cooccurrence.
// ============================================================================================
__global__ __launch_bounds__(256) void item_entropy_kernel(const int32_t* __restrict__ counts, int32_t n, long long n_users,
                                                           double* __restrict__ ent, double* __restrict__ xlx_n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const long long c = counts[i];
    ent[i] = entropy2(c, n_users - c);
  }
  if (i == 0 && xlx_n) *xlx_n = x_log_x(n_users);
}

hipError_t launch_item_entropy(hipStream_t st, const int32_t* counts, int32_t n, long long n_users, double* ent, double* xlx_n) {
  const int blocks = n > 0 ? (n + 255) / 256 : 1;
  hipLaunchKernelGGL(item_entropy_kernel, dim3(blocks), dim3(256), 0, st, counts, n, n_users, ent, xlx_n);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void narrow_counts_kernel(const int32_t* __restrict__ counts, int64_t n, unsigned short* __restrict__ out16,
                                                            int32_t* __restrict__ bad) {
  int over = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = counts[i];
    out16[i] = (unsigned short)c;
    over += (c < 0 || c > 0xffff) ? 1 : 0;
  }
  if (over) atomicAdd(bad, over);
}
hipError_t launch_narrow_counts(hipStream_t st, int n_cu, const int32_t* counts, int64_t n, unsigned short* out16, int32_t* bad) {
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(int32_t), st);
  if (e != hipSuccess || n <= 0) return e;
  int64_t blocks = (n + 255) / 256;
  if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
  hipLaunchKernelGGL(narrow_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, st, counts, n, out16, bad);
  return hipGetLastError();
}

// B' with counts aboard (CcoArgs::b_packed): one streaming pass, four entries per thread and step (one 16-byte load, four count gathers in
// flight, one 16-byte store).  The gathers it makes -- one per ENTRY of B' -- replace one per CANDIDATE of every A'B row: an entry of B' is
// expanded once per item its user holds in A' (~4x on config 4), and here nothing waits on the gather but the store.
// 16-byte non-temporal accesses (the builtins take native vector types, not HIP's int4 class)
typedef int urcco_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 nt_load4(const int32_t* p) {
#ifdef HIPSIM_HOST_BUILD
  return *reinterpret_cast<const int4*>(p);
#else
  const urcco_v4i v = __builtin_nontemporal_load(reinterpret_cast<const urcco_v4i*>(p));
  return make_int4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void nt_store4(int32_t* p, int4 y) {
#ifdef HIPSIM_HOST_BUILD
  *reinterpret_cast<int4*>(p) = y;
#else
  urcco_v4i v;
  v.x = y.x; v.y = y.y; v.z = y.z; v.w = y.w;
  __builtin_nontemporal_store(v, reinterpret_cast<urcco_v4i*>(p));
#endif
}
// cnt: the 16-bit copy of the counts (narrow_counts_kernel: half the table behind the gathers); *bad16 != 0: a count beyond 16 bits -- nothing is packed
__global__ __launch_bounds__(256) void pack_counts_kernel(const int32_t* __restrict__ ci, const int64_t* __restrict__ nnz_dev, int64_t nnz_bound,
                                                          const unsigned short* __restrict__ cnt, const int32_t* __restrict__ bad16, int shift,
                                                          int32_t* __restrict__ out, int32_t* __restrict__ bad, int vec_ok) {
  if (*bad16 != 0) {  // grid-uniform
    if (blockIdx.x == 0 && threadIdx.x == 0) *bad = 1;
    return;
  }
  int64_t nnz = *nnz_dev;
  if (nnz > nnz_bound) nnz = nnz_bound;
  const unsigned limit = 32 - shift >= 16 ? 65536u : (1u << (32 - shift));  // counts must fit the word's spare bits AND the accumulators' 16-bit side arrays
  int n_bad = 0;
  const int64_t nvec = vec_ok ? nnz >> 2 : 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  // two vectors per thread and step: eight count gathers in flight; the streamed words bypass the caches' retention (non-temporal), the count table is
  // what should stay in them
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += 2 * stride) {
    const bool two = v + stride < nvec;
    const int4 x = nt_load4(ci + 4 * v);
    const int4 z = two ? nt_load4(ci + 4 * (v + stride)) : make_int4(0, 0, 0, 0);
    const unsigned c0 = (unsigned)cnt[x.x], c1 = (unsigned)cnt[x.y], c2 = (unsigned)cnt[x.z], c3 = (unsigned)cnt[x.w];
    const unsigned d0 = two ? (unsigned)cnt[z.x] : 0u, d1 = two ? (unsigned)cnt[z.y] : 0u, d2 = two ? (unsigned)cnt[z.z] : 0u, d3 = two ? (unsigned)cnt[z.w] : 0u;
    n_bad += (c0 >= limit) + (c1 >= limit) + (c2 >= limit) + (c3 >= limit) + (d0 >= limit) + (d1 >= limit) + (d2 >= limit) + (d3 >= limit);
    int4 y;
    y.x = (int)((unsigned)x.x | (c0 << shift)); y.y = (int)((unsigned)x.y | (c1 << shift));
    y.z = (int)((unsigned)x.z | (c2 << shift)); y.w = (int)((unsigned)x.w | (c3 << shift));
    nt_store4(out + 4 * v, y);
    if (two) {
      y.x = (int)((unsigned)z.x | (d0 << shift)); y.y = (int)((unsigned)z.y | (d1 << shift));
      y.z = (int)((unsigned)z.z | (d2 << shift)); y.w = (int)((unsigned)z.w | (d3 << shift));
      nt_store4(out + 4 * (v + stride), y);
    }
  }
  for (int64_t e = (nvec << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += stride) {
    const unsigned j = (unsigned)ci[e], c = (unsigned)cnt[j];
    n_bad += c >= limit;
    out[e] = (int)(j | (c << shift));
  }
  if (n_bad) atomicAdd(bad, n_bad);
}
hipError_t launch_pack_counts(hipStream_t st, int n_cu, const int32_t* col_idx, const int64_t* nnz_dev, int64_t nnz_bound, const unsigned short* counts16,
                              const int32_t* bad16, int32_t count_bits, int32_t* out, int32_t* bad) {
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(int32_t), st);
  if (e != hipSuccess || nnz_bound <= 0) return e;
  const int shift = 32 - count_bits;  // the column's bits (count_bits >= 1: shift <= 31)
  int64_t blocks = (nnz_bound / 4 + 255) / 256;
  const int64_t cap = (int64_t)n_cu * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int vec_ok = ((reinterpret_cast<uintptr_t>(col_idx) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  hipLaunchKernelGGL(pack_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, st, col_idx, nnz_dev, nnz_bound, counts16, bad16, shift, out, bad, vec_ok);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void xlx_table_kernel(double* __restrict__ tab) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < XLX_TABLE) tab[x] = x_log_x((long long)x);
}
hipError_t launch_xlx_table(hipStream_t st, double* tab) {
  hipLaunchKernelGGL(xlx_table_kernel, dim3(XLX_TABLE / 256), dim3(256), 0, st, tab);
  return hipGetLastError();
}
// tab[d] = xLogX(n_users - d), d < XLX_TABLE (entries with n_users - d < 0 are never read); behind it
// tab[XLX_TABLE + c] = columnEntropy of a column with c interactions = entropy(c, N - c), evaluated by column_entropy_tab -- the very
// expression the row kernels evaluated per candidate until round 4 (two scattered 8-byte table reads and two subtractions; now one read:
// the CU's address unit, not the arithmetic, is what a candidate's score costs -- profiles/r04_gather_microbench.json)
__global__ __launch_bounds__(256) void xlx_hi_table_kernel(double* __restrict__ tab, const double* __restrict__ xlx_tab, long long n_users) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= XLX_TABLE) return;
  tab[d] = n_users - d >= 0 ? x_log_x(n_users - (long long)d) : 0.0;
  // columnEntropy(c) reads xlx_hi[c] = xLogX(N - c): the value this thread has just produced (x_log_x_hi falls back to the same formula)
  const double hi = n_users - d >= 0 ? x_log_x(n_users - (long long)d) : 0.0;
  tab[XLX_TABLE + d] = d <= n_users ? (x_log_x(n_users) - x_log_x_tab((long long)d, xlx_tab)) - hi : 0.0;
}
hipError_t launch_xlx_hi_table(hipStream_t st, double* tab, const double* xlx_tab, long long n_users) {
  hipLaunchKernelGGL(xlx_hi_table_kernel, dim3(XLX_TABLE / 256), dim3(256), 0, st, tab, xlx_tab, n_users);
  return hipGetLastError();
}

// ============================================================================================
// Expand preparation.  For every entry p of the CSC of A' (user u of some item) it records where u's B' row starts
// and how long it is, then prefix-sums the lengths over the whole CSC:
//     pstart[p] = b_row_ptr[u_p]            wp[p] = sum_{q < p} d_B(u_q)
// One flat, fully parallel gather replaces the per-row pointer chase: inside the SpGEMM the row pointers of B are never
// touched again -- item i's work is the contiguous slice wp[cp[i]] .. wp[cp[i+1]], its upper-bound work
// w_i = wp[cp[i+1]] - wp[cp[i]] (exactly the cooccurrence pairs row i forms) drives binning and work-balanced item
// ranges, and lanes find "their" pairs by searching that slice.
// ============================================================================================
// 32-bit copy of B's row_ptr.  expand_prepare is bound by the fabric traffic of one random row_ptr gather per CSC entry of A'
// (PMC: 415 MB per launch for 4.6M entries); a table of 4 B per user is half as large and stays closer to the L2s.
__global__ __launch_bounds__(256) void narrow_row_ptr_kernel(const int64_t* __restrict__ rp, int64_t n, unsigned* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (unsigned)rp[i];
}

// b_rp32: optional 32-bit copy of b_rp (n_rows_b + 1 entries); used when B holds fewer than 2^32 entries (read on the device)
__global__ __launch_bounds__(256) void expand_prepare_kernel(const int64_t* __restrict__ a_cp, int32_t n_items_a, const int32_t* __restrict__ a_ri,
                                                             const int64_t* __restrict__ b_rp, const unsigned* __restrict__ b_rp32, int64_t n_rows_b,
                                                             int64_t cap, int64_t* __restrict__ pstart, int32_t* __restrict__ plen) {
  const int64_t nnz = a_cp[n_items_a];
  int64_t lim = (nnz / SCAN_TILE + 1) * SCAN_TILE;  // the scan skips tiles that start at or beyond nnz
  if (lim > cap) lim = cap;
  const bool narrow = b_rp32 != nullptr && b_rp[n_rows_b] < ((int64_t)1 << 32);
  // Four grid-stride steps at a time: the four user ids are loaded first, then all eight row_ptr gathers are in flight
  // together (the kernel is a chain of two dependent random loads); every access stays coalesced across the wave.
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t p0 = (int64_t)blockIdx.x * 256 + threadIdx.x; p0 < lim; p0 += stride * 4) {
    int u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t p = p0 + q * stride;
      u[q] = p < nnz ? a_ri[p] : -1;
    }
    int64_t s[4], e[4];
    if (narrow) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s[q] = u[q] >= 0 ? (int64_t)b_rp32[u[q]] : 0;
        e[q] = u[q] >= 0 ? (int64_t)b_rp32[u[q] + 1] : 0;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s[q] = u[q] >= 0 ? b_rp[u[q]] : 0;
        e[q] = u[q] >= 0 ? b_rp[u[q] + 1] : 0;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t p = p0 + q * stride;
      if (p < lim) {
        pstart[p] = s[q];
        plen[p] = (int32_t)(e[q] - s[q]);
      }
    }
  }
}

hipError_t launch_expand_prepare(hipStream_t st, int n_cu, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* a_row_idx,
                                 const int64_t* b_row_ptr, unsigned* b_rp32_scratch, int64_t n_rows_b, int64_t cap, int64_t* pstart, int32_t* plen,
                                 int64_t* wp, int64_t* tile_sums) {
  if (cap > 0) {
    if (b_rp32_scratch) {
      int64_t nb = (n_rows_b + 1 + 255) / 256;
      if (nb > (int64_t)n_cu * 8) nb = (int64_t)n_cu * 8;
      hipLaunchKernelGGL(narrow_row_ptr_kernel, dim3((unsigned)nb), dim3(256), 0, st, b_row_ptr, n_rows_b + 1, b_rp32_scratch);
    }
    int64_t blocks = (cap + 1023) / 1024;
    const int64_t lim = (int64_t)n_cu * 16;
    if (blocks > lim) blocks = lim;
    hipLaunchKernelGGL(expand_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, b_row_ptr,
                       (const unsigned*)b_rp32_scratch, n_rows_b, cap, pstart, plen);
  }
  return launch_scan(st, LoadI32{plen}, cap, wp, tile_sums, a_col_ptr + n_items_a);
}

// --------------------------------------------------------------------------------------------
// Expand preparation for SEVERAL event types at once.  expand_prepare gathers two row_ptr words of B per CSC entry of A' -- one
// scattered 64-byte line per entry and event type, the whole cost of the kernel (0.93 ms per event type on config 4: 40M entries,
// a 40 MB table, fabric-bound).  The secondaries' row pointers are first interleaved per user (32 bits each) so that a CSC entry's single
// gather -- 32 consecutive bytes for four secondaries -- serves every event type.
// --------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) Words4 { unsigned a, b, c, d; };
struct __attribute__((packed, aligned(4))) Words2 { unsigned a, b; };
struct ExpandMultiArgs {
  const int64_t* b_rp[EXPAND_MULTI_MAX];
  int64_t* pstart[EXPAND_MULTI_MAX];
  int32_t* plen[EXPAND_MULTI_MAX];
  int64_t* tsum[EXPAND_MULTI_MAX];  // nullable: the scan-tile sums of plen[d] (launch_expand_scan(..., tile_sums_ready = true))
  int n;
};
// Round 4: the table holds only the STARTS -- T[u][d] = row_ptr_d[u] as 32 bits, u = 0 .. n_rows (the interleaved, narrowed row pointers
// of the secondaries) -- and a length is the next user's start minus this one's: the two records a CSC entry reads are adjacent (32 bytes
// for four secondaries, as before), but the table is HALF the size: 160 MB instead of 320 MB for config 4's 10M users, inside the 256 MiB
// Infinity Cache the gathers otherwise spill from.
__global__ __launch_bounds__(256) void expand_pack_kernel(ExpandMultiArgs a, int64_t n_rows_b, unsigned* __restrict__ T) {
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u <= n_rows_b; u += (int64_t)gridDim.x * 256)
    for (int d = 0; d < a.n; ++d) T[u * a.n + d] = (unsigned)a.b_rp[d][u];
}
template <int N>
__global__ __launch_bounds__(256) void expand_prepare_multi_kernel(const int64_t* __restrict__ a_cp, int32_t n_items_a, const int32_t* __restrict__ a_ri,
                                                                   const unsigned* __restrict__ T, int64_t cap, ExpandMultiArgs a) {
  // A block owns whole SCAN TILES of the CSC entries (round 5): besides pstart / plen it leaves every event type's tile sums of plen
  // (a.tsum[d], when given) -- the first of the three passes of the scans that turn the lengths into the work prefix, which then do
  // not read the lengths a second time.
  __shared__ long long s_part[256 / WAVE][N];
  const int64_t nnz = a_cp[n_items_a];
  int64_t lim = (nnz / SCAN_TILE + 1) * SCAN_TILE;  // the scans skip tiles that start at or beyond nnz
  if (lim > cap) lim = cap;
  const int64_t n_tiles = (lim + SCAN_TILE - 1) / SCAN_TILE;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {  // block-uniform
    const int64_t base = tile * SCAN_TILE;
    long long sum[N];
#pragma unroll
    for (int d = 0; d < N; ++d) sum[d] = 0;
    for (int r = 0; r < SCAN_ITEMS; r += 2) {  // two entries per thread and round: both gathers in flight
      int u[2];
      unsigned v[2][2 * N];  // starts of user u, then of user u + 1: 2 N consecutive words
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t p = base + (int64_t)(r + q) * 256 + threadIdx.x;
        u[q] = p < nnz ? a_ri[p] : -1;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (u[q] >= 0) {
          const unsigned* t = T + (int64_t)u[q] * N;
          if (N == 4) {  // 16-byte aligned: two 16-byte loads
            const uint4 x = *reinterpret_cast<const uint4*>(t), y = *reinterpret_cast<const uint4*>(t + 4);
            v[q][0] = x.x; v[q][1] = x.y; v[q][2] = x.z; v[q][3] = x.w;
            v[q][4 % (2 * N)] = y.x; v[q][5 % (2 * N)] = y.y; v[q][6 % (2 * N)] = y.z; v[q][7 % (2 * N)] = y.w;
          } else {  // 2 N consecutive words, 4-byte aligned: 16-byte loads while they last (global loads only need dword alignment), then 8, then 4
            constexpr int W = 2 * N;
#pragma unroll
            for (int d = 0; d + 4 <= W; d += 4) {
              const Words4 x = *reinterpret_cast<const Words4*>(t + d);
              v[q][d] = x.a; v[q][(d + 1) % W] = x.b; v[q][(d + 2) % W] = x.c; v[q][(d + 3) % W] = x.d;
            }
            if (W % 4 >= 2) {
              const Words2 x = *reinterpret_cast<const Words2*>(t + (W / 4) * 4);
              v[q][(W / 4) * 4 % W] = x.a; v[q][((W / 4) * 4 + 1) % W] = x.b;
            }
            if (W % 2 == 1) v[q][W - 1] = t[W - 1];
          }
        } else {
#pragma unroll
          for (int d = 0; d < 2 * N; ++d) v[q][d] = 0u;
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t p = base + (int64_t)(r + q) * 256 + threadIdx.x;
        if (p < lim) {
#pragma unroll
          for (int d = 0; d < N; ++d) {
            const int32_t len = (int32_t)(v[q][N + d] - v[q][d]);
            a.pstart[d][p] = (int64_t)v[q][d];
            a.plen[d][p] = len;
            sum[d] += (long long)len;
          }
        }
      }
    }
    // the tile's sums: waves by shuffles, the block's four waves through LDS
#pragma unroll
    for (int d = 0; d < N; ++d) {
      unsigned long long x = (unsigned long long)sum[d];
#pragma unroll
      for (int m = 1; m < WAVE; m <<= 1) x += shfl_xor_u64(x, m);
      if ((threadIdx.x & (WAVE - 1)) == 0) s_part[threadIdx.x / WAVE][d] = (long long)x;
    }
    __syncthreads();
    if (threadIdx.x < N && a.tsum[threadIdx.x]) {
      long long tot = 0;
#pragma unroll
      for (int w = 0; w < 256 / WAVE; ++w) tot += s_part[w][threadIdx.x];
      a.tsum[threadIdx.x][tile] = tot;
    }
    __syncthreads();  // s_part is the next tile's
  }
}
// pstart[d][cap], plen[d][cap] for n <= EXPAND_MULTI_MAX event types (every B must hold fewer than 2^32 entries); T: (n_rows_b + 1) * n words of 32 bits
hipError_t launch_expand_prepare_multi(hipStream_t st, int n_cu, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* a_row_idx, int n,
                                       const int64_t* const* b_row_ptr, int64_t n_rows_b, int64_t cap, int64_t* const* pstart, int32_t* const* plen, void* T,
                                       int64_t* const* tsum) {
  if (n < 1 || n > EXPAND_MULTI_MAX) return hipErrorInvalidValue;
  if (cap <= 0) return hipSuccess;
  ExpandMultiArgs a;
  a.n = n;
  for (int d = 0; d < EXPAND_MULTI_MAX; ++d) {
    a.b_rp[d] = d < n ? b_row_ptr[d] : nullptr;
    a.pstart[d] = d < n ? pstart[d] : nullptr;
    a.plen[d] = d < n ? plen[d] : nullptr;
    a.tsum[d] = (d < n && tsum) ? tsum[d] : nullptr;
  }
  int64_t nb = (n_rows_b + 255) / 256;
  if (nb > (int64_t)n_cu * 8) nb = (int64_t)n_cu * 8;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(expand_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, n_rows_b, static_cast<unsigned*>(T));
  int64_t blocks = (cap + SCAN_TILE - 1) / SCAN_TILE;  // a block owns whole scan tiles
  const int64_t lim = (int64_t)n_cu * 16;
  if (blocks > lim) blocks = lim;
  const unsigned* Tc = static_cast<const unsigned*>(T);
  switch (n) {
    case 1: hipLaunchKernelGGL(expand_prepare_multi_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
    case 2: hipLaunchKernelGGL(expand_prepare_multi_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
    case 3: hipLaunchKernelGGL(expand_prepare_multi_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
    case 4: hipLaunchKernelGGL(expand_prepare_multi_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
    case 5: hipLaunchKernelGGL(expand_prepare_multi_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
    case 6: hipLaunchKernelGGL(expand_prepare_multi_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
    case 7: hipLaunchKernelGGL(expand_prepare_multi_kernel<7>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
    default: hipLaunchKernelGGL(expand_prepare_multi_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a_col_ptr, n_items_a, a_row_idx, Tc, cap, a); break;
  }
  return hipGetLastError();
}
// wp = exclusive prefix of plen over cap entries (the second half of launch_expand_prepare, for lengths produced by the multi form)
// tile_sums_ready: tile_sums already holds the sums of plen's scan tiles (expand_prepare_multi left them): the reduce pass is skipped
hipError_t launch_expand_scan(hipStream_t st, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* plen, int64_t cap, int64_t* wp, int64_t* tile_sums,
                              bool tile_sums_ready) {
  return launch_scan(st, LoadI32{plen}, cap, wp, tile_sums, a_col_ptr + n_items_a, tile_sums_ready);
}

__global__ __launch_bounds__(256) void row_work_kernel(int32_t item_lo, int32_t item_hi, const int64_t* __restrict__ a_cp,
                                                       const int64_t* __restrict__ wp, int64_t* __restrict__ work) {
  const int64_t n = (int64_t)item_hi - item_lo;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int64_t i = item_lo + t;
    work[t] = wp[a_cp[i + 1]] - wp[a_cp[i]];
  }
}

hipError_t launch_row_work(hipStream_t st, int n_cu, int32_t item_lo, int32_t item_hi, const int64_t* a_col_ptr, const int64_t* wp, int64_t* work) {
  const int64_t n = (int64_t)item_hi - item_lo;
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)n_cu * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(row_work_kernel, dim3((unsigned)blocks), dim3(256), 0, st, item_lo, item_hi, a_col_ptr, wp, work);
  return hipGetLastError();
}

// ============================================================================================
// Binning (row-tile partitioning of the SpGEMM).  A row goes to the smallest accumulator class that
// (a) is guaranteed to hold its distinct columns AND their 64-bit LLR keys: 3 w < table words, or 3 n_cols_b < table
//     words (then slots are addressed by column and never collide), and packed counts cannot overflow;
// (b) gives it enough lanes: <= 64 pairs and users -> the micro kernel (one pair per lane), <= 512 pairs -> one wave,
//     <= 8192 -> 256 threads (with a 4096- or 8192-word table: the smaller one lets more rows share a CU), else 1024 threads.
// Lists are built by a deterministic tile count / scan / scatter (a global atomic append would serialise
// hundreds of thousands of increments on four addresses).
// ============================================================================================
constexpr int E0 = 1024, E1S = 4096, E1 = 8192, E2S = 16384, E2 = 32768;  // LDS table words: wave / small block / block / half CU / CU
constexpr int MP_KMAX = 256;  // largest k the multi-pass class keeps its running lists for (MP_KMAX_HOST in cco_kernels.h)

// Round 6 layout of an accumulator table of E words (team of T threads).  Insert phase: table_slots(E, T) slots -- two thirds of the table, a multiple of
// T -- of packed (column + 1, count) in words [0, SH), the slots' 16-bit column counts cB (left by the claiming lane: they rode in on the B' words) in the
// SH / 2 words behind them (a pair of slots is one 8-byte access for the zeroing and the compaction sweep, its two counts one word).  Compaction: the D packed words to [0, D) and every candidate's cB into its slot of the KEY array behind them, which the score
// phase reads and then overwrites with the candidate's key: no word more than rounds 1-5 needed (3 D + 3 k + 2 <= E), a third fewer slots.
__host__ __device__ constexpr int table_slots(int E, int T) { return (2 * E / 3) / (2 * T) * (2 * T); }  // (a thread sweeps PAIRS of slots: 8-byte LDS accesses)
constexpr int URCCO_WB1 = 512;
constexpr int URCCO_WB2 = 8192;
__device__ __forceinline__ int choose_bin(long long w, long long ca, int32_t n_cols_b, int32_t count_bits, int32_t k) {
  if (ca <= 0 || w <= 0) return -1;  // no users or no pairs: empty indicator row
  if (count_bits < 31 && ca > ((1ll << count_bits) - 1)) return NBINS - 1;
  if (w <= 64 && ca <= 64) return 0;  // micro: one pair per lane
  int cap_bin = NBINS - 1;
  // a table of E words must hold D packed counts + D 64-bit keys + the k selected (key, col): 3 D + 3 k + 1 <= E,
  // with D <= min(w, n_cols_b)
  const long long dmax = (w < (long long)n_cols_b ? w : (long long)n_cols_b) * 3 + (long long)k * 3 + 2;
  if (dmax <= E0) cap_bin = 1;
  else if (dmax <= E1S) cap_bin = 2;
  else if (dmax <= E1) cap_bin = 3;
  else if (dmax <= E2S) cap_bin = 4;
  else if (dmax <= E2) cap_bin = 5;
  const int work_bin = w <= URCCO_WB1 ? 1 : (w <= URCCO_WB2 ? 2 : 4);  // long rows want more lanes even when a small table would hold them
  return cap_bin > work_bin ? cap_bin : work_bin;
}

// Round 6: the micro class in three sub-lists by row size -- rows of <= 16 pairs and users share a wave four at a time, rows of <= 32 two
// at a time (see cco_rows_micro_kernel).  The binning works on INTERNAL bins (0, 1, 2 = the sub-lists, 3 .. = the other classes); towards
// everything else the micro class stays one bin: bin_off[0 .. NBINS] as before, the sub-lists' starts behind it (bin_off[NBINS + 1], [NBINS + 2]).
constexpr int MICRO_SUBS = 3;
constexpr int IBINS = NBINS - 1 + MICRO_SUBS;
__device__ __forceinline__ int micro_sub(long long w, long long ca) { return (w <= 16 && ca <= 16) ? 0 : ((w <= 32 && ca <= 32) ? 1 : 2); }
__device__ __forceinline__ int internal_bin(int b, long long w, long long ca, int split) {
  return b < 0 ? -1 : (b == 0 ? (split ? micro_sub(w, ca) : MICRO_SUBS - 1) : b + MICRO_SUBS - 1);
}
// The micro class is split into its sub-lists only when the build has enough item rows for three launches to pay: a rank of a sharded build (an eighth
// of config 4's rows) would start 16K waves per sub-list for a handful of passes each (emulated 8-rank build: 3.45 against 3.17 ms of SpGEMM per rank).
// URCCO_MICRO_SPLIT_ROWS (read once) moves the threshold.
static bool micro_split_for(int32_t n_rows) {
  static const long long min_rows = [] { const char* e = getenv("URCCO_MICRO_SPLIT_ROWS"); return e && *e ? atoll(e) : 1000000ll; }();
  return (long long)n_rows >= min_rows;
}
__device__ __forceinline__ int internal_start(const int32_t* __restrict__ bin_off, int ib) {
  return ib == 0 ? bin_off[0] : (ib < MICRO_SUBS ? bin_off[NBINS + ib] : bin_off[ib - (MICRO_SUBS - 1)]);
}
constexpr int BIN_THREADS = 256;
constexpr int BIN_ITEMS = BIN_TILE / BIN_THREADS;  // 4
constexpr int BIN_COLS = IBINS + 2 * NBINS + 1;    // per tile: rows per INTERNAL bin, pairs per bin, users per bin, total pairs
static_assert(BIN_COLS == BIN_COLS_HOST && BIN_OFF_LEN == NBINS + MICRO_SUBS, "scratch sizes of the callers");

__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(int32_t item_lo, int32_t n, const int64_t* __restrict__ work,
                                                                const int32_t* __restrict__ cnt_a, int32_t n_cols_b, int32_t count_bits, int32_t k,
                                                                int64_t* __restrict__ tile_counts, int split) {
  __shared__ long long s_acc[BIN_COLS];
  if (threadIdx.x < BIN_COLS) s_acc[threadIdx.x] = 0;
  __syncthreads();
  int c[IBINS];
  long long pw[NBINS], pu[NBINS];
#pragma unroll
  for (int k = 0; k < IBINS; ++k) c[k] = 0;
#pragma unroll
  for (int k = 0; k < NBINS; ++k) { pw[k] = 0; pu[k] = 0; }
  long long pairs = 0;
#pragma unroll
  for (int q = 0; q < BIN_ITEMS; ++q) {
    const int64_t t = (int64_t)blockIdx.x * BIN_TILE + (int64_t)threadIdx.x * BIN_ITEMS + q;
    if (t < n) {
      const long long w = work[t];
      const long long ca = cnt_a[item_lo + t];
      pairs += w;
      const int b = choose_bin(w, ca, n_cols_b, count_bits, k);
      const int ib = internal_bin(b, w, ca, split);
#pragma unroll
      for (int k = 0; k < IBINS; ++k) c[k] += (ib == k);
#pragma unroll
      for (int k = 0; k < NBINS; ++k) {
        pw[k] += (b == k) ? w : 0;
        pu[k] += (b == k) ? ca : 0;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < IBINS; ++k)
    if (c[k]) atomicAdd((unsigned long long*)&s_acc[k], (unsigned long long)c[k]);
#pragma unroll
  for (int k = 0; k < NBINS; ++k)
    if (pu[k]) {
      atomicAdd((unsigned long long*)&s_acc[IBINS + k], (unsigned long long)pw[k]);
      atomicAdd((unsigned long long*)&s_acc[IBINS + NBINS + k], (unsigned long long)pu[k]);
    }
  if (pairs) atomicAdd((unsigned long long*)&s_acc[IBINS + 2 * NBINS], (unsigned long long)pairs);
  __syncthreads();
  if (threadIdx.x < BIN_COLS) tile_counts[(int64_t)blockIdx.x * BIN_COLS + threadIdx.x] = s_acc[threadIdx.x];
}

// single block: per-column exclusive scan over the tiles (in place), totals -> bin_off / stats.  One wave per column of
// the tile table (rows per internal bin, pairs / users per bin, total pairs), 16 columns at a time.
constexpr int BS_THREADS = 1024;
__global__ __launch_bounds__(BS_THREADS) void bin_scan_kernel(int64_t* __restrict__ tile_counts, int64_t n_tiles, int32_t* __restrict__ bin_off,
                                                              int64_t* __restrict__ stats) {
  __shared__ long long s_tot[BIN_COLS];
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  for (int k = wave; k < BIN_COLS; k += BS_THREADS / WAVE) {  // wave-uniform
    if (k >= IBINS) {
      // pairs / users / total columns: only their TOTALS are used (statistics) -- a plain sum, every lane four loads deep, one reduction at the
      // end (rounds 1-4 ran the same carried shuffle scan over all 22 columns and wrote 15 prefixes nobody read: 54 us per event type)
      long long acc = 0;
      for (int64_t base = 0; base < n_tiles; base += 4 * WAVE) {
        long long v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t i = base + q * WAVE + lane;
          v[q] = i < n_tiles ? tile_counts[i * BIN_COLS + k] : 0;
        }
        acc += (v[0] + v[1]) + (v[2] + v[3]);
      }
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const long long o = shfl_up_i64(acc, d);
        if (lane >= d) acc += o;
      }
      acc = shfl_i64(acc, WAVE - 1);
      if (lane == 0) s_tot[k] = acc;
      continue;
    }
    long long carry = 0;
    for (int64_t base = 0; base < n_tiles; base += WAVE) {  // rows-per-bin columns: exclusive prefix over the tiles = where a tile's rows go
      const int64_t i = base + lane;
      const long long v = i < n_tiles ? tile_counts[i * BIN_COLS + k] : 0;
      long long inc = v;
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const long long o = shfl_up_i64(inc, d);
        if (lane >= d) inc += o;
      }
      if (i < n_tiles) tile_counts[i * BIN_COLS + k] = carry + inc - v;
      carry += shfl_i64(inc, WAVE - 1);
    }
    if (lane == 0) s_tot[k] = carry;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t off = 0;
    for (int ib = 0; ib < IBINS; ++ib) {  // list order = internal bin order: the micro sub-lists, then the other classes
      if (ib == 0) bin_off[0] = off;
      else if (ib < MICRO_SUBS) bin_off[NBINS + ib] = off;
      else bin_off[ib - (MICRO_SUBS - 1)] = off;
      off += (int32_t)s_tot[ib];
    }
    bin_off[NBINS] = off;
    if (stats) {
      for (int k = 0; k < NBINS; ++k) {
        long long rows = 0;
        if (k == 0) for (int q = 0; q < MICRO_SUBS; ++q) rows += s_tot[q];
        else rows = s_tot[k + MICRO_SUBS - 1];
        stats[1 + k] = rows;                                      // rows
        stats[1 + NBINS + k] = s_tot[IBINS + k];                  // pairs
        stats[1 + 2 * NBINS + k] = s_tot[IBINS + NBINS + k];      // users (sum of cA over the bin's rows)
      }
      stats[0] = s_tot[IBINS + 2 * NBINS];
    }
  }
}

__global__ __launch_bounds__(BIN_THREADS) void bin_scatter_kernel(int32_t item_lo, int32_t n, const int64_t* __restrict__ work,
                                                                  const int32_t* __restrict__ cnt_a, int32_t n_cols_b, int32_t count_bits, int32_t k,
                                                                  const int64_t* __restrict__ tile_counts, const int32_t* __restrict__ bin_off,
                                                                  int32_t* __restrict__ bin_rows, int split) {
  __shared__ long long s_wave[SCAN_THREADS / WAVE];
  int b[BIN_ITEMS];
#pragma unroll
  for (int q = 0; q < BIN_ITEMS; ++q) {
    const int64_t t = (int64_t)blockIdx.x * BIN_TILE + (int64_t)threadIdx.x * BIN_ITEMS + q;
    b[q] = -1;
    if (t < n) {
      const long long w = work[t], ca = cnt_a[item_lo + t];
      b[q] = internal_bin(choose_bin(w, ca, n_cols_b, count_bits, k), w, ca, split);
    }
  }
  for (int k = 0; k < IBINS; ++k) {  // block-uniform: one block scan per internal bin
    int c = 0;
#pragma unroll
    for (int q = 0; q < BIN_ITEMS; ++q) c += (b[q] == k);
    long long tot;
    long long pos = block_exclusive_scan(c, s_wave, &tot) + tile_counts[(int64_t)blockIdx.x * BIN_COLS + k] + internal_start(bin_off, k);
#pragma unroll
    for (int q = 0; q < BIN_ITEMS; ++q)
      if (b[q] == k) bin_rows[pos++] = item_lo + (int32_t)((int64_t)blockIdx.x * BIN_TILE + (int64_t)threadIdx.x * BIN_ITEMS + q);
  }
}

hipError_t launch_binning(hipStream_t st, int32_t item_lo, int32_t n, const int64_t* work, const int32_t* cnt_a, int32_t n_cols_b,
                          int32_t count_bits, int32_t k, int64_t* tile_counts, int32_t* bin_off, int32_t* bin_rows, int64_t* stats) {
  if (n <= 0) {
    hipError_t e = hipMemsetAsync(bin_off, 0, sizeof(int32_t) * BIN_OFF_LEN, st);
    if (e == hipSuccess && stats) e = hipMemsetAsync(stats, 0, sizeof(int64_t) * STATS_LEN, st);
    return e;
  }
  const int64_t n_tiles = ((int64_t)n + BIN_TILE - 1) / BIN_TILE;
  const int split = micro_split_for(n) ? 1 : 0;
  hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)n_tiles), dim3(BIN_THREADS), 0, st, item_lo, n, work, cnt_a, n_cols_b, count_bits, k, tile_counts, split);
  hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(BS_THREADS), 0, st, tile_counts, n_tiles, bin_off, stats);
  hipLaunchKernelGGL(bin_scatter_kernel, dim3((unsigned)n_tiles), dim3(BIN_THREADS), 0, st, item_lo, n, work, cnt_a, n_cols_b, count_bits, k,
                     tile_counts, bin_off, bin_rows, split);
  return hipGetLastError();
}

// ============================================================================================
// K4+K5  A'B rows (Gustavson over rows of A') with LDS hash accumulators, fused LLR + top-k.
//
// A team of T threads (one wave, 256 or 1024 threads) owns one item row i at a time:
//   1. zero its table of E packed 32-bit entries  ((col+1) << count_bits) | count
//   2. EXPAND: row i's work is the slice wp[cp[i]] .. wp[cp[i+1]] of the prepared prefix (see expand_prepare).  Users are
//      taken T at a time (coalesced reads of pstart / wp into LDS); the chunk's pairs are dealt out evenly, each lane
//      binary-searches the LDS prefix once for its first pair and then walks B' column indices -- every lane busy, all
//      gathers of a chunk in flight together -- inserting each column: relaxed LDS read, CAS to claim an empty slot,
//      LDS atomic add to count
//   3. COMPACT: occupied slots are packed to the front of the table (registers -> scan -> same LDS), so that
//   4. SCORE runs dense: candidate t gets k11 = count, LLR from the per-item entropies + 4 logs (fp64); self pairs
//      (A'A), zeros and llr < minLLR are dropped; keys and columns stay in registers
//   5. TOP-K: one wave with <= 64 candidates ranks them by counting (shuffle broadcast) and writes each straight to its
//      output position; otherwise repeated argmax over (llr desc, col asc) -- wave shuffles + one LDS hop for T > 64.
// Counts never leave the CU.  Hash = Fibonacci multiplicative; when the table covers all of B's columns slots are
// addressed by column (no probing).  One-wave teams synchronise with wave-level barriers only, so the four teams of a
// block run independent row loops.
// ============================================================================================
struct Best {
  unsigned long long key;  // llr bits (positive doubles order like unsigned integers); 0 = none
  int col;
};
__device__ __forceinline__ bool best_before(unsigned long long ka, int ca, unsigned long long kb, int cb) {
  return ka > kb || (ka == kb && ca < cb);
}

// Rank of (mk, mc) among keys[0 .. n) (columns through col_of) in the order key desc, column asc, by counting.  `n` must be
// wave-uniform and is moved to a scalar register: the trip count, the element index and the LDS offsets then live in the
// scalar unit and the loop unrolls with immediate offsets -- with a lane-valued trip count half of the vector instructions of
// this loop were bookkeeping (three address/counter increments and an exec-mask test per element), and the ranking loops were
// ~45 % of the VALU instructions of a typical one-wave row.  Equal keys are COMMON (an LLR is a function of four small integer
// counts), so the column comparison cannot be left to a rare path (measured: a key-only loop with a second pass for lanes that
// saw their key twice was 8 % slower than the plain loop).
// The table-only evaluation of a candidate's LLR as a WAVE-level decision (round 5).  After the interaction cut every operand of a
// candidate's LLR is small -- k11, k12 = cA - k11, k21 = cB - k11, cB and N - k22 = cA + cB - k11 all sit below the table size -- so
// columnEntropy and the four xLogX terms of matrixEntropy are five table reads.  llr_operands_in_tables is ONE range test over all of
// them; when it holds for every candidate of the wave, llr_from_tables issues the five reads together -- the same five values in the
// same expression order as the general form (llr_of), bit for bit -- in straight-line code; otherwise the wave takes the general form.
// (A per-lane single-check form with one rolled logarithm behind it was slower: profiles/r05_llr_rank_variants_ab.log.)
// k21 is part of the test: the context level guarantees k11 <= cB (post-sampling counts of the same B'), a caller of
// urcco_dev_cco_rows with inconsistent counts_b does not -- cB - k11 would wrap and index ~32 GiB past the table.
__device__ __forceinline__ bool llr_operands_in_tables(unsigned k11, long long ca, unsigned cb, long long n_users, const double* col_ent) {
  const long long k12 = ca - (long long)k11, k21 = (long long)cb - (long long)k11, d22 = ca + k21;  // k22 = n_users - d22
  return col_ent != nullptr && (unsigned long long)(k12 | k21 | (long long)cb | d22) < (unsigned long long)XLX_TABLE && d22 <= n_users;
}
__device__ __forceinline__ double llr_from_tables(double row_entropy, double xlx_n, unsigned k11, unsigned ca, unsigned cb, const double* __restrict__ xlx_tab,
                                                  const double* __restrict__ xlx_hi, const double* __restrict__ col_ent) {
  double t11 = xlx_tab[k11], t12 = xlx_tab[ca - k11], t21 = xlx_tab[cb - k11], t22 = xlx_hi[ca + cb - k11], tce = col_ent[cb];
#ifndef HIPSIM_HOST_BUILD
  asm volatile("" : "+v"(t11), "+v"(t12), "+v"(t21), "+v"(t22), "+v"(tce));  // all five in flight before the first is consumed
#endif
  const double matrix_entropy = (((xlx_n - t11) - t12) - t21) - t22;
  const double s = row_entropy + tce;
  if (s < matrix_entropy) return 0.0; /* round off error */
  return 2.0 * (s - matrix_entropy);
}
// the general form: every operand tested on its own, a logarithm behind every table miss (taken by the waves that hold a candidate outside the tables)
__device__ __forceinline__ double llr_of(double row_entropy, double xlx_n, long long k11, long long ca, long long cb, long long n_users,
                                         const double* __restrict__ xlx_tab, const double* __restrict__ xlx_hi, const double* __restrict__ col_ent) {
  return llr_from_entropies_tab(row_entropy, column_entropy_of(cb, xlx_n, n_users, xlx_tab, xlx_hi, col_ent), xlx_n, k11, ca - k11, cb - k11, n_users - ca - cb + k11, xlx_tab,
                                n_users, xlx_hi);
}

// r += [(ka, ca) sorts before (mk, mc)] -- key desc, column asc -- as the final borrow of a three-word subtraction chain: [ca < mc] enters
// (mk - ka) as its borrow, so the chain ends in [mk < ka] || ([mk == ka] && [ca < mc]).  Three subtract-with-borrow and one add-with-carry
// per element and NO scalar instruction.  The comparison form the compiler makes of best_before -- two 64-bit compares, a 32-bit compare,
// an s_and and an s_or per element, then the add-with-carry -- kept the CU's one scalar unit as busy as its four vector units: ~950 scalar
// against ~1180 vector instructions per row of the one-wave class (profiles/r04_sq_counters_pmc_config4.json), half of them in these loops.
// (__builtin_subc chains are taken apart into the same compares by the compiler: inline assembly it is.)
__device__ __forceinline__ void count_if_before(unsigned& r, unsigned long long ka, unsigned ca, unsigned long long mk, unsigned mc) {
#ifdef HIPSIM_HOST_BUILD
  r += best_before(ka, (int)ca, mk, (int)mc) ? 1u : 0u;
#else
  unsigned t;
  asm("v_sub_co_u32 %1, vcc, %2, %3\n\t"
      "v_subb_co_u32 %1, vcc, %4, %5, vcc\n\t"
      "v_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
      "v_addc_co_u32 %0, vcc, 0, %0, vcc"
      : "+v"(r), "=&v"(t)
      : "v"(ca), "v"(mc), "v"((unsigned)mk), "v"((unsigned)ka), "v"((unsigned)(mk >> 32)), "v"((unsigned)(ka >> 32))
      : "vcc");
#endif
}
template <class ColOf>
__device__ __forceinline__ unsigned rank_by_counting(const unsigned long long* keys, unsigned n_uniform, unsigned long long mk, int mc, ColOf col_of) {
  const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)n_uniform);
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, u = 0;  // four chains
  for (; u + 4 <= n; u += 4) {
    count_if_before(r0, keys[u], (unsigned)col_of(u), mk, (unsigned)mc);
    count_if_before(r1, keys[u + 1], (unsigned)col_of(u + 1), mk, (unsigned)mc);
    count_if_before(r2, keys[u + 2], (unsigned)col_of(u + 2), mk, (unsigned)mc);
    count_if_before(r3, keys[u + 3], (unsigned)col_of(u + 3), mk, (unsigned)mc);
  }
  for (; u < n; ++u) count_if_before(r0, keys[u], (unsigned)col_of(u), mk, (unsigned)mc);
  return (r0 + r1) + (r2 + r3);
}

// Claim-first insert into the packed open-addressing table: one CAS per probe (a new column costs one LDS round trip, a
// known one two), a single rolled loop with one exit (an unrolled probe loop compiles to more exec-mask bookkeeping than
// useful work; measured -4 % on the one-wave class against load-then-CAS).  Returns false only if every slot was probed
// without finding the key or a free slot -- impossible while the binning rule holds (the table always has room for the
// row's distinct columns); the bound keeps a broken invariant from turning into a hung GPU and is reported through
// stats[1 + 4 * NBINS].
// Round 6: NS slots, not a power of two (two thirds of the table: see table_slots) -- the multiplicative hash is reduced to [0, NS) by a
// multiply-high, the probe sequence wraps by a compare -- and the lane that CLAIMS a slot leaves the column's count (it rode in on the B' word) in the
// slot's entry of the 16-bit side array cbv.
template <int NS>
__device__ __forceinline__ bool tab_insert(unsigned* tab, unsigned key, int count_bits, bool ident, unsigned short* cbv, unsigned c_b) {
  unsigned h = ident ? (key - 1u) : __umulhi(key * 0x9E3779B1u, (unsigned)NS);
  const unsigned fresh = (key << count_bits) | 1u;
  // ONE loop condition and no break: with two exits and a result flag the compiler spent ~25 scalar instructions per probe on execution
  // masks (round 5, ISA of the pair loop: the CU's single scalar unit was as loaded as its four vector units).  `left` bounds the probes
  // (a broken binning invariant must not hang the GPU); the add for a known column is predicated, not branched around.
  bool done;
  unsigned left = (unsigned)NS;
#pragma unroll 1
  do {
    const unsigned v = atomicCAS(&tab[h], 0u, fresh);
    const bool hit = (v >> count_bits) == key;
    if (hit) atomicAdd(&tab[h], 1u);
    if (v == 0u) cbv[h] = (unsigned short)c_b;
    done = hit || v == 0u;
    ++h;
    h = h == (unsigned)NS ? 0u : h;
    --left;
  } while (!done && left != 0u);
  return done;
}

// tab_insert for the micro class: the lane whose CAS finds the slot EMPTY owns the new column, and is told which slot that is
// (0xffffffff: the column was known, or -- impossible while the binning rule holds -- no slot was found: *ok false).
__device__ __forceinline__ unsigned tab_insert_claim(unsigned* tab, unsigned key, int count_bits, unsigned mask, int hshift, bool ident, bool* ok) {
  unsigned h = ident ? (key - 1u) : ((key * 0x9E3779B1u) >> hshift);
  const unsigned fresh = (key << count_bits) | 1u;
  unsigned mine = 0xffffffffu;
  bool done;
  unsigned left = mask + 1u;
#pragma unroll 1
  do {
    const unsigned v = atomicCAS(&tab[h], 0u, fresh);
    const bool hit = (v >> count_bits) == key;
    if (hit) atomicAdd(&tab[h], 1u);
    mine = v == 0u ? h : mine;
    done = hit || v == 0u;
    h = (h + 1u) & mask;
    --left;
  } while (!done && left != 0u);
  *ok = done;
  return mine;
}
// rank_by_counting with the elements dealt out to R replicas of the candidates: this lane counts elements first, first + R, ... of
// keys / cols [0, n_pad) -- n_pad a multiple of R (wave-uniform), the padding filled with (key 0, column 0xffffffff), which sorts before
// nothing -- and the caller adds the replicas' counts.  R compile-time: the element offsets are immediates of the LDS reads.
template <int R>
__device__ __forceinline__ unsigned rank_by_counting_strided(const unsigned long long* keys, const unsigned* cols, unsigned first, unsigned n_pad_uniform,
                                                             unsigned long long mk, unsigned mc) {
  const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)n_pad_uniform);
  const unsigned long long* kp = keys + first;
  const unsigned* cp = cols + first;
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, u = 0;  // four chains
  for (; u + 4 * R <= n; u += 4 * R) {
    count_if_before(r0, kp[u], cp[u], mk, mc);
    count_if_before(r1, kp[u + R], cp[u + R], mk, mc);
    count_if_before(r2, kp[u + 2 * R], cp[u + 2 * R], mk, mc);
    count_if_before(r3, kp[u + 3 * R], cp[u + 3 * R], mk, mc);
  }
  for (; u < n; u += R) count_if_before(r0, kp[u], cp[u], mk, mc);
  return (r0 + r1) + (r2 + r3);
}

// "This prefetched register is needed now": an empty asm that reads it makes the compiler place the wait for its load HERE -- ahead
// of the stores that follow -- instead of at its first use in the next row, where the wait would also cover every store issued in
// between (the memory counter retires in order) and so expose the write latency of the row's output at the top of the next row.
// "Recompute what derives from this where it is used": an empty asm that redefines a loop-invariant value inside the loop keeps the compiler
// from hoisting everything computed from it (per-lane LDS addresses of paths only some rows take) into registers that then live -- or spill
// to scratch, and a scratch reload is a memory load the in-order counter waits on -- across the whole row loop.
#ifdef HIPSIM_HOST_BUILD
#define URCCO_OPAQUE(x) ((void)(x))
#else
#define URCCO_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
#ifdef HIPSIM_HOST_BUILD
#define URCCO_SETTLE(x) ((void)(x))
#else
#define URCCO_SETTLE(x) asm volatile("" : "+v"(x) : : "memory")  // "memory": the stores that follow must not be scheduled above it
#endif

// "This kernel argument gets scalar registers of its own": the kernel arguments arrive as 16-dword tuples, the register allocator spills
// and reloads a tuple as a whole, and a wave at eight waves per SIMD has 78 scalar registers for ~90 dwords of arguments -- so the pair
// loop of the one-wave class reloaded SIXTEEN spilled scalars (v_readlane each) per cooccurrence pair to get at the ONE pointer it uses.
// An empty asm that redefines the value cuts it out of its tuple: a pair of its own, spilled -- if at all -- as a pair.
// (A pointer that went through the asm has lost its provenance -- the compiler would address it with FLAT instructions, which also tie up
// the LDS counter --, so pointers make the trip as GLOBAL-address-space pointers: URCCO_OWN_GLOBAL_PTR.)
#ifdef HIPSIM_HOST_BUILD
#define URCCO_OWN_SGPRS(x) ((void)(x))
#define URCCO_OWN_GLOBAL_PTR(T, name, src) T* name = (src)
#else
#define URCCO_OWN_SGPRS(x) asm volatile("" : "+s"(x))
#define URCCO_OWN_GLOBAL_PTR(T, name, src)                                                    \
  T __attribute__((address_space(1)))* name##_as1 = (T __attribute__((address_space(1)))*)(src); \
  asm volatile("" : "+s"(name##_as1));                                                         \
  T* name = (T*)name##_as1
#endif

// LDS hand-off inside ONE wave: DS operations of a wave execute in program order, so a compiler-level fence is all that
// is needed between a lane's write and another lane's read.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int T>
__device__ __forceinline__ void team_sync() {
  if (T == WAVE) wave_sync(); else __syncthreads();
}

// Inclusive prefix sum over the 64 lanes of a wave, entirely in the VALU: four DPP row shifts inside the rows of 16 lanes,
// then the two row broadcasts that carry the row totals upwards.  (A __shfl_up ladder is six dependent ds_bpermute round
// trips through the LDS pipe, each with its own lane-bound bookkeeping; a row of the SpGEMM ran ~5 such ladders.)  Every lane
// of the wave must be active.  One DPP per source line: the test simulator keys wave operations by line.
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return (unsigned)x;
}
// Inclusive prefix MAXIMUM over the 64 lanes (same ladder as wave_inclusive_sum; identity 0).  Every lane must be active.
__device__ __forceinline__ unsigned wave_inclusive_max(unsigned v) {
  int x = (int)v;
  int y;
  y = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  x = (unsigned)y > (unsigned)x ? y : x;
  return (unsigned)x;
}
// value of lane src (per-lane src in [0, 64)): one ds_bpermute, no LDS memory.  Every lane must be active.
__device__ __forceinline__ unsigned wave_gather(unsigned v, unsigned src) { return (unsigned)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }
__device__ __forceinline__ int64_t wave_gather64(int64_t v, unsigned src) {
  const unsigned lo = wave_gather((unsigned)(unsigned long long)v, src);
  const unsigned hi = wave_gather((unsigned)((unsigned long long)v >> 32), src);
  return (int64_t)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
// OR over the 64 lanes of a wave, in the VALU (the inclusive ladder of wave_inclusive_sum; lane 63 ends up with everything), returned as a
// wave-uniform value.  Every lane must be active.  (The butterfly of __shfl_xor it replaces was 6 x 2 ds_bpermute per 64-bit word.)
__device__ __forceinline__ unsigned wave_or(unsigned v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane(x, WAVE - 1);
}
// AND / OR of a 64-bit value over the wave (AND as the complement of the OR of the complements), wave-uniform results
__device__ __forceinline__ void wave_and_or_u64(unsigned long long& kand, unsigned long long& kor) {
  const unsigned nal = wave_or(~(unsigned)kand);
  const unsigned nah = wave_or(~(unsigned)(kand >> 32));
  const unsigned orl = wave_or((unsigned)kor);
  const unsigned orh = wave_or((unsigned)(kor >> 32));
  kand = ~(((unsigned long long)nah << 32) | (unsigned long long)nal);
  kor = ((unsigned long long)orh << 32) | (unsigned long long)orl;
}
// A value every lane of the wave agrees on, moved to a scalar register.  The compiler cannot tell that threadIdx.x / T, or
// anything loaded through it (the row id, its CSC bounds, the chunk's work bounds, counts read back from LDS), is uniform, and
// keeps all arithmetic, addressing and loop control that derives from it in the vector unit -- where every instruction costs a
// wave four issue cycles and the SpGEMM classes are bound by exactly that.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int64_t uni(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (int64_t)lo;
}
// value of lane l (wave-uniform l): one v_readlane, no LDS
__device__ __forceinline__ unsigned wave_read_lane(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
// 64-bit value of lane l (wave-uniform l).  One wave operation per source line: the test simulator keys them by line.
__device__ __forceinline__ int64_t wave_read_lane64(int64_t v, int l) {
  const unsigned lo = wave_read_lane((unsigned)(unsigned long long)v, l);
  const unsigned hi = wave_read_lane((unsigned)((unsigned long long)v >> 32), l);
  return (int64_t)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
// number of set bits of m below this lane
__device__ __forceinline__ unsigned lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// List positions for the lanes that `want` one: the wave's lanes are numbered by a ballot and the wave takes its block of the team's list
// with ONE LDS atomic (T > 64) or none at all (T == 64: the list is the wave's own and its length lives in `wave_count`).  Sixty-four
// lanes each adding 1 to the same LDS word are sixty-four serialised atomics -- on the LDS pipe every wave of the CU shares.
// Wave-uniform control flow only; the positions of one call are consecutive in lane order.
template <int T>
__device__ __forceinline__ unsigned claim_positions(bool want, unsigned* counter, unsigned& wave_count) {
  const unsigned long long m = __ballot(want);
  const unsigned c = (unsigned)__popcll(m);
  unsigned base;
  if (T == WAVE) {
    base = wave_count;
    wave_count += c;
  } else {
    unsigned b = 0u;
    if ((threadIdx.x & (WAVE - 1)) == 0 && c != 0u) b = atomicAdd(counter, c);
    base = wave_read_lane(b, 0);
  }
  return base + lanes_below(m);
}
// exclusive scan of one unsigned per thread across a team of T threads; *total = team sum.  Every thread must call.
template <int T>
__device__ __forceinline__ unsigned team_exclusive_scan(unsigned v, unsigned* s_wsum /*[T / WAVE]*/, unsigned* total) {
  const int lane = threadIdx.x & (WAVE - 1);
  const unsigned inc = wave_inclusive_sum(v);
  if (T == WAVE) {
    *total = wave_read_lane(inc, WAVE - 1);
    return inc - v;
  }
  const int wave = (threadIdx.x % T) / WAVE;
  if (lane == WAVE - 1) s_wsum[wave] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < T / WAVE; ++w) {
    const unsigned sw = s_wsum[w];
    if (w < wave) base += sw;
    tot += sw;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// candidates scored together per lane (their count gathers travel together), per class.  Round 3, config 4 (the count table no
// longer fits an L2): two per lane -5 % on the one-wave and both 256-thread classes, +11 % on the half-CU class, +-0 on the CU class
constexpr int URCCO_U_WAVE = 2;
constexpr int URCCO_U_BS = 2;
constexpr int URCCO_U_B = 2;
constexpr int URCCO_U_H = 1;
constexpr int URCCO_U_C = 1;
constexpr int URCCO_OCC_WAVE = 8;  // blocks of four one-wave teams per CU the one-wave class is compiled for
constexpr int URCCO_OCC_BS = 7;  // blocks per CU the 256-thread / 4Ki class is compiled for (8 = 64 registers: seven of them spill)
constexpr int URCCO_G_WAVE = 2;
constexpr int URCCO_SEL_AMB_WAVE = 64;
constexpr int URCCO_SEL_AMB_BLOCK = 128;
constexpr int URCCO_G_BLOCK = 2;
constexpr int URCCO_SEL_M_BLOCK = 128;  // capacity of the ambiguous set of the teams of several waves (>= URCCO_SEL_AMB_BLOCK)
constexpr int URCCO_G_CU = 2;
// MP ("multi-pass", bin 6): rows no single LDS table can hold -- a hot item of a skewed catalogue pairs with tens of thousands
// of distinct columns -- or whose counts overflow the packed field.  Such a row is accumulated in P = 2^s passes over its
// cooccurrence pairs: pass q keeps the columns with (col mod P) == q, keyed by col div P (so the key narrows by s bits and the
// count field widens by as many), cuts them to their own top k, and merges those into the row's running top k (ranked over
// <= 2k elements).  The exact top k of the row is the top k of the passes' top k's: ties are cut by the full column.  P starts
// from the row's work (1.25 x the expected distinct columns per pass must fit) and doubles whenever a pass still overflows --
// at the latest when ceil(n_cols / P) columns are GUARANTEED to fit, so every row ends.  Round 2 served these rows from dense
// counters in global memory (n_cols x 16 B of scratch per resident block, L2 atomics): 35.9 ms for 16K rows of config 5.
// DBG: the ablation / test switches of CcoArgs::debug exist only in a second instantiation (profiling tools and the race regression tests
// launch it); the production instantiation carries neither their branches nor the scalar register a.debug would occupy -- at eight waves
// per SIMD a wave has 78 scalar registers and the one-wave class spilled 128 of them to vector lanes (round 5: 69 after this and the
// single-check LLR).
// PK: the instantiation for a B' with the columns' counts aboard (CcoArgs::b_packed).  Whether the counts fit is known on the DEVICE only (*pack_bad), so
// the launcher enqueues both instantiations and the one whose turn it is not returns at once -- the price of keeping the other form's registers (the count
// gather's pointers, the word masks as run-time values) out of each: as one kernel with a run-time switch the 256-thread class spilled and the 512-thread
// class lost a wave per SIMD.
template <int T, int E, int U, bool MP = false, bool DBG = false, bool PK = false>
__global__ __launch_bounds__((T < 256 ? 256 : T), (T == 64 ? URCCO_OCC_WAVE : (T == 256 && E == 4096 ? URCCO_OCC_BS : (T == 512 ? 4 : 1)))) void cco_rows_kernel(CcoArgs a, int bin) {
  if ((a.b_packed != nullptr && (a.pk_known != 0 || *a.pack_bad == 0)) != PK) return;  // grid-uniform
  const int dbg = DBG ? a.debug : 0;
  // the arguments the row loop's inner loops use, each in scalar registers of its own (URCCO_OWN_SGPRS)
  // B' with the columns' counts aboard while every count fits (CcoArgs::b_packed), else the plain column indices and the count gather (wave-uniform)
  constexpr bool packed = PK;
  URCCO_OWN_GLOBAL_PTR(const int32_t, b_col_idx, PK ? a.b_packed : a.b_col_idx);
  const int cshift = 32 - a.count_bits;                              // a B' word: column in the low cshift bits, count above
  const unsigned colmask = PK ? (1u << cshift) - 1u : a.b_col_mask;  // (cshift <= 31: count_bits >= 1; the plain form masks only when its words are packed ones)
  URCCO_OWN_GLOBAL_PTR(const unsigned short, cnt_b16, a.cnt_b16);
  URCCO_OWN_GLOBAL_PTR(const int32_t, cnt_b, a.cnt_b);
  URCCO_OWN_GLOBAL_PTR(const double, xlx_tab, a.xlx_tab);
  URCCO_OWN_GLOBAL_PTR(const double, xlx_hi, a.xlx_hi);
  URCCO_OWN_GLOBAL_PTR(const double, col_ent, a.col_ent);
  URCCO_OWN_GLOBAL_PTR(int32_t, out_idx, a.out_idx);
  URCCO_OWN_GLOBAL_PTR(double, out_llr, a.out_llr);
  long long n_users = a.n_users;
  URCCO_OWN_SGPRS(n_users);
  constexpr int BLOCK = T < 256 ? 256 : T;
  constexpr int TEAMS = BLOCK / T;
  constexpr int SH = table_slots(E, T);  // accumulator slots
  constexpr int SPT = SH / T;
  static_assert(SH % (2 * T) == 0 && SH + SH / 2 <= E, "slots and their 16-bit column counts share the table");
  constexpr int NW = T / WAVE;  // waves per team
  constexpr int G = T == WAVE ? URCCO_G_WAVE : (T == 256 ? URCCO_G_BLOCK : URCCO_G_CU);  // column gathers in flight per lane
  constexpr int LOG2E = E == 1024 ? 10 : (E == 4096 ? 12 : (E == 8192 ? 13 : (E == 16384 ? 14 : 15)));
  static_assert((1 << LOG2E) == E, "table size");
  constexpr int LOG2T = T == 64 ? 6 : (T == 256 ? 8 : (T == 512 ? 9 : 10));
  static_assert((1 << LOG2T) == T, "team size");
  __shared__ __attribute__((aligned(16))) unsigned s_tab[TEAMS * E];
  // One-wave teams and the small block class: the chunk operands (insert phase), the select histograms + survivor list (select
  // passes) and the ambiguous / staged survivors (after the passes; they overlay the histograms) are never live together and
  // share ONE region per team.  One-wave class: 19.6 KB of LDS per block instead of 26.8, which with <= 64 VGPRs lets eight
  // blocks (32 waves) share a CU instead of six.
  // select histograms: 256 bins of 16-bit counters, two per word.  Teams of several waves rotate three (pass p counts into
  // one while the previous one is cleared: one team barrier per pass); a one-wave team needs one -- every lane zeroes the
  // two words it has just read.
  constexpr int NH = T == WAVE ? 1 : 3;
  constexpr int SEL_CAP = T == WAVE ? 0 : (T == 256 ? 512 : 2048);  // explicit survivor list (16-bit indices); a wave sweeps its <= 341 candidates directly
  constexpr int SEL_M = T == WAVE ? 64 : URCCO_SEL_M_BLOCK;          // capacity of the ambiguous-set / staged-output arrays
  constexpr int SEL_AMB = T == WAVE ? URCCO_SEL_AMB_WAVE : URCCO_SEL_AMB_BLOCK;  // the cut bin is ranked directly once it holds this many or fewer
  static_assert(SEL_AMB <= SEL_M, "ambiguous set capacity");
  constexpr bool SHARE = T == WAVE || (T == 256 && E == 4096);
  constexpr int SH_INS = T * 8 + (T + 1) * 4;                                                     // ustart | uoff
  constexpr int SH_LST = ((NH * 128 * 4 > SEL_M * 12 ? NH * 128 * 4 : SEL_M * 12) + 7) / 8 * 8;  // histograms or amb_key | amb_col, then the list
  constexpr int SH_SEL = SH_LST + SEL_CAP * 2;
  constexpr int SHARE_WORDS = ((SH_INS > SH_SEL ? SH_INS : SH_SEL) + 7) / 8;
  __shared__ unsigned long long s_share[SHARE ? TEAMS * SHARE_WORDS : 1];
  __shared__ long long s_ustart[SHARE ? 1 : TEAMS * T];
  __shared__ unsigned s_uoff[SHARE ? 1 : TEAMS * (T + 1)];
  __shared__ unsigned s_wsum[NW];
  __shared__ unsigned s_hist[SHARE ? 1 : TEAMS * NH * 128];
  __shared__ unsigned s_selres[TEAMS * 4];
  __shared__ unsigned short s_lst[SHARE ? 1 : TEAMS * (SEL_CAP > 0 ? SEL_CAP : 1)];
  __shared__ unsigned long long s_ambkey[SHARE ? 1 : TEAMS * SEL_M];
  __shared__ unsigned s_ambcol[SHARE ? 1 : TEAMS * SEL_M];
  __shared__ unsigned long long s_selthr[TEAMS * 2];
  // The leading key bytes shared by every candidate of a row need no select pass (LLRs of one row share sign and high
  // exponent bits: typically the whole first pass).  Measured on config 3: -7..9 % for the 256-thread classes, but the
  // extra live registers cost the one-wave class +4 % (spills at its 80-VGPR cap) and the 512/1024-thread classes
  // +0..4 %, so only T == 256 tracks the shared bytes.
  constexpr bool SKIP_SHARED = T == 256;
  __shared__ unsigned long long s_kbits[2 * NW];  // per wave: AND / OR over its valid keys
  __shared__ unsigned s_mpflag;                    // MP: a pass overflowed its table
  __shared__ unsigned long long s_runk[MP ? 2 * MP_KMAX : 1];  // MP: the row's running top k (two buffers: a merge reads one, writes the other)
  __shared__ unsigned s_runc[MP ? 2 * MP_KMAX : 1];

  const int team = TEAMS == 1 ? 0 : uni((int)threadIdx.x / T);  // a team is one wave (T == 64) or the whole block
  const int tl = threadIdx.x % T;
  const int lane = threadIdx.x & (WAVE - 1);
  unsigned* tab = s_tab + team * E;
  unsigned short* cbv = reinterpret_cast<unsigned short*>(tab + SH);   // insert phase: the slots' column counts
  unsigned long long* share = s_share + (SHARE ? team * SHARE_WORDS : 0);
  long long* ustart = SHARE ? reinterpret_cast<long long*>(share) : s_ustart + team * T;
  unsigned* uoff = SHARE ? reinterpret_cast<unsigned*>(share + T) : s_uoff + team * (T + 1);
  unsigned* hist = SHARE ? reinterpret_cast<unsigned*>(share) : s_hist + team * NH * 128;
  unsigned* sel_res = s_selres + team * 4;
  unsigned* nsel = sel_res + 3;
  unsigned short* lst = SHARE ? reinterpret_cast<unsigned short*>(share) + SH_LST / 2 : s_lst + team * (SEL_CAP > 0 ? SEL_CAP : 1);
  unsigned long long* amb_key = SHARE ? share : s_ambkey + team * SEL_M;
  unsigned* amb_col = SHARE ? reinterpret_cast<unsigned*>(share + SEL_M) : s_ambcol + team * SEL_M;
  unsigned long long* sel_thr = s_selthr + team * 2;
  const int list_start = a.bin_off[bin];
  const int list_n = a.bin_off[bin + 1] - list_start;
  const int total_teams = gridDim.x * TEAMS;
  bool ident = (long long)a.n_cols_b * 3 + (long long)a.k * 3 + 2 <= E && a.n_cols_b <= SH;  // the table spans every column of B: slots addressed by column
  int cb = a.count_bits;                                                   // (MP: both follow the row's pass count)
  unsigned cmask = (1u << cb) - 1u;
  unsigned long long cand_acc = 0ull;  // distinct (row, column) candidates scored by this team (statistics)
  const double xlx_n = *a.xlx_n;
  const bool use16 = *a.cnt16_bad == 0;

  // T == 64: teams are independent waves (wave-level sync only).  T > 64: one team per block, loop is block-uniform.
  int li = blockIdx.x * TEAMS + team;
  if (li >= list_n) return;  // team-uniform (a block for T > 64, a wave -- which only ever synchronises with itself -- for T == 64)
  // The chain row id -> CSC bounds -> first-chunk operands, as in the micro class (see there: the memory counter retires in order):
  // every link is issued at the TOP of a row for the rows ahead (ids of the next three rows, bounds two rows ahead, operands one),
  // unconditionally (list positions past the end re-read the last row), into VECTOR registers -- uniform values packed by lane: one
  // register carries three row ids, one pair both bounds of a row -- and read into scalars (readlane / readfirstlane) only where they
  // are consumed, a row later; URCCO_SETTLE collects them after the score phase, ahead of every store of the row's output.
  // (Rounds 1-3: readfirstlane next to each of these loads = a wait for it, twice in a row at the top of every row and once more
  // on the first-chunk operands.)
  const int S = total_teams;
  auto pos_of = [&](int l) { return list_start + (l < list_n ? l : list_n - 1); };
  const unsigned* wp32 = reinterpret_cast<const unsigned*>(a.wp);  // low words: a chunk only uses differences between its own entries
  const int lane3 = lane < 3 ? lane : 2;
  int idv = a.bin_rows[pos_of(li + lane3 * S)];  // lanes 0, 1, 2: ids of this row and the next two
  int64_t cs_c, ce_c;                             // this row's CSC bounds (scalars)
  int64_t bnd_b;                                  // the next row's: lane 0 start, lane 1 end
  {
    const int id0 = (int)wave_read_lane((unsigned)idv, 0);
    const int id1 = (int)wave_read_lane((unsigned)idv, 1);
    cs_c = uni(a.a_col_ptr[id0]);
    ce_c = uni(a.a_col_ptr[id0 + 1]);
    bnd_b = a.a_col_ptr[id1 + (lane & 1)];
  }
  // first-chunk operands of the row about to be processed (low words of the work prefix)
  unsigned pf_w0, pf_w1, pf_wp;
  int64_t pf_start;
  {
    const int64_t c1 = cs_c + T < ce_c ? cs_c + T : ce_c;
    const int64_t pl = cs_c + tl < c1 ? cs_c + tl : c1 - 1;
    pf_w0 = wp32[2 * cs_c];
    pf_w1 = wp32[2 * c1];
    pf_wp = wp32[2 * pl];
    pf_start = a.pstart[pl];
  }
  URCCO_SETTLE(idv); URCCO_SETTLE(bnd_b); URCCO_SETTLE(pf_w0); URCCO_SETTLE(pf_w1); URCCO_SETTLE(pf_wp); URCCO_SETTLE(pf_start);
  int64_t cs_n = 0, ce_n = 0, bnd_c = 0;  // next row's bounds as scalars / the bounds two rows ahead in flight: rotated by the loop's increment
  for (; li < list_n; li += S, cs_c = cs_n, ce_c = ce_n, bnd_b = bnd_c) {
    const int i = (int)wave_read_lane((unsigned)idv, 0);
    const int id2 = (int)wave_read_lane((unsigned)idv, 2);
    const int64_t cs = cs_c, ce = ce_c;
    cs_n = wave_read_lane64(bnd_b, 0);
    ce_n = wave_read_lane64(bnd_b, 1);
    // this row's first-chunk operands leave their registers ...
    const unsigned row_w0 = uni(pf_w0), row_w1 = uni(pf_w1);
    const unsigned my_wp = pf_wp;
    const int64_t my_start = pf_start;
    // ... and the rows ahead take them
    int lz = lane;
    URCCO_OPAQUE(lz);  // (what derives from the lane here is recomputed per row: kept across the row loop it was spilled at 64 registers, and a scratch reload at
                       // the top of a row is a wait for the prefetches just issued)
    idv = a.bin_rows[pos_of(li + (1 + (lz < 3 ? lz : 2)) * S)];
    bnd_c = a.a_col_ptr[id2 + (lz & 1)];
    if (!MP) {  // (the multi-pass rows re-read every chunk once per pass: no prefetched first chunk)
      const int64_t c1 = cs_n + T < ce_n ? cs_n + T : ce_n;
      const int64_t pl = cs_n + tl < c1 ? cs_n + tl : c1 - 1;
      pf_w0 = wp32[2 * cs_n];
      pf_w1 = wp32[2 * c1];
      pf_wp = wp32[2 * pl];
      pf_start = a.pstart[pl];
    }
    // MP: number of passes 2^mp_s, current pass mp_q, entries of the running top k and which of its two buffers is current
    int mp_s = 0;
    unsigned mp_q = 0u, n_run = 0u, run_cur = 0u;
    if (MP) {
      const long long w_row = (long long)(uni(a.wp[ce]) - uni(a.wp[cs]));
      const long long ca_row = a.cnt_a[i];
      const long long dd = w_row < (long long)a.n_cols_b ? w_row : (long long)a.n_cols_b;
      const long long cap = ((long long)E - 3ll * a.k - 2ll) / 3ll;  // distinct columns a pass may hold (packed counts + keys + survivors)
      for (;; ++mp_s) {
        const long long cols_pp = ((long long)a.n_cols_b + (1ll << mp_s) - 1) >> mp_s;  // columns a pass can see
        int kb = 1;
        while ((1ll << kb) <= cols_pp) ++kb;
        const bool count_ok = kb <= 1 || ca_row <= (1ll << (32 - kb)) - 1;
        const long long exp_d = ((dd >> mp_s) + (dd >> (mp_s + 2)) + 1) < cols_pp ? ((dd >> mp_s) + (dd >> (mp_s + 2)) + 1) : cols_pp;
        if (count_ok && exp_d <= cap) break;
      }
    }
  mp_again:  // MP: the next pass, or the row again with twice the passes (team-uniform jumps)
    if (MP) {
      const long long cols_pp = ((long long)a.n_cols_b + (1ll << mp_s) - 1) >> mp_s;
      int kb = 1;
      while ((1ll << kb) <= cols_pp) ++kb;
      cb = 32 - kb;
      cmask = cb >= 32 ? 0xffffffffu : (1u << cb) - 1u;
      ident = cols_pp * 3 + 3ll * a.k + 2ll <= (long long)E && cols_pp <= (long long)SH;
      if (tl == 0) s_mpflag = 0u;
    }
    const unsigned mp_mask = MP ? (1u << mp_s) - 1u : 0u;
#pragma unroll
    for (int q = 0; q < SPT / 2; ++q) *reinterpret_cast<uint2*>(&tab[2 * (tl + q * T)]) = make_uint2(0u, 0u);
    team_sync<T>();
    // ---- 2. expand + accumulate
    for (int64_t c0 = cs; c0 < ce; c0 += T) {  // team-uniform
      const int64_t c1 = c0 + T < ce ? c0 + T : ce;
      const bool pre = !MP && c0 == cs;  // the first chunk's operands were prefetched
      const unsigned w0 = pre ? row_w0 : uni(wp32[2 * c0]);  // low words: the differences below are < 2^32
      const unsigned total = (pre ? row_w1 : uni(wp32[2 * c1])) - w0;
      const int64_t p = c0 + tl;
      if (p < c1) {
        ustart[tl] = pre ? my_start : a.pstart[p];
        uoff[tl] = (pre ? my_wp : wp32[2 * p]) - w0;
      } else {
        uoff[tl] = total;
      }
      if (tl == 0) uoff[T] = total;
      team_sync<T>();
      if (total > 0u) {
        const unsigned per = (total + T - 1) / T;
        const unsigned first = (unsigned)tl * per;
        if (first < total) {
          const unsigned last = first + per < total ? first + per : total;
          int lo = 1, hi = T;  // first idx in [1, T] with uoff[idx] > first (uoff[T] = total > first).  T candidates, halved exactly log2(T) times:
#pragma unroll             // a fixed trip count, selects instead of branches (the data-dependent loop cost four scalar instructions per step)
          for (int step = 0; step < LOG2T; ++step) {
            const int mid = (lo + hi) >> 1;
            const bool gt = uoff[mid] > first;
            hi = gt ? mid : hi;
            lo = gt ? lo : mid + 1;
          }
          int o = lo - 1;
          int64_t pos = ustart[o] + (first - uoff[o]);
          unsigned uend = uoff[o + 1];
          // `per` is team-uniform: the loop counter and its bound live in the scalar unit.  G column gathers are issued
          // before the first of their inserts (a gather that misses L2 costs 1-2 us and a lane's pairs are a chain of them).
          for (unsigned x = 0; x < per; x += G) {
            unsigned jj[G];
            bool on[G];
#pragma unroll
            for (int q = 0; q < G; ++q) {
              const unsigned t = first + x + (unsigned)q;
              on[q] = t < last;
              jj[q] = 0u;
              if (on[q]) {
                if (t >= uend) {  // next user with a non-empty B' row
                  do { ++o; } while (uoff[o + 1] <= t);
                  pos = ustart[o];
                  uend = uoff[o + 1];
                }
                jj[q] = (unsigned)b_col_idx[pos++];
              }
            }
#pragma unroll
            for (int q = 0; q < G; ++q) {
              if (on[q]) {
                const unsigned col = jj[q] & colmask, c_b = packed ? jj[q] >> cshift : 0u;
                if (dbg & 1) {  // ablation: gather only
                  if (jj[q] == 0xffffffffu) tab[0] = 1u;
                } else if (MP) {
                  if ((col & mp_mask) == mp_q && !tab_insert<SH>(tab, (col >> mp_s) + 1u, cb, ident, cbv, c_b)) s_mpflag = 1u;
                } else if (!tab_insert<SH>(tab, col + 1u, cb, ident, cbv, c_b)) {
                  atomicAdd(a.err, 1ull);
                }
              }
            }
          }
        }
      }
      team_sync<T>();  // before the next chunk overwrites ustart / uoff
    }
    // ---- 3. compact the occupied slots to tab[0 .. D); candidate keys will live behind them in the same LDS:
    //         words [kb, kb + 2 D) with kb = D rounded up to even.  The binning rule keeps 3 D + 1 <= E.
    unsigned D;
    //         ... and every candidate's column count goes into its slot of that key array (the score phase reads it, then puts the key there)
    // A thread sweeps PAIRS of neighbouring slots: one 8-byte read for the two packed words, one word for their two counts.
    if (T == WAVE) {  // one wave: positions from ballots (no scan); all reads are issued before the first write
      unsigned v[SPT];
      unsigned cw2[SPT / 2];  // the slots' counts, two per register
#pragma unroll
      for (int q = 0; q < SPT / 2; ++q) {
        const uint2 x = *reinterpret_cast<const uint2*>(&tab[2 * (tl + q * T)]);
        v[2 * q] = x.x;
        v[2 * q + 1] = x.y;
        cw2[q] = *reinterpret_cast<const unsigned*>(&cbv[2 * (tl + q * T)]);
      }
      D = 0;
#pragma unroll
      for (int q = 0; q < SPT; ++q) D += (unsigned)__popcll(__ballot(v[q] != 0u));  // (the keys' base depends on D: counted first, positions below)
      unsigned long long* kk0 = reinterpret_cast<unsigned long long*>(tab + ((D + 1u) & ~1u));
      unsigned at0 = 0;
#pragma unroll
      for (int q = 0; q < SPT; ++q) {
        const unsigned long long m = __ballot(v[q] != 0u);
        if (v[q] != 0u) {
          const unsigned at = at0 + lanes_below(m);
          tab[at] = v[q];
          kk0[at] = (unsigned long long)((q & 1) ? cw2[q >> 1] >> 16 : cw2[q >> 1] & 0xffffu);
        }
        at0 += (unsigned)__popcll(m);
      }
    } else {
      unsigned v[SPT];
      unsigned cw2[SPT / 2];
      unsigned occ = 0;
#pragma unroll
      for (int q = 0; q < SPT / 2; ++q) {
        const uint2 x = *reinterpret_cast<const uint2*>(&tab[2 * (tl + q * T)]);
        v[2 * q] = x.x;
        v[2 * q + 1] = x.y;
        cw2[q] = *reinterpret_cast<const unsigned*>(&cbv[2 * (tl + q * T)]);
        occ += (x.x != 0u) + (x.y != 0u);
      }
      unsigned wpos = team_exclusive_scan<T>(occ, s_wsum, &D);
      D = uni(D);
      team_sync<T>();  // every read of the table precedes every write below
      unsigned long long* kk0 = reinterpret_cast<unsigned long long*>(tab + ((D + 1u) & ~1u));
      // (a multi-pass row's pass may have filled more slots than leave room for their keys: it is abandoned below -- and must not write key slots
      // beyond the table; found by the simulator's bounds-checked build)
      const bool fits = !MP || 3ll * D + 3ll * a.k + 2ll <= (long long)E;  // team-uniform
#pragma unroll
      for (int q = 0; q < SPT; ++q)
        if (v[q] != 0u) {
          tab[wpos] = v[q];
          if (fits) kk0[wpos] = (unsigned long long)((q & 1) ? cw2[q >> 1] >> 16 : cw2[q >> 1] & 0xffffu);
          ++wpos;
        }
    }
    team_sync<T>();
    if (MP) {
      // the pass must leave room for its keys and its survivors behind the packed counts; a pass that does not (or whose table
      // filled up) is abandoned and the row starts over with twice the passes
      if (s_mpflag != 0u || 3ll * D + 3ll * a.k + 2ll > (long long)E) {  // team-uniform
        team_sync<T>();
        ++mp_s;
        mp_q = 0u;
        n_run = 0u;
        goto mp_again;
      }
    }
    cand_acc += D;
    unsigned long long* kk = reinterpret_cast<unsigned long long*>(tab + ((D + 1u) & ~1u));
    // ---- 4. score candidates tl, tl + T, ... (dense); keys go to LDS behind the packed counts
    unsigned n_valid = 0;
    unsigned long long kand = ~0ull, kor = 0ull;  // over this thread's valid keys: the bytes all keys share need no select pass
    {
      const long long ca = a.cnt_a[i];
      const double row_entropy = a.ent_a[i];
      for (unsigned base = 0; base < D; base += U * T) {  // scalar loop control; the column-info gathers of U candidates travel together
        const unsigned t0 = base + (unsigned)tl;
        unsigned vv[U];
        int cbj[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
          const unsigned t = t0 + (unsigned)x * T;
          vv[x] = t < D ? tab[t] : 0u;
          cbj[x] = 0;
          if (vv[x] != 0u) {
            const int j = MP ? (int)((((vv[x] >> cb) - 1u) << mp_s) | mp_q) : (int)(vv[x] >> cb) - 1;
            // the candidate's cB: out of its slot of the key array, where the compaction left it (it came with the B' word) -- or, for a B' without counts aboard, the ONE scattered
            // gather per candidate of rounds 1-5 (ablation 512: a made-up count, no gather)
            cbj[x] = packed ? (int)(unsigned)kk[t] : ((dbg & 512) ? (int)(vv[x] & cmask) + 100 : (use16 ? (int)cnt_b16[j] : cnt_b[j]));
          }
        }
        // every operand of these U candidates of every lane inside the tables: the wave takes the straight-line table form (see llr_from_tables)
        bool in_tables = true;
#pragma unroll
        for (int x = 0; x < U; ++x)
          if (vv[x] != 0u) in_tables = in_tables && llr_operands_in_tables(vv[x] & cmask, ca, (unsigned)cbj[x], n_users, col_ent);
        const bool all_in_tables = !(dbg & 2) && __ballot(!in_tables) == 0ull;  // wave-uniform
#pragma unroll
        for (int x = 0; x < U; ++x) {
          const unsigned t = t0 + (unsigned)x * T;
          if (t < D) {
            const int j = MP ? (int)((((vv[x] >> cb) - 1u) << mp_s) | mp_q) : (int)(vv[x] >> cb) - 1;
            const long long k11 = (long long)(vv[x] & cmask);
            unsigned long long key = 0ull;
            if (!(a.exclude_self && j == i)) {
              const double llr = all_in_tables ? llr_from_tables(row_entropy, xlx_n, (unsigned)k11, (unsigned)ca, (unsigned)cbj[x], xlx_tab, xlx_hi, col_ent)
                                               : ((dbg & 2) ? (double)k11
                                                            : llr_of(row_entropy, xlx_n, k11, ca, (long long)cbj[x], n_users, xlx_tab, xlx_hi, col_ent));
              if (llr > 0.0 && (!a.has_min_llr || llr >= a.min_llr)) key = (unsigned long long)__double_as_longlong(llr);
            }
            kk[t] = key;
            if (key != 0ull) {
              ++n_valid;
              if (SKIP_SHARED) {
                kand &= key;
                kor |= key;
              }
            }
          }
        }
      }
    }
    // what was issued at the top of the row has had the expand and score phases to arrive: collected ahead of the row's output stores
    URCCO_SETTLE(idv); URCCO_SETTLE(bnd_c);
    if (!MP) { URCCO_SETTLE(pf_w0); URCCO_SETTLE(pf_w1); URCCO_SETTLE(pf_wp); URCCO_SETTLE(pf_start); }
    if (SKIP_SHARED) {
      wave_and_or_u64(kand, kor);
      if (lane == 0) {  // published by the barriers inside the scan below
        s_kbits[2 * (tl / WAVE)] = kand;
        s_kbits[2 * (tl / WAVE) + 1] = kor;
      }
    }
    unsigned C;
    team_exclusive_scan<T>(n_valid, s_wsum, &C);
    C = uni(C);
    team_sync<T>();
    // ---- 5. top-k.  Order: key desc, then column asc == (key, ~col) desc as one 96-bit composite.
    //   a. C > k: MSB-first radix select (8-bit digits, LDS histogram) of the k-th composite; stops as soon as the digit
    //      bin that straddles the cut is wanted whole;   b. the <= k survivors are gathered;   c. each is ranked by
    //      counting and written straight to its output position (already in output order).
    const int64_t obase = ((int64_t)(i - a.item_lo)) * a.k;
    unsigned long long thr_key = 0ull;
    unsigned thr_ncol = 0u;
    bool row_done = false;  // team-uniform: the select's finish has already written the row
    if (!(dbg & 4)) {
      if (dbg & 8) {  // ablation: no select (nothing passes)
        if (C > (unsigned)a.k) thr_key = ~0ull;
      } else if (C > (unsigned)a.k) {  // team-uniform
        // MSB-first radix select of the k-th composite, 8-bit digits, LDS histograms (three rotating 256-bin arrays of
        // 16-bit counters, two per word: pass p counts into H[p % 3] while H[(p + 2) % 3] is cleared; every wave repeats
        // the digit search for itself, so a pass costs ONE team barrier).  Two shortcuts keep it to ~3 sweeps:
        //  * once the bin that straddles the cut is small enough its members are copied to an explicit index list and
        //    later passes sweep only that list (teams larger than a wave);
        //  * once it holds <= SEL_M members their (key, col) are copied out and ranked against each other (full
        //    composite, so ties by column are exact); the need-th best becomes the threshold.
        unsigned need = (unsigned)a.k;
        if (a.col_bytes < 4) thr_ncol = 0xffffffffu << (8 * a.col_bytes);  // digits of ~col above the highest used byte are all ones
        int p0 = 0;  // first key byte that differs between candidates
        if (!SKIP_SHARED) {
          // The classes that do not track the shared key bytes while they score (registers) find them here, with one cheap sweep over
          // the keys (an AND and an OR per key, no histogram, no atomics): the LLRs of a row share their sign / exponent byte, so the
          // select's first pass -- a full histogram sweep plus a digit search -- found one bin holding everything and was wasted.
          kand = ~0ull;
          kor = 0ull;
          for (unsigned base = 0; base < D; base += T) {  // scalar loop control
            const unsigned t = base + (unsigned)tl;
            const unsigned long long key = t < D ? kk[t] : 0ull;
            if (key != 0ull) {
              kand &= key;
              kor |= key;
            }
          }
          wave_and_or_u64(kand, kor);
          if (T != WAVE) {
            if (lane == 0) {
              s_kbits[2 * (tl / WAVE)] = kand;
              s_kbits[2 * (tl / WAVE) + 1] = kor;
            }
            team_sync<T>();
          }
        }
        {
          if (T != WAVE) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
              kand &= s_kbits[2 * w];
              kor |= s_kbits[2 * w + 1];
            }
          }
          const unsigned long long kdiff = kand ^ kor;
          p0 = kdiff == 0ull ? 8 : (__clzll((long long)kdiff) >> 3);
          thr_key = p0 == 0 ? 0ull : (kor & ~(p0 >= 8 ? 0ull : (~0ull >> (8 * p0))));
        }
        {
          int tz = tl;
          URCCO_OPAQUE(tz);  // (the address is formed here: hoisted out of the row loop it was the one register the one-wave class spilled to scratch)
          for (int b = tz; b < NH * 128; b += T) hist[b] = 0u;
        }
        if (tl == 0) { sel_res[0] = 0u; sel_res[1] = 0u; }  // list length, ambiguous-set length
        team_sync<T>();
        const int first_col_pass = 8 + (3 - (a.col_bytes - 1));  // column digits above the highest used byte are constant: skip
        bool have_list = false;
        unsigned list_n = 0, prev_cnt = C;
        bool first_pass = true;
        // The rotating histograms are indexed by the ordinal q of the passes that actually run, not by the digit position
        // p: passes are skipped (shared key bytes, constant high column bytes), and a rotation keyed by p would count the
        // first column pass into the buffer the last key pass left full.
        int q = 0;
        for (int p = p0; p < 12; ++p) {  // team-uniform trip count (the breaks below are on values every thread agrees on)
          if (p >= 8 && p < first_col_pass) continue;
          unsigned* H = hist + (q % NH) * 128;
          const bool build = SEL_CAP > 0 && !have_list && !first_pass && prev_cnt <= (unsigned)SEL_CAP;  // this sweep also records the survivors
          const unsigned n_scan = have_list ? list_n : D;
          const int shk = p < 8 ? 56 - 8 * p : 0, shc = p < 8 ? 0 : 24 - 8 * (p - 8);
          unsigned lst_n = 0u;  // (unused: only teams of several waves build a list)
          for (unsigned base = 0; base < n_scan; base += T) {  // scalar loop control, no divergent exits (claim_positions is a wave operation)
            const unsigned idx = base + (unsigned)tl;
            bool match = false;
            unsigned dig = 0u, t = 0u;
            if (idx < n_scan) {
              t = have_list ? (unsigned)lst[idx] : idx;
              const unsigned long long key = kk[t];
              if (key != 0ull) {
                if (p < 8) {
                  match = first_pass || (key >> (shk + 8)) == (thr_key >> (shk + 8));
                  dig = (unsigned)(key >> shk) & 255u;
                } else {
                  const unsigned ncol = ~(unsigned)((int)(tab[t] >> cb) - 1);
                  match = key == thr_key && (p == first_col_pass || (ncol >> (shc + 8)) == (thr_ncol >> (shc + 8)));
                  dig = (ncol >> shc) & 255u;
                }
              }
            }
            if (match) atomicAdd(&H[dig >> 1], 1u << (16 * (dig & 1u)));  // counts < 2^16: D is bounded by the table size
            if (build) {  // team-uniform
              const unsigned pos = claim_positions<T>(match, &sel_res[0], lst_n);
              if (match) lst[pos] = (unsigned short)t;
            }
          }
          team_sync<T>();
          if (build) {
            have_list = true;
            list_n = uni(sel_res[0]);
          }
          first_pass = false;
          // test hook (tests/test_gpu_parity.py::test_select_overlay_race_*): the team's FIRST wave -- it owns the lowest table entries, the
          // ones a tie at the cut selects -- dawdles before it reads the histogram, so that its siblings are far ahead of it: the
          // interleaving the round-3 race needed, made certain
          if (T != WAVE && (dbg & 131072) && tl / WAVE == 0) {
#ifdef HIPSIM_HOST_BUILD
            __builtin_amdgcn_s_sleep(127);
#else
            asm volatile("s_sleep 127\n\ts_sleep 127" ::: "memory");  // "memory": the histogram reads below must not be hoisted above the nap
#endif
          }
          {  // every wave locates the digit that holds the cut: lane l owns the four bins of digit group 63 - l (the highest
             // digits sit in the lowest lanes, so that the count of everything above a group is a PREFIX sum over lanes)
            const int grp = WAVE - 1 - lane;
            const unsigned w01 = H[2 * grp], w23 = H[2 * grp + 1];
            if (NH == 1) {  // one wave: the words just read are this lane's to clear
              H[2 * grp] = 0u;
              H[2 * grp + 1] = 0u;
            } else {
              unsigned* Hz = hist + ((q + NH - 1) % NH) * 128;
              for (int b = tl; b < 128; b += T) Hz[b] = 0u;  // the previous pass's buffer: every wave is past its reads of it
            }
            const unsigned h0 = w01 & 0xffffu, h1 = w01 >> 16, h2 = w23 & 0xffffu, h3 = w23 >> 16;
            const unsigned v4 = h0 + h1 + h2 + h3;
            const unsigned S = wave_inclusive_sum(v4);  // members of this digit group and of every higher one
            const unsigned long long ge = __ballot(S >= need);
            const int L = __ffsll((unsigned long long)ge) - 1;  // the highest group that reaches the cut (ge != 0: S of lane 63 is the whole set)
            unsigned above = S - v4;
            unsigned d, cnt;
            if (above + h3 >= need) { d = 3; cnt = h3; }
            else if (above + h3 + h2 >= need) { d = 2; cnt = h2; above += h3; }
            else if (above + h3 + h2 + h1 >= need) { d = 1; cnt = h1; above += h3 + h2; }
            else { d = 0; cnt = h0; above += h3 + h2 + h1; }
            const unsigned packed = wave_read_lane(((4u * (unsigned)grp + d) << 16) | cnt, L);  // cnt < 2^16
            above = wave_read_lane(above, L);
            d = packed >> 16;
            cnt = packed & 0xffffu;
            need -= above;
            if (p < 8) thr_key |= (unsigned long long)d << shk;
            else thr_ncol |= d << shc;
            prev_cnt = cnt;
            if (cnt == need) break;  // the whole bin is wanted: every composite >= the prefix (low bits zero) is selected
            if (NH == 1) team_sync<T>();  // the histogram just cleared is the next pass's target
            ++q;
          }
          if (prev_cnt <= (unsigned)SEL_AMB) {  // team-uniform
            // finish: copy out the members of the cut bin (they match the prefix through digit p) ...
            // In the SHARE layout amb_key / amb_col OVERLAY the three rotating histograms.  Every wave has run the digit search above for
            // itself, at its own pace: a wave that arrives here first must not write the ambiguous set over histogram words a sibling has
            // yet to read (or is still clearing).  Round 3 shipped without this barrier: the 256-thread small-block class -- the only
            // multi-wave class with the overlay -- then cut a row's top k at a threshold computed from clobbered counts: one or two
            // entries lost at the cut in ~1 build of 50 on config 4, now and then a garbage column and a wild store (the GPU memory
            // fault of profiles/r03_rocprofv3_stats_failure.txt; found by tools/race_hunt.py, profiles/r04_race_hunt.log).  prev_cnt is
            // team-uniform, so every wave takes the barrier.  (debug 262144 skips it: the regression test's negative control.)
            if (SHARE && T != WAVE && !(dbg & 262144)) team_sync<T>();
            // Round 5: when the cut bin AND everything above it (k - need composites) fit the set, they are copied out together and ranked ONCE --
            // the best k of that ranking ARE the row, in output order.  (Before: the bin's members ranked among themselves for the exact threshold,
            // a sweep for the survivors, the survivors ranked again: two sweeps and two rankings, at four vector instructions per compared element,
            // in classes that are bound by vector issue.)  The sweep then covers every candidate: what lies above the bin is not in the index list.
            const bool merged = !MP && !a.unordered && !(dbg & 16) && ((unsigned)a.k - need) + prev_cnt <= (unsigned)SEL_M;  // team-uniform
            const unsigned n_scan2 = (have_list && !merged) ? list_n : D;
            unsigned amb_n = 0u;  // (one-wave teams: the length of the set)
            for (unsigned base = 0; base < n_scan2; base += T) {  // scalar loop control, no divergent exits: claim_positions is a wave operation
              const unsigned idx = base + (unsigned)tl;
              unsigned long long key = 0ull;
              unsigned col = 0u;
              bool match = false;
              if (idx < n_scan2) {
                const unsigned t = (have_list && !merged) ? (unsigned)lst[idx] : idx;
                key = kk[t];
                col = (unsigned)((int)(tab[t] >> cb) - 1);
                if (merged) match = key != 0ull && (p < 8 ? (key >> shk) >= (thr_key >> shk) : (key > thr_key || (key == thr_key && (~col >> shc) >= (thr_ncol >> shc))));
                else match = key != 0ull && (p < 8 ? (key >> shk) == (thr_key >> shk) : (key == thr_key && (~col >> shc) == (thr_ncol >> shc)));
              }
              const unsigned pos = claim_positions<T>(match, &sel_res[1], amb_n);
              if (match) {
                amb_key[pos] = key;
                amb_col[pos] = col;
              }
            }
            team_sync<T>();
            // ... and rank them by counting; the need-th best composite is the exact threshold
            const unsigned m = T == WAVE ? amb_n : uni(sel_res[1]);
            if (merged) {  // m = (k - need) + the bin's members >= k: ranks 0 .. k - 1 are the row
              unsigned long long* selk = kk + D;                          // [k]: the row in output order (the survivors' arrays of the general path)
              unsigned* selc = reinterpret_cast<unsigned*>(selk + a.k);  // [k]
              for (unsigned base = 0; base < m; base += T) {
                const unsigned x = base + (unsigned)tl;
                if (x >= m) continue;
                const unsigned long long mk = amb_key[x];
                const int mc = (int)amb_col[x];
                const unsigned rank = rank_by_counting(amb_key, m, mk, mc, [&](unsigned u) { return (int)amb_col[u]; });
                if (rank < (unsigned)a.k) {
                  selk[rank] = mk;
                  selc[rank] = (unsigned)mc;
                }
              }
              team_sync<T>();
              unsigned tz = (unsigned)tl;
              URCCO_OPAQUE(tz);  // (the stores' per-lane addresses are formed here, not kept -- spilled -- across the row loop)
              for (unsigned t = tz; t < (unsigned)a.k; t += T) {
                out_idx[obase + t] = (int)selc[t];
                out_llr[obase + t] = __longlong_as_double((long long)selk[t]);
              }
              int kout = a.k;
              URCCO_OPAQUE(kout);  // (likewise: the hoisted vector copy of k was spilled, and reloaded behind the row's stores)
              if (tl == 0) a.out_count[i - a.item_lo] = kout;
              row_done = true;
              break;
            }
            for (unsigned base = 0; base < m; base += T) {
              const unsigned x = base + (unsigned)tl;
              if (x >= m) continue;
              const unsigned long long mk = amb_key[x];
              const int mc = (int)amb_col[x];
              const unsigned rank = rank_by_counting(amb_key, m, mk, mc, [&](unsigned u) { return (int)amb_col[u]; });
              if (rank + 1u == need) {
                sel_thr[0] = mk;
                sel_thr[1] = (unsigned long long)(~(unsigned)mc);
              }
            }
            team_sync<T>();
            thr_key = sel_thr[0];
            thr_ncol = (unsigned)sel_thr[1];
            break;
          }
        }
      }
      if (row_done) {  // team-uniform
        team_sync<T>();  // the table is re-zeroed by the next row
        continue;
      }
      if (tl == 0) *nsel = 0u;
      team_sync<T>();
      unsigned long long* selk = kk + D;                                  // [k]
      unsigned* selc = reinterpret_cast<unsigned*>(selk + a.k);          // [k]
      if (a.unordered && !MP) {
        // URCCO_FLAG_UNORDERED_ROWS: the top-k SET of the row, in whatever order the lanes claim output slots -- what
        // Mahout's computeSimilarities returns (a sparse vector has no score order; the reference sorts later, in
        // toStringMapRDD, package.scala:102).  No ranking pass.
        unsigned out_n = 0u;  // (one-wave teams: entries written)
        for (unsigned base = 0; base < D; base += T) {  // scalar loop control, no divergent exits
          const unsigned t = base + (unsigned)tl;
          unsigned long long key = 0ull;
          unsigned col = 0u;
          if (t < D) {
            key = kk[t];
            col = (unsigned)((int)(tab[t] >> cb) - 1);
          }
          const bool sel = key != 0ull && (key > thr_key || (key == thr_key && ~col >= thr_ncol));
          const unsigned pos = claim_positions<T>(sel, nsel, out_n);
          if (sel) {
            out_idx[obase + pos] = (int)col;
            out_llr[obase + pos] = __longlong_as_double((long long)key);
          }
        }
        team_sync<T>();
        if (tl == 0) a.out_count[i - a.item_lo] = (int)(T == WAVE ? out_n : *nsel);
        team_sync<T>();
        continue;
      }
      unsigned sel_n = 0u;  // (one-wave teams: survivors so far)
      for (unsigned base = 0; base < D; base += T) {  // scalar loop control, no divergent exits
        const unsigned t = base + (unsigned)tl;
        unsigned long long key = 0ull;
        unsigned col = 0u;
        if (t < D) {
          key = kk[t];
          col = (unsigned)((int)(tab[t] >> cb) - 1);
        }
        const bool sel = key != 0ull && (key > thr_key || (key == thr_key && ~col >= thr_ncol));
        const unsigned pos = claim_positions<T>(sel, nsel, sel_n);
        if (sel) {
          selk[pos] = key;
          selc[pos] = MP ? ((col << mp_s) | mp_q) : col;  // MP: the pass cut by the column inside the pass, the merge cuts by the full column
        }
      }
      team_sync<T>();
      const unsigned n = (dbg & 16) ? 0u : (T == WAVE ? sel_n : uni(*nsel));  // ablation 16: no ranking / output
      if (MP) {
        // merge the pass's <= k survivors into the running top k: every element of both lists is ranked over both (by counting),
        // the best k land in the other running buffer at their rank -- which is the output order
        const unsigned long long* rk = s_runk + run_cur * MP_KMAX;
        const unsigned* rc = s_runc + run_cur * MP_KMAX;
        unsigned long long* wk = s_runk + (run_cur ^ 1u) * MP_KMAX;
        unsigned* wc = s_runc + (run_cur ^ 1u) * MP_KMAX;
        const unsigned total = n_run + n;
        for (unsigned base = 0; base < total; base += T) {  // scalar loop control
          const unsigned x = base + (unsigned)tl;
          if (x >= total) continue;
          const unsigned long long mk = x < n_run ? rk[x] : selk[x - n_run];
          const int mc = (int)(x < n_run ? rc[x] : selc[x - n_run]);
          const unsigned rank = rank_by_counting(rk, n_run, mk, mc, [&](unsigned u) { return (int)rc[u]; }) +
                                rank_by_counting(selk, n, mk, mc, [&](unsigned u) { return (int)selc[u]; });
          if (rank < (unsigned)a.k) {
            wk[rank] = mk;
            wc[rank] = (unsigned)mc;
          }
        }
        n_run = total < (unsigned)a.k ? total : (unsigned)a.k;
        run_cur ^= 1u;
        team_sync<T>();
      } else {
      // Rank by counting.  Up to SEL_M survivors are put in order in LDS first (the arrays of the select's ambiguous set
      // are free again) and leave as contiguous stores: one element per lane scattered straight to its rank made every
      // store a partial-line write (measured 4x write amplification on the one-wave class).
      const bool staged = n <= (unsigned)SEL_M;
      for (unsigned base = 0; base < n; base += T) {
        const unsigned t = base + (unsigned)tl;
        if (t >= n) continue;
        const unsigned long long mk = selk[t];
        const int mc = (int)selc[t];
        const unsigned rank = rank_by_counting(selk, n, mk, mc, [&](unsigned u) { return (int)selc[u]; });
        if (staged) {
          amb_key[rank] = mk;
          amb_col[rank] = (unsigned)mc;
        } else {
          out_idx[obase + rank] = mc;
          out_llr[obase + rank] = __longlong_as_double((long long)mk);
        }
      }
      if (staged) {
        team_sync<T>();
        for (unsigned base = 0; base < n; base += T) {
          const unsigned t = base + (unsigned)tl;
          if (t >= n) continue;
          out_idx[obase + t] = (int)amb_col[t];
          out_llr[obase + t] = __longlong_as_double((long long)amb_key[t]);
        }
      }
      if (tl == 0) a.out_count[i - a.item_lo] = (int)n;
      }
    }
    if (MP) {
      team_sync<T>();
      if (++mp_q < (1u << mp_s)) goto mp_again;  // team-uniform
      const unsigned long long* rk = s_runk + run_cur * MP_KMAX;
      const unsigned* rc = s_runc + run_cur * MP_KMAX;
      for (unsigned t = (unsigned)tl; t < n_run; t += T) {
        out_idx[obase + t] = (int)rc[t];
        out_llr[obase + t] = __longlong_as_double((long long)rk[t]);
      }
      if (tl == 0) a.out_count[i - a.item_lo] = (int)n_run;
    }
    team_sync<T>();  // the table is re-zeroed by the next row
  }
  // statistics (only while stage timing is on): spread over CAND_SLOTS words -- every team of the grid adding to ONE address was
  // 16K serialised L2 atomics, +0.4 ms per launch
  if (a.cand && tl == 0 && cand_acc != 0ull) atomicAdd(&a.cand[(blockIdx.x * TEAMS + team) & (CAND_SLOTS - 1)], cand_acc);
}

// --------------------------------------------------------------------------------------------
// Micro rows (bin 0): <= 64 users and <= 64 cooccurrence pairs -- more than half of all item rows under a Zipf
// catalogue.  One pair per lane, an accumulator of four words per lane, at most one candidate per lane ranked by counting:
// no scans, no chunk loop, no selection passes, few registers (8 waves/SIMD).
// The row body (round 5; the rounds 1-4 form -- binary search per pair, compaction sweep, one ranking replica -- is
// profiles/r05_micro_v2_wave_llr_ab.log's "v1"): the class is bound by vector-instruction issue (~310 per row, 76 % of the issue
// slots of config 4's launches), so the row body is built around instruction count:
//  * a pair finds its user by a mark + prefix maximum (one LDS atomic, one LDS read, DPP steps, two lane gathers) instead of a
//    seven-step binary search over LDS;
//  * the lane whose insert CLAIMS a column owns the candidate: it reads the finished count from its own slot, clears the slot (the
//    accumulator is zero between rows without a zeroing pass) and scores it -- no compaction sweep over the slots;
//  * few candidates are ranked by two or four replicas of the candidates that each count every second (fourth) element.
// Round 6: S ROWS PER WAVE.  The average micro row of config 4 holds 33 pairs -- half of the wave's lanes idled through ~310 vector
// instructions --, 61 % of the class's rows hold <= 32 pairs and users and 33 % <= 16.  The binning pass sorts the class into three
// sub-lists (bin_off[NBINS + 1], [NBINS + 2]); rows of the first share a wave four at a time (S = 4, 16 lanes and 64 accumulator words
// each), rows of the second two at a time, the rest keep a wave to themselves.  A sub-row is a SEGMENT of L = 64 / S lanes: everything
// per row (ids, bounds, operands, counts, entropy, output base) is a per-lane value that is uniform inside a segment; prefix maxima
// stop at segment boundaries (the DPP row broadcasts that would cross them are left out), ballots are masked to the segment, lane
// gathers address inside it, and the ranking loops run to the LARGEST candidate count of the wave's rows over sentinel-padded lists.
// --------------------------------------------------------------------------------------------
constexpr int URCCO_OCC_MICRO = 8;  // blocks of four waves per CU the micro class is compiled for
template <int L> struct MicroGeom {
  static constexpr int S = WAVE / L;
  static constexpr int TW = 256 / S;                 // accumulator words per row
  static constexpr int LOG2TW = S == 1 ? 8 : (S == 2 ? 7 : 6);
  static constexpr int LIST = L + 4;                 // candidate columns / keys per row (with room for the padding of the ranking loops)
  static constexpr int MARKS = L + 2;                // pair -> user marks per row (offsets 0 .. L)
  static constexpr int ROW_WORDS = LIST + 2 * LIST + MARKS;  // (LIST even: the 64-bit keys behind the columns stay 8-byte aligned)
  static constexpr int WORDS = (256 + S * ROW_WORDS + 3) / 4 * 4;
};
// inclusive prefix maximum inside segments of L lanes (wave_inclusive_max without the row broadcasts that cross a segment boundary)
template <int L>
__device__ __forceinline__ unsigned seg_inclusive_max(unsigned v) {
  if (L == WAVE) return wave_inclusive_max(v);
  int x = (int)v;
  int y;
  y = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x = (unsigned)y > (unsigned)x ? y : x;
  y = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x = (unsigned)y > (unsigned)x ? y : x;
  if (L == 32) {
    y = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3: the upper half of either 32-lane segment
    x = (unsigned)y > (unsigned)x ? y : x;
  }
  return (unsigned)x;
}
// the largest number of set bits of m inside one segment of L lanes (wave-uniform m: scalar arithmetic)
template <int L>
__device__ __forceinline__ unsigned seg_max_popc(unsigned long long m) {
  if (L == WAVE) return (unsigned)__popcll(m);
  unsigned best = 0;
#pragma unroll
  for (int q = 0; q < WAVE / L; ++q) {
    const unsigned c = (unsigned)__popcll((m >> (q * L)) & ((1ull << L) - 1ull));
    best = c > best ? c : best;
  }
  return best;
}

template <int L, bool DBG, bool PK = false>
__global__ __launch_bounds__(256, (L == WAVE ? URCCO_OCC_MICRO : URCCO_OCC_MICRO - 2)) void cco_rows_micro_kernel(CcoArgs a) {
  if ((a.b_packed != nullptr && (a.pk_known != 0 || *a.pack_bad == 0)) != PK) return;  // grid-uniform: the other instantiation's turn (see cco_rows_kernel)
  using G = MicroGeom<L>;
  constexpr int S = G::S;
  const int dbg = DBG ? a.debug : 0;
  // (plain argument pointers here: with scalar registers of their own -- URCCO_OWN_GLOBAL_PTR, as in cco_rows_kernel -- this class spilled five
  // VECTOR registers to scratch and ran 3 % slower, profiles/r05_sgpr_diet_variants_ab.log)
  const int32_t* bin_rows = a.bin_rows;
  const int64_t* a_col_ptr = a.a_col_ptr;
  const int64_t* pstart = a.pstart;
  // B' with the columns' counts aboard while every count fits (CcoArgs::b_packed): the lane that claims a column has the column's count in the
  // very word it inserted -- no gather; else the plain column indices and one scattered count gather per candidate (wave-uniform)
  constexpr bool packed = PK;
  const int32_t* b_col_idx = PK ? a.b_packed : a.b_col_idx;
  const int cshift = 32 - a.count_bits;
  const unsigned colmask = PK ? (1u << cshift) - 1u : a.b_col_mask;
  const int32_t* cnt_a = a.cnt_a;
  const double* ent_a = a.ent_a;
  const double* xlx_tab = a.xlx_tab;
  const double* xlx_hi = a.xlx_hi;
  const double* col_ent = a.col_ent;
  int32_t* out_idx = a.out_idx;
  double* out_llr = a.out_llr;
  int32_t* out_count = a.out_count;
  long long n_users = a.n_users;
  constexpr int TEAMS = 256 / WAVE;
  // Wave LDS layout (words): [0,256) the S accumulators (zero between rows), then per row: candidate columns [LIST], their 64-bit keys [LIST],
  // the pair -> user marks [MARKS]
  __shared__ __attribute__((aligned(16))) unsigned s_tab[TEAMS * G::WORDS];
  const int team = threadIdx.x / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  const int seg = lane / L, sl = lane % L;  // this lane's row of the wave's S rows, and its position in the row's segment
  const unsigned long long seg_mask = L == WAVE ? ~0ull : (((1ull << (L % WAVE)) - 1ull) << (seg * L));
  unsigned* tab_w = s_tab + team * G::WORDS;
  unsigned* tab = tab_w + seg * G::TW;
  unsigned* cand = tab_w + 256 + seg * G::ROW_WORDS;
  unsigned long long* kkm = reinterpret_cast<unsigned long long*>(cand + G::LIST);
  unsigned* marks = cand + 3 * G::LIST;
  // this instantiation's sub-list of the class: rows of <= 16 / <= 32 / <= 64 pairs and users (bin_off, see the binning pass)
  const int list_start = L == 16 ? a.bin_off[0] : a.bin_off[NBINS + (L == 32 ? 1 : 2)];
  const int list_n = (L == 16 ? a.bin_off[NBINS + 1] : (L == 32 ? a.bin_off[NBINS + 2] : a.bin_off[1])) - list_start;
  const int n_groups = (list_n + S - 1) / S;  // a wave takes S consecutive rows of the list at a time
  const int total_teams = gridDim.x * TEAMS;
  const bool ident = a.n_cols_b <= G::TW;
  const int cb = a.count_bits;
  const unsigned cmask = (1u << cb) - 1u;
  const double xlx_n = *a.xlx_n;
  const bool use16 = *a.cnt16_bad == 0;
  unsigned long long cand_acc = 0ull;  // candidates scored by this wave (statistics)

  int li = blockIdx.x * TEAMS + team;
  if (li >= n_groups) return;  // (wave-level synchronisation only: a wave without rows may leave)
  // The row loop is a chain of dependent gathers (row id -> CSC bounds -> per-user operands -> B' columns -> column counts), and
  // the memory counter retires IN ORDER: a wave that waits for any load waits for every older one, and for every older store.  So
  //  * every link of the chain is issued at the TOP of a row, for the rows ahead -- the row id three rows ahead, the CSC bounds two,
  //    the operands one -- where the wait for this row's B' columns (the one unavoidable long wait) covers them all;
  //  * they are issued unconditionally (a branch with loads in it makes the compiler wait for everything where the paths join): list
  //    positions past the end re-read the last row (and are marked dead), lanes beyond a row's users its last user;
  //  * nothing is touched where it is loaded (a conversion next to a load is a wait for it), and everything is collected
  //    (URCCO_SETTLE) just before the row's output stores, so that the next row never waits behind those stores.
  // Round 4 found the rounds 1-3 form of this loop waiting three times per row for loads it had issued as "prefetches".
  const int stride = total_teams;
  auto row_at = [&](int g) { const int l = g * S + seg; return bin_rows[list_start + (l < list_n ? l : list_n - 1)]; };
  const unsigned* wp32 = reinterpret_cast<const unsigned*>(a.wp);  // low words: a row only uses differences (<= 64) between its own entries
  const unsigned* cnt_words = use16 ? reinterpret_cast<const unsigned*>(a.cnt_b16) : reinterpret_cast<const unsigned*>(a.cnt_b);  // the column counts, read a word at a time
  int i_cur = row_at(li);              // this row
  int i_n1 = row_at(li + stride);      // the next one: id ...
  int i_n2 = row_at(li + 2 * stride);  // (two ahead: id only)
  int64_t cs1 = a_col_ptr[i_n1], ce1 = a_col_ptr[i_n1 + 1];  // ... and CSC bounds
  // operands of the row about to be processed; wp[cs] is what the segment's first lane reads as its user's entry
  unsigned pf_w1, pf_wp;
  int64_t pf_start;
  int pf_ca;  // as loaded: widened where it is used
  double pf_ent;
  int n_cur;  // users of the row about to be processed
  {
    const int64_t cs0 = a_col_ptr[i_cur], ce0 = a_col_ptr[i_cur + 1];
    n_cur = (int)(ce0 - cs0);
    const int64_t pl = sl < n_cur ? cs0 + sl : ce0 - 1;
    pf_w1 = wp32[2 * ce0];
    pf_wp = wp32[2 * pl];
    pf_start = pstart[pl];
    pf_ca = cnt_a[i_cur];
    pf_ent = ent_a[i_cur];
  }
  // (collected here as at the end of every row: with a load still pending on ONE way into the loop header the compiler waits there
  // for everything in flight on every pass)
  URCCO_SETTLE(pf_w1); URCCO_SETTLE(pf_wp); URCCO_SETTLE(pf_start); URCCO_SETTLE(pf_ca); URCCO_SETTLE(pf_ent);
  URCCO_SETTLE(cs1); URCCO_SETTLE(ce1); URCCO_SETTLE(i_n2);
#pragma unroll
  for (int q = 0; q < 4; ++q) tab_w[lane + q * WAVE] = 0u;  // the accumulators: zero between rows (a candidate's owner clears its slot)
  for (; li < n_groups; li += stride) {  // each wave runs its own row loop: wave-level sync only
    const int i = i_cur;
    const bool live = S == 1 || li * S + seg < list_n;  // (the last group of a sub-list may be short: its dead segments hold no user and no pair)
    // this row's operands leave their registers ...
    const unsigned w0 = S == 1 ? wave_read_lane(pf_wp, 0) : wave_gather(pf_wp, (unsigned)(lane & ~(L - 1)));  // wp[cs]
    const unsigned total = live ? pf_w1 - w0 : 0u;  // <= L by the binning rule
    const bool owns_user = live && sl < n_cur;
    const long long ca = (long long)pf_ca;
    const double row_entropy = pf_ent;
    const int64_t my_start = owns_user ? pf_start : 0;
    const unsigned my_off = owns_user ? pf_wp - w0 : total;
    // ... and the rows ahead take them: id of row + 3, bounds of row + 2, operands of row + 1
    int i_n3 = row_at(li + 3 * stride);
    int64_t cs2 = a_col_ptr[i_n2], ce2 = a_col_ptr[i_n2 + 1];
    n_cur = (int)(ce1 - cs1);
    {
      const int64_t pl = sl < n_cur ? cs1 + sl : ce1 - 1;
      pf_w1 = wp32[2 * ce1];
      pf_wp = wp32[2 * pl];
      pf_start = pstart[pl];
      pf_ca = cnt_a[i_n1];
      pf_ent = ent_a[i_n1];
    }
    // ---- pair -> user: user u marks the first pair of its B' row with u (the LAST user of an offset is the one whose row is not
    // empty); a pair's user is the largest mark at or below it
    marks[sl] = 0u;
    wave_sync();
    if (owns_user) atomicMax(&marks[my_off], (unsigned)sl);  // my_off <= total <= L: marks has L + 2 words
    wave_sync();
    const unsigned o = seg_inclusive_max<L>(marks[sl]);
    const int64_t base_o = wave_gather64(my_start - (int64_t)my_off, (unsigned)(seg * L) + o);  // B' position of pair p of user o: base + p
    // ---- insert; the claiming lane owns the candidate
    unsigned slot = 0xffffffffu;
    unsigned jj = 0u;  // this lane's B' word: the column, and (packed) the column's count
    if ((unsigned)sl < total) {
      jj = (unsigned)b_col_idx[base_o + sl];
      if (!(dbg & 1)) {
        bool ok;
        slot = tab_insert_claim(tab, (jj & colmask) + 1u, cb, (unsigned)(G::TW - 1), 32 - G::LOG2TW, ident, &ok);
        if (!ok) atomicAdd(a.err, 1ull);
      }
    }
    wave_sync();
    const bool is_cand = slot != 0xffffffffu;
    const unsigned long long cand_mask = __ballot(is_cand);
    const unsigned D = (unsigned)__popcll(cand_mask & seg_mask);  // this row's candidates
    const unsigned D_max = seg_max_popc<L>(cand_mask);            // the wave's largest row (wave-uniform: loop bounds)
    cand_acc += (unsigned)__popcll(cand_mask);
    unsigned long long mk = 0ull;
    unsigned vv = 0u, cb_raw = 0u;
    if (is_cand) {  // the finished count leaves the accumulator, the slot is zero again, and the count gather is issued (ONE 4-byte load whichever
                    // width the counts have; nothing reads it before the block below)
      vv = tab[slot];
      tab[slot] = 0u;
      const int j = (int)(vv >> cb) - 1;
      if (!packed) cb_raw = cnt_words[use16 ? j >> 1 : j];
    }
    if (D + (unsigned)sl < (unsigned)G::LIST) {  // padding of the ranking loop's element list behind the row's candidates: sorts before nothing
      kkm[D + (unsigned)sl] = 0ull;
      cand[D + (unsigned)sl] = 0xffffffffu;
    }
    bool in_tables = true;
    if (is_cand) {
      const int j = (int)(vv >> cb) - 1;
      const long long k11 = (long long)(vv & cmask);
      const unsigned cbj = packed ? jj >> cshift : ((dbg & 512) ? (unsigned)k11 + 100u : (use16 ? ((j & 1) ? cb_raw >> 16 : cb_raw & 0xffffu) : cb_raw));
      in_tables = llr_operands_in_tables((unsigned)k11, ca, cbj, n_users, col_ent);
    }
    // Every operand of every candidate inside the tables (always, once the interaction cut has capped the counts): the wave takes the
    // straight-line form -- five table reads in flight together, no logarithm behind a divergent branch.  Wave-uniform test.
    const bool all_in_tables = !(dbg & 2) && __ballot(is_cand && !in_tables) == 0ull;
    if (is_cand) {
      const int j = (int)(vv >> cb) - 1;
      const long long k11 = (long long)(vv & cmask);
      if (!(a.exclude_self && j == i)) {
        const unsigned cbj = packed ? jj >> cshift : ((dbg & 512) ? (unsigned)k11 + 100u : (use16 ? ((j & 1) ? cb_raw >> 16 : cb_raw & 0xffffu) : cb_raw));
        const double llr = all_in_tables ? llr_from_tables(row_entropy, xlx_n, (unsigned)k11, (unsigned)ca, cbj, xlx_tab, xlx_hi, col_ent)
                                         : ((dbg & 2) ? (double)k11
                                                      : llr_of(row_entropy, xlx_n, k11, ca, (long long)cbj, n_users, xlx_tab, xlx_hi, col_ent));
        if (llr > 0.0 && (!a.has_min_llr || llr >= a.min_llr)) mk = (unsigned long long)__double_as_longlong(llr);
      }
      const unsigned pos = lanes_below(cand_mask & seg_mask);
      kkm[pos] = mk;
      cand[pos] = (unsigned)j;
    }
    wave_sync();
    const unsigned long long valid_mask = __ballot(mk != 0ull);
    const int n_valid = __popcll(valid_mask & seg_mask);
    const bool all_fit_k = (int)seg_max_popc<L>(valid_mask) <= a.k;  // wave-uniform
    auto settle_prefetch = [&]() {  // the next row's operands have had the score phase to arrive: collect them before the output stores
      URCCO_SETTLE(pf_w1); URCCO_SETTLE(pf_wp); URCCO_SETTLE(pf_start); URCCO_SETTLE(pf_ca); URCCO_SETTLE(pf_ent);
      URCCO_SETTLE(cs2); URCCO_SETTLE(ce2); URCCO_SETTLE(i_n3);
    };
    if (a.unordered && all_fit_k && !(dbg & 4)) {  // every candidate is emitted: no ranking needed (wave-uniform)
      const int64_t obase = ((int64_t)(i - a.item_lo)) * a.k;
      settle_prefetch();
      if (mk != 0ull) {
        const unsigned opos = lanes_below(valid_mask & seg_mask);
        out_idx[obase + opos] = (int)(vv >> cb) - 1;
        out_llr[obase + opos] = __longlong_as_double((long long)mk);
      }
      if (sl == 0 && live) out_count[i - a.item_lo] = n_valid;
    } else if (!(dbg & 4)) {
      const int64_t obase = ((int64_t)(i - a.item_lo)) * a.k;
      // candidates dense by lane, replicated while they fit twice / four times into the row's segment: replica q counts elements q, q + R, ...
      unsigned rank;
      unsigned long long rk;
      unsigned rc;
      unsigned ln = (unsigned)sl;
      URCCO_OPAQUE(ln);  // the per-lane LDS addresses of the three forms are computed here, not kept across the row loop
      if (D_max <= (unsigned)(L / 4)) {  // wave-uniform
        const unsigned c = ln & (unsigned)(L / 4 - 1);
        rk = kkm[c];
        rc = cand[c];
        rank = rank_by_counting_strided<4>(kkm, cand, ln / (unsigned)(L / 4), (D_max + 3u) & ~3u, rk, rc);
        rank += (unsigned)__shfl_xor((int)rank, L / 4);
        rank += (unsigned)__shfl_xor((int)rank, L / 2);
      } else if (D_max <= (unsigned)(L / 2)) {
        const unsigned c = ln & (unsigned)(L / 2 - 1);
        rk = kkm[c];
        rc = cand[c];
        rank = rank_by_counting_strided<2>(kkm, cand, ln / (unsigned)(L / 2), (D_max + 1u) & ~1u, rk, rc);
        rank += (unsigned)__shfl_xor((int)rank, L / 2);
      } else {
        rk = kkm[ln];
        rc = cand[ln];
        rank = rank_by_counting_strided<1>(kkm, cand, 0u, D_max, rk, rc);
      }
      // the row is put in order in LDS (every lane has its element in registers: in place) and leaves as contiguous stores
      wave_sync();
      const unsigned n_out = (unsigned)(n_valid < a.k ? n_valid : a.k);
      if ((unsigned)sl < D && rk != 0ull && rank < n_out) {
        cand[rank] = rc;
        kkm[rank] = rk;
      }
      wave_sync();
      settle_prefetch();
      if ((unsigned)sl < n_out) {
        out_idx[obase + sl] = (int)cand[sl];
        out_llr[obase + sl] = __longlong_as_double((long long)kkm[sl]);
      }
      if (sl == 0 && live) out_count[i - a.item_lo] = (int)n_out;
    } else {
      settle_prefetch();
    }
    i_cur = i_n1;
    i_n1 = i_n2;
    i_n2 = i_n3;
    cs1 = cs2;
    ce1 = ce2;
    wave_sync();
  }
  if (a.cand && lane == 0 && cand_acc != 0ull) atomicAdd(&a.cand[(blockIdx.x * TEAMS + team) & (CAND_SLOTS - 1)], cand_acc);
}

// --------------------------------------------------------------------------------------------
// Global-accumulator variant (bin 4): rows whose distinct columns cannot be bounded below an LDS table or
// whose counts overflow the packed entry.  One 1024-thread block per row; a dense int32 counter array per
// resident block (zero on entry, restored to zero by the claim walk), candidates spilled to global scratch,
// top-k by k strictly-descending argmax sweeps.  Correct for any row; only meant for the rare heavy ones.
// --------------------------------------------------------------------------------------------
constexpr int GB_THREADS = 1024;
constexpr int GSEL_K = 1024;  // survivors held in LDS by the radix-select form of the top-k

__global__ __launch_bounds__(GB_THREADS) void cco_rows_global_kernel(CcoArgs a) {
  constexpr int NW = GB_THREADS / WAVE;
  __shared__ unsigned long long s_pkey[2][NW];
  __shared__ int s_pcol[2][NW];
  __shared__ int s_ncand;
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_sel[4];
  __shared__ int s_nsel;
  __shared__ unsigned long long s_selk[GSEL_K];
  __shared__ unsigned s_selc[GSEL_K];
  const int bin = NBINS - 1;
  const int list_start = a.bin_off[bin];
  const int list_n = a.bin_off[bin + 1] - list_start;
  int32_t* cnt = a.g_counts + (int64_t)blockIdx.x * a.n_cols_b;
  unsigned long long* ckey = a.g_cand_key + (int64_t)blockIdx.x * a.n_cols_b;
  int32_t* ccol = a.g_cand_col + (int64_t)blockIdx.x * a.n_cols_b;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int G = 1 << a.g_log2;
  const int grp = threadIdx.x >> a.g_log2, gl = threadIdx.x & (G - 1), ngrp = GB_THREADS >> a.g_log2;
  const double xlx_n = *a.xlx_n;
  for (int li = blockIdx.x; li < list_n; li += gridDim.x) {  // block-uniform
    const int i = a.bin_rows[list_start + li];
    const int64_t cs = a.a_col_ptr[i], ce = a.a_col_ptr[i + 1];
    if (threadIdx.x == 0) s_ncand = 0;
    for (int64_t p = cs + grp; p < ce; p += ngrp) {
      const int64_t s = a.pstart[p], e = s + (a.wp[p + 1] - a.wp[p]);
      for (int64_t q = s + gl; q < e; q += G) atomicAdd(&cnt[(unsigned)a.b_col_idx[q] & a.b_col_mask], 1);
    }
    __syncthreads();
    const long long ca = a.cnt_a[i];
    const double row_entropy = a.ent_a[i];
    for (int64_t p = cs + grp; p < ce; p += ngrp) {
      const int64_t s = a.pstart[p], e = s + (a.wp[p + 1] - a.wp[p]);
      for (int64_t q = s + gl; q < e; q += G) {
        const int j = (int)((unsigned)a.b_col_idx[q] & a.b_col_mask);
        const long long k11 = atomicExch(&cnt[j], 0);  // exactly one lane claims (and clears) each column
        if (k11 > 0 && !(a.exclude_self && j == i)) {
          const long long cbj = a.cnt_b[j];
          const double llr = llr_from_entropies_tab(row_entropy, column_entropy_of(cbj, xlx_n, a.n_users, a.xlx_tab, a.xlx_hi, a.col_ent), xlx_n, k11, ca - k11, cbj - k11,
                                                    a.n_users - ca - cbj + k11, a.xlx_tab, a.n_users, a.xlx_hi);
          if (llr > 0.0 && (!a.has_min_llr || llr >= a.min_llr)) {
            const int pos = atomicAdd(&s_ncand, 1);
            ckey[pos] = (unsigned long long)__double_as_longlong(llr);
            ccol[pos] = j;
          }
        }
      }
    }
    __syncthreads();
    const int ncand = s_ncand;
    const int64_t obase = ((int64_t)(i - a.item_lo)) * a.k;
    if (a.k <= GSEL_K) {
      // ---- top-k: MSB-first radix select (8-bit digits, one 256-bin LDS histogram) of the k-th (llr, ~col) composite over
      // the candidates in global scratch -- at most twelve coalesced sweeps instead of k argmax sweeps (these rows have
      // tens of thousands of candidates: under config 5's skew the k = 50 sweeps were 60 % of a 330 us row)
      unsigned long long thr_key = 0ull;
      unsigned thr_ncol = 0u;
      if (ncand > a.k) {  // block-uniform
        unsigned need = (unsigned)a.k;
        if (a.col_bytes < 4) thr_ncol = 0xffffffffu << (8 * a.col_bytes);  // digits of ~col above the highest used byte are all ones
        const int first_col_pass = 8 + (3 - (a.col_bytes - 1));
        for (int p = 0; p < 12; ++p) {  // block-uniform trip count (the break below is on a value every thread agrees on)
          if (p >= 8 && p < first_col_pass) continue;
          if (threadIdx.x < 256) s_hist[threadIdx.x] = 0u;
          __syncthreads();
          const int shk = p < 8 ? 56 - 8 * p : 0, shc = p < 8 ? 0 : 24 - 8 * (p - 8);
          for (int t = threadIdx.x; t < ncand; t += GB_THREADS) {
            const unsigned long long key = ckey[t];
            bool match;
            unsigned dig;
            if (p < 8) {
              match = p == 0 || (key >> (shk + 8)) == (thr_key >> (shk + 8));
              dig = (unsigned)(key >> shk) & 255u;
            } else {
              const unsigned ncol = ~(unsigned)ccol[t];
              match = key == thr_key && (p == first_col_pass || (ncol >> (shc + 8)) == (thr_ncol >> (shc + 8)));
              dig = (ncol >> shc) & 255u;
            }
            if (match) atomicAdd(&s_hist[dig], 1u);
          }
          __syncthreads();
          if (threadIdx.x == 0) {  // the digit that holds the cut, scanning from the top
            unsigned above = 0u, d = 255u;
            for (;; --d) {
              if (above + s_hist[d] >= need || d == 0u) break;
              above += s_hist[d];
            }
            s_sel[0] = d;
            s_sel[1] = s_hist[d];
            s_sel[2] = above;
          }
          __syncthreads();
          const unsigned d = s_sel[0], cnt_d = s_sel[1];
          need -= s_sel[2];
          if (p < 8) thr_key |= (unsigned long long)d << shk;
          else thr_ncol |= d << shc;
          if (cnt_d == need) break;  // the whole bin is wanted: every composite >= the prefix (low bits zero) is selected
        }
      }
      if (threadIdx.x == 0) s_nsel = 0;
      __syncthreads();
      for (int t = threadIdx.x; t < ncand; t += GB_THREADS) {
        const unsigned long long key = ckey[t];
        const unsigned col = (unsigned)ccol[t];
        if (key > thr_key || (key == thr_key && ~col >= thr_ncol)) {
          const int pos = atomicAdd(&s_nsel, 1);
          if (a.unordered) {
            a.out_idx[obase + pos] = (int)col;
            a.out_llr[obase + pos] = __longlong_as_double((long long)key);
          } else {
            s_selk[pos] = key;
            s_selc[pos] = col;
          }
        }
      }
      __syncthreads();
      const int n = s_nsel;
      if (!a.unordered)
        for (int t = threadIdx.x; t < n; t += GB_THREADS) {  // rank by counting, straight to the output position
          const unsigned long long mk = s_selk[t];
          const int mc = (int)s_selc[t];
          const int rank = (int)rank_by_counting(s_selk, (unsigned)n, mk, mc, [&](unsigned u) { return (int)s_selc[u]; });
          a.out_idx[obase + rank] = mc;
          a.out_llr[obase + rank] = __longlong_as_double((long long)mk);
        }
      if (threadIdx.x == 0) a.out_count[i - a.item_lo] = n;
      __syncthreads();
      continue;
    }
    // k beyond the LDS survivor arrays: k strictly-descending argmax sweeps
    unsigned long long last_key = ~0ull;
    int last_col = -1;
    int emitted = 0;
    for (int r = 0; r < a.k; ++r) {
      unsigned long long wk = 0ull;
      int wc = 0x7fffffff;
      for (int t = threadIdx.x; t < ncand; t += GB_THREADS) {
        const unsigned long long kk = ckey[t];
        const int cc = ccol[t];
        // strictly after the previous winner in (llr desc, col asc) order
        if ((kk < last_key || (kk == last_key && cc > last_col)) && best_before(kk, cc, wk, wc)) {
          wk = kk;
          wc = cc;
        }
      }
#pragma unroll
      for (int m = WAVE / 2; m >= 1; m >>= 1) {
        const unsigned long long ok = shfl_xor_u64(wk, m);
        const int oc = __shfl_xor(wc, m);
        if (best_before(ok, oc, wk, wc)) {
          wk = ok;
          wc = oc;
        }
      }
      if (lane == 0) {
        s_pkey[r & 1][wv] = wk;
        s_pcol[r & 1][wv] = wc;
      }
      __syncthreads();
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) {
        const unsigned long long ok = s_pkey[r & 1][w2];
        const int oc = s_pcol[r & 1][w2];
        if (best_before(ok, oc, wk, wc)) {
          wk = ok;
          wc = oc;
        }
      }
      if (wk == 0ull) break;
      if (threadIdx.x == 0) {
        a.out_idx[obase + r] = wc;
        a.out_llr[obase + r] = __longlong_as_double((long long)wk);
      }
      last_key = wk;
      last_col = wc;
      ++emitted;
    }
    if (threadIdx.x == 0) a.out_count[i - a.item_lo] = emitted;
    __syncthreads();
  }
}

// resident blocks per CU of each LDS-accumulator kernel (registers / LDS decide), so that the persistent grids fill
// the chip exactly once
static int blocks_per_cu(int bin) {
  static std::atomic<int> cache[7];  // zero-initialised; a racing first call computes the same value twice
  if (cache[bin].load(std::memory_order_relaxed) == 0) {
    int n = 0;
    hipError_t e = hipErrorUnknown;
    if (bin == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (cco_rows_micro_kernel<WAVE, false, false>), 256, 0);
    if (bin == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cco_rows_kernel<64, E0, URCCO_U_WAVE>, 256, 0);
    if (bin == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cco_rows_kernel<256, E1S, URCCO_U_BS>, 256, 0);
    if (bin == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cco_rows_kernel<256, E1, URCCO_U_B>, 256, 0);
    if (bin == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cco_rows_kernel<512, E2S, URCCO_U_H>, 512, 0);
    if (bin == 5) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cco_rows_kernel<1024, E2, URCCO_U_C>, 1024, 0);
    if (bin == 6) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cco_rows_kernel<1024, E2, URCCO_U_C, true>, 1024, 0);
    cache[bin].store((e == hipSuccess && n > 0) ? n : 1, std::memory_order_relaxed);
  }
  return cache[bin].load(std::memory_order_relaxed);
}

hipError_t launch_cco_rows_bin(hipStream_t st, int n_cu, const CcoArgs& args, int bin, int32_t n_rows) {
  // Persistent grids sized to the chip; each kernel reads its own row list length from bin_off on the device,
  // so no host synchronisation sits between binning and the SpGEMM.
  // Several times as many blocks as fit the chip (tunable per class through URCCO_GRID_FACTORS="f0,f1,..,f5" for measurements):
  // the later ones start as blocks of the first wave retire, which evens out the classes' ragged ends -- rows are dealt out by a
  // static stride, so a block's share of heavy rows is luck -- and lets short kernels of the other event types' streams in: a grid
  // that exactly fills the chip locks them out until it ends (measured: single-block kernels of another stream waited 0.2 ms).
  // Round 2 (config 3): 2x, and 3x / 4x / 8x measured no better.  Round 5 (config 4 / 5, after the row kernels had lost a third of
  // their time): 8x for the four big classes and 4x for the 512/1024-thread classes = -0.55 / -0.8 ms per build, every class's own
  // time included (profiles/r05_grid_factors_ab.log); bounded by one row loop per 32 item rows of the build (small builds and the ranks of a sharded
  // build keep the 2x: a row loop's start-up -- three rows of prefetches -- is not free; at an eighth of config 4's rows 4x measured 0.12 ms
  // per rank slower than 2x).
  struct Factors {  // initialised once, thread-safely: every event type's enqueueing thread comes through here in the first build
    int f[7] = {8, 8, 8, 8, 4, 4, 2};
    Factors() {
      if (const char* e = getenv("URCCO_GRID_FACTORS")) {
        int v[6];
        if (sscanf(e, "%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6)
          for (int b = 0; b < 6; ++b)
            if (v[b] >= 1 && v[b] <= 64) f[b] = v[b];
      }
    }
  };
  static const Factors factors;
  const int* factor = factors.f;
  // the micro class's sub-lists of shared waves (two / four rows per wave and pass) hold a third of the class's passes each at most: smaller grids,
  // so that a wave still runs several passes behind its start-up (three rows of prefetches).  URCCO_MICRO_GRID="f32,f16" for measurements.
  struct MicroFactors {
    int f32 = 3, f16 = 2;
    MicroFactors() {
      if (const char* e = getenv("URCCO_MICRO_GRID")) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && a <= 64 && b >= 1 && b <= 64) { f32 = a; f16 = b; }
      }
    }
  };
  static const MicroFactors micro_factors;
  auto grid = [&](int b, int f = 0) {
    const long long fill = (long long)n_cu * blocks_per_cu(b);  // blocks resident at once
    const long long teams = b <= 1 ? 4 : 1;                       // row loops per block (micro / one-wave classes: four one-wave teams)
    long long cap = ((long long)n_rows + teams * 32 - 1) / (teams * 32);
    if (cap < 2 * fill) cap = 2 * fill;  // (as rounds 2-4)
    long long blocks = fill * (f > 0 ? f : factor[b]);
    if (blocks > cap) blocks = cap;
    return dim3((unsigned)blocks);
  };
  const bool dbgk = (args.debug & (1 | 2 | 4 | 8 | 16 | 512 | 131072 | 262144)) != 0;  // the ablation / test switches live in the DBG instantiations only
  // A B' with counts aboard: BOTH instantiations are enqueued -- whether the counts fit is a device-side fact, the one whose turn it is not returns at
  // once.  The DBG instantiations exist for the plain form only (the ablation switches price the count gather among other things).
  CcoArgs plain = args;
  plain.b_packed = nullptr;
  const bool both = args.b_packed != nullptr && !dbgk && !args.pk_known;  // (pk_known: the host knows the counts are aboard -- only that instantiation)
#define URCCO_LAUNCH_ROWS(TT, EE, UU, MPF, GRID, BLK, BINARG)                                                                      \
  do {                                                                                                                           \
    if (dbgk) hipLaunchKernelGGL((cco_rows_kernel<TT, EE, UU, MPF, true, false>), GRID, dim3(BLK), 0, st, plain, BINARG);          \
    else {                                                                                                                       \
      if (both || args.pk_known) hipLaunchKernelGGL((cco_rows_kernel<TT, EE, UU, MPF, false, true>), GRID, dim3(BLK), 0, st, args, BINARG); \
      if (!args.pk_known) hipLaunchKernelGGL((cco_rows_kernel<TT, EE, UU, MPF, false, false>), GRID, dim3(BLK), 0, st, args, BINARG); \
    }                                                                                                                            \
  } while (0)
#define URCCO_LAUNCH_MICRO(LL, GRID)                                                                                  \
  do {                                                                                                                \
    if (dbgk) hipLaunchKernelGGL((cco_rows_micro_kernel<LL, true, false>), GRID, dim3(256), 0, st, plain);              \
    else {                                                                                                            \
      if (both || args.pk_known) hipLaunchKernelGGL((cco_rows_micro_kernel<LL, false, true>), GRID, dim3(256), 0, st, args); \
      if (!args.pk_known) hipLaunchKernelGGL((cco_rows_micro_kernel<LL, false, false>), GRID, dim3(256), 0, st, args);  \
    }                                                                                                                 \
  } while (0)
  switch (bin) {
    case 0:  // the class's three sub-lists, the rows that keep a wave to themselves first (each kernel reads its own list bounds on the device)
      URCCO_LAUNCH_MICRO(64, grid(0));
      if (micro_split_for(n_rows)) {  // (the same rule as the binning pass: without the split the two sub-lists are empty)
        URCCO_LAUNCH_MICRO(32, grid(0, micro_factors.f32));
        URCCO_LAUNCH_MICRO(16, grid(0, micro_factors.f16));
      }
      break;
    case 1: URCCO_LAUNCH_ROWS(64, E0, URCCO_U_WAVE, false, grid(1), 256, 1); break;
    case 2: URCCO_LAUNCH_ROWS(256, E1S, URCCO_U_BS, false, grid(2), 256, 2); break;
    case 3: URCCO_LAUNCH_ROWS(256, E1, URCCO_U_B, false, grid(3), 256, 3); break;
    case 4: URCCO_LAUNCH_ROWS(512, E2S, URCCO_U_H, false, grid(4), 512, 4); break;
    case 5: URCCO_LAUNCH_ROWS(1024, E2, URCCO_U_C, false, grid(5), 1024, 5); break;
    default:
      if (args.g_blocks > 0) hipLaunchKernelGGL(cco_rows_global_kernel, dim3((unsigned)args.g_blocks), dim3(GB_THREADS), 0, st, plain);
      else URCCO_LAUNCH_ROWS(1024, E2, URCCO_U_C, true, grid(6), 1024, 6);
      break;
  }
#undef URCCO_LAUNCH_ROWS
#undef URCCO_LAUNCH_MICRO
  return hipGetLastError();
}

// stats[1 + 3 * NBINS + bin] = indicator entries emitted by the rows of each bin (profiling aid, deterministic block reduce)
__global__ __launch_bounds__(256) void bin_out_stats_kernel(const int32_t* __restrict__ bin_rows, const int32_t* __restrict__ bin_off,
                                                            int32_t item_lo, const int32_t* __restrict__ out_count, const unsigned long long* __restrict__ cand,
                                                            int64_t* __restrict__ stats) {
  __shared__ long long s_wave[SCAN_THREADS / WAVE];
  const int bin = blockIdx.x;
  if (bin == 0 && blockIdx.y == 0 && threadIdx.x == 0 && cand) {  // distinct (row, column) candidates scored: the slots the row kernels added to
    long long c = 0;
    for (int q = 0; q < CAND_SLOTS; ++q) c += (long long)cand[q];
    stats[2 + 4 * NBINS] = c;
  }
  long long v = 0;
  for (int t = bin_off[bin] + blockIdx.y * 256 + threadIdx.x; t < bin_off[bin + 1]; t += 256 * gridDim.y) v += out_count[bin_rows[t] - item_lo];
  long long tot;
  block_exclusive_scan(v, s_wave, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd((unsigned long long*)&stats[1 + 3 * NBINS + bin], (unsigned long long)tot);
}
hipError_t launch_bin_out_stats(hipStream_t st, const int32_t* bin_rows, const int32_t* bin_off, int32_t item_lo, const int32_t* out_count,
                                const unsigned long long* cand, int64_t* stats) {
  hipLaunchKernelGGL(bin_out_stats_kernel, dim3(NBINS, 128), dim3(256), 0, st, bin_rows, bin_off, item_lo, out_count, cand, stats);
  return hipGetLastError();
}

// ============================================================================================
// Strided top-k rows -> CSR
// ============================================================================================
__global__ __launch_bounds__(256) void compact_indicators_kernel(int32_t n_rows, int32_t k, const int32_t* __restrict__ count,
                                                                 const int32_t* __restrict__ idx, const double* __restrict__ llr,
                                                                 const int64_t* __restrict__ row_ptr, int32_t* __restrict__ out_idx,
                                                                 double* __restrict__ out_llr) {
  const int64_t total = (int64_t)n_rows * k;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / k;
    const int s = (int)(t - r * k);
    if (s < count[r]) {
      const int64_t o = row_ptr[r] + s;
      out_idx[o] = idx[t];
      out_llr[o] = llr[t];
    }
  }
}

hipError_t launch_compact_indicators(hipStream_t st, int32_t n_rows, int32_t k, const int32_t* count, const int32_t* idx,
                                     const double* llr, const int64_t* row_ptr, int32_t* out_idx, double* out_llr) {
  const int64_t total = (int64_t)n_rows * k;
  if (total == 0) return hipSuccess;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(compact_indicators_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, k, count, idx, llr, row_ptr, out_idx, out_llr);
  return hipGetLastError();
}

// ============================================================================================
// Work-balanced item ranges: bounds[p] = first item whose exclusive work prefix >= p * total / n_parts
// ============================================================================================
__global__ void partition_kernel(int32_t n_items, const int64_t* __restrict__ work_prefix, int32_t n_parts, int32_t* __restrict__ bounds) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_parts) return;
  if (p == 0) { bounds[0] = 0; return; }
  if (p == n_parts) { bounds[p] = n_items; return; }
  const long long total = work_prefix[n_items];
  const long long target = (total / n_parts) * p + ((total % n_parts) * p) / n_parts;  // floor(total * p / n_parts) without overflow
  int lo = 0, hi = n_items;  // first i with prefix[i] >= target
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (work_prefix[mid] >= target) hi = mid; else lo = mid + 1;
  }
  bounds[p] = lo;
}

hipError_t launch_partition(hipStream_t st, int32_t n_items, const int64_t* work_prefix, int32_t n_parts, int32_t* bounds) {
  hipLaunchKernelGGL(partition_kernel, dim3(1), dim3(64 * ((n_parts + 64) / 64)), 0, st, n_items, work_prefix, n_parts, bounds);
  return hipGetLastError();
}

// ============================================================================================
// Multi-GPU exchange helpers.
//  * Row lengths of a CSR shard -- what travels in the all-gather-v next to the column indices; the receiver rebuilds row_ptr
//    with one scan over the concatenated lengths.  Written twice, as int32 and as uint16: the record a rank publishes about its
//    down-sampled shard is {rows, nnz, rows whose length does not fit 16 bits}, and the host sends the 16-bit copy when that
//    last figure is zero on every rank.
//  * CSC fragments of the primary: every rank transposes ITS user shard (all columns, shard-local user ids); the slice of
//    that CSC belonging to the item range of rank q is contiguous, so it is sent as it lies (entries + 16-bit column lengths)
//    and rank q merges the W fragments it receives into the CSC of its range -- no rank ever walks the whole of A' to pick
//    its columns out.
// ============================================================================================
__global__ __launch_bounds__(256) void row_lengths_kernel(int64_t n_rows, const int64_t* __restrict__ rp, int32_t* __restrict__ len,
                                                          unsigned short* __restrict__ len16, int64_t* __restrict__ sizes) {
  int over = 0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
    const int64_t l = rp[r + 1] - rp[r];
    len[r] = (int32_t)l;
    if (len16) {
      len16[r] = (unsigned short)l;
      over += l > 0xffff ? 1 : 0;
    }
  }
  if (sizes && over) atomicAdd(reinterpret_cast<unsigned long long*>(sizes + 2), (unsigned long long)over);
  if (sizes && blockIdx.x == 0 && threadIdx.x == 0) {
    sizes[0] = n_rows;
    sizes[1] = rp[n_rows];
  }
}
hipError_t launch_row_lengths(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, int32_t* len, unsigned short* len16, int64_t* sizes) {
  int64_t blocks = (n_rows + 255) / 256;
  const int64_t cap = (int64_t)n_cu * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (sizes) {
    hipError_t e = hipMemsetAsync(sizes, 0, sizeof(int64_t) * EXCH_SIZES, st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(row_lengths_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, row_ptr, len, len16, sizes);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void counts_over_limit_kernel(const int32_t* __restrict__ counts, int64_t n, unsigned limit, unsigned long long* __restrict__ out) {
  int over = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) over += (unsigned)counts[i] >= limit ? 1 : 0;
  if (over) atomicAdd(out, (unsigned long long)over);
}
hipError_t launch_counts_over_limit(hipStream_t st, int n_cu, const int32_t* counts, int64_t n, int32_t count_bits, int64_t* out) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(int64_t), st);
  if (e != hipSuccess || n <= 0) return e;
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)n_cu * 8;
  if (blocks > cap) blocks = cap;
  const unsigned limit = count_bits >= 16 ? 65536u : (1u << count_bits);
  hipLaunchKernelGGL(counts_over_limit_kernel, dim3((unsigned)blocks), dim3(256), 0, st, counts, n, limit, reinterpret_cast<unsigned long long*>(out));
  return hipGetLastError();
}

struct LoadU16 {
  const unsigned short* p;
  __device__ __forceinline__ long long operator()(int64_t i) const { return p[i]; }
  __device__ __forceinline__ void load8(int64_t i, long long* x) const {
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      const uint4 a = *reinterpret_cast<const uint4*>(p + i);
      x[0] = a.x & 0xffffu; x[1] = a.x >> 16; x[2] = a.y & 0xffffu; x[3] = a.y >> 16;
      x[4] = a.z & 0xffffu; x[5] = a.z >> 16; x[6] = a.w & 0xffffu; x[7] = a.w >> 16;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = p[i + q];
    }
  }
};
hipError_t launch_scan_u16(hipStream_t st, const unsigned short* in, int64_t n, int64_t* out, int64_t* tile_sums) {
  return launch_scan(st, LoadU16{in}, n, out, tile_sums);
}

// --------------------------------------------------------------------------------------------
// Row-filtered exchange of the down-sampled matrices (round 4).  Rank q multiplies the CSC of ITS item range of A' with B': it
// reads B' row u only for users that hold an item of that range -- ~40 % of all users at 8 ranks.  Which users those are is known
// to the rank that owns them (it holds their rows of A' and every rank holds the bounds): no request travels.  Per user a mask
// of the ranks that need it; per event type the shard's rows are then packed per destination and sent by all-to-all-v -- a row
// nobody's range touches is not sent at all, and a destination receives a length of 0 for a row it does not need (the rebuilt
// matrix keeps every user's row, empty where it was not sent: the SpGEMM never looks those up).
//   need_mask        mask[u] bit q: row u of A' (shard) holds a column of [bounds[q], bounds[q + 1])            (W <= 64)
//   masked_lengths   mlen[q * n + u] = mask[u] bit q ? len(row u of B') : 0     -> scan -> where every sent row starts, and
//   peer_totals      to_nnz[q] = column indices destined for rank q
//   pack_rows        the rows, destination-major, in user order (16 lanes per user)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void need_mask_kernel(int64_t n_rows, const int64_t* __restrict__ a_rp, const int32_t* __restrict__ a_ci,
                                                        const int32_t* __restrict__ bounds, int world, unsigned long long* __restrict__ mask) {
  __shared__ int s_b[65];
  for (int t = threadIdx.x; t <= world; t += 256) s_b[t] = bounds[t];
  __syncthreads();
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n_rows; u += (int64_t)gridDim.x * 256) {
    unsigned long long m = 0ull;
    for (int64_t p = a_rp[u]; p < a_rp[u + 1]; ++p) {
      const int c = a_ci[p];
      int lo = 0, hi = world;  // last q with bounds[q] <= c  (bounds[0] = 0 <= c < bounds[world])
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_b[mid] <= c) lo = mid; else hi = mid;
      }
      m |= 1ull << lo;
    }
    mask[u] = m;
  }
}
hipError_t launch_need_mask(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* a_row_ptr, const int32_t* a_col_idx, const int32_t* bounds, int world,
                            unsigned long long* mask) {
  if (world > 64) return hipErrorInvalidValue;
  if (n_rows == 0) return hipSuccess;
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
  hipLaunchKernelGGL(need_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, a_row_ptr, a_col_idx, bounds, world, mask);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void masked_lengths_kernel(int64_t n_rows, const int64_t* __restrict__ rp, const unsigned long long* __restrict__ mask, int world,
                                                             int32_t* __restrict__ mlen) {
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n_rows; u += (int64_t)gridDim.x * 256) {
    const int32_t l = (int32_t)(rp[u + 1] - rp[u]);
    const unsigned long long m = mask[u];
    for (int q = 0; q < world; ++q) mlen[(int64_t)q * n_rows + u] = ((m >> q) & 1ull) ? l : 0;
  }
}
__global__ void peer_totals_kernel(int world, int64_t n_rows, const int64_t* __restrict__ off, int64_t* __restrict__ to_nnz) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < world) to_nnz[q] = off[(int64_t)(q + 1) * n_rows] - off[(int64_t)q * n_rows];
}
// mlen [world * n_rows] int32, off [world * n_rows + 1] int64 (exclusive scan of mlen), to_nnz [world]; tile_sums: scan scratch for world * n_rows values
hipError_t launch_masked_lengths(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const unsigned long long* mask, int world, int32_t* mlen, int64_t* off,
                                 int64_t* tile_sums, int64_t* to_nnz) {
  if (n_rows > 0) {
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
    hipLaunchKernelGGL(masked_lengths_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, row_ptr, mask, world, mlen);
  }
  hipError_t e = launch_scan_i32(st, mlen, (int64_t)world * n_rows, off, tile_sums);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(peer_totals_kernel, dim3((unsigned)((world + 63) / 64)), dim3(64), 0, st, world, n_rows, off, to_nnz);
  return hipGetLastError();
}
// One block per (tile of PK_ROWS consecutive rows, destination): the tile's slice of the destination's offsets (the scanned masked
// lengths: a row the destination does not need has length 0) and its row starts are staged in LDS, then the threads walk the tile's OUTPUT
// entries -- consecutive lanes write consecutive words of the send buffer and, inside a row, read consecutive words of the shard; an
// entry finds its row by a binary search of the staged offsets.  Two memory round trips per block, whatever the rows' lengths.
// (Round 4 walked a row's destinations one after the other inside a 16-lane group, round 5's first form gave every (row, destination)
// pair eight lanes: 1.3 and 1.45 ms per rank of config 4 at 8 ranks -- one row at a time per group, three dependent loads each:
// profiles/r05_emulated_ranks_w8_kernel_table_config4.txt.)
constexpr int PK_ROWS = 512;
__global__ __launch_bounds__(256) void pack_rows_kernel(int64_t n_rows, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci, int world,
                                                        const int64_t* __restrict__ off, int32_t* __restrict__ pack) {
  __shared__ unsigned s_o[PK_ROWS + 1];  // offsets relative to the tile's first output entry
  __shared__ long long s_src[PK_ROWS];   // where the row starts in the shard
  const int q = blockIdx.y;
  const int64_t u0 = (int64_t)blockIdx.x * PK_ROWS;
  const int nr = (int)(n_rows - u0 < PK_ROWS ? n_rows - u0 : PK_ROWS);
  const int64_t* oq = off + (int64_t)q * n_rows + u0;  // off holds world * n_rows + 1 entries: oq[nr] exists for the last tile of the last destination too
  const int64_t base = oq[0];
  for (int r = threadIdx.x; r <= nr; r += 256) {
    s_o[r] = (unsigned)(oq[r] - base);
    if (r < nr) s_src[r] = rp[u0 + r];
  }
  __syncthreads();
  const unsigned n_out = s_o[nr];
  for (unsigned e0 = threadIdx.x; e0 < n_out; e0 += 4 * 256) {  // four entries per thread and round: their gathers travel together
    int32_t v[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const unsigned e = e0 + (unsigned)x * 256u;
      v[x] = 0;
      if (e < n_out) {
        int lo = 0, hi = nr;  // last r in [0, nr) with s_o[r] <= e  (s_o[0] = 0 <= e < s_o[nr])
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (s_o[mid] <= e) lo = mid; else hi = mid;
        }
        v[x] = ci[s_src[lo] + (long long)(e - s_o[lo])];
      }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const unsigned e = e0 + (unsigned)x * 256u;
      if (e < n_out) pack[base + e] = v[x];
    }
  }
}
hipError_t launch_pack_rows(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, const unsigned long long* mask, int world,
                            const int64_t* off, int32_t* pack) {
  (void)n_cu; (void)mask;  // (the masked lengths behind `off` already say which rows travel)
  if (n_rows == 0 || world <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((n_rows + PK_ROWS - 1) / PK_ROWS), (unsigned)world), dim3(256), 0, st, n_rows, row_ptr, col_idx, world, off, pack);
  return hipGetLastError();
}

// rec[0 .. W] = entry offsets of the local CSC at the range bounds, rec[W + 1 .. 2W + 1] = the bounds,
// rec[2W + 2] = local column lengths that do not fit 16 bits
__global__ void frag_record_kernel(int32_t world, const int32_t* __restrict__ bounds, const int64_t* __restrict__ l_cp, const int32_t* __restrict__ bad,
                                   int64_t* __restrict__ rec) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > world) return;
  const int32_t b = bounds[p];
  rec[p] = l_cp[b];
  rec[world + 1 + p] = b;
  if (p == 0) rec[2 * world + 2] = bad ? *bad : 0;
}
hipError_t launch_frag_record(hipStream_t st, int32_t world, const int32_t* bounds, const int64_t* l_cp, const int32_t* bad, int64_t* rec) {
  hipLaunchKernelGGL(frag_record_kernel, dim3((unsigned)((world + 64) / 64)), dim3(64), 0, st, world, bounds, l_cp, bad, rec);
  return hipGetLastError();
}

// Merge of the fragments received for the item range [lo, lo + n_range): lens[p * n_range + j] = length of column lo + j in the
// shard of rank p, src_off = exclusive scan of lens in that (rank-major) order = where that run starts in `ents` (the fragments
// lie one behind the other in rank order); a_cp = CSC pointers of the range (scan of the all-reduced column counts).  The
// shard-local user ids become global ones: + the rows of the ranks before p (sizes[EXCH_SIZES * q] = rows of rank q).  Runs are
// placed in rank order, so a column ascends in the user id if the fragments did.  16 lanes per column.
constexpr int FRAG_LANES = 16;
template <typename L>
__global__ __launch_bounds__(256) void frag_place_kernel(int32_t world, int32_t lo, int32_t n_range, const L* __restrict__ lens,
                                                         const int64_t* __restrict__ src_off, const int32_t* __restrict__ ents,
                                                         const int64_t* __restrict__ a_cp, const int64_t* __restrict__ sizes,
                                                         int32_t* __restrict__ a_ri) {
  const int gl = threadIdx.x & (FRAG_LANES - 1);
  const int64_t groups = (int64_t)gridDim.x * (256 / FRAG_LANES);
  for (int64_t j = (int64_t)blockIdx.x * (256 / FRAG_LANES) + threadIdx.x / FRAG_LANES; j < n_range; j += groups) {
    int64_t dst = a_cp[lo + j];
    int64_t base = 0;
    for (int p = 0; p < world; ++p) {
      const int64_t at = (int64_t)p * n_range + j;
      const int64_t n = (int64_t)lens[at];
      const int64_t src = src_off[at];
      for (int64_t t = gl; t < n; t += FRAG_LANES) a_ri[dst + t] = (int32_t)(ents[src + t] + base);
      dst += n;
      base += sizes[(int64_t)EXCH_SIZES * p];
    }
  }
}
hipError_t launch_frag_place(hipStream_t st, int n_cu, int32_t world, int32_t lo, int32_t n_range, const void* lens, int wire16, const int64_t* src_off,
                             const int32_t* ents, const int64_t* a_cp, const int64_t* sizes, int32_t* a_ri) {
  if (n_range <= 0) return hipSuccess;
  int64_t blocks = ((int64_t)n_range + (256 / FRAG_LANES) - 1) / (256 / FRAG_LANES);
  if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
  if (wire16)
    hipLaunchKernelGGL((frag_place_kernel<unsigned short>), dim3((unsigned)blocks), dim3(256), 0, st, world, lo, n_range,
                       static_cast<const unsigned short*>(lens), src_off, ents, a_cp, sizes, a_ri);
  else
    hipLaunchKernelGGL((frag_place_kernel<int32_t>), dim3((unsigned)blocks), dim3(256), 0, st, world, lo, n_range, static_cast<const int32_t*>(lens),
                       src_off, ents, a_cp, sizes, a_ri);
  return hipGetLastError();
}

// ============================================================================================
// Boundary checks of a caller-supplied CSR (the host level hands over JVM arrays): row_ptr monotone inside [0, nnz],
// column indices inside [0, n_cols) and strictly increasing inside a row (the precondition of every kernel above:
// an out-of-range column would become an out-of-bounds atomic, a duplicate would inflate the counts).  2^g lanes walk a
// row; err[0] counts violations.  row_ptr is checked before col_idx is touched, so a corrupt row_ptr cannot send the
// walk out of bounds.
// ============================================================================================
__global__ __launch_bounds__(256) void validate_csr_kernel(int64_t n_rows, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci, int64_t nnz,
                                                           int32_t n_cols, int g_log2, int64_t rp0, unsigned long long* __restrict__ err) {
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t groups_per_block = 256 >> g_log2;
  unsigned bad = 0;
  for (int64_t r = (int64_t)blockIdx.x * groups_per_block + (threadIdx.x >> g_log2); r < n_rows; r += (int64_t)gridDim.x * groups_per_block) {
    const int64_t s = rp[r] - rp0, e = rp[r + 1] - rp0;
    if (s < 0 || e < s || e > nnz) {
      bad += gl == 0;
      continue;
    }
    for (int64_t p = s + gl; p < e; p += G) {
      const int j = ci[p];
      if (j < 0 || j >= n_cols || (p > s && ci[p - 1] >= j)) ++bad;
    }
  }
  if (bad) atomicAdd(err, (unsigned long long)bad);
}
hipError_t launch_validate_csr(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz, int32_t n_cols,
                               int g_log2, int64_t rp0, unsigned long long* err) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t gpb = 256 >> g_log2;
  int64_t blocks = (n_rows + gpb - 1) / gpb;
  const int64_t cap = (int64_t)n_cu * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(validate_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, row_ptr, col_idx, nnz, n_cols, g_log2, rp0, err);
  return hipGetLastError();
}

// p[i] -= delta (a row_ptr slice of a user shard re-based to start at 0); p2 (nullable): out[i] = p[i] + add (indicator row_ptr
// slices of the GPUs of one process re-based onto the concatenated output)
__global__ __launch_bounds__(256) void rebase_kernel(int64_t* __restrict__ p, int64_t n, int64_t delta) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] -= delta;
}
// One device word -> host-mapped pinned memory by a STORE of the GPU, not by a copy: a D2H copy of eight bytes queues on the copy engine behind
// whatever results another event type is bringing over (round 6, host level: config 4's last event type waited 39 ms for its boundary check).
__global__ void publish_word_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst_mapped) { *dst_mapped = *src; }
hipError_t launch_publish_word(hipStream_t st, const unsigned long long* src, unsigned long long* dst_mapped) {
  hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(1), 0, st, src, dst_mapped);
  return hipGetLastError();
}
hipError_t launch_rebase_i64(hipStream_t st, int n_cu, int64_t* p, int64_t n, int64_t delta) {
  if (n <= 0 || delta == 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)n_cu * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(rebase_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n, delta);
  return hipGetLastError();
}

// ============================================================================================
// PopModel.calcPopular / calcTrending / calcHot (reference src/main/scala/PopModel.scala:113-179): per-item counts of the
// events whose time lies in one of up to three consecutive half-open intervals [bounds[b], bounds[b + 1]) -- the reference
// counts each interval with its own PEventStore.find(startTime, untilTime) + groupByKey; here one pass over the event
// stream fills all the interval histograms.  item < 0 = event without a target item (or of another event name): skipped.
// Hot items take millions of increments, so the block-level LDS cache of K1 sits in front of the L2 atomics.
// ============================================================================================
struct PopBounds { long long b[4]; };
__global__ __launch_bounds__(CC_THREADS) void pop_counts_kernel(int64_t n, const int32_t* __restrict__ item, const int64_t* __restrict__ t_ms,
                                                              int32_t n_items, int n_buckets, PopBounds bounds, int32_t* __restrict__ counts) {
  __shared__ int s_key[CC_SLOTS];
  __shared__ int s_cnt[CC_SLOTS];
  for (int s = threadIdx.x; s < CC_SLOTS; s += CC_THREADS) {
    s_key[s] = 0;
    s_cnt[s] = 0;
  }
  __syncthreads();
  for (int64_t e = (int64_t)blockIdx.x * CC_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * CC_THREADS) {
    const int i = item[e];
    if (i < 0 || i >= n_items) continue;
    const long long t = t_ms[e];
    int b = -1;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < n_buckets && t >= bounds.b[k] && t < bounds.b[k + 1]) b = k;
    if (b >= 0) cc_insert(s_key, s_cnt, counts, b * n_items + i);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < CC_SLOTS; s += CC_THREADS)
    if (s_key[s] != 0) atomicAdd(&counts[s_key[s] - 1], s_cnt[s]);
}
hipError_t launch_pop_counts(hipStream_t st, int n_cu, int64_t n, const int32_t* item, const int64_t* t_ms, int32_t n_items, int n_buckets,
                             const int64_t* bounds, int32_t* counts) {
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)n_items * (size_t)n_buckets, st);
  if (e != hipSuccess || n == 0) return e;
  PopBounds pb;
  for (int k = 0; k < 4; ++k) pb.b[k] = k <= n_buckets ? bounds[k] : bounds[n_buckets];
  int64_t blocks = (n + (int64_t)CC_THREADS * 8 - 1) / ((int64_t)CC_THREADS * 8);
  const int64_t cap = (int64_t)n_cu * 4;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pop_counts_kernel, dim3((unsigned)blocks), dim3(CC_THREADS), 0, st, n, item, t_ms, n_items, n_buckets, pb, counts);
  return hipGetLastError();
}

// ============================================================================================
// test hooks
// ============================================================================================
__global__ void llr_test_kernel(int64_t n, const int64_t* a, const int64_t* b, const int64_t* ab, const int64_t* nu, double* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = llr_full(a[i], b[i], ab[i], nu[i]);
}
__global__ void u01_test_kernel(int64_t n, uint32_t seed, const int32_t* row, const int32_t* col, double* out, int rng32) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rng32 ? u01_mix32(seed, (uint32_t)row[i], (uint32_t)col[i]) : u01_hash(seed, (uint32_t)row[i], (uint32_t)col[i]);
}
hipError_t launch_llr_test(hipStream_t st, int64_t n, const int64_t* a, const int64_t* b, const int64_t* ab, const int64_t* nu, double* out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(llr_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, a, b, ab, nu, out);
  return hipGetLastError();
}
hipError_t launch_u01_test(hipStream_t st, int64_t n, uint32_t seed, const int32_t* row, const int32_t* col, double* out, int rng32) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(u01_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, seed, row, col, out, rng32);
  return hipGetLastError();
}

}  // namespace urcco
