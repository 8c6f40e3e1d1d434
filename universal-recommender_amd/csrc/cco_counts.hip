// cco_counts.hip -- numNonZeroElementsPerColumn (three forms by matrix size), the public scans, PopModel interval histograms
// Part of the hand-written gfx950 (MI355X / CDNA4) kernels of the Correlated Cross-Occurrence model build: see cco_common.h for the map of the stages.
#include "cco_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cco_common.h"
#include "cco_device.h"


namespace urcco {

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

hipError_t launch_scan_i32(hipStream_t st, const int32_t* in, int64_t n, int64_t* out, int64_t* tile_sums) {
  return launch_scan(st, LoadI32{in}, n, out, tile_sums);
}
hipError_t launch_scan_i64(hipStream_t st, const int64_t* in, int64_t n, int64_t* out, int64_t* tile_sums) {
  return launch_scan(st, LoadI64{in}, n, out, tile_sums);
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
// weight[b] (zeroed here) = entries of bucket b over all parts of a part-local layout: loc_t[(n_buckets + 1) x n_parts] transposed slice starts
hipError_t launch_slice_weights(hipStream_t st, const unsigned short* loc_t, int n_buckets, int64_t n_parts, long long* weight) {
  hipError_t we = hipMemsetAsync(weight, 0, sizeof(long long) * (size_t)n_buckets, st);
  if (we != hipSuccess) return we;
  const unsigned wsplit = (unsigned)(n_parts >= 8192 ? 8 : (n_parts >= 1024 ? 4 : 1));
  hipLaunchKernelGGL(pl_weights_kernel, dim3((unsigned)n_buckets, wsplit), dim3(256), 0, st, loc_t, n_parts, reinterpret_cast<unsigned long long*>(weight));
  return hipGetLastError();
}

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
    hipError_t we = launch_slice_weights(st, loc_t, n_buckets, n_parts, weight);
    if (we != hipSuccess) return we;
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


}  // namespace urcco
