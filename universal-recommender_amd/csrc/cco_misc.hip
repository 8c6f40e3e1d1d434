// cco_misc.hip -- indicator compaction, work-balanced item ranges, multi-GPU exchange helpers, boundary checks, test hooks
// Part of the hand-written gfx950 (MI355X / CDNA4) kernels of the Correlated Cross-Occurrence model build: see cco_common.h for the map of the stages.
#include "cco_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cco_common.h"
#include "cco_device.h"


namespace urcco {

// ============================================================================================
// Strided top-k rows -> CSR
// ============================================================================================
// A wave moves CI_ROWS consecutive rows per step, a lane one entry of each (k <= 64: one round; more: a loop): the row's count and output offset are
// wave-uniform scalar loads, every load of the step is issued before its first store, no division.  (Rounds 1-6a: one thread per (row, slot) of the strided
// buffer with a 64-bit division each -- ~500 M of them per build of config 4 for 287 M entries: 1.9 ms, as much issue- as memory-bound.)
constexpr int CI_ROWS = 4;
__global__ __launch_bounds__(256) void compact_indicators_kernel(int32_t n_rows, int32_t k, const int32_t* __restrict__ count,
                                                                 const int32_t* __restrict__ idx, const double* __restrict__ llr,
                                                                 const int64_t* __restrict__ row_ptr, int32_t* __restrict__ out_idx,
                                                                 double* __restrict__ out_llr) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t n_waves = (int64_t)gridDim.x * (256 / WAVE);
  const int64_t wave = (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE));
  for (int64_t r0 = wave * CI_ROWS; r0 < n_rows; r0 += n_waves * CI_ROWS) {
    int c[CI_ROWS];
    int64_t o[CI_ROWS];
#pragma unroll
    for (int q = 0; q < CI_ROWS; ++q) {
      const int64_t r = r0 + q < n_rows ? r0 + q : (int64_t)n_rows - 1;
      c[q] = r0 + q < n_rows ? count[r] : 0;
      o[q] = row_ptr[r];
    }
    int32_t vi[CI_ROWS];
    double vl[CI_ROWS];
#pragma unroll
    for (int q = 0; q < CI_ROWS; ++q)
      if (lane < c[q]) {
        vi[q] = idx[(r0 + q) * k + lane];
        vl[q] = llr[(r0 + q) * k + lane];
      }
#pragma unroll
    for (int q = 0; q < CI_ROWS; ++q)
      if (lane < c[q]) {
        out_idx[o[q] + lane] = vi[q];
        out_llr[o[q] + lane] = vl[q];
      }
#pragma unroll
    for (int q = 0; q < CI_ROWS; ++q)
      for (int s = lane + WAVE; s < c[q]; s += WAVE) {  // k > 64
        out_idx[o[q] + s] = idx[(r0 + q) * k + s];
        out_llr[o[q] + s] = llr[(r0 + q) * k + s];
      }
  }
}

hipError_t launch_compact_indicators(hipStream_t st, int32_t n_rows, int32_t k, const int32_t* count, const int32_t* idx,
                                     const double* llr, const int64_t* row_ptr, int32_t* out_idx, double* out_llr) {
  if ((int64_t)n_rows * k == 0) return hipSuccess;
  int64_t blocks = ((int64_t)n_rows + (256 / WAVE) * CI_ROWS - 1) / ((256 / WAVE) * CI_ROWS);
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
