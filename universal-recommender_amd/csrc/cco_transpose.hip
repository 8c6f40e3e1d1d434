// cco_transpose.hip -- the A.t of A.t %*% B (cursor-atomic and part-local forms), row work from a user shard
// Part of the hand-written gfx950 (MI355X / CDNA4) kernels of the Correlated Cross-Occurrence model build: see cco_common.h for the map of the stages.
#include "cco_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cco_common.h"
#include "cco_device.h"


namespace urcco {

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
  hipError_t we = launch_slice_weights(st, loc_t, (int)n_buckets, n_parts, weight);  // (cco_counts.hip: the column counts' part-local layout has the same slice table)
  if (we != hipSuccess) return we;
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


}  // namespace urcco
