// cco_expand.hip -- per-item entropies, 16-bit counts, B' words with counts aboard, expand preparation, row work
// Part of the hand-written gfx950 (MI355X / CDNA4) kernels of the Correlated Cross-Occurrence model build: see cco_common.h for the map of the stages.
#include "cco_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cco_common.h"
#include "cco_device.h"


namespace urcco {

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
  const unsigned limit = 32 - shift >= 16 ? 65536u : (1u << (32 - shift));  // counts must fit the word's spare bits AND 16 bits of a candidate-list word
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


}  // namespace urcco
