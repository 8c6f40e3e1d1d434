// Hand-written gfx950 kernels of the device-side Preparator: dictionaries and
// binary CSR matrices straight from (user key, item key) event streams in HBM.
//
// Replaces what UR's own IndexedDatasetSpark builders do with three `distinct().collect()` + broadcast round trips and a
// string-keyed `groupByKey` per event type (reference src/main/scala/Preparator.scala:102-158, :160-214):
//   dictionary   key -> dense id, ids in order of FIRST APPEARANCE in the stream (decision D8), optionally only keys that
//                occur at least min_count times (`minEventsPerUser`, raw events, duplicates included: :129-132)
//   lookup       ids of a stream against an existing dictionary (-1 = not in it: :173-179 drop such events)
//   CSR build    (row id, column id) pairs -> rows with sorted, duplicate-free columns (`setQuick(col, 1.0)`: :146, :205)
// Keys are 64-bit (the host hashes its strings, or passes integer ids); ~0 is reserved for "empty slot".
// First-appearance ids need no sort: every slot keeps the smallest position of its key (atomicMin); an event is a
// "first" iff its position equals that minimum; the id of a key is the number of firsts before its own first -- one
// exclusive scan over the stream.
// Wave = 64 lanes.  Wave-level primitives only under wave-uniform control flow.
#include "cco_kernels.h"

namespace urcco {

namespace {
constexpr int IG_WAVE = 64;
constexpr unsigned long long IG_EMPTY = ~0ull;

__device__ __forceinline__ unsigned long long ig_mix(unsigned long long x) {  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// slot of `key`, or the empty slot where its probe sequence ends
__device__ __forceinline__ unsigned long long ig_find(const KeyTable& t, unsigned long long key) {
  unsigned long long slot = ig_mix(key) & t.mask;
  for (;;) {
    const unsigned long long cur = t.keys[slot];
    if (cur == key || cur == IG_EMPTY) return slot;
    slot = (slot + 1) & t.mask;
  }
}
}  // namespace

// Popular keys are hit by millions of events (the top item of a Zipf catalogue by ~1 % of the stream): three atomics per event
// on the key's slot -- claim, position minimum, count -- serialise on one L2 address each (measured: 6.2 ms per dictionary,
// two thirds of the whole ingest).  A key never changes once written, the position minimum only decreases and the count only
// matters up to `need` (min_count), so each of the three is read first and the atomic issued only if it can still change
// something; a stale read (L1 is not coherent with other CUs' atomics) costs one unnecessary atomic, never a wrong result.
__global__ __launch_bounds__(256) void ig_insert_kernel(KeyTable t, int64_t n, const unsigned long long* __restrict__ keys,
                                                        const int32_t* __restrict__ select, unsigned need) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
    if (select && select[p] < 0) continue;
    const unsigned long long key = keys[p];
    unsigned long long slot = ig_mix(key) & t.mask;
    for (;;) {
      unsigned long long cur = t.keys[slot];
      if (cur == IG_EMPTY) cur = atomicCAS(&t.keys[slot], IG_EMPTY, key);
      if (cur == IG_EMPTY || cur == key) {
        if ((unsigned)p < t.minpos[slot]) atomicMin(&t.minpos[slot], (unsigned)p);
        if (need > 1u && t.count[slot] < need) atomicAdd(&t.count[slot], 1u);  // exact below `need`, at least `need` from there on
        break;
      }
      slot = (slot + 1) & t.mask;
    }
  }
}

// flag[p] = 1 iff event p is the first appearance of a key that occurs at least min_count times
__global__ __launch_bounds__(256) void ig_first_flags_kernel(KeyTable t, int64_t n, const unsigned long long* __restrict__ keys,
                                                             const int32_t* __restrict__ select, unsigned min_count, int32_t* __restrict__ flag) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
    int f = 0;
    if (!(select && select[p] < 0)) {
      const unsigned long long slot = ig_find(t, keys[p]);
      f = t.minpos[slot] == (unsigned)p && (min_count <= 1u || t.count[slot] >= min_count);
    }
    flag[p] = f;
  }
}

__global__ __launch_bounds__(256) void ig_assign_kernel(KeyTable t, int64_t n, const unsigned long long* __restrict__ keys,
                                                        const int32_t* __restrict__ flag, const int64_t* __restrict__ prefix,
                                                        int64_t* __restrict__ first_pos) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
    if (!flag[p]) continue;
    const int64_t id = prefix[p];
    t.id[ig_find(t, keys[p])] = (int32_t)id;
    first_pos[id] = p;
  }
}

__global__ __launch_bounds__(256) void ig_lookup_kernel(KeyTable t, int64_t n, const unsigned long long* __restrict__ keys,
                                                        const int32_t* __restrict__ select, int32_t* __restrict__ ids) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
    int32_t id = -1;
    if (!(select && select[p] < 0)) {
      const unsigned long long key = keys[p];
      const unsigned long long slot = ig_find(t, key);
      if (t.keys[slot] == key) id = t.id[slot];
    }
    ids[p] = id;
  }
}

static unsigned ig_grid(int64_t n, int n_cu) {
  int64_t b = (n + 255) / 256;
  const int64_t cap = (int64_t)n_cu * 16;
  if (b > cap) b = cap;
  return (unsigned)(b < 1 ? 1 : b);
}

hipError_t launch_dictionary_build(hipStream_t st, int n_cu, KeyTable t, int64_t n, const unsigned long long* keys, const int32_t* select,
                                   int32_t min_count, int32_t* flag, int64_t* prefix, int64_t* tile_sums, int64_t* first_pos) {
  if (n == 0) return hipMemsetAsync(prefix, 0, sizeof(int64_t), st);
  hipLaunchKernelGGL(ig_insert_kernel, dim3(ig_grid(n, n_cu)), dim3(256), 0, st, t, n, keys, select, (unsigned)(min_count < 1 ? 1 : min_count));
  hipLaunchKernelGGL(ig_first_flags_kernel, dim3(ig_grid(n, n_cu)), dim3(256), 0, st, t, n, keys, select, (unsigned)(min_count < 1 ? 1 : min_count), flag);
  hipError_t e = launch_scan_i32(st, flag, n, prefix, tile_sums);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ig_assign_kernel, dim3(ig_grid(n, n_cu)), dim3(256), 0, st, t, n, keys, flag, prefix, first_pos);
  return hipGetLastError();
}

// collision check: the check key (a second, independent hash of the same string) of every position must equal the check key
// of its id's first occurrence; err[0] counts the positions where it does not
__global__ __launch_bounds__(256) void ig_verify_kernel(KeyTable t, int64_t n, const unsigned long long* __restrict__ keys, const int32_t* __restrict__ select,
                                                        const unsigned long long* __restrict__ check, const unsigned long long* __restrict__ ref_check,
                                                        const int64_t* __restrict__ first_pos, unsigned long long* __restrict__ err) {
  unsigned bad = 0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
    if (select && select[p] < 0) continue;
    const unsigned long long key = keys[p];
    const unsigned long long slot = ig_find(t, key);
    if (t.keys[slot] != key) continue;
    const int32_t id = t.id[slot];
    if (id >= 0 && check[p] != ref_check[first_pos[id]]) ++bad;  // ref_check: check keys of the stream the dictionary was built from
  }
  if (bad) atomicAdd(err, (unsigned long long)bad);
}
hipError_t launch_dictionary_verify(hipStream_t st, int n_cu, KeyTable t, int64_t n, const unsigned long long* keys, const int32_t* select,
                                    const unsigned long long* check, const unsigned long long* ref_check, const int64_t* first_pos, unsigned long long* err) {
  hipError_t e = hipMemsetAsync(err, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess || n == 0) return e;
  hipLaunchKernelGGL(ig_verify_kernel, dim3(ig_grid(n, n_cu)), dim3(256), 0, st, t, n, keys, select, check, ref_check, first_pos, err);
  return hipGetLastError();
}

hipError_t launch_dictionary_lookup(hipStream_t st, int n_cu, KeyTable t, int64_t n, const unsigned long long* keys, const int32_t* select,
                                    int32_t* ids) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(ig_lookup_kernel, dim3(ig_grid(n, n_cu)), dim3(256), 0, st, t, n, keys, select, ids);
  return hipGetLastError();
}

// ============================================================================================
// (row, col) pairs -> binary CSR.  count per row (L2 atomics) -> scan -> scatter by cursor -> per row: sort ascending,
// drop duplicates (rows of <= 64 raw entries: bitonic network in the registers of one wave; longer rows: one block,
// bitonic in LDS up to 4096 entries, in global memory beyond) -> scan of the final lengths -> compaction.
// ============================================================================================
__global__ __launch_bounds__(256) void ig_count_rows_kernel(int64_t n, const int32_t* __restrict__ rows, const int32_t* __restrict__ cols,
                                                            int32_t* __restrict__ cnt) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256)
    if (rows[p] >= 0 && cols[p] >= 0) atomicAdd(&cnt[rows[p]], 1);
}

__global__ __launch_bounds__(256) void ig_scatter_rows_kernel(int64_t n, const int32_t* __restrict__ rows, const int32_t* __restrict__ cols,
                                                              const int64_t* __restrict__ raw_ptr, int32_t* __restrict__ cursor,
                                                              int32_t* __restrict__ tmp) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
    const int r = rows[p], c = cols[p];
    if (r >= 0 && c >= 0) tmp[raw_ptr[r] + atomicAdd(&cursor[r], 1)] = c;
  }
}

// one wave per row with <= 64 raw entries: sort, unique, write back to the row's start, len[r] = distinct columns.
// Rows with more entries get len[r] = -1 (the block kernel takes them).
__global__ __launch_bounds__(256) void ig_sort_rows_wave_kernel(int64_t n_rows, const int64_t* __restrict__ raw_ptr, int32_t* __restrict__ tmp,
                                                                int32_t* __restrict__ len) {
  const int lane = threadIdx.x & (IG_WAVE - 1);
  const int64_t n_waves = (int64_t)gridDim.x * (256 / IG_WAVE);
  for (int64_t r = (int64_t)blockIdx.x * (256 / IG_WAVE) + threadIdx.x / IG_WAVE; r < n_rows; r += n_waves) {  // wave-uniform
    const int64_t s = raw_ptr[r];
    const int64_t L = raw_ptr[r + 1] - s;
    if (L > IG_WAVE) {
      if (lane == 0) len[r] = -1;
      continue;
    }
    int v = lane < L ? tmp[s + lane] : 0x7fffffff;
    for (int k2 = 2; k2 <= IG_WAVE; k2 <<= 1) {
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        const int o = __shfl_xor(v, j);
        const bool keep_small = ((lane & j) == 0) == ((lane & k2) == 0);  // lower lane of an ascending block
        if (keep_small ? o < v : o > v) v = o;
      }
    }
    const int prev = __shfl_up(v, 1);
    const bool fresh = lane < L && (lane == 0 || v != prev);
    const unsigned long long m = __ballot(fresh);
    if (fresh) tmp[s + __popcll(m & ((1ull << lane) - 1ull))] = v;
    if (lane == 0) len[r] = __popcll(m);
  }
}

constexpr int IG_LDS_ROW = 4096;

// one block per row with > 64 raw entries (grid-stride over all rows; the test is block-uniform)
__global__ __launch_bounds__(256) void ig_sort_rows_block_kernel(int64_t n_rows, const int64_t* __restrict__ raw_ptr, int32_t* __restrict__ tmp,
                                                                 int32_t* __restrict__ len) {
  __shared__ int s_v[IG_LDS_ROW];
  __shared__ int s_wsum[256 / IG_WAVE];
  __shared__ int s_total;
  const int lane = threadIdx.x & (IG_WAVE - 1), wave = threadIdx.x / IG_WAVE;
  for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const int64_t s = raw_ptr[r];
    const int64_t L = raw_ptr[r + 1] - s;
    if (L <= IG_WAVE) continue;  // block-uniform
    int64_t P = 1;
    while (P < L) P <<= 1;
    int* row = tmp + s;
    const bool in_lds = L <= IG_LDS_ROW;
    if (in_lds) {
      for (int64_t t = threadIdx.x; t < P; t += 256) s_v[t] = t < L ? row[t] : 0x7fffffff;
      __syncthreads();
      for (int64_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (int64_t j = k2 >> 1; j > 0; j >>= 1) {
          for (int64_t t = threadIdx.x; t < P; t += 256) {
            const int64_t u = t ^ j;
            if (u > t) {
              const int a = s_v[t], b = s_v[u];
              const bool asc = (t & k2) == 0;
              if (asc ? a > b : a < b) { s_v[t] = b; s_v[u] = a; }
            }
          }
          __syncthreads();
        }
      }
    } else {
      // In global memory, as P = 2^ceil(log2 L) entries whose tail [L, P) is +inf.  The ascending-only formulation of the
      // bitonic network (partner = t ^ (k2 - 1) on the first step of a level, t ^ j afterwards; the smaller value always
      // goes to the lower index) never moves a real value into the padding, so exchanges that touch it are skipped.
      for (int64_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (int64_t j = k2 >> 1; j > 0; j >>= 1) {
          for (int64_t t = threadIdx.x; t < P; t += 256) {
            const int64_t u = (j == (k2 >> 1)) ? (t ^ (k2 - 1)) : (t ^ j);
            if (u > t && u < L) {
              const int a = row[t], b = row[u];
              if (a > b) { row[t] = b; row[u] = a; }
            }
          }
          __syncthreads();
        }
      }
    }
    // unique: keep an entry iff it differs from its predecessor; positions by block scan, chunk by chunk
    int carry = 0;
    for (int64_t base = 0; base < L; base += 256) {  // block-uniform
      const int64_t t = base + threadIdx.x;
      int v = 0, pv = 0;
      bool fresh = false;
      if (t < L) {
        v = in_lds ? s_v[t] : row[t];
        pv = t == 0 ? 0 : (in_lds ? s_v[t - 1] : row[t - 1]);
        fresh = t == 0 || v != pv;
      }
      const unsigned long long m = __ballot(fresh);
      if (lane == 0) s_wsum[wave] = __popcll(m);
      __syncthreads();
      int before = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < 256 / IG_WAVE; ++w) {
        const int c = s_wsum[w];
        if (w < wave) before += c;
        tot += c;
      }
      const int pos = carry + before + __popcll(m & ((1ull << lane) - 1ull));
      __syncthreads();  // every read of row[base .. base+256) and of s_wsum precedes the writes below
      if (fresh) row[pos] = v;  // pos <= t: in-place compaction towards the front, chunk by chunk
      carry += tot;
      __syncthreads();
    }
    if (threadIdx.x == 0) len[r] = carry;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void ig_compact_rows_kernel(int64_t n_rows, const int64_t* __restrict__ raw_ptr, const int32_t* __restrict__ tmp,
                                                              const int64_t* __restrict__ out_rp, int32_t* __restrict__ out_ci) {
  const int lane = threadIdx.x & (IG_WAVE - 1);
  const int64_t n_waves = (int64_t)gridDim.x * (256 / IG_WAVE);
  for (int64_t r = (int64_t)blockIdx.x * (256 / IG_WAVE) + threadIdx.x / IG_WAVE; r < n_rows; r += n_waves) {
    const int64_t s = raw_ptr[r], d = out_rp[r];
    const int64_t L = out_rp[r + 1] - d;
    for (int64_t t = lane; t < L; t += IG_WAVE) out_ci[d + t] = tmp[s + t];
  }
}

hipError_t launch_csr_from_pairs(hipStream_t st, int n_cu, int64_t n, const int32_t* rows, const int32_t* cols, int64_t n_rows, int32_t* cnt,
                                 int64_t* raw_ptr, int32_t* tmp, int64_t* tile_sums, int64_t* out_row_ptr, int32_t* out_col_idx) {
  hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(n_rows > 0 ? n_rows : 1), st);
  if (e != hipSuccess) return e;
  if (n > 0) hipLaunchKernelGGL(ig_count_rows_kernel, dim3(ig_grid(n, n_cu)), dim3(256), 0, st, n, rows, cols, cnt);
  e = launch_scan_i32(st, cnt, n_rows, raw_ptr, tile_sums);
  if (e != hipSuccess) return e;
  if (n_rows == 0) return hipMemsetAsync(out_row_ptr, 0, sizeof(int64_t), st);
  e = hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)n_rows, st);  // now the scatter cursors
  if (e != hipSuccess) return e;
  if (n > 0) hipLaunchKernelGGL(ig_scatter_rows_kernel, dim3(ig_grid(n, n_cu)), dim3(256), 0, st, n, rows, cols, raw_ptr, cnt, tmp);
  const unsigned wgrid = ig_grid(n_rows * IG_WAVE, n_cu);
  hipLaunchKernelGGL(ig_sort_rows_wave_kernel, dim3(wgrid), dim3(256), 0, st, n_rows, raw_ptr, tmp, cnt);  // cnt becomes the final lengths
  int64_t bgrid = n_rows < (int64_t)n_cu * 8 ? n_rows : (int64_t)n_cu * 8;
  hipLaunchKernelGGL(ig_sort_rows_block_kernel, dim3((unsigned)(bgrid < 1 ? 1 : bgrid)), dim3(256), 0, st, n_rows, raw_ptr, tmp, cnt);
  e = launch_scan_i32(st, cnt, n_rows, out_row_ptr, tile_sums);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ig_compact_rows_kernel, dim3(wgrid), dim3(256), 0, st, n_rows, raw_ptr, tmp, out_row_ptr, out_col_idx);
  return hipGetLastError();
}

}  // namespace urcco
