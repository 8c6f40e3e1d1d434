// cco_rowscan.hip -- sampleDownAndBinarize: the CSR row scan
// Part of the hand-written gfx950 (MI355X / CDNA4) kernels of the Correlated Cross-Occurrence model build: see cco_common.h for the map of the stages.
#include "cco_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cco_common.h"
#include "cco_device.h"


namespace urcco {

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


}  // namespace urcco
