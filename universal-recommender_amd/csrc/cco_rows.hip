// cco_rows.hip -- binning + A.t %*% B (LDS hash accumulators) fused with computeSimilarities (LLR + top-k): micro, accumulator classes, multi-pass, dense global
// Part of the hand-written gfx950 (MI355X / CDNA4) kernels of the Correlated Cross-Occurrence model build: see cco_common.h for the map of the stages.
#include "cco_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cco_common.h"
#include "cco_device.h"


namespace urcco {

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
// 2 T -- of packed (column + 1, count) in words [0, SH), and behind them the row's CANDIDATE LIST: one word (slot | cB << 16) per claimed slot, appended by
// the claiming lane (cB rode in on its B' word).  Compaction: the candidates' packed words to [0, D), in list order, and every candidate's cB into its slot
// of the KEY array behind them, which the score phase reads and then overwrites with the candidate's key: no word more than rounds 1-5 needed
// (3 D + 3 k + 2 <= E; the list's E - SH >= E / 3 words hold every D the rule admits), a third fewer slots.
__host__ __device__ constexpr int table_slots(int E, int T) { return (2 * E / 3) / (2 * T) * (2 * T); }  // (the zeroing writes PAIRS of slots: 8-byte LDS accesses)
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
// multiply-high, the probe sequence wraps by a compare -- and the lane that CLAIMS a slot is told so: it owns the new candidate and appends (slot, the
// column's count -- it rode in on the B' word --) to the row's candidate list in the third of the table behind the slots (see the pair loop).
template <int NS>
__device__ __forceinline__ bool tab_insert(unsigned* tab, unsigned key, int count_bits, bool ident, unsigned& claimed_at) {
  unsigned h = ident ? (key - 1u) : __umulhi(key * 0x9E3779B1u, (unsigned)NS);
  const unsigned fresh = (key << count_bits) | 1u;
  // ONE loop condition and no break: with two exits and a result flag the compiler spent ~25 scalar instructions per probe on execution
  // masks (round 5, ISA of the pair loop: the CU's single scalar unit was as loaded as its four vector units).  `left` bounds the probes
  // (a broken binning invariant must not hang the GPU); the add for a known column is predicated, not branched around.
  // claimed_at: the slot this lane's CAS found EMPTY (the lane then owns the new candidate: it appends it to the row's candidate list), else unchanged.
  bool done;
  unsigned left = (unsigned)NS;
#pragma unroll 1
  do {
    const unsigned v = atomicCAS(&tab[h], 0u, fresh);
    const bool hit = (v >> count_bits) == key;
    if (hit) atomicAdd(&tab[h], 1u);
    claimed_at = v == 0u ? h : claimed_at;
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
  static_assert(SH % (2 * T) == 0 && SH + SH / 2 <= E && SH <= 65536, "slots and the candidate list share the table; a slot index takes 16 bits of a list word");
  constexpr unsigned CAND_CAP = (unsigned)(E - SH);               // list capacity: >= E / 3 > the most candidates the capacity rule (3 D + 3 k + 2 <= E) admits
  constexpr int CPT = ((E - 5) / 3 + T - 1) / T;                  // candidates a thread moves in the compaction at most
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
  constexpr bool SKIP_SHARED = T >= 256;  // (round 6, after the packed form freed registers: the 512/1024-thread classes -1..2 % with it; the one-wave class still +2 %: profiles/r06_gathers_in_flight_ab.log)
  __shared__ unsigned long long s_kbits[2 * NW];  // per wave: AND / OR over its valid keys
  __shared__ unsigned s_mpflag;                    // MP: a pass overflowed its table
  __shared__ unsigned s_ncand;                     // teams of several waves: length of the candidate list
  __shared__ unsigned long long s_runk[MP ? 2 * MP_KMAX : 1];  // MP: the row's running top k (two buffers: a merge reads one, writes the other)
  __shared__ unsigned s_runc[MP ? 2 * MP_KMAX : 1];

  const int team = TEAMS == 1 ? 0 : uni((int)threadIdx.x / T);  // a team is one wave (T == 64) or the whole block
  const int tl = threadIdx.x % T;
  const int lane = threadIdx.x & (WAVE - 1);
  unsigned* tab = s_tab + team * E;
  unsigned* cand = tab + SH;  // insert phase: the row's candidate list -- (slot, column count) of every claimed slot, in the order the claims were made
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
    unsigned n_cand = 0u;  // one-wave teams: length of the candidate list (wave-uniform)
    if (T != WAVE && tl == 0) s_ncand = 0u;
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
        // Every lane runs the pair loop (`per` is team-uniform; a lane without pairs has an empty range): the claims of a round are
        // appended to the candidate list by wave operations.
        const unsigned per = (total + T - 1) / T;
        const unsigned first = (unsigned)tl * per;
        const bool act = first < total;
        const unsigned last = act ? (first + per < total ? first + per : total) : first;
        int o = 0;
        int64_t pos = 0;
        unsigned uend = 0u;
        if (act) {
          int lo = 1, hi = T;  // first idx in [1, T] with uoff[idx] > first (uoff[T] = total > first).  T candidates, halved exactly log2(T) times:
#pragma unroll             // a fixed trip count, selects instead of branches (the data-dependent loop cost four scalar instructions per step)
          for (int step = 0; step < LOG2T; ++step) {
            const int mid = (lo + hi) >> 1;
            const bool gt = uoff[mid] > first;
            hi = gt ? mid : hi;
            lo = gt ? lo : mid + 1;
          }
          o = lo - 1;
          pos = ustart[o] + (first - uoff[o]);
          uend = uoff[o + 1];
        }
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
          unsigned at[G];  // the slot a lane's insert CLAIMED (found empty), else ~0u
#pragma unroll
          for (int q = 0; q < G; ++q) {
            at[q] = ~0u;
            if (on[q]) {
              const unsigned col = jj[q] & colmask;
              if (dbg & 1) {  // ablation: gather only
                if (jj[q] == 0xffffffffu) tab[0] = 1u;
              } else if (MP) {
                if ((col & mp_mask) == mp_q && !tab_insert<SH>(tab, (col >> mp_s) + 1u, cb, ident, at[q])) s_mpflag = 1u;
              } else if (!tab_insert<SH>(tab, col + 1u, cb, ident, at[q])) {
                atomicAdd(a.err, 1ull);
              }
            }
          }
          // The lane that claimed a slot owns the new candidate and appends (slot | cB << 16) to the row's list -- cB rode in on its B' word --, at
          // the position a ballot gives it: the compaction then moves CANDIDATES instead of sweeping SLOTS (rounds 1-6a: every thread read SH / T
          // slots, counted the occupied ones twice and scanned; 223 vector instructions per row of the one-wave class for this and the key stores).
          unsigned long long cm[G];
          unsigned n_new = 0u;
#pragma unroll
          for (int q = 0; q < G; ++q) {
            cm[q] = __ballot(at[q] != ~0u);
            n_new += (unsigned)__popcll(cm[q]);
          }
          if (n_new != 0u) {  // wave-uniform
            unsigned base;
            if (T == WAVE) {
              base = n_cand;
              n_cand += n_new;
            } else {
              unsigned b = 0u;
              if (lane == 0) b = atomicAdd(&s_ncand, n_new);
              base = wave_read_lane(b, 0);
            }
#pragma unroll
            for (int q = 0; q < G; ++q) {
              const unsigned idx = base + lanes_below(cm[q]);
              if (at[q] != ~0u) {
                if (idx < CAND_CAP) cand[idx] = at[q] | ((packed ? jj[q] >> cshift : 0u) << 16);
                else if (MP) s_mpflag = 1u;        // (a pass with more candidates than it has room for is abandoned below)
                else atomicAdd(a.err, 1ull);       // (cannot happen while the binning rule holds)
              }
              base += (unsigned)__popcll(cm[q]);
            }
          }
        }
      }
      team_sync<T>();  // before the next chunk overwrites ustart / uoff
    }
    // ---- 3. the candidates' packed words go to tab[0 .. D), in list order; candidate keys will live behind them in the same LDS:
    //         words [kb, kb + 2 D) with kb = D rounded up to even.  The binning rule keeps 3 D + 1 <= E.
    //         ... and every candidate's column count goes into its slot of that key array (the score phase reads it, then puts the key there)
    // Thread tl moves candidates tl, tl + T, ...: list word -> slot -> packed word, every read of the table and of the list ahead of the first write.
    unsigned D = T == WAVE ? n_cand : uni(s_ncand);  // (teams of several waves: the expand loop's last barrier published it)
    if (D > CAND_CAP) D = CAND_CAP;                   // (an abandoned pass / a broken invariant: flagged above)
    {
      unsigned cw[CPT], cv[CPT];
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        const unsigned t = (unsigned)tl + (unsigned)q * T;
        cw[q] = 0u;
        cv[q] = 0u;
        if (t < D) {  // (a team-uniform test of q * T < D around this measured no better)
          cw[q] = cand[t];
          cv[q] = tab[cw[q] & 0xffffu];
        }
      }
      team_sync<T>();  // every read of the table and the list precedes every write below (one wave: its LDS accesses execute in order -- wave_sync costs nothing there)
      unsigned long long* kk0 = reinterpret_cast<unsigned long long*>(tab + ((D + 1u) & ~1u));
      // (a multi-pass row's pass may hold more candidates than leave room for their keys: it is abandoned below -- and must not write key slots
      // beyond the table; found by the simulator's bounds-checked build)
      const bool fits = !MP || 3ll * D + 3ll * a.k + 2ll <= (long long)E;  // team-uniform
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        const unsigned t = (unsigned)tl + (unsigned)q * T;
        if (t < D) {
          tab[t] = cv[q];
          if (fits) kk0[t] = (unsigned long long)(cw[q] >> 16);
        }
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


}  // namespace urcco
