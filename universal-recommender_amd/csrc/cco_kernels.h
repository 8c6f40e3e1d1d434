// Host-callable launchers of the gfx950 CCO kernels (cco_counts / cco_rowscan / cco_transpose / cco_expand / cco_rows / cco_misc .hip).  Every pointer is a device pointer;
// every launcher only enqueues on `st`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urcco {

constexpr int NBINS = 7;  // accumulator classes: 0 micro (one wave, <= 64 pairs), 1 wave-LDS (64 thr, 1024 words),
                          // 2 small-block-LDS (256 thr, 4096 words), 3 block-LDS (256 thr, 8192 words),
                          // 4 half-CU-LDS (512 thr, 16384 words), 5 CU-LDS (1024 thr, 32768 words),
                          // 6 multi-pass CU-LDS (rows no single table holds; dense global counters when k > MP_KMAX_HOST)
constexpr int MP_KMAX_HOST = 256;  // largest k the multi-pass class serves (== MP_KMAX in cco_rows.hip)

// Geometry the host side needs for scratch sizing.
constexpr int SCAN_TILE = 2048;          // elements per scan tile (256 threads x 8)
constexpr int DS_TILE = 4096;            // entries per down-sample tile (256 threads x 4 x 4)
constexpr int GLOBAL_BIN_BLOCKS = 128;    // persistent blocks of the global-accumulator kernel (upper bound)
constexpr int BIN_TILE = 1024;           // items per binning tile
constexpr int XLX_TABLE_HOST = 4096;     // entries of the small-integer xLogX table (== XLX_TABLE in cco_device.h)
constexpr int BIN_COLS_HOST = 3 * NBINS + 3;  // int64 per binning tile (rows per internal bin -- the micro class has three sub-lists --, pairs and users per bin, total)
constexpr int BIN_OFF_LEN = NBINS + 3;        // bin_off: [0 .. NBINS] list offsets of the classes, then the starts of the micro class's second and third sub-list
constexpr int CAND_SLOTS = 64;           // words the row kernels spread their candidate counts over (see CcoArgs::cand)
constexpr int STATS_LEN = 32;            // [0] pairs, then NBINS each of rows / pairs / users / out entries per bin, [1 + 4 NBINS] table overflows

struct CcoArgs {
  // row lists per bin
  const int32_t* bin_rows;   // item ids grouped by bin
  const int32_t* bin_off;    // [BIN_OFF_LEN] offsets into bin_rows
  // matrices
  const int64_t* a_col_ptr;  // CSC of A': users of item i are entries [a_col_ptr[i], a_col_ptr[i+1])
  const int64_t* pstart;     // per CSC entry: start of that user's B' row in b_col_idx
  const int64_t* wp;         // per CSC entry (+1): exclusive prefix of B' row lengths over the CSC
  const int32_t* b_col_idx;
  // Round 6: B' with the column's post-sampling count riding in the spare bits of the column word -- word = col | cB << (32 - count_bits) -- so that a
  // candidate's cB arrives with the pair that claims its accumulator slot instead of through one scattered 2-byte gather per candidate (more than half
  // of the SpGEMM classes' line fills: profiles/r06_fetch_by_phase.txt).  Nullable; used while *pack_bad == 0 (every count fits its 32 - key bits).
  const int32_t* b_packed;
  const int32_t* pack_bad;   // [1] counts that do not fit the spare bits (then b_col_idx + the count gather serve the build)
  int32_t pk_known;          // 1: the HOST knows b_packed is good (a sharded build learns it with the shard sizes): pack_bad is not read, one instantiation is launched
  uint32_t b_col_mask;       // b_col_idx[e] & b_col_mask = the column (0xffffffff unless b_col_idx itself holds packed words: the rows a sharded build received)
  const int32_t* cnt_a;
  const int32_t* cnt_b;
  const double* ent_a;       // rowEntropy per item of A
  const unsigned short* cnt_b16;  // 16-bit copy of cnt_b (a quarter of the bytes behind the one gather per candidate); valid while *cnt16_bad == 0
  const int32_t* cnt16_bad;  // [1] number of counts beyond 16 bits (then cnt_b is gathered)
  const double* xlx_n;       // [1] xLogX(N)
  const double* xlx_tab;     // [XLX_TABLE_HOST] xLogX of small integers
  const double* xlx_hi;      // [XLX_TABLE_HOST] xLogX(n_users - d): the k22 term without a logarithm
  const double* col_ent;     // [XLX_TABLE_HOST] columnEntropy of a column with d interactions, for the N of the build (behind xlx_hi in the same allocation)
  int32_t debug;             // ablation switches for profiling (0 in production): 1 = gather only, 2 = no LLR, 4 = no top-k
  long long n_users;
  int32_t n_cols_b;
  int32_t item_lo;
  int32_t exclude_self;
  int32_t k;
  int32_t has_min_llr;
  double min_llr;
  int32_t count_bits;        // packed LDS entry = ((col+1) << count_bits) | count
  int32_t col_bytes;         // bytes needed for a column index of B (1..4): digits of the top-k tie break
  int32_t g_log2;            // lanes cooperating on one user's B row = 1 << g_log2
  int32_t unordered;         // 1: rows carry their top-k set in arbitrary order (no ranking pass)
  // outputs (strided by k)
  int32_t* out_count;
  int32_t* out_idx;
  double* out_llr;
  unsigned long long* err;   // stats[1 + 4 * NBINS]: LDS table overflows (must stay 0)
  unsigned long long* cand;  // nullable [CAND_SLOTS], zero on entry: distinct (row, column) candidates scored, spread over the slots (statistics)
  // global-accumulator scratch (bin 3)
  int32_t* g_counts;         // [GLOBAL_BIN_BLOCKS][n_cols_b] zero on entry, zero on exit
  unsigned long long* g_cand_key;  // [GLOBAL_BIN_BLOCKS][n_cols_b]
  int32_t* g_cand_col;       // [GLOBAL_BIN_BLOCKS][n_cols_b]
  int32_t g_blocks;          // > 0: bin 6 runs on the dense global-accumulator kernel with this many resident blocks; 0: multi-pass LDS class
};

hipError_t launch_column_counts(hipStream_t st, int n_cu, const int32_t* col_idx, int64_t nnz, int32_t n_cols, int32_t* counts);
// Atomic-free variant for large matrices: returns 0 scratch bytes when the plain kernel should be used instead
// (small nnz or more than 1024 buckets of 8192 columns).  nnz_dev (nullable): device-side nnz <= nnz.
constexpr int64_t PH_MIN_NNZ = 1 << 20;
int64_t column_counts_scratch_bytes(int64_t nnz, int32_t n_cols);
hipError_t launch_column_counts_partitioned(hipStream_t st, const int32_t* col_idx, int64_t nnz, const int64_t* nnz_dev, int32_t n_cols,
                                            int32_t* counts, char* scratch);

// scans: out[i] = sum_{t<i} in[t], out[n] = total.  tile_sums scratch: ceil(n / SCAN_TILE) + 1 int64.
hipError_t launch_scan_i32(hipStream_t st, const int32_t* in, int64_t n, int64_t* out, int64_t* tile_sums);
// part-local layouts (column counts, transposition): weight[b] = entries of bucket b over all parts, from the transposed slice table loc_t[(n_buckets + 1) x n_parts]
hipError_t launch_slice_weights(hipStream_t st, const unsigned short* loc_t, int n_buckets, int64_t n_parts, long long* weight);

// CSR row scan.  Scratch: thresholds [n_cols + n_cols / 8 + 2] u64 (the 8-byte thresholds, then their one-byte prefixes), tile_rows [tiles + 1] i64, flags [tiles * DS_TILE / 64] u64,
// tile_count [tiles + 1] i64 (exclusive offsets after launch_downsample_scan), tiles = ceil(nnz / DS_TILE)
hipError_t launch_downsample_flags(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz,
                                   int32_t n_cols, const int32_t* raw_counts, unsigned long long* thresholds, uint32_t seed, int32_t max_n,
                                   int row_rate_mode, int64_t row_base, int64_t* tile_rows, unsigned long long* flags, int64_t* tile_count,
                                   int32_t* post_counts, int debug);
hipError_t launch_downsample_scan(hipStream_t st, int64_t nnz, int64_t* tile_count);
hipError_t launch_downsample_compact(hipStream_t st, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz,
                                     const int64_t* tile_rows, const unsigned long long* flags, const int64_t* tile_off, int64_t* out_row_ptr,
                                     int32_t* out_col_idx);

// only columns in [col_lo, col_hi) are transposed (col_ptr must come from launch_scan_i32_range with the same range)
hipError_t launch_scan_i32_range(hipStream_t st, const int32_t* in, int64_t n, int32_t lo, int32_t hi, int64_t* out, int64_t* tile_sums);
hipError_t launch_transpose(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int g_log2,
                            const int64_t* col_ptr, int32_t* cursor, int32_t* out_row_idx, int32_t col_lo, int32_t col_hi);
// two-level counting sort for large matrices (part-local partition + placement): returns 0 scratch bytes when the cursor-atomic kernel should be used
int64_t transpose_scratch_bytes(int64_t n_rows, int64_t nnz, int32_t n_cols);
hipError_t launch_transpose_partitioned(hipStream_t st, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz, int32_t n_cols,
                                        const int64_t* col_ptr, int32_t* cursor /* [n_cols] zero */, int32_t* out_row_idx, int32_t col_lo, int32_t col_hi, char* scratch);
hipError_t launch_row_work_csr(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* a_row_ptr, const int32_t* a_col_idx,
                               const int64_t* b_row_ptr, int g_log2, int32_t n_items_a, int64_t* work);

// ---- device-side Preparator (ingest_kernels.hip) ------------------------------------------------------
// open-addressing dictionary of 64-bit keys: capacity = mask + 1 (power of two, >= 2 x keys), ~0 = empty slot
struct KeyTable {
  unsigned long long* keys;  // [capacity] initialised to ~0
  unsigned* minpos;          // [capacity] smallest stream position of the key, initialised to ~0
  unsigned* count;           // [capacity] occurrences, initialised to 0
  int32_t* id;               // [capacity] dense id (first-appearance order), -1 = none
  unsigned long long mask;
};
// flag: int32[n] scratch, prefix: int64[n + 1] scratch (prefix[n] = number of ids), first_pos: int64[>= ids] out
hipError_t launch_dictionary_build(hipStream_t st, int n_cu, KeyTable t, int64_t n, const unsigned long long* keys, const int32_t* select,
                                   int32_t min_count, int32_t* flag, int64_t* prefix, int64_t* tile_sums, int64_t* first_pos);
hipError_t launch_dictionary_lookup(hipStream_t st, int n_cu, KeyTable t, int64_t n, const unsigned long long* keys, const int32_t* select,
                                    int32_t* ids);
hipError_t launch_dictionary_verify(hipStream_t st, int n_cu, KeyTable t, int64_t n, const unsigned long long* keys, const int32_t* select,
                                    const unsigned long long* check, const unsigned long long* ref_check, const int64_t* first_pos, unsigned long long* err);
// cnt: int32[n_rows] scratch, raw_ptr: int64[n_rows + 1] scratch, tmp: int32[n] scratch
hipError_t launch_csr_from_pairs(hipStream_t st, int n_cu, int64_t n, const int32_t* rows, const int32_t* cols, int64_t n_rows, int32_t* cnt,
                                 int64_t* raw_ptr, int32_t* tmp, int64_t* tile_sums, int64_t* out_row_ptr, int32_t* out_col_idx);

// len[n_rows] = row lengths, len16 (nullable) = the same as uint16; sizes (nullable) = {n_rows, nnz, rows longer than 65535}
constexpr int EXCH_SIZES = 4;  // == URCCO_EXCH_SIZES of include/urcco.h (checked in urcco_internal.h)  // (round 6: [3] = columns whose count does not fit a packed B' word -- launch_counts_over_limit)
hipError_t launch_row_lengths(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, int32_t* len, unsigned short* len16, int64_t* sizes);
// out[0] = number of counts that a B' word of a matrix with these counts cannot carry (>= 2^min(count_bits, 16)): a fact of the count TABLE, the same on
// every rank of a sharded build once the counts are all-reduced -- so every rank decides alike whether the rows it sends travel with their counts aboard
hipError_t launch_counts_over_limit(hipStream_t st, int n_cu, const int32_t* counts, int64_t n, int32_t count_bits, int64_t* out);
// row-filtered exchange (cco_misc.hip): per-user masks of the ranks whose item range a row of A' touches, per-destination masked row
// lengths + their scan + the totals per destination, and the packing of the rows per destination
hipError_t launch_need_mask(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* a_row_ptr, const int32_t* a_col_idx, const int32_t* bounds, int world,
                            unsigned long long* mask);
hipError_t launch_masked_lengths(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const unsigned long long* mask, int world, int32_t* mlen, int64_t* off,
                                 int64_t* tile_sums, int64_t* to_nnz);
hipError_t launch_pack_rows(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, const unsigned long long* mask, int world,
                            const int64_t* off, int32_t* pack);
hipError_t launch_scan_u16(hipStream_t st, const unsigned short* in, int64_t n, int64_t* out, int64_t* tile_sums);
// CSC fragments of the primary (multi-GPU): the record a rank publishes ((2 * world + 3) int64) and the merge of received fragments
hipError_t launch_frag_record(hipStream_t st, int32_t world, const int32_t* bounds, const int64_t* l_cp, const int32_t* bad, int64_t* rec);
hipError_t launch_frag_place(hipStream_t st, int n_cu, int32_t world, int32_t lo, int32_t n_range, const void* lens, int wire16, const int64_t* src_off,
                             const int32_t* ents, const int64_t* a_cp, const int64_t* sizes, int32_t* a_ri);

// boundary checks of a caller-supplied CSR (rp0 = value of row_ptr[0] of the slice); err[0] += violations
hipError_t launch_validate_csr(hipStream_t st, int n_cu, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz, int32_t n_cols,
                               int g_log2, int64_t rp0, unsigned long long* err);
hipError_t launch_rebase_i64(hipStream_t st, int n_cu, int64_t* p, int64_t n, int64_t delta);
// *dst_mapped = *src, dst_mapped = the device address of host-mapped pinned memory (no copy engine involved)
hipError_t launch_publish_word(hipStream_t st, const unsigned long long* src, unsigned long long* dst_mapped);

// PopModel interval histograms: counts[b * n_items + i] = events of item i with bounds[b] <= t < bounds[b + 1], b < n_buckets <= 3
hipError_t launch_pop_counts(hipStream_t st, int n_cu, int64_t n, const int32_t* item, const int64_t* t_ms, int32_t n_items, int n_buckets,
                             const int64_t* bounds, int32_t* counts);

// out[e] = col_idx[e] | counts16[col_idx[e]] << (32 - count_bits) for e < *nnz_dev (<= nnz_bound); counts16 / bad16 as launch_narrow_counts leaves them;
// bad[0] (zeroed here) = counts that do not fit (or 1 when *bad16 != 0: nothing packed)
hipError_t launch_pack_counts(hipStream_t st, int n_cu, const int32_t* col_idx, const int64_t* nnz_dev, int64_t nnz_bound, const unsigned short* counts16,
                              const int32_t* bad16, int32_t count_bits, int32_t* out, int32_t* bad);
hipError_t launch_xlx_table(hipStream_t st, double* tab);
hipError_t launch_xlx_hi_table(hipStream_t st, double* tab /*[2 * XLX_TABLE_HOST]: xlx_hi, then col_ent*/, const double* xlx_tab, long long n_users);
hipError_t launch_item_entropy(hipStream_t st, const int32_t* counts, int32_t n, long long n_users, double* ent, double* xlx_n);
// out16[i] = counts[i] (low 16 bits); bad[0] = number of counts that do not fit
hipError_t launch_narrow_counts(hipStream_t st, int n_cu, const int32_t* counts, int64_t n, unsigned short* out16, int32_t* bad);  // n: 64-bit (world x shard rows)

// pstart[cap], plen[cap] (scratch), wp[cap + 1]; cap >= nnz(A'); tile_sums scratch as for scans over cap elements
hipError_t launch_expand_prepare(hipStream_t st, int n_cu, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* a_row_idx,
                                 const int64_t* b_row_ptr, unsigned* b_rp32_scratch /* nullable: n_rows_b + 1 words */, int64_t n_rows_b, int64_t cap,
                                 int64_t* pstart, int32_t* plen, int64_t* wp, int64_t* tile_sums);
// the same for up to EXPAND_MULTI_MAX event types with ONE gather per CSC entry: T = scratch of n_rows_b * n * 8 bytes
constexpr int EXPAND_MULTI_MAX = 8;
hipError_t launch_expand_prepare_multi(hipStream_t st, int n_cu, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* a_row_idx, int n,
                                       const int64_t* const* b_row_ptr, int64_t n_rows_b, int64_t cap, int64_t* const* pstart, int32_t* const* plen, void* T,
                                       int64_t* const* tsum /* nullable; tsum[d]: ceil(cap / 2048) + 2 words: the scan-tile sums of plen[d] */);
hipError_t launch_expand_scan(hipStream_t st, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* plen, int64_t cap, int64_t* wp, int64_t* tile_sums,
                              bool tile_sums_ready = false);
hipError_t launch_row_work(hipStream_t st, int n_cu, int32_t item_lo, int32_t item_hi, const int64_t* a_col_ptr, const int64_t* wp, int64_t* work);

// binning: tile_counts scratch [(ceil(n/BIN_TILE)+1) * BIN_COLS_HOST] int64;
// bin_off[BIN_OFF_LEN] int32, bin_rows[n] int32, stats[STATS_LEN] int64.
hipError_t launch_binning(hipStream_t st, int32_t item_lo, int32_t n, const int64_t* work, const int32_t* cnt_a, int32_t n_cols_b,
                          int32_t count_bits, int32_t k, int64_t* tile_counts, int32_t* bin_off, int32_t* bin_rows, int64_t* stats);

hipError_t launch_cco_rows_bin(hipStream_t st, int n_cu, const CcoArgs& args, int bin, int32_t n_rows /* item rows of the build: bounds the grids */);
hipError_t launch_bin_out_stats(hipStream_t st, const int32_t* bin_rows, const int32_t* bin_off, int32_t item_lo, const int32_t* out_count,
                                const unsigned long long* cand, int64_t* stats);

hipError_t launch_compact_indicators(hipStream_t st, int32_t n_rows, int32_t k, const int32_t* count, const int32_t* idx,
                                     const double* llr, const int64_t* row_ptr, int32_t* out_idx, double* out_llr);

// splits: bounds[p] = first item whose exclusive work prefix >= p * total / n_parts (prefix = scan of work)
hipError_t launch_partition(hipStream_t st, int32_t n_items, const int64_t* work_prefix, int32_t n_parts, int32_t* bounds);
hipError_t launch_scan_i64(hipStream_t st, const int64_t* in, int64_t n, int64_t* out, int64_t* tile_sums);

hipError_t launch_llr_test(hipStream_t st, int64_t n, const int64_t* a, const int64_t* b, const int64_t* ab, const int64_t* nu, double* out);
hipError_t launch_u01_test(hipStream_t st, int64_t n, uint32_t seed, const int32_t* row, const int32_t* col, double* out, int rng32 = 0);

}  // namespace urcco
