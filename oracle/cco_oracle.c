/* CPU ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the reference's
 * Correlated Cross-Occurrence arithmetic on CSR inputs.  Same algorithm as oracle/cco_oracle.py
 * (which carries the full list of reference citations and of reference-unpinned decisions D4..D14);
 * this file exists so that parity tests at 10^5..10^8 pairs and bench.py's cpu_baseline finish in seconds.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Follows (reference = /root/reference, Mahout 0.13.0 = un-vendored dependency, build.sbt:15,34-40):
 *   orc_llr                 Mahout math/stats/LogLikelihood.java logLikelihoodRatio + SimilarityAnalysis.logLikelihoodRatio
 *   orc_downsample          Mahout SimilarityAnalysis.sampleDownAndBinarize (called via URAlgorithm.scala:323-328, :336-340)
 *   orc_column_counts       Mahout numNonZeroElementsPerColumn
 *   orc_transpose + orc_cco_rows   Mahout `A.t %*% B` + SimilarityAnalysis.computeSimilarities; output order of
 *                           package.scala:102 (score desc) with the canonical tie rule D7 (column index asc)
 * PARITY PINNING: see oracle/cco_oracle.py header -- pinned through the reference's two integration goldens
 * (membership) and known-answer LLRs; everything finer is reference-unpinned and defined here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_ROW_RATE_MAHOUT_INT_DIV 0
#define ORC_ROW_RATE_FRACTIONAL 1
#define ORC_RNG_MIX32 0x100 /* OR-ed into row_rate_mode: the 32-bit form of the down-sampling RNG (D10 b) instead of the 53-bit one */

/* ---- LogLikelihood.java ---------------------------------------------------------------- */
static inline double x_log_x(int64_t x) { return x == 0 ? 0.0 : (double)x * log((double)x); }
static inline double entropy2(int64_t a, int64_t b) { return x_log_x(a + b) - x_log_x(a) - x_log_x(b); }
static inline double entropy4(int64_t a, int64_t b, int64_t c, int64_t d) {
  return x_log_x(a + b + c + d) - x_log_x(a) - x_log_x(b) - x_log_x(c) - x_log_x(d);
}

double orc_llr_k(int64_t k11, int64_t k12, int64_t k21, int64_t k22) {
  double row_entropy = entropy2(k11 + k12, k21 + k22);
  double column_entropy = entropy2(k11 + k21, k12 + k22);
  double matrix_entropy = entropy4(k11, k12, k21, k22);
  if (row_entropy + column_entropy < matrix_entropy) return 0.0; /* round off error */
  return 2.0 * (row_entropy + column_entropy - matrix_entropy);
}

/* SimilarityAnalysis.logLikelihoodRatio(numInteractionsWithA, ..WithB, ..WithAandB, numInteractions) */
double orc_llr(int64_t with_a, int64_t with_b, int64_t with_ab, int64_t n) {
  return orc_llr_k(with_ab, with_a - with_ab, with_b - with_ab, n - with_a - with_b + with_ab);
}

/* ---- D10: stateless RNG, identical to cco_oracle.py::u01 ------------------------------- */
double orc_u01(uint32_t seed, uint32_t row, uint32_t col) {
  uint64_t x = ((uint64_t)row << 32) | (uint64_t)col;
  x ^= (uint64_t)seed * 0x9E3779B97F4A7C15ull;
  x += 0x9E3779B97F4A7C15ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* D10 (b), ORC_RNG_MIX32: identical to cco_oracle.py::mix32 / u01_mix32 */
uint32_t orc_mix32(uint32_t seed, uint32_t row, uint32_t col) {
  uint32_t x = col ^ (row * 0x9E3779B1u + seed * 0x85EBCA77u + 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
double orc_u01_mix32(uint32_t seed, uint32_t row, uint32_t col) { return (double)orc_mix32(seed, row, col) * (1.0 / 4294967296.0); }

void orc_column_counts(int64_t nnz, const int32_t* col_idx, int32_t n_cols, int32_t* counts) {
  memset(counts, 0, sizeof(int32_t) * (size_t)n_cols);
  for (int64_t e = 0; e < nnz; ++e) counts[col_idx[e]]++;
}

/* keep decision of sampleDownAndBinarize for one interaction (r = global row index) */
static inline int orc_keep(uint32_t seed, int64_t r, int32_t j, int64_t n_row, const int32_t* raw_counts, int32_t max_n, int row_rate_mode) {
  int64_t capped = n_row < max_n ? n_row : max_n;
  double per_row = (row_rate_mode & 0xff) == ORC_ROW_RATE_MAHOUT_INT_DIV ? (double)(capped / n_row) /* Int / Int (D9) */
                                                                         : (double)capped / (double)n_row;
  double n_thing = (double)raw_counts[j];
  double per_thing = (n_thing < (double)max_n ? n_thing : (double)max_n) / n_thing;
  double rate = per_row < per_thing ? per_row : per_thing;
  return ((row_rate_mode & ORC_RNG_MIX32) ? orc_u01_mix32(seed, (uint32_t)r, (uint32_t)j) : orc_u01(seed, (uint32_t)r, (uint32_t)j)) <= rate;
}

/* sampleDownAndBinarize.  raw_counts = column counts of the RAW matrix (all users, D11).
 * row_base = global index of local row 0 (sharded inputs keep the same RNG stream).
 * out_rp has n_rows+1 entries, out_ci capacity = nnz.  Returns kept nnz.
 * The RNG is stateless (D10), so rows are independent: counted in parallel, prefix-summed, written in parallel --
 * the result is the one the sequential loop over rows gives. */
int64_t orc_downsample(int64_t n_rows, const int64_t* rp, const int32_t* ci, const int32_t* raw_counts, uint32_t seed,
                       int32_t max_n, int row_rate_mode, int64_t row_base, int64_t* out_rp, int32_t* out_ci) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 4096)
#endif
  for (int64_t r = 0; r < n_rows; ++r) {
    int64_t n_row = rp[r + 1] - rp[r], kept = 0;
    for (int64_t e = rp[r]; e < rp[r + 1]; ++e) kept += orc_keep(seed, row_base + r, ci[e], n_row, raw_counts, max_n, row_rate_mode);
    out_rp[r + 1] = kept;
  }
  out_rp[0] = 0;
  for (int64_t r = 0; r < n_rows; ++r) out_rp[r + 1] += out_rp[r];
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 4096)
#endif
  for (int64_t r = 0; r < n_rows; ++r) {
    int64_t n_row = rp[r + 1] - rp[r], w = out_rp[r];
    for (int64_t e = rp[r]; e < rp[r + 1]; ++e)
      if (orc_keep(seed, row_base + r, ci[e], n_row, raw_counts, max_n, row_rate_mode)) out_ci[w++] = ci[e];
  }
  return out_rp[n_rows];
}

/* CSR (n_rows x n_cols) -> CSC; rows inside a column ascending.  Threads own disjoint column ranges and each walks
 * the whole matrix in row order (streaming, no shared writes), which keeps the ascending order of the sequential loop. */
void orc_transpose(int64_t n_rows, const int64_t* rp, const int32_t* ci, int32_t n_cols, int64_t* col_ptr, int32_t* row_idx) {
  memset(col_ptr, 0, sizeof(int64_t) * ((size_t)n_cols + 1));
  int64_t nnz = rp[n_rows];
  for (int64_t e = 0; e < nnz; ++e) col_ptr[ci[e] + 1]++;
  for (int32_t j = 0; j < n_cols; ++j) col_ptr[j + 1] += col_ptr[j];
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_cols > 0 ? n_cols : 1));
  memcpy(cur, col_ptr, sizeof(int64_t) * (size_t)n_cols);
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    int t = 0, nt = 1;
#ifdef _OPENMP
    t = omp_get_thread_num(); nt = omp_get_num_threads();
#endif
    if (nnz < (1 << 22)) nt = 1;             /* small matrices: one pass by thread 0 */
    if (t < nt) {
      /* column ranges of ~equal nnz */
      int32_t lo = 0, hi = n_cols;
      if (nt > 1) {
        int64_t t0 = nnz / nt * t, t1 = t == nt - 1 ? nnz : nnz / nt * (t + 1);
        int32_t a = 0, b = n_cols;
        while (a < b) { int32_t m = a + (b - a) / 2; if (col_ptr[m] >= t0) b = m; else a = m + 1; }
        lo = a; a = 0; b = n_cols;
        while (a < b) { int32_t m = a + (b - a) / 2; if (col_ptr[m] >= t1) b = m; else a = m + 1; }
        hi = t == nt - 1 ? n_cols : a;
      }
      for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t e = rp[r]; e < rp[r + 1]; ++e) {
          int32_t j = ci[e];
          if (j >= lo && j < hi) row_idx[cur[j]++] = (int32_t)r;
        }
    }
  }
  free(cur);
}

typedef struct { double llr; int32_t j; } cand_t;

/* canonical order D7: score desc, column asc; returns <0 when a ranks before b */
static inline int cand_before(const cand_t* a, const cand_t* b) {
  if (a->llr != b->llr) return a->llr > b->llr;
  return a->j < b->j;
}
static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa; const cand_t* b = (const cand_t*)pb;
  if (cand_before(a, b)) return -1;
  if (cand_before(b, a)) return 1;
  return 0;
}

/* Rows [item_lo, item_hi) of A'B, LLR-scored and cut to the top k (computeSimilarities).
 *   a_col_ptr/a_row_idx : CSC of down-sampled A (users of each item)
 *   b_rp/b_ci           : CSR of down-sampled B (items of each user), n_cols_b columns
 *   cnt_a/cnt_b         : post-sampling column counts; n_users = nrow of the DRMs
 *   exclude_self        : 1 for A'A (crossCooccurrence = false)
 *   has_min_llr/min_llr : D12
 * Output is strided: row r (= item_lo + r) holds out_count[r] <= k entries at out_idx/out_llr[r*k ..],
 * sorted (llr desc, col asc); llr == 0.0 entries are dropped after the cut (D5).
 * Returns the number of cooccurrence pairs formed (sum over (i,u) of d_B(u)). */
int64_t orc_cco_rows(int32_t item_lo, int32_t item_hi, const int64_t* a_col_ptr, const int32_t* a_row_idx,
                     const int64_t* b_rp, const int32_t* b_ci, int32_t n_cols_b, const int32_t* cnt_a, const int32_t* cnt_b,
                     int64_t n_users, int exclude_self, int32_t k, int has_min_llr, double min_llr, int32_t* out_count,
                     int32_t* out_idx, double* out_llr, int n_threads) {
  int64_t pairs_total = 0;
#ifdef _OPENMP
  /* a num_threads clause, NOT omp_set_num_threads: the setter is process-wide and sticky -- a one-thread call (the small parity
   * tests) used to leave every later orc_downsample / orc_transpose, and orc_max_threads itself, single-threaded for the rest of
   * the process: the 10M-row tests then ran 6-9x longer when they came after the small ones */
  const int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt) reduction(+ : pairs_total)
#endif
  {
    int32_t* acc = (int32_t*)calloc((size_t)(n_cols_b > 0 ? n_cols_b : 1), sizeof(int32_t));
    int64_t cap = 1024;
    int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    cand_t* cands = (cand_t*)malloc(sizeof(cand_t) * (size_t)cap);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (int32_t i = item_lo; i < item_hi; ++i) {
      int64_t nt = 0;
      for (int64_t p = a_col_ptr[i]; p < a_col_ptr[i + 1]; ++p) {
        int32_t u = a_row_idx[p];
        pairs_total += b_rp[u + 1] - b_rp[u];
        for (int64_t e = b_rp[u]; e < b_rp[u + 1]; ++e) {
          int32_t j = b_ci[e];
          if (acc[j]++ == 0) {
            if (nt == cap) {
              cap *= 2;
              touched = (int32_t*)realloc(touched, sizeof(int32_t) * (size_t)cap);
              cands = (cand_t*)realloc(cands, sizeof(cand_t) * (size_t)cap);
            }
            touched[nt++] = j;
          }
        }
      }
      int64_t nc = 0;
      for (int64_t t = 0; t < nt; ++t) {
        int32_t j = touched[t];
        int32_t k11 = acc[j];
        acc[j] = 0;
        if (exclude_self && j == i) continue; /* D4 */
        double llr = orc_llr(cnt_a[i], cnt_b[j], k11, n_users);
        if (has_min_llr && !(llr >= min_llr)) continue; /* D12 */
        cands[nc].llr = llr; cands[nc].j = j; ++nc;
      }
      qsort(cands, (size_t)nc, sizeof(cand_t), cand_cmp);
      int64_t r = (int64_t)(i - item_lo);
      int32_t w = 0;
      for (int64_t t = 0; t < nc && t < k; ++t) {
        if (cands[t].llr == 0.0) continue; /* D5: a 0.0 never materialises in the sparse row */
        out_idx[r * k + w] = cands[t].j;
        out_llr[r * k + w] = cands[t].llr;
        ++w;
      }
      out_count[r] = w;
    }
    free(acc); free(touched); free(cands);
  }
  return pairs_total;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
