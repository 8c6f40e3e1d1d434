/* urcco.h -- C ABI of liburcco: MI355X-native (gfx950 / CDNA4, hand-written HIP) Correlated
 * Cross-Occurrence model build, the drop-in for the two Mahout calls the Universal Recommender makes
 * from URAlgorithm.calcAll:
 *
 *   SimilarityAnalysis.cooccurrencesIDSs(...)            reference src/main/scala/URAlgorithm.scala:323-329
 *   SimilarityAnalysis.crossOccurrenceDownsampled(...)   reference src/main/scala/URAlgorithm.scala:343-346
 *
 * (Mahout 0.13.0: math-scala/.../math/cf/SimilarityAnalysis.scala; un-vendored dependency, build.sbt:15,34-40.)
 *
 * Two levels:
 *   1. HOST level (what a JNI shim binds): host CSR in, host indicator CSR out.
 *        urcco_cooccurrences_idss / urcco_cross_occurrence_downsampled / urcco_free_indicators
 *   2. DEVICE level (what the per-GPU ranks of a multi-GPU job and bench.py drive): every pointer is a
 *      device pointer, nothing is copied, collectives between stages are the caller's (RCCL through
 *      torch.distributed).  Stage functions only enqueue on the session's HIP stream unless they say
 *      "synchronises".
 *
 * Conventions: plain C, no exceptions or aborts cross the boundary; every function returns a
 * urcco_status (0 = OK) and leaves a message for urcco_last_error() (thread local).  Matrices are
 * binary: values are implicit 1 (Preparator.scala:146,205 stores 1.0 for every event).
 * The library has no CPU fallback: without a HIP device every compute entry point returns URCCO_NO_DEVICE.
 */
#ifndef URCCO_H
#define URCCO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URCCO_VERSION 305 /* 0.3.5: urcco_dev_pack_counts, urcco_dev_cco_rows_packed (304: URCCO_BUSY, urcco_cross_occurrence_cancel_any) */

typedef enum urcco_status {
  URCCO_OK = 0,
  URCCO_BAD_ARG = 1,     /* the reference throws IllegalArgumentException for these (URAlgorithm.scala:225,232) */
  URCCO_OOM_HOST = 2,
  URCCO_OOM_DEVICE = 3,
  URCCO_HIP_ERROR = 4,
  URCCO_INTERNAL = 5,
  URCCO_NO_DEVICE = 6,
  URCCO_RCCL_ERROR = 7,  /* a collective failed, or more than one GPU was asked for and librccl could not be loaded */
  URCCO_BUSY = 8         /* the process-wide context holds ANOTHER thread's staged build (stage: not finished within the wait; cancel: not the caller's to discard) */
} urcco_status;

/* D9: how sampleDownAndBinarize's perRowSampleRate is evaluated */
#define URCCO_ROW_RATE_MAHOUT_INT_DIV 0 /* Int / Int as in Mahout 0.13.0: rows with more than max non-zeros are dropped */
#define URCCO_ROW_RATE_FRACTIONAL 1     /* min(max, n) / n in floating point */
/* The down-sampling RNG (oracle decision D10: the reference pins none -- Mahout draws from a java.util.Random per Spark partition), OR-ed
 * into row_rate_mode wherever that travels (urcco_options.row_rate_mode, urcco_dev_downsample): a stateless uniform keyed by
 * (seed, global row, column), identical in oracle/cco_oracle.{py,c} and on the device. */
#define URCCO_RNG_SPLITMIX53 0         /* 64-bit splitmix finaliser, 53-bit uniform (default) */
#define URCCO_RNG_MIX32 0x100          /* 32-bit: column xor (row, seed) key, two-round multiply-xorshift; ~10 instead of ~25 instructions per interaction.
                                        * Known structure: the seed and the row enter the key LINEARLY (key = row * C1 + seed * C2 + C3), so two seeds give the
                                        * same stream up to a constant shift of the row index, and hash(row1, col) == hash(row2, col ^ key1 ^ key2): different seeds are
                                        * relabelings of ONE sample family, not independent samples of the whole matrix.  Per-entry uniformity and the per-column keep
                                        * rates are unaffected (tests/test_oracle.py); a caller that needs independent re-samples across seeds uses the default. */

/* One IndexedDataset.matrix (user x item, binary), rows = the shared user dictionary
 * (Preparator.scala:44-87).  row_ptr has n_rows + 1 entries; col_idx is sorted and unique inside a row. */
typedef struct urcco_csr {
  int64_t n_rows;
  int64_t n_cols;
  const int64_t* row_ptr;
  const int32_t* col_idx;
} urcco_csr;

/* Mahout DownsamplableCrossOccurrenceDataset(iD, maxElementsPerRow, maxInterestingElements, minLLROpt)
 * as URAlgorithm.scala:334-341 fills it from engine.json `indicators[i]`. */
typedef struct urcco_dataset {
  urcco_csr matrix;
  int32_t max_elements_per_row;     /* indicators[i].maxItemsPerUser      (default 500, URAlgorithm.scala:54) */
  int32_t max_interesting_elements; /* indicators[i].maxCorrelatorsPerItem (default 50,  URAlgorithm.scala:56) */
  double min_llr;                   /* indicators[i].minLLR */
  int32_t has_min_llr;              /* 0 = None */
  int32_t reserved;
} urcco_dataset;

typedef struct urcco_options {
  int32_t device;        /* HIP device ordinal of the first GPU */
  int32_t row_rate_mode; /* URCCO_ROW_RATE_* | URCCO_RNG_* */
  int32_t n_gpus;        /* GPUs of THIS process to use, starting at `device`; 0 = every visible one (engine.json "numGPUs") */
  int32_t flags;         /* URCCO_FLAG_* */
  int32_t reserved[4];
} urcco_options;
#define URCCO_FLAG_SINGLE_STREAM 1  /* run the event types back to back on one HIP stream per GPU (profiling) */
#define URCCO_FLAG_FORCE_EXCHANGE 2 /* run the multi-GPU exchange path (collectives, work-balanced ranges) even with one rank */
#define URCCO_FLAG_UNORDERED_ROWS 4 /* indicator rows carry their top-k SET in unspecified (run-dependent) order: what Mahout's
                                       computeSimilarities returns -- a sparse vector has no score order, the reference sorts
                                       later (toStringMapRDD, package.scala:102) and a JNI host re-inserts by index anyway.
                                       Saves the in-kernel ranking pass.  Default off: rows ordered (llr desc, col asc). */

#define URCCO_FLAG_EMULATE_RANKS 8  /* MEASUREMENT ONLY (bench.py --emulate-ranks): the context's n_gpus ranks all run on the ONE device
                                       `device`, one after the other on a single stream (every kernel alone on the GPU), so that the per-rank
                                       critical path of a W-rank build -- user-range input phase, own-shard transposition, masks, packing,
                                       fragment merge, fused expand, item-range SpGEMM -- can be timed without W GPUs.  Needs caller-supplied
                                       collectives (urcco_comm_config.collectives; RCCL cannot put two ranks on one device) and the device
                                       level (urcco_context_build_device).  Results are the real W-rank results; xGMI time is not in them. */

/* One returned IndexedDataset: rows = items of the primary matrix A (rowIDs = A.columnIDs), columns = items
 * of B_i (columnIDs = B_i.columnIDs), values = raw LLR.  Inside a row entries are ordered (llr desc, col asc)
 * -- the order package.scala:102 (`sortBy(-score)`) produces, with the canonical tie rule.  Library-owned
 * host memory; release with urcco_free_indicators. */
typedef struct urcco_indicators {
  int64_t n_rows;
  int64_t n_cols;
  int64_t nnz;
  int64_t* row_ptr; /* n_rows + 1 */
  int32_t* col_idx;
  double* llr;
} urcco_indicators;

#define URCCO_N_BINS 7
typedef struct urcco_dataset_stats {
  int64_t nnz_raw;     /* interactions before down-sampling */
  int64_t nnz_sampled; /* after sampleDownAndBinarize */
  int64_t pairs;       /* cooccurrence pairs formed: sum_u d_A'(u) * d_B'(u) (the metric's unit) */
  int64_t nnz_out;     /* indicator entries emitted */
  int64_t rows_by_bin[7]; /* item rows per accumulator class: micro, wave, small block, block, half CU, CU, global (URCCO_N_BINS) */
  double ms_total;     /* device time of this dataset's stages (HIP events) */
} urcco_dataset_stats;

int urcco_version(void);
int urcco_device_count(void); /* 0 when no HIP device is visible */
/* Tears down the process-wide context the one-shot entry points below create on first use (streams, scratch arenas,
 * pinned staging buffers, RCCL communicators).  Safe to call when nothing was created; the next call re-creates it. */
int urcco_shutdown(void);
const char* urcco_last_error(void);
const char* urcco_status_string(int status);

/* ---- HOST level ------------------------------------------------------------------------------------- */

/* SimilarityAnalysis.cooccurrencesIDSs(indexedDatasets, randomSeed, maxInterestingItemsPerThing,
 * maxNumInteractions): datasets[0] is the primary.  out[n_datasets] receives A'A then A'B_i.
 * stats may be NULL, else stats[n_datasets]. */
int urcco_cooccurrences_idss(const urcco_csr* datasets, int32_t n_datasets, int32_t random_seed,
                             int32_t max_interesting_items_per_thing, int32_t max_num_interactions,
                             const urcco_options* options, urcco_indicators* out, urcco_dataset_stats* stats);

/* SimilarityAnalysis.crossOccurrenceDownsampled(datasets, randomSeed). */
int urcco_cross_occurrence_downsampled(const urcco_dataset* datasets, int32_t n_datasets, int32_t random_seed,
                                       const urcco_options* options, urcco_indicators* out, urcco_dataset_stats* stats);

void urcco_free_indicators(urcco_indicators* indicators, int32_t n);

/* The same call in two halves, for a caller whose input arrays are only pinned for a short while -- the JNI shim holds them
 * between Get/ReleasePrimitiveArrayCritical, which locks the JVM's garbage collector out (jni/urcco_jni.cpp):
 *   _stage   returns as soon as the library has finished READING datasets[*].matrix (every byte sits in its pinned staging ring
 *            or in HBM; the model build is already running behind the copies).  The caller may release its arrays.
 *   _finish  waits for the build and hands out the indicator matrices (n_datasets = the staged count).
 * Exactly one _finish per successful _stage, from the same thread or another; urcco_shutdown abandons a staged build.
 * urcco_cross_occurrence_downsampled == _stage followed by _finish.  out[] is zeroed on entry of every function that takes
 * it, before any fallible step: on failure the caller owns nothing and must free nothing. */
int urcco_cross_occurrence_stage(const urcco_dataset* datasets, int32_t n_datasets, int32_t random_seed, const urcco_options* options);
int urcco_cross_occurrence_finish(urcco_indicators* out, int32_t n_datasets, urcco_dataset_stats* stats);
/* The process-wide context behind these calls holds ONE build at a time: a second thread's _stage waits until the first thread's
 * _finish (at most URCCO_STAGE_WAIT_S seconds, default 600, then URCCO_BUSY -- a status of its own, so that a caller can tell "occupied"
 * from a real internal fault), the same thread staging twice is URCCO_BAD_ARG.
 * A _finish whose n_datasets differs from the staged count DISCARDS the staged build (URCCO_BAD_ARG; the context is free again).
 * _cancel discards the CALLING thread's own staged build (a caller that gave up between the halves calls it so that others do not wait);
 * nothing staged is a no-op, another thread's build is left alone (URCCO_BUSY): a thread that merely timed out in _stage must not be able
 * to throw away the owner's legitimately staged build.  _cancel_any discards whatever is staged, whoever staged it -- for a supervisor
 * that KNOWS the owner is gone (a thread that staged and died); the owner's _finish then reports "nothing staged". */
int urcco_cross_occurrence_cancel(void);
int urcco_cross_occurrence_cancel_any(void);

/* ---- CONTEXT level: the persistent form of the host level, and the multi-GPU build ------------------------------
 * A context owns, per GPU, one HIP stream + scratch arena per event type, every intermediate and output buffer (grown
 * on demand, reused by the next build), pinned staging memory and -- with more than one rank -- a communicator.  The
 * one-shot functions above run on a lazily created process-wide context (urcco_shutdown frees it); a long-lived host
 * (the Spark driver JVM) may also hold its own.  Not re-entrant: one build at a time per context, as URAlgorithm.train
 * calls it (URAlgorithm.scala:292-306, driver thread).
 *
 * Ranks: a build runs on world_size GPUs ("ranks").  Either ONE process drives them all (what a JVM does: comm = NULL,
 * n_gpus of the options, collectives through RCCL communicators created with ncclCommInitAll), or one process per GPU
 * (what bench.py does under torch.distributed.run: world_size / first_rank / nccl_unique_id filled in, n_gpus = 1).
 * Users are range-sharded over the ranks for the input phase (column counts, sampleDownAndBinarize), items of the
 * primary matrix are range-partitioned by work for the compute phase; the exchange between the two is
 * two all-reduces of column counts per event type, one all-reduce of the row-work key and one all-gather-v of the
 * down-sampled shards. */
typedef struct urcco_context urcco_context;

/* Replacement for the built-in RCCL collectives (the CPU test-suite runs the multi-rank build over `gloo` through this;
 * product callers pass NULL).  `rank` = global rank of the calling GPU; calls between group_start and group_end are
 * one collective each across all ranks and may complete at group_end -- the offset / count arrays handed to a callback stay
 * valid until that group_end returns (the data buffers until the stream has passed the collective).  stream = the HIP stream the
 * buffers are produced / consumed on.  Return 0 on success. */
typedef struct urcco_collectives {
  size_t struct_size; /* sizeof(urcco_collectives) of the header the CALLER was compiled against (since 302): a struct shorter than
                         the library's is rejected by urcco_context_create instead of being read past its end */
  void* user;
  int (*group_start)(void* user);
  int (*group_end)(void* user);
  /* in place; dtype 0 = int32, 1 = int64 */
  int (*all_reduce_sum)(void* user, int32_t rank, void* buf, int64_t count, int32_t dtype, void* stream);
  /* recv + byte_offsets[r] receives byte_counts[r] bytes = rank r's send buffer; send holds byte_counts[rank] bytes */
  int (*all_gather_v)(void* user, int32_t rank, const void* send, void* recv, const int64_t* byte_offsets,
                      const int64_t* byte_counts, void* stream);
  /* send + send_offsets[q] holds send_counts[q] bytes for rank q; recv + recv_offsets[p] receives recv_counts[p] bytes
   * from rank p (the CSC fragments of the primary).  Since 301. */
  int (*all_to_all_v)(void* user, int32_t rank, const void* send, const int64_t* send_offsets, const int64_t* send_counts,
                      void* recv, const int64_t* recv_offsets, const int64_t* recv_counts, void* stream);
} urcco_collectives;

typedef struct urcco_comm_config {
  int32_t world_size;         /* ranks of the job; 0 = the options' n_gpus (single process) */
  int32_t first_rank;         /* global rank of this process's first GPU */
  const void* nccl_unique_id; /* URCCO_UNIQUE_ID_BYTES from urcco_comm_unique_id on one rank, distributed by the caller
                                 (NULL when one process holds every rank) */
  const urcco_collectives* collectives; /* NULL = RCCL */
} urcco_comm_config;
#define URCCO_UNIQUE_ID_BYTES 128
int urcco_comm_unique_id(void* out /*[URCCO_UNIQUE_ID_BYTES]*/);

int urcco_context_create(const urcco_options* options, const urcco_comm_config* comm, urcco_context** out);
void urcco_context_destroy(urcco_context* ctx);
int32_t urcco_context_local_gpus(const urcco_context* ctx);

/* SimilarityAnalysis.crossOccurrenceDownsampled on a context: host CSR in (caller-owned, pageable is fine: staged
 * through pinned memory by a few copy threads while the GPU already works), host indicator CSR out (pinned memory of the
 * context's pool; release with urcco_free_indicators).  Single-process contexts only. */
int urcco_context_cross_occurrence(urcco_context* ctx, const urcco_dataset* datasets, int32_t n_datasets, int32_t random_seed,
                                   urcco_indicators* out, urcco_dataset_stats* stats);
/* ... and its two halves (see urcco_cross_occurrence_stage): out[n_datasets of the stage call] */
int urcco_context_stage(urcco_context* ctx, const urcco_dataset* datasets, int32_t n_datasets, int32_t random_seed);
int urcco_context_finish(urcco_context* ctx, urcco_indicators* out, urcco_dataset_stats* stats);

/* The same build with the matrices already resident in HBM -- what bench.py times.  Per event type and per local GPU
 * one user-range shard (rows [row_base, row_base + n_rows) of the n_users_total x n_cols matrix). */
typedef struct urcco_dev_shard {
  int64_t n_rows;
  int64_t row_base;
  const int64_t* row_ptr; /* device, n_rows + 1, starts at 0 */
  const int32_t* col_idx; /* device */
  int64_t nnz;            /* == row_ptr[n_rows] (known to the host) */
} urcco_dev_shard;
typedef struct urcco_dev_dataset {
  int64_t n_cols;
  int32_t max_elements_per_row;
  int32_t max_interesting_elements;
  double min_llr;
  int32_t has_min_llr;
  int32_t reserved;
  const urcco_dev_shard* shards; /* [urcco_context_local_gpus] */
} urcco_dev_dataset;
/* One indicator matrix slice: rows = items [item_lo, item_hi) of A.  Device memory owned by the context, valid until
 * its next build or destruction. */
typedef struct urcco_dev_result {
  int32_t item_lo, item_hi;
  const int64_t* row_ptr;  /* item_hi - item_lo + 1 */
  const int32_t* col_idx;
  const double* llr;
  const int64_t* stats;    /* URCCO_STATS_LEN, see urcco_dev_cco_rows */
  /* The down-sampled B this GPU multiplied with, sampled_rows + 1 row pointers.  One rank: the whole matrix.  Several ranks: a row per
   * user of the job, but (row-filtered exchange, the default) only the rows of users holding an item of THIS rank's item range are
   * filled -- every other row is empty, and sampled_row_ptr[sampled_rows] is this rank's filtered entry count, not the matrix's.  Debug
   * bit 16384 (urcco_context_set_debug) exchanges every row instead.  The exchange is also unfiltered when the job has more than 64
   * ranks or runs on caller-supplied collectives without all_to_all_v. */
  const int64_t* sampled_row_ptr;
  const int32_t* sampled_col_idx;
  int64_t sampled_rows;
  /* entries of the WHOLE down-sampled matrix, over all ranks (host-known after the exchange); -1 on a one-rank build, where the host
   * never reads a size back: there it equals sampled_row_ptr[sampled_rows] on the device */
  int64_t sampled_nnz_total;
  /* ABI 305: sampled_col_idx[e] & sampled_col_mask is the column index.  -1 (every bit) on a one-rank build.  On a sharded build the rows a rank received
   * travelled with their column's post-sampling count in the bits above the column (the words the row kernels consume: see urcco_dev_pack_counts) whenever
   * every count fits; the mask then keeps the column's bits. */
  int32_t sampled_col_mask;
} urcco_dev_result;
/* input_stream (nullable hipStream_t): the stream the shards were produced on -- the build waits for it on the device.
 * out[d * local_gpus + g].  Returns after ENQUEUEING (except for the one blocking read of range bounds / shard sizes
 * with more than one rank); follow with urcco_context_wait_stream or urcco_context_synchronize. */
int urcco_context_build_device(urcco_context* ctx, const urcco_dev_dataset* datasets, int32_t n_datasets, int64_t n_users_total,
                               int32_t random_seed, void* input_stream, urcco_dev_result* out);
int urcco_context_wait_stream(urcco_context* ctx, void* stream); /* `stream` waits (on the device) for the last build */
int urcco_context_synchronize(urcco_context* ctx);               /* the host waits */
/* per-stage timing / ablation switches of every session of the context (see urcco_session_set_timing / _set_debug) */
int urcco_context_set_timing(urcco_context* ctx, int32_t enable);
int urcco_context_get_timings(urcco_context* ctx, double* ms, int64_t* launches);
/* the same for ONE local GPU (rank first_rank + local_gpu): what bench.py --emulate-ranks reads per rank */
int urcco_context_get_timings_gpu(urcco_context* ctx, int32_t local_gpu, double* ms, int64_t* launches);
int urcco_context_set_debug(urcco_context* ctx, int32_t flags);
int urcco_context_set_flags(urcco_context* ctx, int32_t flags); /* URCCO_FLAG_* */

/* ---- DEVICE level ----------------------------------------------------------------------------------- */

typedef struct urcco_session urcco_session;

/* stream: a hipStream_t (NULL = the library creates its own).  The session owns a scratch arena that grows on
 * demand and is reused by every stage. */
int urcco_session_create(int32_t device, void* stream, urcco_session** out);
void urcco_session_destroy(urcco_session* s);
int urcco_session_synchronize(urcco_session* s);
/* Per-stage device timing with HIP events on the session's stream (what bench.py's roofline numbers are made of).
 * set_timing resets the accumulators; get_timings synchronises and returns, per stage id, the summed event time
 * in ms and the number of timed launches. */
#define URCCO_N_STAGES 17
enum {
  URCCO_STAGE_COLUMN_COUNTS = 0,
  URCCO_STAGE_DOWNSAMPLE_FLAGS = 1,
  URCCO_STAGE_DOWNSAMPLE_SCAN = 2,
  URCCO_STAGE_DOWNSAMPLE_COMPACT = 3,
  URCCO_STAGE_TRANSPOSE = 4,
  URCCO_STAGE_ROW_WORK = 5,
  URCCO_STAGE_BINNING = 6,
  URCCO_STAGE_ENTROPY = 7,
  URCCO_STAGE_CCO_BIN0 = 8,  /* micro rows (<= 64 pairs)    */
  URCCO_STAGE_CCO_BIN1 = 9,  /* wave-LDS accumulator rows   */
  URCCO_STAGE_CCO_BIN2 = 10, /* small block-LDS rows        */
  URCCO_STAGE_CCO_BIN3 = 11, /* block-LDS accumulator rows  */
  URCCO_STAGE_CCO_BIN4 = 12, /* half-CU-LDS rows            */
  URCCO_STAGE_CCO_BIN5 = 13, /* CU-LDS accumulator rows     */
  URCCO_STAGE_CCO_BIN6 = 14, /* global accumulator rows     */
  URCCO_STAGE_COMPACT_INDICATORS = 15,
  URCCO_STAGE_EXCHANGE = 16 /* several ranks only: row lengths, need masks, masked lengths, packing per destination, row_ptr rebuild of what was received */
};
int urcco_session_set_timing(urcco_session* s, int32_t enable);
/* Profiling aid: kernel ablation switches; results are meaningless when non-zero.  0 in production.
 * SpGEMM rows: 1 = gather only, 2 = no LLR, 4 = no top-k, 8 = no select, 16 = no rank / output.
 * CSR row scan: 32 = cheap hash, 64 = no threshold gather, 128 = no entry -> row lookup.
 * Test hooks of the top-k select (tests/test_gpu_parity.py::test_select_overlay_race_*): 131072 = the first wave of every multi-wave
 * team sleeps before it reads the select histogram, 262144 = skip the barrier in front of the ambiguous-set copy-out (round 3's race). */
int urcco_session_set_debug(urcco_session* s, int32_t flags);
int urcco_session_get_timings(urcco_session* s, double* ms /*[URCCO_N_STAGES]*/, int64_t* launches /*[URCCO_N_STAGES]*/);
/* bytes of device scratch currently held */
int64_t urcco_session_scratch_bytes(const urcco_session* s);
/* Fault-hunting aid.  With URCCO_DEBUG_MARKS=1 in the environment every launch group of every session leaves "begun" /
 * "finished" marks in pinned host memory (stream-ordered); this prints them to stderr (the library's own SIGABRT handler does
 * the same when the HSA runtime aborts the process on a GPU memory fault).  Without the variable it prints nothing. */
void urcco_debug_dump_marks(void);

/* numNonZeroElementsPerColumn: counts[n_cols] = occurrences of each column id in col_idx[0..nnz).
 * counts is overwritten. */
int urcco_dev_column_counts(urcco_session* s, int64_t nnz, const int32_t* col_idx, int32_t n_cols, int32_t* counts);

/* sampleDownAndBinarize on a row shard.  raw_counts are the column counts of the WHOLE raw matrix (all shards
 * summed).  row_base = global index of local row 0 (keys the stateless RNG so results do not depend on
 * sharding).  out_row_ptr[n_rows+1], out_col_idx[capacity >= nnz], post_counts[n_cols] (nullable) is
 * overwritten with the kept entries' column counts of THIS shard.  nnz_out (device int64, nullable) receives
 * the kept total (it is also out_row_ptr[n_rows]). */
int urcco_dev_downsample(urcco_session* s, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx,
                         int64_t nnz, int32_t n_cols, const int32_t* raw_counts, int32_t seed,
                         int32_t max_elements_per_row, int32_t row_rate_mode, int64_t row_base,
                         int64_t* out_row_ptr, int32_t* out_col_idx, int32_t* post_counts);

/* CSR -> CSC of a (down-sampled) matrix.  counts[n_cols] = its column counts.  out_col_ptr[n_cols+1],
 * out_row_idx[nnz]; order inside a column is unspecified (only integer sums are formed from it).  Only columns in
 * [col_lo, col_hi) are materialised (the others get empty CSC columns): a rank of a multi-GPU job transposes just the
 * item range it owns. */
int urcco_dev_transpose(urcco_session* s, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx,
                        int64_t nnz, int32_t n_cols, const int32_t* counts, int32_t col_lo, int32_t col_hi,
                        int64_t* out_col_ptr, int32_t* out_row_idx);

/* Upper-bound work per item row of A'B: work[i - item_lo] = sum over users u of item i of d_B(u)
 * (= the cooccurrence pairs row i forms).  Used for accumulator binning and for work-balanced item ranges. */
int urcco_dev_row_work(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a, const int64_t* a_col_ptr,
                       const int32_t* a_row_idx, int64_t nnz_a_bound, const int64_t* b_row_ptr, int64_t* work);

/* The same per-item work from a USER shard, before any rank holds the whole matrix: work[i] = sum over the shard's
 * users u that hold item i (rows of a_*) of d_B(u) (rows of b_row_ptr, same users).  Summed over the ranks it equals
 * urcco_dev_row_work's result.  work[n_items_a] is overwritten. */
int urcco_dev_row_work_csr(urcco_session* s, int64_t n_rows, const int64_t* a_row_ptr, const int32_t* a_col_idx,
                           int64_t nnz_a, const int64_t* b_row_ptr, int32_t n_items_a, int64_t* work);

/* Splits items [0, n_items) into n_parts contiguous ranges of ~equal summed work.  bounds_host[n_parts+1]
 * (host memory).  Synchronises. */
int urcco_dev_partition(urcco_session* s, int32_t n_items, const int64_t* work, int32_t n_parts,
                        int32_t* bounds_host);

/* Multi-GPU: the CSC of the item range [item_lo, item_hi) of the primary from the fragments of `world` user shards.  Rank p
 * transposed its own shard (urcco_dev_transpose over all columns: shard-local user ids) and sent the slice of that CSC
 * that belongs to the range: lens[p * (item_hi - item_lo) + j] = length of column item_lo + j in shard p (uint16 when
 * wire16, else int32), entries = the W slices one behind the other in rank order (n_entries in all).
 * sizes[URCCO_EXCH_SIZES * p] = rows of shard p (the record of the exchange, ABI 305: rows, nnz, rows longer than 65535, counts that do not fit a
 * packed B' word -- four int64 per shard, three before 305); counts[n_items] = the
 * all-reduced column counts.  out_col_ptr[n_items + 1] (columns outside the range are empty), out_row_idx = global
 * user ids, ascending inside a column when the fragments were.  Replaces the pass every rank used to make over the
 * whole gathered A' to pick its columns out (Spark: the shuffle inside `A.t %*% B`, URAlgorithm.scala:323-346). */
int urcco_dev_merge_fragments(urcco_session* s, int32_t world, int32_t item_lo, int32_t item_hi, int32_t n_items,
                              const void* lens, int32_t wire16, const int32_t* entries, int64_t n_entries,
                              const int64_t* sizes, const int32_t* counts, int64_t* out_col_ptr, int32_t* out_row_idx);

/* Rows [item_lo, item_hi) of A'B, LLR scored, cut to the top k (computeSimilarities fused onto the SpGEMM).
 *   a_col_ptr/a_row_idx   CSC of down-sampled A (nnz_a_bound >= its nnz: sizes scratch, no host sync needed)
 *   b_row_ptr/b_col_idx   CSR of down-sampled B
 *   counts_a/counts_b     post-sampling column counts     n_users               nrow of the DRMs (N)
 *   exclude_self          1 for A'A (crossCooccurrence = false)
 * Outputs (strided, row r = item_lo + r): out_count[r] entries at out_idx/out_llr[r*k ..], sorted
 * (llr desc, col asc).  stats_dev (nullable, device int64[URCCO_STATS_LEN]): [0] pairs, then URCCO_N_BINS entries each
 * of rows / pairs / users (sum of cA) / emitted entries per accumulator bin (the last group only while timing is
 * enabled), then [1 + 4 * URCCO_N_BINS] accumulator-table overflows (an internal invariant: must be 0). */
#define URCCO_STATS_LEN 32
#define URCCO_EXCH_SIZES 4 /* int64 words of a shard's record in urcco_dev_merge_fragments' sizes */
int urcco_dev_cco_rows(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a,
                       const int64_t* a_col_ptr, const int32_t* a_row_idx, int64_t nnz_a_bound,
                       const int64_t* b_row_ptr, const int32_t* b_col_idx, int32_t n_cols_b, const int32_t* counts_a,
                       const int32_t* counts_b, int64_t n_users, int32_t exclude_self, int32_t k,
                       int32_t has_min_llr, double min_llr, int32_t* out_count, int32_t* out_idx,
                       double* out_llr, int64_t* stats_dev);

/* B' with the columns' counts aboard (round 6): out_packed[e] = b_col_idx[e] | counts_b[b_col_idx[e]] << key_bits for the entries of B' (b_row_ptr[n_rows_b] of
 * them, read on the device; nnz_b_bound >= that sizes the launch; out_packed holds nnz_b_bound words), key_bits = the bits of the value n_cols_b.
 * out_bad[0] = number of entries whose count does not fit the word's 32 - key_bits spare bits or 16 bits.  urcco_dev_cco_rows_packed is
 * urcco_dev_cco_rows for a B' that comes with such a copy: while out_bad[0] == 0 the row kernels read a candidate's cB off the word that claims its
 * accumulator slot instead of gathering counts_b[col] once per candidate -- more than half of the SpGEMM's cache-line fills; otherwise,
 * or with b_packed == pack_bad == NULL, it IS urcco_dev_cco_rows.  The context level packs every B' itself.  Results are identical either way. */
int urcco_dev_pack_counts(urcco_session* s, int64_t n_rows_b, const int64_t* b_row_ptr, const int32_t* b_col_idx, int64_t nnz_b_bound, const int32_t* counts_b,
                          int32_t n_cols_b, int32_t* out_packed, int32_t* out_bad);
int urcco_dev_cco_rows_packed(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a,
                              const int64_t* a_col_ptr, const int32_t* a_row_idx, int64_t nnz_a_bound,
                              const int64_t* b_row_ptr, const int32_t* b_col_idx, int32_t n_cols_b, const int32_t* counts_a,
                              const int32_t* counts_b, int64_t n_users, int32_t exclude_self, int32_t k,
                              int32_t has_min_llr, double min_llr, int32_t* out_count, int32_t* out_idx,
                              double* out_llr, int64_t* stats_dev, const int32_t* b_packed, const int32_t* pack_bad);

/* Strided top-k rows -> CSR.  out_row_ptr[n_rows+1]; out_col_idx/out_llr capacity n_rows*k. */
int urcco_dev_compact_indicators(urcco_session* s, int32_t n_rows, int32_t k, const int32_t* count,
                                 const int32_t* idx, const double* llr, int64_t* out_row_ptr,
                                 int32_t* out_col_idx, double* out_llr);

/* PopModel.calcPopular / calcTrending / calcHot (reference src/main/scala/PopModel.scala:113-179, consumed by
 * URAlgorithm.getRanksRDD, URAlgorithm.scala:537-560): per-item counts of the events whose time (ms since the epoch) lies in
 * one of n_intervals (1..3) consecutive half-open intervals [bounds_host[b], bounds_host[b + 1]) -- PEventStore.find's
 * startTime (inclusive) / untilTime (exclusive).  item_ids[e] < 0 = no target item / other event name: skipped.
 * counts (device): int32[n_intervals * n_items], overwritten; counts[b * n_items + i].  The host derives the ranks:
 * popular = counts, trending = newer - older, hot = (newer - middle) - (middle - older), each over the items present in
 * every interval it joins (universal-recommender_amd/pop_model.py). */
int urcco_dev_pop_counts(urcco_session* s, int64_t n_events, const int32_t* item_ids, const int64_t* times_ms, int32_t n_items,
                         int32_t n_intervals, const int64_t* bounds_host, int32_t* counts);

/* Test hooks (device level): LLR of SimilarityAnalysis.logLikelihoodRatio evaluated by the device code for
 * n argument tuples; u01 of the down-sampling RNG.  All pointers device. */
int urcco_dev_llr(urcco_session* s, int64_t n, const int64_t* with_a, const int64_t* with_b, const int64_t* with_ab,
                  const int64_t* n_users, double* out);
int urcco_dev_u01(urcco_session* s, int64_t n, int32_t seed, const int32_t* row, const int32_t* col, double* out);
int urcco_dev_u01_rng(urcco_session* s, int64_t n, int32_t seed, const int32_t* row, const int32_t* col, int32_t rng /* URCCO_RNG_* */, double* out);

/* ---- DEVICE level: Preparator (reference src/main/scala/Preparator.scala:44-87, :102-158, :160-214) -------------
 * Dictionaries and binary CSR matrices from event streams of 64-bit keys resident in HBM (the host hashes its id
 * strings or passes integer ids; the value ~0 is reserved).  Dense ids follow FIRST APPEARANCE in the stream
 * (oracle decision D8), so the host recovers the id -> string dictionary from `first_pos` alone. */
typedef struct urcco_key_table urcco_key_table;

/* Host helper: keys[i] = XXH64(bytes[offsets[i] .. offsets[i + 1]), seed) for n strings laid end to end (UTF-8); the value ~0
 * is remapped to 0 (reserved by the device dictionary).  A few host threads for large n.  The same function with a
 * second seed gives the independent check keys of urcco_dev_dictionary_verify. */
int urcco_hash_strings(const uint8_t* bytes, const int64_t* offsets, int64_t n, uint64_t seed, uint64_t* keys);

/* BiDictionary over keys[0..n): id = rank of the key's first position among the first positions of the keys that
 * occur at least min_count times (`minEventsPerUser` counts RAW events, Preparator.scala:129-132); other keys get no id.
 * select (nullable, device int32[n]): positions with select[p] < 0 do not take part (events of dropped users,
 * Preparator.scala:173-179).  first_pos (device int64, capacity n): first_pos[id] = stream position of the id's first
 * occurrence.  *n_ids is written on the host (the call synchronises the stream). */
int urcco_dev_dictionary_build(urcco_session* s, int64_t n, const uint64_t* keys, const int32_t* select, int32_t min_count,
                               int64_t* first_pos, urcco_key_table** table, int64_t* n_ids);
/* ids[p] = dense id of keys[p], or -1 (key without id, or select[p] < 0).  No host synchronisation. */
int urcco_dev_dictionary_lookup(urcco_session* s, const urcco_key_table* table, int64_t n, const uint64_t* keys,
                                const int32_t* select, int32_t* ids);
/* Collision check of a dictionary built from 64-bit hashes: check_keys[p] = a SECOND, independent hash of the same string
 * (another seed).  Counts the positions whose check key differs from the check key of their id's first occurrence, i.e. two
 * different strings that the first hash merged into one id (probability ~n^2 / 2^65 per dictionary); *n_mismatch is written
 * on the host (synchronises).  first_pos = the array urcco_dev_dictionary_build filled. */
int urcco_dev_dictionary_verify(urcco_session* s, const urcco_key_table* table, int64_t n, const uint64_t* keys, const int32_t* select,
                                const uint64_t* check_keys, const int64_t* first_pos, int64_t* n_mismatch);
/* The same check for ANOTHER stream looked up in the dictionary (Preparator.scala:173-179: the user keys of a secondary event
 * type against the user dictionary of the primary): dict_check_keys = the check keys of the stream the dictionary was built
 * from (first_pos indexes it).  A secondary-event user string whose 64-bit key collides with a primary user's would otherwise
 * be merged into that user silently. */
int urcco_dev_dictionary_verify_against(urcco_session* s, const urcco_key_table* table, int64_t n, const uint64_t* keys, const int32_t* select,
                                        const uint64_t* check_keys, const uint64_t* dict_check_keys, const int64_t* first_pos, int64_t* n_mismatch);
/* Frees the table (hipFree: waits for device work still using it). */
void urcco_key_table_destroy(urcco_key_table* table);
/* IndexedDatasetSpark's row assembly: (row id, column id) pairs (pairs with a negative id are skipped) -> binary CSR
 * with sorted, duplicate-free columns (`setQuick(col, 1.0)`, Preparator.scala:146, :205).  out_row_ptr: int64[n_rows + 1],
 * out_col_idx: capacity >= n.  nnz (nullable, host): written after a stream synchronisation when not NULL. */
int urcco_dev_csr_from_pairs(urcco_session* s, int64_t n, const int32_t* rows, const int32_t* cols, int64_t n_rows,
                             int64_t* out_row_ptr, int32_t* out_col_idx, int64_t* nnz);

#ifdef __cplusplus
}
#endif
#endif /* URCCO_H */
