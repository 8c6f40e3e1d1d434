// C ABI of liburcco (include/urcco.h): session / scratch management, the device-level stage functions and the
// host-level one-shot entry points that stand in for Mahout's SimilarityAnalysis.cooccurrencesIDSs and
// crossOccurrenceDownsampled (reference call sites: src/main/scala/URAlgorithm.scala:323-329, :343-346).
// No CPU fallback: without a HIP device the compute entry points return URCCO_NO_DEVICE.
#include <memory>

#include "urcco_internal.h"

using namespace urcco_detail;

#include <signal.h>
#include <unistd.h>

#include <mutex>

namespace urcco_detail {
char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}

// ---- fault-hunting aids (urcco_internal.h: URCCO_DEBUG_MARKS / URCCO_DEBUG_POISON) ----------------------------------
const DebugCfg& debug_cfg() {
  static const DebugCfg cfg = [] {
    DebugCfg c;
    const char* m = getenv("URCCO_DEBUG_MARKS");
    const char* p = getenv("URCCO_DEBUG_POISON");
    c.marks = m && *m && *m != '0';
    c.poison = p && *p && *p != '0';
    return c;
  }();
  return cfg;
}

namespace {
__global__ void debug_mark_kernel(unsigned* slot, unsigned value) {
  __hip_atomic_store(slot, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int MARK_SESSIONS = 64;
struct MarkSlot { unsigned* marks; void* stream; int device; };
MarkSlot g_mark_slots[MARK_SESSIONS];  // read by the signal handler: plain array, entries published by their `marks` pointer
std::mutex g_mark_mu;
struct sigaction g_prev_abrt;
void put(const char* s) { (void)!write(2, s, strlen(s)); }
void put_u(unsigned long long v, int base = 10) {
  char b[24];
  int n = 0;
  do { const int d = (int)(v % (unsigned)base); b[n++] = (char)(d < 10 ? '0' + d : 'a' + d - 10); v /= (unsigned)base; } while (v && n < 23);
  char o[24];
  for (int i = 0; i < n; ++i) o[i] = b[n - 1 - i];
  o[n] = 0;
  put(o);
}
const char* const STAGE_NAME[URCCO_N_STAGES] = {"column_counts", "downsample_flags", "downsample_scan", "downsample_compact", "transpose", "row_work", "binning",
                                               "entropy", "cco_bin0", "cco_bin1", "cco_bin2", "cco_bin3", "cco_bin4", "cco_bin5", "cco_bin6", "compact_indicators", "exchange"};
void dump_marks(const char* why) {  // async-signal-safe: write(2) only
  put("[urcco marks] "); put(why); put(": last launch groups per session (ordinal:stage begun / finished)\n");
  for (int i = 0; i < MARK_SESSIONS; ++i) {
    unsigned* m = g_mark_slots[i].marks;
    if (!m) continue;
    const unsigned b = __atomic_load_n(&m[0], __ATOMIC_RELAXED), f = __atomic_load_n(&m[1], __ATOMIC_RELAXED);
    put("[urcco marks]   session "); put_u((unsigned)i); put(" dev "); put_u((unsigned)g_mark_slots[i].device);
    put(" stream 0x"); put_u((unsigned long long)(uintptr_t)g_mark_slots[i].stream, 16);
    put(": begun "); put_u(b >> 8); put(":"); put((b & 255u) < URCCO_N_STAGES ? STAGE_NAME[b & 255u] : "?");
    put("  finished "); put_u(f >> 8); put(":"); put((f & 255u) < URCCO_N_STAGES ? STAGE_NAME[f & 255u] : "?");
    put(b == f ? "  (idle)\n" : "  <-- IN FLIGHT\n");
  }
}
void on_abort(int sig) {
  dump_marks("SIGABRT");
  sigaction(SIGABRT, &g_prev_abrt, nullptr);
  raise(sig);
}
}  // namespace

void debug_register(urcco_session* s) {
  if (!debug_cfg().marks) return;
  std::lock_guard<std::mutex> g(g_mark_mu);
  static bool installed = false;
  if (!installed) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, &g_prev_abrt);
    installed = true;
  }
  unsigned* m = nullptr;
  if (hipHostMalloc((void**)&m, 64, 0) != hipSuccess || !m) return;
  m[0] = m[1] = 0u;
  for (int i = 0; i < MARK_SESSIONS; ++i)
    if (!g_mark_slots[i].marks) {
      g_mark_slots[i].stream = (void*)s->stream;
      g_mark_slots[i].device = s->device;
      __atomic_store_n(&g_mark_slots[i].marks, m, __ATOMIC_RELEASE);
      s->marks = m;
      return;
    }
  (void)hipHostFree(m);
}
void debug_unregister(urcco_session* s) {
  if (!s->marks) return;
  std::lock_guard<std::mutex> g(g_mark_mu);
  for (int i = 0; i < MARK_SESSIONS; ++i)
    if (g_mark_slots[i].marks == s->marks) __atomic_store_n(&g_mark_slots[i].marks, (unsigned*)nullptr, __ATOMIC_RELEASE);
  (void)hipHostFree(s->marks);
  s->marks = nullptr;
}
void debug_mark(urcco_session* s, int which, int stage) {
  if (which == 0) ++s->mark_seq;
  hipLaunchKernelGGL(debug_mark_kernel, dim3(1), dim3(1), 0, s->stream, s->marks + which, (s->mark_seq << 8) | (unsigned)(stage & 255));
}
void debug_poison(void* p, size_t bytes, hipStream_t st, bool async) {
#ifdef HIPSIM_HOST_BUILD
  if (hipsim::guard_on()) return;  // the simulator's own guard mode poisons (and keeps PROT_NONE pages inside the arena)
#endif
  if (!p || !bytes) return;
  if (async) (void)hipMemsetAsync(p, 0x7f, bytes, st);
  else (void)hipMemset(p, 0x7f, bytes);
}
}  // namespace urcco_detail

extern "C" {

int urcco_version(void) { return URCCO_VERSION; }

int urcco_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* urcco_last_error(void) { return err_buf(); }

const char* urcco_status_string(int status) {
  switch (status) {
    case URCCO_OK: return "OK";
    case URCCO_BAD_ARG: return "BAD_ARG";
    case URCCO_OOM_HOST: return "OOM_HOST";
    case URCCO_OOM_DEVICE: return "OOM_DEVICE";
    case URCCO_HIP_ERROR: return "HIP_ERROR";
    case URCCO_INTERNAL: return "INTERNAL";
    case URCCO_NO_DEVICE: return "NO_DEVICE";
    case URCCO_RCCL_ERROR: return "RCCL_ERROR";
    default: return "UNKNOWN";
  }
}

void urcco_debug_dump_marks(void) {
  if (debug_cfg().marks) dump_marks("dump");
}

int urcco_session_create(int32_t device, void* stream, urcco_session** out) {
  if (!out) return fail(URCCO_BAD_ARG, "urcco_session_create: out is NULL");
  *out = nullptr;
  const int n = urcco_device_count();
  if (n <= 0) return fail(URCCO_NO_DEVICE, "no HIP device visible (liburcco has no CPU fallback)");
  if (device < 0 || device >= n) return fail(URCCO_BAD_ARG, "device %d out of range [0,%d)", device, n);
  HIPC(hipSetDevice(device));
  urcco_session* s = new (std::nothrow) urcco_session();
  if (!s) return fail(URCCO_OOM_HOST, "session alloc");
  s->device = device;
  if (stream) {
    s->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreate(&s->stream);
    if (e != hipSuccess) { delete s; return hip_fail(e, "hipStreamCreate"); }
    s->own_stream = true;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cu = prop.multiProcessorCount;
  debug_register(s);
  *out = s;
  return URCCO_OK;
}

void urcco_session_destroy(urcco_session* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  (void)hipStreamSynchronize(s->stream);
  debug_unregister(s);
  if (s->arena) (void)hipFree(s->arena);
  if (s->xlx_tab) (void)hipFree(s->xlx_tab);
  if (s->xlx_hi) (void)hipFree(s->xlx_hi);
  if (s->g_counts) { (void)hipFree(s->g_counts); (void)hipFree(s->g_cand_key); (void)hipFree(s->g_cand_col); }
  s->collect();
  for (hipEvent_t e : s->free_events) (void)hipEventDestroy(e);
  if (s->own_stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

int urcco_session_synchronize(urcco_session* s) {
  if (!s) return fail(URCCO_BAD_ARG, "session is NULL");
  HIPC(hipStreamSynchronize(s->stream));
  return URCCO_OK;
}

int urcco_session_set_debug(urcco_session* s, int32_t flags) {
  if (!s) return fail(URCCO_BAD_ARG, "session is NULL");
  s->debug = flags;
  return URCCO_OK;
}

int urcco_session_set_timing(urcco_session* s, int32_t enable) {
  if (!s) return fail(URCCO_BAD_ARG, "session is NULL");
  s->collect();
  s->timing = enable != 0;
  for (int i = 0; i < URCCO_N_STAGES; ++i) { s->acc_ms[i] = 0; s->acc_n[i] = 0; }
  return URCCO_OK;
}

int urcco_session_get_timings(urcco_session* s, double* ms, int64_t* launches) {
  if (!s || !ms || !launches) return fail(URCCO_BAD_ARG, "urcco_session_get_timings: bad argument");
  s->collect();
  for (int i = 0; i < URCCO_N_STAGES; ++i) { ms[i] = s->acc_ms[i]; launches[i] = s->acc_n[i]; }
  return URCCO_OK;
}

int64_t urcco_session_scratch_bytes(const urcco_session* s) {
  if (!s) return 0;
  return (int64_t)s->arena_cap + (int64_t)s->g_cap * 16;
}

int urcco_dev_column_counts(urcco_session* s, int64_t nnz, const int32_t* col_idx, int32_t n_cols, int32_t* counts) {
  if (!s || nnz < 0 || n_cols < 0 || (nnz > 0 && !col_idx) || (n_cols > 0 && !counts)) return fail(URCCO_BAD_ARG, "urcco_dev_column_counts: bad argument");
  if (n_cols == 0) return URCCO_OK;
  const int64_t ph_bytes = urcco::column_counts_scratch_bytes(nnz, n_cols);
  if (ph_bytes > 0) URC(s->reserve((size_t)ph_bytes));
  s->begin(URCCO_STAGE_COLUMN_COUNTS);
  if (ph_bytes > 0)
    HIPC(urcco::launch_column_counts_partitioned(s->stream, col_idx, nnz, nullptr, n_cols, counts, s->take<char>((size_t)ph_bytes)));
  else
    HIPC(urcco::launch_column_counts(s->stream, s->n_cu, col_idx, nnz, n_cols, counts));
  s->end();
  return URCCO_OK;
}

int urcco_dev_downsample(urcco_session* s, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz, int32_t n_cols,
                         const int32_t* raw_counts, int32_t seed, int32_t max_elements_per_row, int32_t row_rate_mode, int64_t row_base,
                         int64_t* out_row_ptr, int32_t* out_col_idx, int32_t* post_counts) {
  if (!s || n_rows < 0 || nnz < 0 || n_cols < 0 || !row_ptr || !out_row_ptr || (nnz > 0 && (!col_idx || !raw_counts || !out_col_idx)))
    return fail(URCCO_BAD_ARG, "urcco_dev_downsample: bad argument");
  if (max_elements_per_row <= 0) return fail(URCCO_BAD_ARG, "maxElementsPerRow must be positive, got %d", max_elements_per_row);
  if ((row_rate_mode & ~URCCO_RNG_MIX32) != URCCO_ROW_RATE_MAHOUT_INT_DIV && (row_rate_mode & ~URCCO_RNG_MIX32) != URCCO_ROW_RATE_FRACTIONAL)
    return fail(URCCO_BAD_ARG, "unknown row_rate_mode %d", row_rate_mode);
  if (post_counts && n_cols > 0) HIPC(hipMemsetAsync(post_counts, 0, sizeof(int32_t) * (size_t)n_cols, s->stream));
  if (nnz == 0) {
    HIPC(hipMemsetAsync(out_row_ptr, 0, sizeof(int64_t) * (size_t)(n_rows + 1), s->stream));
    return URCCO_OK;
  }
  // post-sampling column counts: large matrices are counted after compaction by the atomic-free partitioned histogram
  // (its length is read on the device: out_row_ptr[n_rows]); small ones by L2 atomics inside the scan kernel
  const int64_t ph_bytes = post_counts ? urcco::column_counts_scratch_bytes(nnz, n_cols) : 0;
  const int64_t ds_tiles = (nnz + urcco::DS_TILE - 1) / urcco::DS_TILE;
  const size_t n_words = (size_t)ds_tiles * (urcco::DS_TILE / 64);
  const size_t thr_words = (size_t)n_cols + (size_t)n_cols / 8 + 2;  // 8-byte thresholds + their one-byte prefixes
  URC(s->reserve(urcco_session::need(thr_words, 8) + urcco_session::need((size_t)ds_tiles + 1, 8) * 2 + urcco_session::need(n_words, 8) +
                 (size_t)ph_bytes + 256));
  unsigned long long* thresholds = s->take<unsigned long long>(thr_words);
  int64_t* tile_rows = s->take<int64_t>((size_t)ds_tiles + 1);
  int64_t* tile_count = s->take<int64_t>((size_t)ds_tiles + 1);
  unsigned long long* flags = s->take<unsigned long long>(n_words);
  s->begin(URCCO_STAGE_DOWNSAMPLE_FLAGS);
  HIPC(urcco::launch_downsample_flags(s->stream, s->n_cu, n_rows, row_ptr, col_idx, nnz, n_cols, raw_counts, thresholds, (uint32_t)seed,
                                      max_elements_per_row, row_rate_mode, row_base, tile_rows, flags, tile_count,
                                      ph_bytes > 0 ? nullptr : post_counts, s->debug));
  s->end();
  s->begin(URCCO_STAGE_DOWNSAMPLE_SCAN);
  HIPC(urcco::launch_downsample_scan(s->stream, nnz, tile_count));
  s->end();
  s->begin(URCCO_STAGE_DOWNSAMPLE_COMPACT);
  HIPC(urcco::launch_downsample_compact(s->stream, n_rows, row_ptr, col_idx, nnz, tile_rows, flags, tile_count, out_row_ptr, out_col_idx));
  s->end();
  if (ph_bytes > 0) {
    s->begin(URCCO_STAGE_COLUMN_COUNTS);
    HIPC(urcco::launch_column_counts_partitioned(s->stream, out_col_idx, nnz, out_row_ptr + n_rows, n_cols, post_counts, s->take<char>((size_t)ph_bytes)));
    s->end();
  }
  return URCCO_OK;
}

int urcco_dev_transpose(urcco_session* s, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx, int64_t nnz, int32_t n_cols,
                        const int32_t* counts, int32_t col_lo, int32_t col_hi, int64_t* out_col_ptr, int32_t* out_row_idx) {
  if (!s || n_rows < 0 || nnz < 0 || n_cols < 0 || !row_ptr || !out_col_ptr || (n_cols > 0 && !counts) || (nnz > 0 && (!col_idx || !out_row_idx)) ||
      col_lo < 0 || col_hi < col_lo || col_hi > n_cols)
    return fail(URCCO_BAD_ARG, "urcco_dev_transpose: bad argument");
  const int64_t n_tiles = ((int64_t)n_cols + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  const int64_t tr_bytes = (s->debug & 256) ? 0 : urcco::transpose_scratch_bytes(n_rows, nnz, n_cols);  // 256: profiling, force the cursor-atomic kernel
  URC(s->reserve(urcco_session::need((size_t)n_cols, 4) + urcco_session::need((size_t)n_tiles + 2, 8) + (size_t)tr_bytes + 256));
  int32_t* cursor = s->take<int32_t>((size_t)n_cols);
  int64_t* tile_sums = s->take<int64_t>((size_t)n_tiles + 2);
  s->begin(URCCO_STAGE_TRANSPOSE);
  HIPC(urcco::launch_scan_i32_range(s->stream, counts, n_cols, col_lo, col_hi, out_col_ptr, tile_sums));
  if (nnz > 0 && n_rows > 0) {
    int g = ceil_log2_i64((nnz + n_rows - 1) / n_rows);
    if (g < 1) g = 1;
    if (g > 6) g = 6;
    HIPC(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)n_cols, s->stream));
    if (tr_bytes > 0) {
      HIPC(urcco::launch_transpose_partitioned(s->stream, n_rows, row_ptr, col_idx, nnz, n_cols, out_col_ptr, cursor, out_row_idx, col_lo, col_hi,
                                               s->take<char>((size_t)tr_bytes)));
    } else {
      HIPC(urcco::launch_transpose(s->stream, s->n_cu, n_rows, row_ptr, col_idx, g, out_col_ptr, cursor, out_row_idx, col_lo, col_hi));
    }
  }
  s->end();
  return URCCO_OK;
}

int urcco_dev_row_work_csr(urcco_session* s, int64_t n_rows, const int64_t* a_row_ptr, const int32_t* a_col_idx, int64_t nnz_a, const int64_t* b_row_ptr,
                           int32_t n_items_a, int64_t* work) {
  if (!s || n_rows < 0 || nnz_a < 0 || n_items_a < 0 || !a_row_ptr || !b_row_ptr || (nnz_a > 0 && !a_col_idx) || (n_items_a > 0 && !work))
    return fail(URCCO_BAD_ARG, "urcco_dev_row_work_csr: bad argument");
  int g = n_rows > 0 ? ceil_log2_i64((nnz_a + n_rows - 1) / n_rows) : 1;
  if (g < 1) g = 1;
  if (g > 6) g = 6;
  s->begin(URCCO_STAGE_ROW_WORK);
  HIPC(urcco::launch_row_work_csr(s->stream, s->n_cu, n_rows, a_row_ptr, a_col_idx, b_row_ptr, g, n_items_a, work));
  s->end();
  return URCCO_OK;
}

int urcco_dev_row_work(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a, const int64_t* a_col_ptr, const int32_t* a_row_idx,
                       int64_t nnz_a_bound, const int64_t* b_row_ptr, int64_t* work) {
  if (!s || item_lo < 0 || item_hi < item_lo || item_hi > n_items_a || nnz_a_bound < 0 || !a_col_ptr || !b_row_ptr || (item_hi > item_lo && !work))
    return fail(URCCO_BAD_ARG, "urcco_dev_row_work: bad argument");
  const int64_t cap = nnz_a_bound;
  const int64_t n_tiles = (cap + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  URC(s->reserve(urcco_session::need((size_t)cap, 8) + urcco_session::need((size_t)cap, 4) + urcco_session::need((size_t)cap + 1, 8) +
                 urcco_session::need((size_t)n_tiles + 2, 8)));
  int64_t* pstart = s->take<int64_t>((size_t)cap);
  int32_t* plen = s->take<int32_t>((size_t)cap);
  int64_t* wp = s->take<int64_t>((size_t)cap + 1);
  int64_t* tile_sums = s->take<int64_t>((size_t)n_tiles + 2);
  s->begin(URCCO_STAGE_ROW_WORK);
  HIPC(urcco::launch_expand_prepare(s->stream, s->n_cu, a_col_ptr, n_items_a, a_row_idx, b_row_ptr, nullptr, 0, cap, pstart, plen, wp, tile_sums));
  HIPC(urcco::launch_row_work(s->stream, s->n_cu, item_lo, item_hi, a_col_ptr, wp, work));
  s->end();
  return URCCO_OK;
}

int urcco_dev_partition(urcco_session* s, int32_t n_items, const int64_t* work, int32_t n_parts, int32_t* bounds_host) {
  if (!s || n_items < 0 || n_parts <= 0 || n_parts > 4096 || !bounds_host || (n_items > 0 && !work)) return fail(URCCO_BAD_ARG, "urcco_dev_partition: bad argument");
  int32_t* bounds = nullptr;
  URC(urcco_detail::partition_dev(s, n_items, work, n_parts, nullptr, &bounds));
  HIPC(hipMemcpyAsync(bounds_host, bounds, sizeof(int32_t) * (size_t)(n_parts + 1), hipMemcpyDeviceToHost, s->stream));
  HIPC(hipStreamSynchronize(s->stream));
  return URCCO_OK;
}

int urcco_dev_merge_fragments(urcco_session* s, int32_t world, int32_t item_lo, int32_t item_hi, int32_t n_items, const void* lens, int32_t wire16,
                              const int32_t* entries, int64_t n_entries, const int64_t* sizes, const int32_t* counts, int64_t* out_col_ptr,
                              int32_t* out_row_idx) {
  if (!s || world <= 0 || item_lo < 0 || item_hi < item_lo || item_hi > n_items || n_entries < 0 || !sizes || !out_col_ptr || (n_items > 0 && !counts) ||
      (item_hi > item_lo && !lens) || (n_entries > 0 && (!entries || !out_row_idx)))
    return fail(URCCO_BAD_ARG, "urcco_dev_merge_fragments: bad argument");
  const int32_t n_range = item_hi - item_lo;
  const int64_t n_runs = (int64_t)world * n_range;
  const int64_t t_items = ((int64_t)n_items + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE, t_runs = (n_runs + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  URC(s->reserve(urcco_session::need((size_t)n_runs + 1, 8) + urcco_session::need((size_t)(t_items > t_runs ? t_items : t_runs) + 2, 8)));
  int64_t* src_off = s->take<int64_t>((size_t)n_runs + 1);
  int64_t* tile_sums = s->take<int64_t>((size_t)(t_items > t_runs ? t_items : t_runs) + 2);
  s->begin(URCCO_STAGE_TRANSPOSE);
  HIPC(urcco::launch_scan_i32_range(s->stream, counts, n_items, item_lo, item_hi, out_col_ptr, tile_sums));
  if (wire16) {
    HIPC(urcco::launch_scan_u16(s->stream, static_cast<const unsigned short*>(lens), n_runs, src_off, tile_sums));
  } else {
    HIPC(urcco::launch_scan_i32(s->stream, static_cast<const int32_t*>(lens), n_runs, src_off, tile_sums));
  }
  HIPC(urcco::launch_frag_place(s->stream, s->n_cu, world, item_lo, n_range, lens, wire16, src_off, entries, out_col_ptr, sizes, out_row_idx));
  s->end();
  return URCCO_OK;
}

int urcco_dev_cco_rows(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a, const int64_t* a_col_ptr,
                       const int32_t* a_row_idx, int64_t nnz_a_bound, const int64_t* b_row_ptr, const int32_t* b_col_idx, int32_t n_cols_b,
                       const int32_t* counts_a, const int32_t* counts_b, int64_t n_users, int32_t exclude_self, int32_t k,
                       int32_t has_min_llr, double min_llr, int32_t* out_count, int32_t* out_idx, double* out_llr, int64_t* stats_dev) {
  return urcco_detail::cco_rows_impl(s, item_lo, item_hi, n_items_a, a_col_ptr, a_row_idx, nnz_a_bound, b_row_ptr, b_col_idx, n_cols_b, counts_a, counts_b, n_users,
                                     exclude_self, k, has_min_llr, min_llr, out_count, out_idx, out_llr, stats_dev, nullptr, nullptr);
}

int urcco_dev_pack_counts(urcco_session* s, int64_t n_rows_b, const int64_t* b_row_ptr, const int32_t* b_col_idx, int64_t nnz_b_bound, const int32_t* counts_b,
                          int32_t n_cols_b, int32_t* out_packed, int32_t* out_bad) {
  return urcco_detail::pack_counts(s, b_row_ptr, n_rows_b, b_col_idx, nnz_b_bound, counts_b, n_cols_b, out_packed, out_bad);
}

int urcco_dev_cco_rows_packed(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a, const int64_t* a_col_ptr,
                              const int32_t* a_row_idx, int64_t nnz_a_bound, const int64_t* b_row_ptr, const int32_t* b_col_idx, int32_t n_cols_b,
                              const int32_t* counts_a, const int32_t* counts_b, int64_t n_users, int32_t exclude_self, int32_t k,
                              int32_t has_min_llr, double min_llr, int32_t* out_count, int32_t* out_idx, double* out_llr, int64_t* stats_dev,
                              const int32_t* b_packed, const int32_t* pack_bad) {
  if ((b_packed == nullptr) != (pack_bad == nullptr)) return fail(URCCO_BAD_ARG, "urcco_dev_cco_rows_packed: b_packed and pack_bad come together");
  return urcco_detail::cco_rows_impl(s, item_lo, item_hi, n_items_a, a_col_ptr, a_row_idx, nnz_a_bound, b_row_ptr, b_col_idx, n_cols_b, counts_a, counts_b, n_users,
                                     exclude_self, k, has_min_llr, min_llr, out_count, out_idx, out_llr, stats_dev, nullptr, nullptr, nullptr, b_packed, pack_bad);
}

}  // extern "C"

namespace urcco_detail {

// bounds stay on the device: `bounds_dev` if given, else arena scratch returned through *bounds_out (valid until the session's next reserve)
int partition_dev(urcco_session* s, int32_t n_items, const int64_t* work, int32_t n_parts, int32_t* bounds_dev, int32_t** bounds_out) {
  const int64_t n_tiles = ((int64_t)n_items + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  URC(s->reserve(urcco_session::need((size_t)n_items + 1, 8) + urcco_session::need((size_t)n_tiles + 2, 8) + urcco_session::need((size_t)n_parts + 1, 4)));
  int64_t* prefix = s->take<int64_t>((size_t)n_items + 1);
  int64_t* tile_sums = s->take<int64_t>((size_t)n_tiles + 2);
  int32_t* bounds = bounds_dev ? bounds_dev : s->take<int32_t>((size_t)n_parts + 1);
  HIPC(urcco::launch_scan_i64(s->stream, work, n_items, prefix, tile_sums));
  HIPC(urcco::launch_partition(s->stream, n_items, prefix, n_parts, bounds));
  if (bounds_out) *bounds_out = bounds;
  return URCCO_OK;
}

// One secondary's share of a fused expand preparation (see urcco_expand_multi) may be handed in as pre_pstart / pre_plen.
int cco_rows_impl(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a, const int64_t* a_col_ptr, const int32_t* a_row_idx, int64_t nnz_a_bound,
                  const int64_t* b_row_ptr, const int32_t* b_col_idx, int32_t n_cols_b, const int32_t* counts_a, const int32_t* counts_b, int64_t n_users,
                  int32_t exclude_self, int32_t k, int32_t has_min_llr, double min_llr, int32_t* out_count, int32_t* out_idx, double* out_llr, int64_t* stats_dev,
                  const int64_t* pre_pstart, const int32_t* pre_plen, int64_t* pre_tile_sums, const int32_t* b_packed, const int32_t* pack_bad, bool pk_known) {
  if (!s || item_lo < 0 || item_hi < item_lo || item_hi > n_items_a || n_cols_b < 0 || n_users < 0 || nnz_a_bound < 0 || !a_col_ptr || !b_row_ptr)
    return fail(URCCO_BAD_ARG, "urcco_dev_cco_rows: bad argument");
  if (k <= 0) return fail(URCCO_BAD_ARG, "maxInterestingElements must be positive, got %d", k);
  const int32_t n = item_hi - item_lo;
  if (n == 0) {
    if (stats_dev) HIPC(hipMemsetAsync(stats_dev, 0, sizeof(int64_t) * URCCO_STATS_LEN, s->stream));
    return URCCO_OK;
  }
  if (!out_count || !out_idx || !out_llr || !counts_a || (n_cols_b > 0 && !counts_b)) return fail(URCCO_BAD_ARG, "urcco_dev_cco_rows: NULL buffer");
  HIPC(hipMemsetAsync(out_count, 0, sizeof(int32_t) * (size_t)n, s->stream));
  if (n_cols_b == 0 || n_users == 0) {
    if (stats_dev) HIPC(hipMemsetAsync(stats_dev, 0, sizeof(int64_t) * URCCO_STATS_LEN, s->stream));
    return URCCO_OK;
  }
  // packed LDS entry: key = col + 1 in the high bits, count in the low bits
  int key_bits = 1;
  while (((int64_t)1 << key_bits) <= (int64_t)n_cols_b) ++key_bits;  // values 1..n_cols_b
  const int count_bits = 32 - key_bits;
  if (count_bits < 1) return fail(URCCO_BAD_ARG, "n_cols_b %d too large for the packed accumulator", n_cols_b);
  // bin 6 = the multi-pass LDS class; only a k beyond its running lists (or beyond any LDS table) falls back to the dense
  // global-accumulator kernel and pays for its n_cols x 16 B of scratch per resident block
  const bool dense_bin6 = k > urcco::MP_KMAX_HOST || 3ll * k + 5 > 32768ll;
  if (dense_bin6) URC(s->ensure_global_bin(n_cols_b));
  if (!s->xlx_tab) {
    HIPC(hipMalloc((void**)&s->xlx_tab, sizeof(double) * urcco::XLX_TABLE_HOST));
    HIPC(urcco::launch_xlx_table(s->stream, s->xlx_tab));
  }
  if (!s->xlx_hi) HIPC(hipMalloc((void**)&s->xlx_hi, sizeof(double) * 2 * urcco::XLX_TABLE_HOST));  // xLogX(N - d), then columnEntropy(c)
  if (s->xlx_hi_n != n_users) {
    HIPC(urcco::launch_xlx_hi_table(s->stream, s->xlx_hi, s->xlx_tab, n_users));
    s->xlx_hi_n = n_users;
  }
  const int64_t n_tiles = ((int64_t)n + urcco::BIN_TILE - 1) / urcco::BIN_TILE;
  const int64_t cap = nnz_a_bound;
  const int64_t p_tiles = (cap + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  URC(s->reserve(urcco_session::need((size_t)cap, 8) + urcco_session::need((size_t)cap, 4) + urcco_session::need((size_t)cap + 1, 8) +
                 urcco_session::need((size_t)p_tiles + 2, 8) +
                 urcco_session::need((size_t)n, 8) + urcco_session::need((size_t)(n_tiles + 1) * urcco::BIN_COLS_HOST, 8) +
                 urcco_session::need(urcco::BIN_OFF_LEN, 4) + urcco_session::need((size_t)n, 4) + urcco_session::need((size_t)n_items_a, 8) +
                 urcco_session::need((size_t)n_cols_b, 2) + urcco_session::need(1, 4) + urcco_session::need(urcco::CAND_SLOTS, 8) + urcco_session::need(1, 8) + urcco_session::need(URCCO_STATS_LEN, 8) +
                 urcco_session::need((size_t)n_users + 1, 4)));
  int64_t* own_pstart = s->take<int64_t>((size_t)cap);
  int32_t* own_plen = s->take<int32_t>((size_t)cap);
  const int64_t* pstart = pre_pstart ? pre_pstart : own_pstart;
  int64_t* wp = s->take<int64_t>((size_t)cap + 1);
  int64_t* p_tile_sums = s->take<int64_t>((size_t)p_tiles + 2);
  int64_t* work = s->take<int64_t>((size_t)n);
  int64_t* tile_counts = s->take<int64_t>((size_t)(n_tiles + 1) * urcco::BIN_COLS_HOST);
  int32_t* bin_off = s->take<int32_t>(urcco::BIN_OFF_LEN);
  int32_t* bin_rows = s->take<int32_t>((size_t)n);
  double* ent_a = s->take<double>((size_t)n_items_a);
  unsigned short* cnt_b16 = s->take<unsigned short>((size_t)n_cols_b);
  int32_t* cnt16_bad = s->take<int32_t>(1);
  unsigned long long* cand = s->take<unsigned long long>(urcco::CAND_SLOTS);
  double* xlx_n = s->take<double>(1);
  int64_t* stats = stats_dev ? stats_dev : s->take<int64_t>(URCCO_STATS_LEN);
  unsigned* b_rp32 = s->take<unsigned>((size_t)n_users + 1);

  s->begin(URCCO_STAGE_ROW_WORK);
  if (pre_pstart && pre_plen)
    HIPC(urcco::launch_expand_scan(s->stream, a_col_ptr, n_items_a, pre_plen, cap, wp, pre_tile_sums ? pre_tile_sums : p_tile_sums, pre_tile_sums != nullptr));
  else
    HIPC(urcco::launch_expand_prepare(s->stream, s->n_cu, a_col_ptr, n_items_a, a_row_idx, b_row_ptr, b_rp32, n_users, cap, own_pstart, own_plen, wp, p_tile_sums));
  HIPC(urcco::launch_row_work(s->stream, s->n_cu, item_lo, item_hi, a_col_ptr, wp, work));
  s->end();
  s->begin(URCCO_STAGE_BINNING);
  HIPC(hipMemsetAsync(stats, 0, sizeof(int64_t) * URCCO_STATS_LEN, s->stream));
  HIPC(urcco::launch_binning(s->stream, item_lo, n, work, counts_a, n_cols_b, count_bits, k, tile_counts, bin_off, bin_rows, stats));
  s->end();
  s->begin(URCCO_STAGE_ENTROPY);
  HIPC(urcco::launch_item_entropy(s->stream, counts_a, n_items_a, n_users, ent_a, xlx_n));
  HIPC(urcco::launch_narrow_counts(s->stream, s->n_cu, counts_b, n_cols_b, cnt_b16, cnt16_bad));
  s->end();

  urcco::CcoArgs a;
  a.bin_rows = bin_rows; a.bin_off = bin_off;
  a.a_col_ptr = a_col_ptr; a.pstart = pstart; a.wp = wp; a.b_col_idx = b_col_idx;
  // debug 1048576: the count gather of rounds 1-5 (A/B, tests).  pk_known: b_col_idx itself holds packed words (the rows a sharded build received
  // travelled with their counts aboard -- the host learnt with the shard sizes that every count fits): no plain copy exists, every reader masks
  a.b_packed = pk_known ? b_col_idx : ((b_packed && pack_bad && !(s->debug & 1048576)) ? b_packed : nullptr);
  a.pack_bad = pack_bad;
  a.pk_known = pk_known ? 1 : 0;
  a.b_col_mask = pk_known ? (key_bits >= 32 ? 0xffffffffu : (1u << key_bits) - 1u) : 0xffffffffu;
  a.cnt_a = counts_a; a.cnt_b = counts_b; a.ent_a = ent_a; a.cnt_b16 = cnt_b16; a.cnt16_bad = cnt16_bad; a.xlx_n = xlx_n; a.xlx_tab = s->xlx_tab; a.xlx_hi = s->xlx_hi; a.col_ent = s->xlx_hi + urcco::XLX_TABLE_HOST; a.debug = s->debug;
  a.n_users = n_users; a.n_cols_b = n_cols_b; a.item_lo = item_lo; a.exclude_self = exclude_self ? 1 : 0; a.k = k;
  a.has_min_llr = has_min_llr ? 1 : 0; a.min_llr = min_llr; a.count_bits = count_bits;
  a.col_bytes = n_cols_b <= (1 << 8) ? 1 : (n_cols_b <= (1 << 16) ? 2 : (n_cols_b <= (1 << 24) ? 3 : 4));
  a.unordered = (s->unordered_rows || (s->debug & 32768)) ? 1 : 0;  // debug 32768: per-class measurement of the flag
  a.g_log2 = 4;  // 16 lanes stream one user's B' row: 64 B segments, matches the ~10-40 item rows the cut leaves
  a.out_count = out_count; a.out_idx = out_idx; a.out_llr = out_llr;
  a.err = reinterpret_cast<unsigned long long*>(stats + 1 + 4 * urcco::NBINS);
  a.cand = s->timing ? cand : nullptr;
  if (s->timing) HIPC(hipMemsetAsync(cand, 0, sizeof(unsigned long long) * urcco::CAND_SLOTS, s->stream));
  a.g_counts = s->g_counts; a.g_cand_key = s->g_cand_key; a.g_cand_col = s->g_cand_col; a.g_blocks = dense_bin6 ? s->g_blocks : 0;
  // Heaviest classes first (global, whole-CU, half-CU, ...): they have few, long rows and end raggedly; the fine-grained
  // one-wave and micro classes run last and finish sharply -- and, with a stream per event type, fill the heavy classes'
  // tails of the other event types instead of leaving a tail of their own.  debug 65536 restores the ascending order.
  for (int step = 0; step < urcco::NBINS; ++step) {
    const int bin = (s->debug & 65536) ? step : urcco::NBINS - 1 - step;
    s->begin(URCCO_STAGE_CCO_BIN0 + bin);
    HIPC(urcco::launch_cco_rows_bin(s->stream, s->n_cu, a, bin, n));
    s->end();
  }
  if (s->timing) HIPC(urcco::launch_bin_out_stats(s->stream, bin_rows, bin_off, item_lo, out_count, cand, stats));
  return URCCO_OK;
}

int pack_counts(urcco_session* s, const int64_t* b_row_ptr, int64_t n_rows_b, const int32_t* b_col_idx, int64_t nnz_bound, const int32_t* counts_b, int32_t n_cols_b,
                int32_t* out, int32_t* bad) {
  if (!s || !b_row_ptr || n_rows_b < 0 || nnz_bound < 0 || !out || !bad || (nnz_bound > 0 && (!b_col_idx || !counts_b))) return fail(URCCO_BAD_ARG, "pack_counts: bad argument");
  int key_bits = 1;  // as cco_rows_impl: the bits of a (column + 1) key; the column itself fits them too
  while (((int64_t)1 << key_bits) <= (int64_t)n_cols_b) ++key_bits;
  const int count_bits = 32 - key_bits;
  if (count_bits < 1) return fail(URCCO_BAD_ARG, "n_cols_b %d too large for the packed accumulator", n_cols_b);
  // the gathers read the 16-bit copy of the counts (half the table: 4 MB for a 2M-item catalogue) -- arena scratch, needed until the pack kernel has run
  URC(s->reserve(urcco_session::need((size_t)n_cols_b + 8, 2) + urcco_session::need(1, 4) + 256));
  unsigned short* c16 = s->take<unsigned short>((size_t)n_cols_b + 8);
  int32_t* bad16 = s->take<int32_t>(1);
  s->begin(URCCO_STAGE_ENTROPY);
  HIPC(urcco::launch_narrow_counts(s->stream, s->n_cu, counts_b, n_cols_b, c16, bad16));
  HIPC(urcco::launch_pack_counts(s->stream, s->n_cu, b_col_idx, b_row_ptr + n_rows_b, nnz_bound, c16, bad16, count_bits, out, bad));
  s->end();
  return URCCO_OK;
}

// Expand preparation of n secondaries in one pass over the CSC of A' (cco_expand.hip, expand_prepare_multi): pstart[d] / plen[d] hold
// cap entries each.  The interleaved (start, length) table lives in the session's arena for the duration of the launch.
int expand_multi(urcco_session* s, int n, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* a_row_idx, int64_t cap, const int64_t* const* b_row_ptr,
                 int64_t n_users, int64_t* const* pstart, int32_t* const* plen, int64_t* const* tile_sums) {
  if (!s || n < 1 || n > urcco::EXPAND_MULTI_MAX || !a_col_ptr || cap < 0) return fail(URCCO_BAD_ARG, "expand_multi: bad argument");
  URC(s->reserve(urcco_session::need(((size_t)n_users + 2) * (size_t)n, 4) + 256));
  void* T = s->take<unsigned>(((size_t)n_users + 2) * (size_t)n);  // n_users + 1 records of n starts (+ one record of slack: the last user's 2 n-word read)
  s->begin(URCCO_STAGE_ROW_WORK);
  HIPC(urcco::launch_expand_prepare_multi(s->stream, s->n_cu, a_col_ptr, n_items_a, a_row_idx, n, b_row_ptr, n_users, cap, pstart, plen, T, tile_sums));
  s->end();
  return URCCO_OK;
}

}  // namespace urcco_detail

extern "C" {

int urcco_dev_compact_indicators(urcco_session* s, int32_t n_rows, int32_t k, const int32_t* count, const int32_t* idx, const double* llr,
                                 int64_t* out_row_ptr, int32_t* out_col_idx, double* out_llr) {
  if (!s || n_rows < 0 || k <= 0 || !out_row_ptr || (n_rows > 0 && (!count || !idx || !llr || !out_col_idx || !out_llr)))
    return fail(URCCO_BAD_ARG, "urcco_dev_compact_indicators: bad argument");
  const int64_t n_tiles = ((int64_t)n_rows + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  URC(s->reserve(urcco_session::need((size_t)n_tiles + 2, 8)));
  int64_t* tile_sums = s->take<int64_t>((size_t)n_tiles + 2);
  s->begin(URCCO_STAGE_COMPACT_INDICATORS);
  HIPC(urcco::launch_scan_i32(s->stream, count, n_rows, out_row_ptr, tile_sums));
  HIPC(urcco::launch_compact_indicators(s->stream, n_rows, k, count, idx, llr, out_row_ptr, out_col_idx, out_llr));
  s->end();
  return URCCO_OK;
}

struct urcco_key_table {
  urcco::KeyTable t{};
  int64_t capacity = 0;
  int device = 0;
};

void urcco_key_table_destroy(urcco_key_table* table) {
  if (!table) return;
  (void)hipSetDevice(table->device);
  if (table->t.keys) (void)hipFree(table->t.keys);
  if (table->t.minpos) (void)hipFree(table->t.minpos);
  if (table->t.count) (void)hipFree(table->t.count);
  if (table->t.id) (void)hipFree(table->t.id);
  delete table;
}

int urcco_dev_dictionary_build(urcco_session* s, int64_t n, const uint64_t* keys, const int32_t* select, int32_t min_count,
                               int64_t* first_pos, urcco_key_table** table, int64_t* n_ids) {
  if (!s || n < 0 || (n > 0 && (!keys || !first_pos)) || !table || !n_ids) return fail(URCCO_BAD_ARG, "urcco_dev_dictionary_build: bad argument");
  if (n >= ((int64_t)1 << 32) - 1) return fail(URCCO_BAD_ARG, "urcco_dev_dictionary_build: at most 2^32 - 2 events per stream, got %lld", (long long)n);
  *table = nullptr;
  std::unique_ptr<urcco_key_table, void (*)(urcco_key_table*)> tab(new (std::nothrow) urcco_key_table(), urcco_key_table_destroy);
  if (!tab) return fail(URCCO_OOM_HOST, "key table alloc");
  tab->device = s->device;
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  tab->capacity = cap;
  tab->t.mask = (unsigned long long)cap - 1ull;
  HIPC(hipMalloc((void**)&tab->t.keys, sizeof(unsigned long long) * (size_t)cap));
  HIPC(hipMalloc((void**)&tab->t.minpos, sizeof(unsigned) * (size_t)cap));
  HIPC(hipMalloc((void**)&tab->t.count, sizeof(unsigned) * (size_t)cap));
  HIPC(hipMalloc((void**)&tab->t.id, sizeof(int32_t) * (size_t)cap));
  HIPC(hipMemsetAsync(tab->t.keys, 0xFF, sizeof(unsigned long long) * (size_t)cap, s->stream));
  HIPC(hipMemsetAsync(tab->t.minpos, 0xFF, sizeof(unsigned) * (size_t)cap, s->stream));
  HIPC(hipMemsetAsync(tab->t.count, 0, sizeof(unsigned) * (size_t)cap, s->stream));
  HIPC(hipMemsetAsync(tab->t.id, 0xFF, sizeof(int32_t) * (size_t)cap, s->stream));
  const int64_t n_tiles = (n + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  URC(s->reserve(urcco_session::need((size_t)n, 4) + urcco_session::need((size_t)n + 1, 8) + urcco_session::need((size_t)n_tiles + 2, 8)));
  int32_t* flag = s->take<int32_t>((size_t)n);
  int64_t* prefix = s->take<int64_t>((size_t)n + 1);
  int64_t* tile_sums = s->take<int64_t>((size_t)n_tiles + 2);
  HIPC(urcco::launch_dictionary_build(s->stream, s->n_cu, tab->t, n, reinterpret_cast<const unsigned long long*>(keys), select, min_count, flag, prefix,
                                      tile_sums, first_pos));
  int64_t ids = 0;
  HIPC(hipMemcpyAsync(&ids, prefix + n, sizeof(int64_t), hipMemcpyDeviceToHost, s->stream));
  HIPC(hipStreamSynchronize(s->stream));
  *n_ids = ids;
  *table = tab.release();
  return URCCO_OK;
}

int urcco_dev_dictionary_lookup(urcco_session* s, const urcco_key_table* table, int64_t n, const uint64_t* keys, const int32_t* select,
                                int32_t* ids) {
  if (!s || !table || n < 0 || (n > 0 && (!keys || !ids))) return fail(URCCO_BAD_ARG, "urcco_dev_dictionary_lookup: bad argument");
  HIPC(urcco::launch_dictionary_lookup(s->stream, s->n_cu, table->t, n, reinterpret_cast<const unsigned long long*>(keys), select, ids));
  return URCCO_OK;
}

int urcco_dev_dictionary_verify_against(urcco_session* s, const urcco_key_table* table, int64_t n, const uint64_t* keys, const int32_t* select,
                                        const uint64_t* check_keys, const uint64_t* dict_check_keys, const int64_t* first_pos, int64_t* n_mismatch) {
  if (!s || !table || n < 0 || !n_mismatch || (n > 0 && (!keys || !check_keys || !dict_check_keys || !first_pos)))
    return fail(URCCO_BAD_ARG, "urcco_dev_dictionary_verify: bad argument");
  URC(s->reserve(urcco_session::need(1, 8)));
  unsigned long long* err = s->take<unsigned long long>(1);
  HIPC(urcco::launch_dictionary_verify(s->stream, s->n_cu, table->t, n, reinterpret_cast<const unsigned long long*>(keys), select,
                                       reinterpret_cast<const unsigned long long*>(check_keys), reinterpret_cast<const unsigned long long*>(dict_check_keys),
                                       first_pos, err));
  unsigned long long bad = 0;
  HIPC(hipMemcpyAsync(&bad, err, sizeof(bad), hipMemcpyDeviceToHost, s->stream));
  HIPC(hipStreamSynchronize(s->stream));
  *n_mismatch = (int64_t)bad;
  return URCCO_OK;
}

int urcco_dev_dictionary_verify(urcco_session* s, const urcco_key_table* table, int64_t n, const uint64_t* keys, const int32_t* select,
                                const uint64_t* check_keys, const int64_t* first_pos, int64_t* n_mismatch) {
  return urcco_dev_dictionary_verify_against(s, table, n, keys, select, check_keys, check_keys, first_pos, n_mismatch);
}

int urcco_dev_csr_from_pairs(urcco_session* s, int64_t n, const int32_t* rows, const int32_t* cols, int64_t n_rows, int64_t* out_row_ptr,
                             int32_t* out_col_idx, int64_t* nnz) {
  if (!s || n < 0 || n_rows < 0 || !out_row_ptr || (n > 0 && (!rows || !cols || !out_col_idx)))
    return fail(URCCO_BAD_ARG, "urcco_dev_csr_from_pairs: bad argument");
  const int64_t n_tiles = (n_rows + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE;
  URC(s->reserve(urcco_session::need((size_t)n_rows, 4) + urcco_session::need((size_t)n_rows + 1, 8) + urcco_session::need((size_t)n, 4) +
                 urcco_session::need((size_t)n_tiles + 2, 8)));
  int32_t* cnt = s->take<int32_t>((size_t)n_rows);
  int64_t* raw_ptr = s->take<int64_t>((size_t)n_rows + 1);
  int32_t* tmp = s->take<int32_t>((size_t)n);
  int64_t* tile_sums = s->take<int64_t>((size_t)n_tiles + 2);
  HIPC(urcco::launch_csr_from_pairs(s->stream, s->n_cu, n, rows, cols, n_rows, cnt, raw_ptr, tmp, tile_sums, out_row_ptr, out_col_idx));
  if (nnz) {
    HIPC(hipMemcpyAsync(nnz, out_row_ptr + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost, s->stream));
    HIPC(hipStreamSynchronize(s->stream));
  }
  return URCCO_OK;
}

int urcco_dev_pop_counts(urcco_session* s, int64_t n_events, const int32_t* item_ids, const int64_t* times_ms, int32_t n_items, int32_t n_intervals,
                         const int64_t* bounds_host, int32_t* counts) {
  if (!s || n_events < 0 || n_items < 0 || n_intervals < 1 || n_intervals > 3 || !bounds_host || (n_events > 0 && (!item_ids || !times_ms)) ||
      (n_items > 0 && !counts))
    return fail(URCCO_BAD_ARG, "urcco_dev_pop_counts: bad argument");
  for (int k = 0; k < n_intervals; ++k)
    if (bounds_host[k + 1] < bounds_host[k]) return fail(URCCO_BAD_ARG, "urcco_dev_pop_counts: interval bounds must not decrease");
  if ((int64_t)n_items * n_intervals > 0x7ffffff0ll) return fail(URCCO_BAD_ARG, "urcco_dev_pop_counts: too many items");
  if (n_items == 0) return URCCO_OK;
  HIPC(urcco::launch_pop_counts(s->stream, s->n_cu, n_events, item_ids, times_ms, n_items, n_intervals, bounds_host, counts));
  return URCCO_OK;
}

int urcco_dev_llr(urcco_session* s, int64_t n, const int64_t* with_a, const int64_t* with_b, const int64_t* with_ab, const int64_t* n_users,
                  double* out) {
  if (!s || n < 0) return fail(URCCO_BAD_ARG, "urcco_dev_llr: bad argument");
  HIPC(urcco::launch_llr_test(s->stream, n, with_a, with_b, with_ab, n_users, out));
  return URCCO_OK;
}

int urcco_dev_u01(urcco_session* s, int64_t n, int32_t seed, const int32_t* row, const int32_t* col, double* out) {
  if (!s || n < 0) return fail(URCCO_BAD_ARG, "urcco_dev_u01: bad argument");
  HIPC(urcco::launch_u01_test(s->stream, n, (uint32_t)seed, row, col, out));
  return URCCO_OK;
}

int urcco_dev_u01_rng(urcco_session* s, int64_t n, int32_t seed, const int32_t* row, const int32_t* col, int32_t rng, double* out) {
  if (!s || n < 0 || (rng != URCCO_RNG_SPLITMIX53 && rng != URCCO_RNG_MIX32)) return fail(URCCO_BAD_ARG, "urcco_dev_u01_rng: bad argument");
  HIPC(urcco::launch_u01_test(s->stream, n, (uint32_t)seed, row, col, out, rng == URCCO_RNG_MIX32 ? 1 : 0));
  return URCCO_OK;
}


}  // extern "C"
