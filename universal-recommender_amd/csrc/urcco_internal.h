// Internals shared by the C-ABI translation units of liburcco (urcco_api.hip: sessions + device-level stages;
// urcco_context.hip: persistent contexts, the host level, the multi-GPU build).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/urcco.h"
#include "cco_kernels.h"
static_assert(urcco::EXCH_SIZES == URCCO_EXCH_SIZES && urcco::STATS_LEN == URCCO_STATS_LEN, "include/urcco.h and cco_kernels.h agree");

struct urcco_session;
namespace urcco_detail {

char* err_buf();  // thread-local message buffer of urcco_last_error (512 bytes)

inline int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return status;
}

inline int hip_fail(hipError_t e, const char* what) {
  return fail(e == hipErrorOutOfMemory ? URCCO_OOM_DEVICE : URCCO_HIP_ERROR, "%s: %s", what, hipGetErrorString(e));
}

#define HIPC(expr)                                                  \
  do {                                                              \
    hipError_t _e = (expr);                                         \
    if (_e != hipSuccess) return urcco_detail::hip_fail(_e, #expr); \
  } while (0)

#define URC(expr)                  \
  do {                             \
    int _s = (expr);               \
    if (_s != URCCO_OK) return _s; \
  } while (0)

// No C++ exception crosses the C ABI: every extern "C" entry point that can allocate runs its body through this.
template <typename F>
inline int guarded(F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(URCCO_OOM_HOST, "out of host memory");
  } catch (const std::exception& e) {
    return fail(URCCO_INTERNAL, "unexpected exception: %s", e.what());
  } catch (...) {
    return fail(URCCO_INTERNAL, "unexpected exception");
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Fault-hunting aids, off unless the environment asks (read once; urcco_api.hip):
//   URCCO_DEBUG_MARKS=1   flight recorder: every launch group of every session writes "begun" / "finished" marks (build ordinal, stage)
//                         into pinned host memory, in stream order; a SIGABRT handler (the HSA runtime aborts the process on a GPU
//                         memory fault) prints every session's last marks, so the launch groups in flight at the fault are known
//   URCCO_DEBUG_POISON=1  every fresh device allocation of the library and the WHOLE scratch arena at every reserve() are filled with
//                         0x7f bytes (stream-ordered): a kernel that consumes memory nobody wrote meets an index ~2^31 elements away
//                         (or a 64-bit offset beyond the address space) instead of a stale but plausible value
struct DebugCfg { bool marks = false, poison = false; };
const DebugCfg& debug_cfg();
void debug_mark(urcco_session* s, int which /*0: begun, 1: finished*/, int stage);
void debug_register(urcco_session* s);
void debug_unregister(urcco_session* s);
void debug_poison(void* p, size_t bytes, hipStream_t st, bool async);

inline int ceil_log2_i64(int64_t v) {
  int l = 0;
  while (((int64_t)1 << l) < v) ++l;
  return l;
}

}  // namespace urcco_detail

namespace urcco_detail {
// urcco_dev_cco_rows with one secondary's share of a fused expand preparation handed in (both NULL: it prepares its own)
int cco_rows_impl(urcco_session* s, int32_t item_lo, int32_t item_hi, int32_t n_items_a, const int64_t* a_col_ptr, const int32_t* a_row_idx, int64_t nnz_a_bound,
                  const int64_t* b_row_ptr, const int32_t* b_col_idx, int32_t n_cols_b, const int32_t* counts_a, const int32_t* counts_b, int64_t n_users,
                  int32_t exclude_self, int32_t k, int32_t has_min_llr, double min_llr, int32_t* out_count, int32_t* out_idx, double* out_llr, int64_t* stats_dev,
                  const int64_t* pre_pstart, const int32_t* pre_plen, int64_t* pre_tile_sums = nullptr /* the scan-tile sums of pre_plen, left by expand_multi */,
                  const int32_t* b_packed = nullptr /* B' with the columns' counts aboard (launch_pack_counts) ... */, const int32_t* pack_bad = nullptr /* ... and its verdict */,
                  bool pk_known = false /* b_col_idx itself holds such words and the host knows they are good (sharded builds) */);
// B' with counts aboard for cco_rows_impl: out[e] = b_col_idx[e] | counts_b[b_col_idx[e]] << key bits, e < b_row_ptr[n_rows_b] (<= nnz_bound); bad[0] = counts that do not fit
int pack_counts(urcco_session* s, const int64_t* b_row_ptr, int64_t n_rows_b, const int32_t* b_col_idx, int64_t nnz_bound, const int32_t* counts_b, int32_t n_cols_b,
                int32_t* out, int32_t* bad);
int partition_dev(urcco_session* s, int32_t n_items, const int64_t* work, int32_t n_parts, int32_t* bounds_dev, int32_t** bounds_out);
int expand_multi(urcco_session* s, int n, const int64_t* a_col_ptr, int32_t n_items_a, const int32_t* a_row_idx, int64_t cap, const int64_t* const* b_row_ptr,
                 int64_t n_users, int64_t* const* pstart, int32_t* const* plen, int64_t* const* tile_sums = nullptr /* [d]: expand_tile_words(cap) words */);
inline size_t expand_tile_words(int64_t cap) { return (size_t)((cap + urcco::SCAN_TILE - 1) / urcco::SCAN_TILE + 2); }
}  // namespace urcco_detail

using namespace urcco_detail;

struct urcco_session {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 256;
  char* arena = nullptr;
  size_t arena_cap = 0;
  size_t arena_off = 0;
  // persistent zeroed dense counters + candidate scratch of the global-accumulator kernel
  int32_t* g_counts = nullptr;
  unsigned long long* g_cand_key = nullptr;
  int32_t* g_cand_col = nullptr;
  int64_t g_cols = 0;
  double* xlx_tab = nullptr;  // xLogX of small integers (N-independent), filled once
  double* xlx_hi = nullptr;   // xLogX(N - d) for the N of the last build
  long long xlx_hi_n = -1;
  int debug = 0;              // kernel ablation switches (profiling only)
  unsigned* marks = nullptr;  // URCCO_DEBUG_MARKS: pinned host words [0] last launch group begun, [1] last finished ((ordinal << 8) | stage)
  unsigned mark_seq = 0;
  int unordered_rows = 0;     // URCCO_FLAG_UNORDERED_ROWS of the owning context
  // optional per-stage HIP-event timing (bench.py's roofline numbers)
  bool timing = false;
  struct Rec { int stage; hipEvent_t e0, e1; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> free_events;
  double acc_ms[URCCO_N_STAGES] = {0};
  int64_t acc_n[URCCO_N_STAGES] = {0};

  // Timing bookkeeping never throws across the C ABI: an allocation failure just drops the sample.
  hipEvent_t get_event() noexcept {
    if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void begin(int stage) noexcept {
    cur_stage = stage;
    if (marks) debug_mark(this, 0, stage);
    if (!timing) return;
    try {
      recs.reserve(recs.size() + 1);
      free_events.reserve(free_events.size() + 2 * (recs.size() + 1));
    } catch (...) {
      open_rec = false;
      return;
    }
    Rec r{stage, get_event(), get_event()};
    if (!r.e0 || !r.e1) { open_rec = false; return; }
    (void)hipEventRecord(r.e0, stream);
    recs.push_back(r);
    open_rec = true;
  }
  bool open_rec = false;
  int cur_stage = 0;
  void end() noexcept {
    if (marks) debug_mark(this, 1, cur_stage);
    if (!timing || !open_rec || recs.empty()) return;
    (void)hipEventRecord(recs.back().e1, stream);
    open_rec = false;
  }
  void collect() {
    (void)hipStreamSynchronize(stream);
    for (const Rec& r : recs) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { acc_ms[r.stage] += ms; acc_n[r.stage] += 1; }
      free_events.push_back(r.e0);
      free_events.push_back(r.e1);
    }
    recs.clear();
  }

  int reserve(size_t bytes) {
    arena_off = 0;
#ifdef HIPSIM_HOST_BUILD  // test-only host simulator (tests/hostsim): sub-buffers end at guard pages under HIPSIM_GUARD=1
    if (hipsim::guard_on()) {
      hipsim::arena_unguard(arena, arena_cap);
      bytes += 64 * hipsim::GUARD_PAGE;  // callers that reserve raw byte counts (one block carved by the launcher) do not go through need()
    }
#endif
    if (bytes <= arena_cap) {
      if (debug_cfg().poison) debug_poison(arena, arena_cap, stream, true);  // what the previous stage left behind is not an input of this one
      return URCCO_OK;
    }
    if (arena) {
      HIPC(hipStreamSynchronize(stream));
      HIPC(hipFree(arena));
      arena = nullptr;
      arena_cap = 0;
    }
    const size_t want = align_up(bytes + bytes / 4, (size_t)1 << 20);
    HIPC(hipMalloc((void**)&arena, want));
    arena_cap = want;
    if (debug_cfg().poison) debug_poison(arena, arena_cap, stream, true);
    return URCCO_OK;
  }
  template <typename T>
  T* take(size_t n) {
#ifdef HIPSIM_HOST_BUILD
    if (hipsim::guard_on()) {
      T* q = reinterpret_cast<T*>(hipsim::arena_place(arena, arena_off, (n ? n : 1) * sizeof(T), &arena_off));
      if (arena_off > arena_cap) { fprintf(stderr, "hipsim guard: arena overflow (%zu > %zu)\n", arena_off, arena_cap); abort(); }
      return q;
    }
#endif
    const size_t bytes = align_up((n ? n : 1) * sizeof(T), 256);
    char* p = arena + arena_off;
    arena_off += bytes;
    return reinterpret_cast<T*>(p);
  }
  static size_t need(size_t n, size_t elem) {
#ifdef HIPSIM_HOST_BUILD
    if (hipsim::guard_on()) return align_up((n ? n : 1) * elem, hipsim::GUARD_PAGE) + 2 * hipsim::GUARD_PAGE;
#endif
    return align_up((n ? n : 1) * elem, 256);
  }

  // Dense per-block counters of the global-accumulator class: g_blocks x n_cols_b x 16 B.  The block count shrinks with
  // the width of B so that the scratch stays within 1 GiB per session (4 GiB from 1M columns on: 128 blocks at 2M, 26 at 10M, never fewer
  // than 2): the class serves the rows no LDS table can hold -- a handful under a Zipf catalogue, thousands under config
  // 5's hot head, where the number of resident blocks is what its throughput scales with.
  int g_blocks = 0;
  size_t g_cap = 0;  // elements allocated
  int ensure_global_bin(int64_t n_cols_b) {
    // wide column spaces (>= 1M columns: the 10M x 2M configurations) are where thousands of rows land in this class
    const int64_t budget = n_cols_b >= (1 << 20) ? ((int64_t)4096 << 20) : ((int64_t)1024 << 20);
    int64_t blocks = budget / ((n_cols_b > 0 ? n_cols_b : 1) * 16);
    if (blocks > urcco::GLOBAL_BIN_BLOCKS) blocks = urcco::GLOBAL_BIN_BLOCKS;
    if (blocks < 2) blocks = 2;
    const size_t n = (size_t)blocks * (size_t)n_cols_b;
    if (n <= g_cap && g_counts) {
      // the counters are zero between launches whatever the geometry (every claim walk restores them)
      g_blocks = (int)(g_cap / (size_t)(n_cols_b > 0 ? n_cols_b : 1) < (size_t)urcco::GLOBAL_BIN_BLOCKS ? g_cap / (size_t)(n_cols_b > 0 ? n_cols_b : 1)
                                                                                                         : (size_t)urcco::GLOBAL_BIN_BLOCKS);
      g_cols = n_cols_b;
      return URCCO_OK;
    }
    HIPC(hipStreamSynchronize(stream));
    if (g_counts) { HIPC(hipFree(g_counts)); HIPC(hipFree(g_cand_key)); HIPC(hipFree(g_cand_col)); }
    g_counts = nullptr; g_cols = 0; g_cap = 0;
    HIPC(hipMalloc((void**)&g_counts, n * sizeof(int32_t)));
    HIPC(hipMalloc((void**)&g_cand_key, n * sizeof(unsigned long long)));
    HIPC(hipMalloc((void**)&g_cand_col, n * sizeof(int32_t)));
    HIPC(hipMemsetAsync(g_counts, 0, n * sizeof(int32_t), stream));
    g_cols = n_cols_b;
    g_cap = n;
    g_blocks = (int)blocks;
    return URCCO_OK;
  }
};

