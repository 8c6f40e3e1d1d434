// Persistent contexts of liburcco (include/urcco.h, "CONTEXT level"): the host-level entry points that stand in for
// Mahout's SimilarityAnalysis.cooccurrencesIDSs / crossOccurrenceDownsampled (reference call sites
// src/main/scala/URAlgorithm.scala:323-329, :343-346), the device-resident build bench.py times, and the multi-GPU build
// (user-range input phase, work-balanced item ranges, RCCL exchange -- SURVEY.md 8e) driven from C++ so that a single
// JVM process reaches every GPU of the node.
//
// Per GPU the context keeps one urcco_session (HIP stream + scratch arena) per event type and every intermediate /
// output buffer; buffers only grow, so from the second build on a model build allocates nothing.  One host thread
// enqueues everything: with several GPUs in one process it walks them phase by phase and brackets every collective in a
// group (RCCL requires that of a thread that owns more than one rank).
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include "urcco_internal.h"

using namespace urcco_detail;

namespace {

// ---------------------------------------------------------------------------------------------------------
// RCCL, bound at run time (librccl.so.1 -- the copy already in the process when PyTorch loaded one, else ROCm's).  Only
// multi-rank contexts need it; a single-GPU build never touches it.
// ---------------------------------------------------------------------------------------------------------
struct Rccl {
  typedef struct { char internal[URCCO_UNIQUE_ID_BYTES]; } UniqueId;
  typedef void* Comm;
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommInitAll)(Comm*, int, const int*) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  // rccl.h: ncclInt8 = 0, ncclInt32 = 2, ncclInt64 = 4; ncclSum = 0
  static constexpr int kInt8 = 0, kInt32 = 2, kInt64 = 4, kSum = 0;

  static Rccl* get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
      }
      if (!r.handle) return;
      auto sym = [&](const char* n) { return dlsym(r.handle, n); };
      r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
      r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
      r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
      r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
      r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
      r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
      r.Send = (decltype(r.Send))sym("ncclSend");
      r.Recv = (decltype(r.Recv))sym("ncclRecv");
      r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.AllReduce || !r.Send || !r.Recv)
        r.handle = nullptr;
    });
    return r.handle ? &r : nullptr;
  }
};

int rccl_fail(Rccl* r, int code, const char* what) {
  return fail(URCCO_RCCL_ERROR, "%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(code) : "RCCL error");
}
#define RCCLC(r, expr)                            \
  do {                                            \
    int _c = (expr);                              \
    if (_c != 0) return rccl_fail((r), _c, #expr); \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// buffers
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct DBuf {  // device buffer that only grows (hipFree synchronises the device: growth happens on the first builds only)
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap && p) return URCCO_OK;
    if (p) HIPC(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = n + n / 16 + 64;
#ifdef HIPSIM_HOST_BUILD  // test-only host simulator: no slack, so that an overrun meets the guard page
    if (hipsim::guard_on()) want = n ? n : 1;
#endif
    HIPC(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    if (debug_cfg().poison) debug_poison(p, want * sizeof(T), nullptr, false);
    return URCCO_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Pinned host memory handed out as indicator arrays.  Process-wide: blocks may outlive the context that filled them
// (the caller releases them with urcco_free_indicators whenever it is done).
struct PinnedPool {
  struct Block { void* p; size_t cap; bool used; };
  std::mutex mu;
  std::vector<Block> blocks;
  void* get(size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    bytes = bytes ? bytes : 1;
    int best = -1;
    for (size_t i = 0; i < blocks.size(); ++i)
      if (!blocks[i].used && blocks[i].cap >= bytes && (best < 0 || blocks[i].cap < blocks[(size_t)best].cap)) best = (int)i;
    if (best >= 0) {
      blocks[(size_t)best].used = true;
      return blocks[(size_t)best].p;
    }
    void* p = nullptr;
    const size_t want = align_up(bytes + bytes / 8, 4096);
    if (hipHostMalloc(&p, want, 0) != hipSuccess || !p) return nullptr;
    blocks.push_back(Block{p, want, true});
    return p;
  }
  bool put(void* p) {  // false: not one of ours
    std::lock_guard<std::mutex> g(mu);
    for (Block& b : blocks)
      if (b.p == p) {
        b.used = false;
        return true;
      }
    return false;
  }
  void trim() {  // frees the blocks nobody holds
    std::lock_guard<std::mutex> g(mu);
    std::vector<Block> keep;
    for (Block& b : blocks) {
      if (b.used) keep.push_back(b);
      else (void)hipHostFree(b.p);
    }
    blocks.swap(keep);
  }
};
PinnedPool& pinned_pool() {
  static PinnedPool* pool = new PinnedPool();  // never destroyed: blocks may be returned during process teardown
  return *pool;
}

// ---------------------------------------------------------------------------------------------------------
// per event type, per GPU
// ---------------------------------------------------------------------------------------------------------
// buffer sets of the primary's shared products (see DevState); measured: 2, 3 and 4 sets give the same back-to-back build rate
constexpr int A_SETS = 2;

struct EvState {
  urcco_session* s = nullptr;   // own stream + arena (borrowed from DevState::sessions)
  DBuf<int64_t> in_rp;          // host level: the staged raw shard
  DBuf<int32_t> in_ci;
  DBuf<int32_t> raw, post;      // column counts before / after sampling (whole matrix once the all-reduce has run)
  DBuf<int64_t> s_rp;           // down-sampled shard
  DBuf<int32_t> s_ci;
  DBuf<int32_t> deg, f_deg;     // exchange: row lengths of the shard / of the whole matrix
  DBuf<unsigned short> deg16, f_deg16;  // ... as they travel when every row of every shard fits 16 bits
  DBuf<int64_t> f_rp;           // whole down-sampled matrix (multi-rank builds)
  DBuf<int32_t> f_ci;
  DBuf<int64_t> sizes;          // [EXCH_SIZES * world] (rows, nnz', rows longer than 65535) of every rank's shard; own record at [EXCH_SIZES * rank]
  DBuf<int64_t> scan_tmp;
  DBuf<int32_t> o_count, o_idx; // strided top-k
  DBuf<double> o_llr;
  DBuf<int64_t> c_rp;           // indicator CSR of this GPU's item range
  DBuf<int32_t> c_idx;
  DBuf<double> c_llr;
  DBuf<int64_t> stats;
  DBuf<unsigned long long> verr;
  DBuf<int32_t> b_pk;           // B' with the columns' counts aboard (CcoArgs::b_packed): rebuilt per build from b_ci + the post-sampling counts ...
  DBuf<int32_t> pk_bad;         // ... and [1] counts that did not fit
  DBuf<int32_t> s_pk;           // sharded builds: this rank's own down-sampled shard with the counts aboard -- what it SENDS when every count fits (the
                                // receivers' f_ci then holds packed words: b_pk_known)
  bool b_pk_known = false;      // b_ci holds packed words and the host knows it (see cco_rows_impl)
  uint32_t b_col_mask = 0xffffffffu;  // b_ci[e] & b_col_mask = the column
  unsigned long long* h_verr = nullptr;  // host-mapped pinned word the boundary check's result is STORED to by the GPU (host level, one GPU) ...
  unsigned long long* h_verr_dev = nullptr;  // ... and its device address
  int ensure_h_verr() {
    if (h_verr) return URCCO_OK;
    HIPC(hipHostMalloc((void**)&h_verr, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    HIPC(hipHostGetDevicePointer((void**)&h_verr_dev, h_verr, 0));
    return URCCO_OK;
  }
  DBuf<int64_t> pre_pstart;     // this event type's share of the fused expand preparation (builds with >= 2 secondaries)
  DBuf<int32_t> pre_plen;
  DBuf<int64_t> pre_tsum;       // ... and the scan-tile sums of pre_plen the fused pass leaves (the expand scan then skips its reduce pass)
  // row-filtered exchange: the shard's row lengths masked per destination [W][rows] (int32 / as they travel when 16 bits do), their
  // exclusive scan (where every sent row starts in `pack`), the rows packed per destination, and to_nnz[p * W + q] = column indices
  // rank p sends to rank q (own row computed here, the others gathered)
  DBuf<int32_t> mlen, pack;
  DBuf<unsigned short> mlen16;
  DBuf<int32_t> mlen_bad;
  DBuf<int64_t> moff, mtmp, to_nnz;
  hipEvent_t ev_sampled = nullptr, ev_done = nullptr, ev_rp = nullptr;
  hipEvent_t ev_cons[A_SETS] = {};  // ev_cons[q]: the A'B_d of the last build that used the primary's buffer set q has finished
  bool cons_valid[A_SETS] = {};
  // facts of the current build
  const int64_t* b_rp = nullptr;  // the B this GPU multiplies with
  const int32_t* b_ci = nullptr;
  int64_t b_rows = 0, b_nnz_bound = 0;
  void release() {
    in_rp.release(); in_ci.release(); raw.release(); post.release(); s_rp.release(); s_ci.release(); deg.release(); f_deg.release();
    deg16.release(); f_deg16.release();
    f_rp.release(); f_ci.release(); sizes.release(); scan_tmp.release(); o_count.release(); o_idx.release(); o_llr.release(); c_rp.release();
    c_idx.release(); c_llr.release(); stats.release(); verr.release(); b_pk.release(); pk_bad.release(); s_pk.release(); pre_pstart.release(); pre_plen.release(); pre_tsum.release();
    mlen.release(); pack.release(); mlen16.release(); mlen_bad.release(); moff.release(); mtmp.release(); to_nnz.release();
    if (h_verr) (void)hipHostFree(h_verr);
    h_verr = nullptr;
    h_verr_dev = nullptr;
    if (ev_sampled) (void)hipEventDestroy(ev_sampled);
    if (ev_done) (void)hipEventDestroy(ev_done);
    if (ev_rp) (void)hipEventDestroy(ev_rp);
    for (int q = 0; q < A_SETS; ++q) {
      if (ev_cons[q]) (void)hipEventDestroy(ev_cons[q]);
      ev_cons[q] = nullptr;
      cons_valid[q] = false;
    }
    ev_sampled = ev_done = ev_rp = nullptr;
  }
};

struct DevState {
  int device = 0;
  int rank = 0;  // global rank
  int n_cu = 256;
  std::vector<urcco_session*> sessions;
  std::vector<EvState> ev;
  // What every event type's stream reads of the primary -- its CSC and its post-sampling column counts -- exists twice:
  // consecutive builds alternate, so the next build's primary chain (stream 0) may overwrite one set while the A'B_d of the
  // previous build (streams 1..) still read the other.  A set is reused two builds later, behind the ev_cons events.
  DBuf<int64_t> a_cp[A_SETS];
  DBuf<int32_t> a_ri[A_SETS];
  DBuf<int32_t> a_post[A_SETS];
  int par = 0;
  DBuf<int64_t> work;
  DBuf<int32_t> bounds;
  // CSC fragments of the primary (several ranks): the shard's own column counts and CSC, their 16-bit lengths, the record
  // every rank publishes ((2 W + 3) int64: entry offsets at the range bounds, the bounds, lengths that do not fit 16 bits),
  // and what arrives for this GPU's item range
  DBuf<int32_t> l_cnt, l_ri, len_bad, f_ent;
  DBuf<int64_t> l_cp, rec;
  DBuf<unsigned short> len16;
  DBuf<char> f_len;
  DBuf<unsigned long long> need;  // row-filtered exchange: per local user the ranks whose item range its row of A' touches
  hipEvent_t need_ready = nullptr;
  hipEvent_t a_ready = nullptr, in_ready = nullptr, b_expanded = nullptr;
  Rccl::Comm comm = nullptr;
  int32_t item_lo = 0, item_hi = 0;
  int64_t a_ents = 0;  // entries of the primary's CSC over [item_lo, item_hi) (known exactly when it was merged from fragments)
};

struct Shard {  // one device-resident user-range shard handed to the pipelines
  int64_t n_rows = 0, row_base = 0, nnz = 0;
  const int64_t* rp = nullptr;
  const int32_t* ci = nullptr;
};
struct DsParams {
  int64_t n_cols = 0;
  int32_t max_rows = 500, k = 50, has_min_llr = 0;
  double min_llr = 0.0;
};

// One enqueueing thread per local GPU.  A build is a few hundred launches per GPU; issued by one host thread for eight GPUs
// the launch rate (~4 us each) would exceed the build itself, so the phases that only enqueue kernels run on all GPUs at
// once; the collective phases stay on the calling thread (one thread owning several ranks must group them anyway).
struct DevWorkers {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable start_cv, done_cv;
  const std::function<int(size_t)>* job = nullptr;
  uint64_t generation = 0;
  size_t pending = 0;
  bool stop = false;
  std::vector<int> status;
  std::vector<std::string> message;
  bool sequential = false;  // URCCO_FLAG_EMULATE_RANKS: the ranks share one device and one stream; their phases are enqueued one after the other

  void start(size_t n) {
    status.assign(n, URCCO_OK);
    message.assign(n, std::string());
    if (n <= 1 || sequential) return;
    for (size_t g = 0; g < n; ++g)
      threads.emplace_back([this, g] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<int(size_t)>* f = nullptr;
          {
            std::unique_lock<std::mutex> lk(mu);
            start_cv.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            f = job;
          }
          const int st = guarded([&] { return (*f)(g); });
          {
            std::lock_guard<std::mutex> lk(mu);
            status[g] = st;
            if (st != URCCO_OK) message[g] = err_buf();
            if (--pending == 0) done_cv.notify_all();
          }
        }
      });
  }
  // f(g) for every local GPU g; returns the first failure (its message copied into the caller's error buffer)
  int run(const std::function<int(size_t)>& f) {
    const size_t n = status.size();
    if (n <= 1) return n == 1 ? f(0) : URCCO_OK;
    if (sequential) {
      for (size_t g = 0; g < n; ++g) URC(f(g));
      return URCCO_OK;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      job = &f;
      pending = n;
      ++generation;
    }
    start_cv.notify_all();
    {
      std::unique_lock<std::mutex> lk(mu);
      done_cv.wait(lk, [&] { return pending == 0; });
    }
    for (size_t g = 0; g < n; ++g)
      if (status[g] != URCCO_OK) return fail(status[g], "%s", message[g].c_str());
    return URCCO_OK;
  }
  ~DevWorkers() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    start_cv.notify_all();
    for (std::thread& t : threads) t.join();
  }
};

// URCCO_TRACE_HOST=1: wall-clock marks of the host-level call on stderr (where the milliseconds of a PCIe-inclusive call go)
struct HostTrace {
  bool on = getenv("URCCO_TRACE_HOST") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what, int d = -1) const {
    if (!on) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[urcco host] %8.3f ms  %s%s\n", ms, what, d >= 0 ? (std::string(" ") + std::to_string(d)).c_str() : "");
  }
};

// Pinned staging ring of ONE GPU: pageable host memory -> pinned slot -> H2D on a stream of that GPU.  The slot events are
// created under hipSetDevice(device): HIP rejects an event recorded on a stream of another device.
constexpr size_t STAGE_CHUNK = (size_t)8 << 20;
struct StageRing {
  struct Slot { hipEvent_t ev = nullptr; std::atomic<long long> gen{0}; };
  int device = 0;
  char* base = nullptr;
  std::vector<std::unique_ptr<Slot>> slots;
  std::atomic<long long> next_chunk{0};  // global chunk counter: chunk q uses slot q % n_slots in generation q / n_slots
  int ensure(int dev, size_t n_slots) {
    device = dev;
    HIPC(hipSetDevice(dev));
    if (slots.size() >= n_slots && base) return URCCO_OK;
    release();
    HIPC(hipHostMalloc((void**)&base, n_slots * STAGE_CHUNK, hipHostMallocPortable));
    while (slots.size() < n_slots) {
      slots.emplace_back(new Slot());
      HIPC(hipEventCreateWithFlags(&slots.back()->ev, hipEventDisableTiming));
    }
    return URCCO_OK;
  }
  void release() {
    if (base || !slots.empty()) (void)hipSetDevice(device);
    for (auto& s : slots)
      if (s->ev) (void)hipEventDestroy(s->ev);
    slots.clear();
    next_chunk = 0;
    if (base) (void)hipHostFree(base);
    base = nullptr;
  }
  ~StageRing() { release(); }
};

struct InputGate;
// What urcco_context_stage leaves behind for urcco_context_finish: the caller's arrays are not touched again.
struct PendingBuild {
  int n_ds = 0;
  int64_t n_users = 0;
  int32_t seed = 0;
  std::vector<DsParams> ps;
  std::vector<std::vector<Shard>> sh;
  std::vector<int64_t> nnz_raw;
  std::unique_ptr<InputGate> gate;
  std::thread builder;        // one GPU: enqueues the build while the staging goes on
  int build_st = URCCO_OK;
  std::string build_msg;
  bool build_deferred = false;  // several GPUs / exchange path: the build is issued by urcco_context_finish
  // one GPU: every event type's results are brought to the host by the thread that enqueued its chain, the moment the chain ends --
  // under the uploads and the SpGEMMs of the other event types (pinned blocks of the process-wide pool; handed out by finish)
  std::vector<urcco_indicators> res;
  std::vector<char> res_done;
  void* stats_block = nullptr;  // pinned int64 [(URCCO_STATS_LEN + 1) * n_ds]: per-event statistics, then nnz' of B_d
  HostTrace trace;
  ~PendingBuild();
};

}  // namespace

struct urcco_context {
  std::unique_ptr<DevWorkers> workers{new DevWorkers()};
  std::vector<DevState> devs;
  int world = 1, first_rank = 0;
  int row_rate_mode = URCCO_ROW_RATE_MAHOUT_INT_DIV;
  int flags = 0, debug = 0;
  bool timing = false;
  bool have_cb = false;
  urcco_collectives cb{};
  Rccl* rccl = nullptr;
  std::vector<int32_t> h_bounds;                // host copy of the item range bounds of the last multi-rank build
  std::vector<int64_t> h_sizes;
  // staging (host level): one pinned ring per GPU (slot events belong to the device whose stream records them)
  std::vector<std::unique_ptr<StageRing>> rings;
  int copy_threads = 4;
  std::unique_ptr<PendingBuild> pending;  // between urcco_context_stage and urcco_context_finish

  hipStream_t emu_stream = nullptr;  // URCCO_FLAG_EMULATE_RANKS: the one stream every session of every rank runs on
  bool emulate() const { return (flags & URCCO_FLAG_EMULATE_RANKS) != 0; }
  bool exchange() const { return world > 1 || (flags & URCCO_FLAG_FORCE_EXCHANGE); }
  bool single_stream() const { return (flags & URCCO_FLAG_SINGLE_STREAM) != 0; }

  // ---- collectives ------------------------------------------------------------------------------------
  int group_start() {
    if (have_cb) return cb.group_start(cb.user) == 0 ? URCCO_OK : fail(URCCO_RCCL_ERROR, "collectives.group_start failed");
    RCCLC(rccl, rccl->GroupStart());
    return URCCO_OK;
  }
  int group_end() {
    if (have_cb) return cb.group_end(cb.user) == 0 ? URCCO_OK : fail(URCCO_RCCL_ERROR, "collectives.group_end failed");
    RCCLC(rccl, rccl->GroupEnd());
    return URCCO_OK;
  }
  int all_reduce(DevState& D, void* buf, int64_t count, int dtype, hipStream_t st) {
    if (count <= 0) return URCCO_OK;
    if (have_cb) return cb.all_reduce_sum(cb.user, D.rank, buf, count, dtype, (void*)st) == 0 ? URCCO_OK : fail(URCCO_RCCL_ERROR, "collectives.all_reduce_sum failed");
    RCCLC(rccl, rccl->AllReduce(buf, buf, (size_t)count, dtype == 0 ? Rccl::kInt32 : Rccl::kInt64, Rccl::kSum, D.comm, st));
    return URCCO_OK;
  }
  int all_gather_v(DevState& D, const void* send, void* recv, const int64_t* off, const int64_t* cnt, hipStream_t st) {
    if (have_cb) return cb.all_gather_v(cb.user, D.rank, send, recv, off, cnt, (void*)st) == 0 ? URCCO_OK : fail(URCCO_RCCL_ERROR, "collectives.all_gather_v failed");
    RCCLC(rccl, rccl->GroupStart());
    for (int p = 0; p < world; ++p) {
      if (cnt[D.rank] > 0) RCCLC(rccl, rccl->Send(send, (size_t)cnt[D.rank], Rccl::kInt8, p, D.comm, st));
      if (cnt[p] > 0) RCCLC(rccl, rccl->Recv((char*)recv + off[p], (size_t)cnt[p], Rccl::kInt8, p, D.comm, st));
    }
    RCCLC(rccl, rccl->GroupEnd());
    return URCCO_OK;
  }
  int all_to_all_v(DevState& D, const void* send, const int64_t* soff, const int64_t* scnt, void* recv, const int64_t* roff, const int64_t* rcnt, hipStream_t st) {
    if (have_cb) {
      if (!cb.all_to_all_v) return fail(URCCO_BAD_ARG, "collectives.all_to_all_v is required (urcco.h, since 301)");
      return cb.all_to_all_v(cb.user, D.rank, send, soff, scnt, recv, roff, rcnt, (void*)st) == 0 ? URCCO_OK : fail(URCCO_RCCL_ERROR, "collectives.all_to_all_v failed");
    }
    RCCLC(rccl, rccl->GroupStart());
    for (int p = 0; p < world; ++p) {
      if (scnt[p] > 0) RCCLC(rccl, rccl->Send((const char*)send + soff[p], (size_t)scnt[p], Rccl::kInt8, p, D.comm, st));
      if (rcnt[p] > 0) RCCLC(rccl, rccl->Recv((char*)recv + roff[p], (size_t)rcnt[p], Rccl::kInt8, p, D.comm, st));
    }
    RCCLC(rccl, rccl->GroupEnd());
    return URCCO_OK;
  }
};

namespace {

int set_dev(const DevState& D) {
  HIPC(hipSetDevice(D.device));
  return URCCO_OK;
}
// The context-level entry points walk the context's GPUs; the calling thread's current device is put back when they return (a caller --
// a test, a tool, a JVM thread -- that held a session or a stream of another device would otherwise launch on the wrong one afterwards).
struct CallerDevice {
  int dev = -1;
  CallerDevice() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~CallerDevice() { if (dev >= 0) (void)hipSetDevice(dev); }
};

urcco_session* sess_of(urcco_context* c, DevState& D, int d) { return D.sessions[c->single_stream() ? 0 : (size_t)d]; }

int ensure_events(urcco_context* c, DevState& D, int n_ds) {
  URC(set_dev(D));
  while ((int)D.sessions.size() < n_ds) {
    urcco_session* s = nullptr;
    URC(urcco_session_create(D.device, c->emulate() ? (void*)c->emu_stream : nullptr, &s));
    s->debug = c->debug;
    s->timing = c->timing;
    s->unordered_rows = (c->flags & URCCO_FLAG_UNORDERED_ROWS) ? 1 : 0;
    D.sessions.push_back(s);
  }
  if ((int)D.ev.size() < n_ds) D.ev.resize((size_t)n_ds);
  for (int d = 0; d < n_ds; ++d) {
    EvState& E = D.ev[(size_t)d];
    E.s = sess_of(c, D, d);
    E.s->unordered_rows = (c->flags & URCCO_FLAG_UNORDERED_ROWS) ? 1 : 0;
    if (!E.ev_sampled) HIPC(hipEventCreateWithFlags(&E.ev_sampled, hipEventDisableTiming));
    if (!E.ev_done) HIPC(hipEventCreateWithFlags(&E.ev_done, hipEventDisableTiming));
    for (int q = 0; q < A_SETS; ++q)
      if (!E.ev_cons[q]) HIPC(hipEventCreateWithFlags(&E.ev_cons[q], hipEventDisableTiming));
    if (!E.ev_rp) HIPC(hipEventCreateWithFlags(&E.ev_rp, hipEventDisableTiming));
  }
  if (!D.a_ready) HIPC(hipEventCreateWithFlags(&D.a_ready, hipEventDisableTiming));
  if (!D.in_ready) HIPC(hipEventCreateWithFlags(&D.in_ready, hipEventDisableTiming));
  if (!D.b_expanded) HIPC(hipEventCreateWithFlags(&D.b_expanded, hipEventDisableTiming));
  if (!D.need_ready) HIPC(hipEventCreateWithFlags(&D.need_ready, hipEventDisableTiming));
  return URCCO_OK;
}

// column counts + sampleDownAndBinarize of one shard on its event stream; with several ranks the two count vectors are
// all-reduced by the caller between the halves
int stage_raw_counts(DevState& D, EvState& E, const Shard& sh, const DsParams& p) {
  URC(E.raw.ensure((size_t)p.n_cols + 1));
  return urcco_dev_column_counts(E.s, sh.nnz, sh.ci, (int32_t)p.n_cols, E.raw.p);
}
// post-sampling column counts of event type d: the primary's live in the build's buffer set (read by every stream)
DBuf<int32_t>& post_of(DevState& D, int d) { return d == 0 ? D.a_post[D.par] : D.ev[(size_t)d].post; }

int stage_downsample(urcco_context* c, DevState& D, int d, const Shard& sh, const DsParams& p, int32_t seed) {
  EvState& E = D.ev[(size_t)d];
  URC(E.s_rp.ensure((size_t)sh.n_rows + 1));
  URC(E.s_ci.ensure((size_t)sh.nnz + 4));
  URC(post_of(D, d).ensure((size_t)p.n_cols + 1));
  return urcco_dev_downsample(E.s, sh.n_rows, sh.rp, sh.ci, sh.nnz, (int32_t)p.n_cols, E.raw.p, seed, p.max_rows, c->row_rate_mode, sh.row_base, E.s_rp.p,
                              E.s_ci.p, post_of(D, d).p);
}

// A'B_d for the GPU's item range + strided -> CSR, on event d's stream
int stage_rows(DevState& D, EvState& E, EvState& A, int d, const DsParams& pa, const DsParams& p, int64_t n_users, int64_t a_nnz_bound, bool pre_expanded = false) {
  const int32_t n = D.item_hi - D.item_lo;
  const size_t strided = (size_t)(n > 0 ? n : 1) * (size_t)p.k;
  URC(E.o_count.ensure((size_t)n + 1));
  URC(E.o_idx.ensure(strided));
  URC(E.o_llr.ensure(strided));
  URC(E.c_rp.ensure((size_t)n + 2));
  URC(E.c_idx.ensure(strided));
  URC(E.c_llr.ensure(strided));
  URC(E.stats.ensure(URCCO_STATS_LEN));
  // B' with the columns' counts aboard (round 6): one pass over the matrix this GPU multiplies with, with the FINAL post-sampling counts (all-reduced in a
  // sharded build) -- the row kernels then read a candidate's cB off the word that claims its slot instead of gathering it
  // (a sharded build packs on the SENDING side -- a rank's own shard is 1 / W of the matrix, what it receives about half of it -- and learns with the
  // shard sizes whether every count fits: the rows then arrive packed, b_pk_known)
  if (!E.b_pk_known) {
    URC(E.b_pk.ensure((size_t)E.b_nnz_bound + 4));
    URC(E.pk_bad.ensure(1));
    URC(pack_counts(E.s, E.b_rp, E.b_rows, E.b_ci, E.b_nnz_bound, post_of(D, d).p, (int32_t)p.n_cols, E.b_pk.p, E.pk_bad.p));
  }
  URC(cco_rows_impl(E.s, D.item_lo, D.item_hi, (int32_t)pa.n_cols, D.a_cp[D.par].p, D.a_ri[D.par].p, a_nnz_bound, E.b_rp, E.b_ci, (int32_t)p.n_cols,
                    post_of(D, 0).p, post_of(D, d).p, n_users, d == 0 ? 1 : 0, p.k, p.has_min_llr, p.min_llr, E.o_count.p, E.o_idx.p, E.o_llr.p, E.stats.p,
                    pre_expanded ? E.pre_pstart.p : nullptr, pre_expanded ? E.pre_plen.p : nullptr, pre_expanded ? E.pre_tsum.p : nullptr,
                    E.b_pk_known ? nullptr : E.b_pk.p, E.b_pk_known ? nullptr : E.pk_bad.p, E.b_pk_known));
  URC(urcco_dev_compact_indicators(E.s, n, p.k, E.o_count.p, E.o_idx.p, E.o_llr.p, E.c_rp.p, E.c_idx.p, E.c_llr.p));
  HIPC(hipEventRecord(E.ev_done, E.s->stream));
  HIPC(hipEventRecord(E.ev_cons[D.par], E.s->stream));
  E.cons_valid[D.par] = true;
  return URCCO_OK;
}

// Host level, one GPU: the build is enqueued WHILE the caller's matrices are still being staged.  The staging thread signals
// an event type once all of its copies and its boundary check are enqueued on that event type's stream; the thread that
// enqueues the event type's chain then waits for the stream (the copies have landed), reads the check's verdict and only
// then lets a kernel consume the matrix.  The primary's chain and A'A thus run under the upload of the larger secondaries.
struct InputGate {
  const HostTrace* trace = nullptr;
  std::function<int(int)> after_chain;  // called by the thread that has just enqueued event type d's chain (see PendingBuild::res)
  std::vector<int> order;               // the order the event types are staged in (primary first): a single enqueueing thread follows it
  std::vector<std::promise<int>> staged;
  std::vector<std::shared_future<int>> fut;
  std::vector<char> released;
  explicit InputGate(int n) : staged((size_t)n), fut((size_t)n), released((size_t)n, 0) {
    for (int d = 0; d < n; ++d) fut[(size_t)d] = staged[(size_t)d].get_future().share();
  }
  void release(int d, int status) {
    released[(size_t)d] = 1;
    staged[(size_t)d].set_value(status);
  }
  int wait(DevState& D, int d) {
    const int st = fut[(size_t)d].get();
    if (st != URCCO_OK) return fail(st, "dataset %d: staging failed", d);
    EvState& E = D.ev[(size_t)d];
    // the check's result was STORED to host-mapped memory by the last kernel of the staging chain (stage_event): waiting for the stream is all
    // it takes -- an 8-byte D2H copy here queued on the copy engine behind the other event types' results (round 6: 39 ms on config 4)
    HIPC(hipStreamSynchronize(E.s->stream));
    const unsigned long long bad = E.h_verr ? *(volatile unsigned long long*)E.h_verr : 0ull;
    if (bad)
      return fail(URCCO_BAD_ARG, "dataset %d: %llu invalid entries (row_ptr not monotone, or col_idx out of [0, n_cols) / not strictly increasing inside a row)", d, bad);
    if (trace) trace->mark("matrices landed and checked", d);
    return URCCO_OK;
  }
};

// ---------------------------------------------------------------------------------------------------------
// one rank, nothing to exchange: every event type on its own stream; B_d is sampled while A is sampled and transposed,
// every A'B_d runs behind an event on A's CSC; the heaviest event type is enqueued first
// ---------------------------------------------------------------------------------------------------------
int build_single(urcco_context* c, DevState& D, const std::vector<Shard>& sh, const std::vector<DsParams>& ps_, int64_t n_users, int32_t seed,
                 InputGate* gate) {
  const int n_ds = (int)sh.size();
  URC(set_dev(D));
  EvState& A = D.ev[0];
  D.item_lo = 0;
  D.item_hi = (int32_t)ps_[0].n_cols;
  URC(D.a_cp[D.par].ensure((size_t)ps_[0].n_cols + 2));
  URC(D.a_ri[D.par].ensure((size_t)sh[0].nnz + 4));
  std::atomic<bool> a_ok{false};  // the primary's chain up to its CSC was enqueued successfully
  // A build is ~150 launches of mostly short kernels: enqueued by ONE host thread the first ~0.5 ms of every build are
  // launch-bound (measured: the primary's stream idles 0.4 ms between its transposition and its SpGEMM while the host is
  // still enqueueing the other event types).  With a stream per event type each secondary gets its own enqueueing thread:
  // B_d's chain (counts, sampling) -> wait for A's CSC event -> A'B_d.  The event must have been recorded before a
  // stream is told to wait for it, hence the host-side hand-off.
  const bool threaded = n_ds > 1 && !c->single_stream();
  std::promise<void> a_recorded;
  std::shared_future<void> a_recorded_f = a_recorded.get_future().share();
  // With two or more secondaries their expand preparation is FUSED: one pass over the CSC of A' gathers an interleaved
  // (start, length) record per user and writes every secondary's pstart / plen (one scattered sector per CSC entry instead of one
  // line per entry AND event type).  It runs on the first secondary's stream once every secondary has been sampled and A'
  // transposed; the primary's own A'A does not wait for it.  (debug 4096: every event type prepares its own, as in round 2.)
  // Not under the host level's gate: there the secondaries land one after the other over tens of milliseconds, and a pass that needs
  // ALL of them sampled would hold every A'B_d back until the last upload has finished (measured on config 4, round 4: view's matrices
  // landed at 63 ms, its chain was enqueued at 102 ms -- the fused pass saves 2 ms of GPU time and cost 40 ms of wall time).
  bool fuse = n_ds >= 3 && n_ds - 1 <= urcco::EXPAND_MULTI_MAX && !(c->debug & 4096) && gate == nullptr;
  for (int d = 1; d < n_ds; ++d) fuse = fuse && sh[(size_t)d].nnz < ((int64_t)1 << 32);
  // Round 5: the PRIMARY's expand preparation rides on the same pass (A'A reads the down-sampled A as its B): its own pass was a second
  // scattered gather per CSC entry of A' -- 0.95 ms on config 4 against 1.1 ms for the four secondaries together -- and the price is that
  // A'A starts once every secondary has been sampled instead of right behind the transposition (URCCO_FOLD_PRIMARY=0: as before).
  static const bool fold_env = [] { const char* e = getenv("URCCO_FOLD_PRIMARY"); return !(e && e[0] == '0'); }();
  const bool fold = fuse && fold_env && n_ds <= urcco::EXPAND_MULTI_MAX && sh[0].nnz < ((int64_t)1 << 32);
  const int f0 = fold ? 0 : 1;  // first event type of the fused pass
  std::vector<std::promise<int>> sampled((size_t)n_ds);
  std::vector<std::shared_future<int>> sampled_f((size_t)n_ds);
  for (int d = 0; d < n_ds; ++d) sampled_f[(size_t)d] = sampled[(size_t)d].get_future().share();
  std::promise<int> expanded;
  std::shared_future<int> expanded_f = expanded.get_future().share();
  const int64_t a_cap = sh[0].nnz;
  auto fused_expand = [&]() -> int {  // on the first secondary's stream; every secondary's ev_sampled and a_ready have been recorded
    EvState& L = D.ev[1];
    if (L.s != A.s) HIPC(hipStreamWaitEvent(L.s->stream, D.a_ready, 0));
    std::vector<const int64_t*> rp((size_t)(n_ds - f0));
    std::vector<int64_t*> ps((size_t)(n_ds - f0));
    std::vector<int32_t*> pl((size_t)(n_ds - f0));
    std::vector<int64_t*> ts((size_t)(n_ds - f0));
    for (int d = f0; d < n_ds; ++d) {
      EvState& E = D.ev[(size_t)d];
      if (d > 0 && E.s != L.s) HIPC(hipStreamWaitEvent(L.s->stream, E.ev_sampled, 0));  // (the primary: a_ready, above)
      URC(E.pre_pstart.ensure((size_t)a_cap + 1));
      URC(E.pre_plen.ensure((size_t)a_cap + 1));
      URC(E.pre_tsum.ensure(expand_tile_words(a_cap)));
      ts[(size_t)(d - f0)] = E.pre_tsum.p;
      rp[(size_t)(d - f0)] = E.s_rp.p;
      ps[(size_t)(d - f0)] = E.pre_pstart.p;
      pl[(size_t)(d - f0)] = E.pre_plen.p;
    }
    URC(expand_multi(L.s, n_ds - f0, D.a_cp[D.par].p, (int32_t)ps_[0].n_cols, D.a_ri[D.par].p, a_cap, rp.data(), n_users, ps.data(), pl.data(), ts.data()));
    HIPC(hipEventRecord(D.b_expanded, L.s->stream));
    return URCCO_OK;
  };
  auto sample_secondary = [&](int d) -> int {
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    if (gate) URC(gate->wait(D, d));
    URC(stage_raw_counts(D, E, sh[(size_t)d], ps_[(size_t)d]));
    URC(stage_downsample(c, D, d, sh[(size_t)d], ps_[(size_t)d], seed));
    HIPC(hipEventRecord(E.ev_sampled, E.s->stream));
    return URCCO_OK;
  };
  auto rows_secondary = [&](int d) -> int {
    EvState& E = D.ev[(size_t)d];
    if (E.s != A.s) HIPC(hipStreamWaitEvent(E.s->stream, D.a_ready, 0));
    if (fuse && E.s != D.ev[1].s) HIPC(hipStreamWaitEvent(E.s->stream, D.b_expanded, 0));
    E.b_rp = E.s_rp.p;
    E.b_ci = E.s_ci.p;
    E.b_pk_known = false;
    E.b_col_mask = 0xffffffffu;
    E.b_rows = sh[(size_t)d].n_rows;
    E.b_nnz_bound = sh[(size_t)d].nnz;
    URC(stage_rows(D, E, A, d, ps_[0], ps_[(size_t)d], n_users, sh[0].nnz, fuse));
    if (gate && gate->trace) gate->trace->mark("chain enqueued", d);
    // (one enqueueing thread for everything -- URCCO_FLAG_SINGLE_STREAM: the downloads wait until every chain is enqueued, see below)
    if (threaded && gate && gate->after_chain) URC(gate->after_chain(d));
    return URCCO_OK;
  };
  // one enqueueing thread per secondary (see above); host-side hand-offs make sure an event has been RECORDED before a stream is
  // told to wait for it, and every promise is fulfilled on every path so that nobody waits forever
  struct Fulfil {  // a promise that is kept even if the code in between throws
    std::promise<int>* p;
    bool done = false;
    void set(int v) { if (p && !done) { done = true; p->set_value(v); } }
    ~Fulfil() { set(URCCO_INTERNAL); }
  };
  auto secondary_thread = [&](int d) -> int {
    Fulfil my_sample{fuse ? &sampled[(size_t)d] : nullptr};
    Fulfil my_expand{fuse && d == 1 ? &expanded : nullptr};
    int st = sample_secondary(d);
    my_sample.set(st);
    if (st == URCCO_OK) a_recorded_f.wait();
    if (fuse && d == 1) {
      for (int d2 = 2; d2 < n_ds && st == URCCO_OK; ++d2) st = sampled_f[(size_t)d2].get();
      if (st == URCCO_OK) st = a_ok ? fused_expand() : URCCO_INTERNAL;
      my_expand.set(st);
    } else if (fuse && st == URCCO_OK) {
      st = expanded_f.get();
    }
    if (st != URCCO_OK) return st;
    if (!a_ok) return URCCO_INTERNAL;
    return rows_secondary(d);
  };
  std::vector<std::thread> workers;
  std::vector<int> status((size_t)n_ds, URCCO_OK);
  std::vector<std::string> message((size_t)n_ds);
  if (threaded)
    for (int d = 1; d < n_ds; ++d)
      workers.emplace_back([&, d] {
        status[(size_t)d] = guarded([&] { return secondary_thread(d); });
        if (status[(size_t)d] != URCCO_OK) message[(size_t)d] = err_buf();  // the message lives in the worker's thread-local buffer
      });
  int st = [&]() -> int {
    if (gate) URC(gate->wait(D, 0));
    URC(stage_raw_counts(D, A, sh[0], ps_[0]));
    URC(stage_downsample(c, D, 0, sh[0], ps_[0], seed));
    URC(urcco_dev_transpose(A.s, sh[0].n_rows, A.s_rp.p, A.s_ci.p, sh[0].nnz, (int32_t)ps_[0].n_cols, post_of(D, 0).p, 0, (int32_t)ps_[0].n_cols,
                            D.a_cp[D.par].p, D.a_ri[D.par].p));
    HIPC(hipEventRecord(D.a_ready, A.s->stream));
    return URCCO_OK;
  }();
  a_ok = st == URCCO_OK;
  a_recorded.set_value();  // also on failure: the workers must not wait forever
  auto rows_primary = [&]() -> int {
    A.b_rp = A.s_rp.p;
    A.b_ci = A.s_ci.p;
    A.b_pk_known = false;
    A.b_col_mask = 0xffffffffu;
    A.b_rows = sh[0].n_rows;
    A.b_nnz_bound = sh[0].nnz;
    if (fold && A.s != D.ev[1].s) HIPC(hipStreamWaitEvent(A.s->stream, D.b_expanded, 0));
    URC(stage_rows(D, A, A, 0, ps_[0], ps_[0], n_users, sh[0].nnz, fold));
    if (gate && gate->trace) gate->trace->mark("chain enqueued", 0);
    return URCCO_OK;
  };
  if (st == URCCO_OK && !(fold && !threaded)) {  // (folded and one enqueueing thread: behind the fused pass, below)
    if (fold) st = expanded_f.get();  // the fused pass has been enqueued (its event recorded) by the first secondary's thread
    if (st == URCCO_OK) st = rows_primary();
    if (st == URCCO_OK && threaded && gate && gate->after_chain) st = gate->after_chain(0);
  }
  for (std::thread& t : workers) t.join();
  // A worker's failure first, with the WORKER's message (it lives in that thread's error buffer): with the folded expand pass the main thread
  // takes its status from expanded_f -- a secondary's sampling error arrives here as a bare code, and returning it at once left
  // urcco_last_error() empty or stale for the real failure.
  for (int d = 1; d < n_ds; ++d)
    if (status[(size_t)d] != URCCO_OK && !message[(size_t)d].empty()) return fail(status[(size_t)d], "%s", message[(size_t)d].c_str());
  if (st != URCCO_OK) return st;
  if (!threaded) {
    // ONE thread enqueues everything: a download (two blocking stream waits) between two chains would turn the build into chain, D2H, chain, D2H ...
    //  -- so every chain is enqueued first, the secondaries in the order their uploads were staged (largest first, gate->order), and the
    // results are brought over afterwards, in the order the chains end
    std::vector<int> order;
    for (int d = 1; d < n_ds; ++d) order.push_back(d);
    if (gate && !gate->order.empty()) {
      order.clear();
      for (int d : gate->order)
        if (d >= 1 && d < n_ds) order.push_back(d);
    }
    for (int d : order) URC(sample_secondary(d));
    if (fuse) URC(fused_expand());
    if (fold) URC(rows_primary());
    for (int d : order) URC(rows_secondary(d));
    if (gate && gate->after_chain) {
      URC(gate->after_chain(0));
      for (int d : order) URC(gate->after_chain(d));
    }
  }
  for (int d = 1; d < n_ds; ++d)
    if (status[(size_t)d] != URCCO_OK) return fail(status[(size_t)d], "%s", message[(size_t)d].c_str());
  return URCCO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// several ranks (or the forced exchange path): SURVEY.md 8e.  `L` = this process's GPUs; every phase walks them.
// ---------------------------------------------------------------------------------------------------------
constexpr int XS = urcco::EXCH_SIZES;
// the primary's CSC of a rank's item range comes from fragments (default) or, for A/B runs (debug bit 8192), from the pass every
// rank makes over the whole gathered A'
bool fragments(const urcco_context* c) { return !(c->debug & 8192); }
// Row-filtered exchange of the down-sampled matrices (cco_misc.hip, "Row-filtered exchange"): a rank receives the rows of B' only
// of the users that hold an item of ITS range.  Needs the ranges on the device before any whole-matrix work (the fragments route) and
// an all-to-all-v; debug bit 16384 restores the all-gather of every row (A/B).
bool filtered(const urcco_context* c) { return fragments(c) && c->world <= 64 && !(c->debug & 16384) && (!c->have_cb || c->cb.all_to_all_v != nullptr); }

// this shard's part of the filtered exchange of event type d, up to the sizes: per-destination masked row lengths, their scan, the
// totals per destination (own row of E.to_nnz); on event d's stream, behind the masks (D.need_ready)
int filtered_sizes(urcco_context* c, DevState& D, int d, const Shard& s) {
  EvState& E = D.ev[(size_t)d];
  const int W = c->world;
  const size_t n = (size_t)s.n_rows;
  URC(E.mlen.ensure((size_t)W * n + 8));
  URC(E.moff.ensure((size_t)W * n + 2));
  URC(E.mtmp.ensure((size_t)W * n / urcco::SCAN_TILE + 4));
  URC(E.to_nnz.ensure((size_t)W * (size_t)W));
  if (E.s != D.ev[0].s) HIPC(hipStreamWaitEvent(E.s->stream, D.need_ready, 0));
  E.s->begin(URCCO_STAGE_EXCHANGE);
  HIPC(urcco::launch_masked_lengths(E.s->stream, D.n_cu, s.n_rows, E.s_rp.p, D.need.p, W, E.mlen.p, E.moff.p, E.mtmp.p, E.to_nnz.p + (size_t)W * (size_t)D.rank));
  E.s->end();
  return URCCO_OK;
}
// ... and the tiny all-gather of those totals (W int64 per rank), on event d's stream
int gather_filtered_sizes(urcco_context* c, int d) {
  const int W = c->world;
  std::vector<int64_t> off((size_t)W), cnt((size_t)W, 8 * (int64_t)W);
  for (int r = 0; r < W; ++r) off[(size_t)r] = 8 * (int64_t)W * r;
  URC(c->group_start());
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    URC(c->all_gather_v(D, E.to_nnz.p + (size_t)W * (size_t)D.rank, E.to_nnz.p, off.data(), cnt.data(), E.s->stream));
  }
  URC(c->group_end());
  return URCCO_OK;
}

int input_phase(urcco_context* c, int d, const std::vector<std::vector<Shard>>& sh, const std::vector<DsParams>& ps, int32_t seed) {
  const DsParams& p = ps[(size_t)d];
  URC(c->workers->run([&](size_t g) -> int {
    DevState& D = c->devs[g];
    URC(set_dev(D));
    return stage_raw_counts(D, D.ev[(size_t)d], sh[(size_t)d][g], p);
  }));
  URC(c->group_start());
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    URC(c->all_reduce(D, D.ev[(size_t)d].raw.p, p.n_cols, 0, D.ev[(size_t)d].s->stream));
  }
  URC(c->group_end());
  URC(c->workers->run([&](size_t g) -> int {
    DevState& D = c->devs[g];
    URC(set_dev(D));
    URC(stage_downsample(c, D, d, sh[(size_t)d][g], p, seed));
    if (d == 0 && fragments(c)) {  // the shard's own column counts = the column lengths of its CSC, before the all-reduce adds the others'
      URC(D.l_cnt.ensure((size_t)p.n_cols + 1));
      if (p.n_cols > 0) HIPC(hipMemcpyAsync(D.l_cnt.p, post_of(D, 0).p, sizeof(int32_t) * (size_t)p.n_cols, hipMemcpyDeviceToDevice, D.ev[0].s->stream));
    }
    return URCCO_OK;
  }));
  URC(c->group_start());
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    URC(c->all_reduce(D, post_of(D, d).p, p.n_cols, 0, D.ev[(size_t)d].s->stream));
  }
  URC(c->group_end());
  // row lengths (what travels) + the (rows, nnz', long rows) record of the shard, gathered over the ranks
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    const Shard& s = sh[(size_t)d][(size_t)(&D - c->devs.data())];
    URC(E.deg.ensure((size_t)s.n_rows + 1));
    URC(E.deg16.ensure((size_t)s.n_rows + 8));
    URC(E.sizes.ensure((size_t)XS * (size_t)c->world));
    E.s->begin(URCCO_STAGE_EXCHANGE);
    HIPC(urcco::launch_row_lengths(E.s->stream, D.n_cu, s.n_rows, E.s_rp.p, E.deg.p, E.deg16.p, E.sizes.p + XS * D.rank));
    // (never: debug 1048576, and the primary of a build that gathers and transposes the WHOLE A' on every rank -- debug 8192 -- reads the received words as columns)
    const bool no_pack = (c->debug & 1048576) != 0 || (d == 0 && !fragments(c));
    // round 6: can a B' word of this event type carry its column's count?  A fact of the all-reduced count table -- every rank finds the same answer
    // and learns the others' with the shard sizes -- and, when it can, this rank's shard is packed HERE, before it travels (debug 1048576: never)
    {
      int key_bits = 1;
      while (((int64_t)1 << key_bits) <= p.n_cols) ++key_bits;
      const int count_bits = 32 - key_bits;
      if (count_bits < 1 || no_pack) {
        HIPC(hipMemsetAsync(E.sizes.p + XS * D.rank + 3, 1, sizeof(int64_t), E.s->stream));  // (any non-zero value says "plain")
      } else {
        HIPC(urcco::launch_counts_over_limit(E.s->stream, D.n_cu, post_of(D, d).p, p.n_cols, count_bits, E.sizes.p + XS * D.rank + 3));
      }
    }
    E.s->end();
    if (!no_pack) {
      URC(E.s_pk.ensure((size_t)s.nnz + 4));
      URC(E.pk_bad.ensure(1));
      URC(pack_counts(E.s, E.s_rp.p, s.n_rows, E.s_ci.p, s.nnz, post_of(D, d).p, (int32_t)p.n_cols, E.s_pk.p, E.pk_bad.p));
    }
  }
  std::vector<int64_t> off((size_t)c->world), cnt((size_t)c->world, 8 * XS);
  for (int r = 0; r < c->world; ++r) off[(size_t)r] = 8 * XS * (int64_t)r;
  URC(c->group_start());
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    URC(c->all_gather_v(D, E.sizes.p + XS * D.rank, E.sizes.p, off.data(), cnt.data(), E.s->stream));
  }
  URC(c->group_end());
  if (d > 0 && filtered(c)) {  // the ranges are on the device by now (the primary's exchange has been issued): what goes to whom
    URC(c->workers->run([&](size_t g) -> int {
      DevState& D = c->devs[g];
      URC(set_dev(D));
      return filtered_sizes(c, D, d, sh[(size_t)d][g]);
    }));
    URC(gather_filtered_sizes(c, d));
  }
  return URCCO_OK;
}

// What the host learns about the primary's fragments in the one blocking read of a multi-rank build
struct FragPlan {
  bool on = false, wire16 = true;
  std::vector<int32_t> bounds;         // [W + 1]
  std::vector<int64_t> cpb;            // [W][W + 1]: entry offsets of rank p's CSC at the bounds
};

// reads event d's shard sizes on the host (waits for that event's stream only; for the primary with fragments the same read
// brings every rank's fragment record), exchanges the down-sampled shards and rebuilds the whole matrix on every GPU
int exchange_phase(urcco_context* c, int d, const std::vector<std::vector<Shard>>& sh, const std::vector<DsParams>& ps, int64_t n_users,
                   std::vector<int64_t>& sizes /*out [XS * world]*/, FragPlan* fp = nullptr) {
  const int W = c->world;
  const int R = 2 * W + 3;
  const bool filt = filtered(c);
  sizes.assign((size_t)XS * (size_t)W, 0);
  std::vector<int64_t> recs, T;  // T[p * W + q]: column indices rank p sends to rank q (filtered exchange)
  {
    DevState& D = c->devs[0];
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    if (filt) {
      T.assign((size_t)W * (size_t)W, 0);
      HIPC(hipMemcpyAsync(T.data(), E.to_nnz.p, sizeof(int64_t) * (size_t)W * (size_t)W, hipMemcpyDeviceToHost, E.s->stream));
    }
    HIPC(hipMemcpyAsync(sizes.data(), E.sizes.p, sizeof(int64_t) * (size_t)XS * (size_t)W, hipMemcpyDeviceToHost, E.s->stream));
    if (fp && fp->on) {
      recs.assign((size_t)R * (size_t)W, 0);
      HIPC(hipMemcpyAsync(recs.data(), D.rec.p, sizeof(int64_t) * (size_t)R * (size_t)W, hipMemcpyDeviceToHost, E.s->stream));
    }
    HIPC(hipStreamSynchronize(E.s->stream));
  }
  std::vector<int64_t> off_r((size_t)W), cnt_r((size_t)W), off_c((size_t)W), cnt_c((size_t)W);
  int64_t rows = 0, nnz = 0;
  bool deg16 = true;
  for (int r = 0; r < W; ++r) deg16 = deg16 && sizes[(size_t)XS * r + 2] == 0;
  bool packed_ok = true;  // the rows travel with their counts aboard: every rank's verdict on the (same) count table (input_phase)
  for (int r = 0; r < W; ++r) packed_ok = packed_ok && sizes[(size_t)XS * r + 3] == 0;
  uint32_t col_mask = 0xffffffffu;
  if (packed_ok) {
    int key_bits = 1;
    while (((int64_t)1 << key_bits) <= ps[(size_t)d].n_cols) ++key_bits;
    col_mask = key_bits >= 32 ? 0xffffffffu : (1u << key_bits) - 1u;
  }
  const int64_t deg_bytes = deg16 ? 2 : 4;
  for (int r = 0; r < W; ++r) {
    off_r[(size_t)r] = rows * deg_bytes;
    cnt_r[(size_t)r] = sizes[(size_t)XS * r] * deg_bytes;
    off_c[(size_t)r] = nnz * 4;
    cnt_c[(size_t)r] = sizes[(size_t)XS * r + 1] * 4;
    rows += sizes[(size_t)XS * r];
    nnz += sizes[(size_t)XS * r + 1];
  }
  if (rows != n_users) return fail(URCCO_BAD_ARG, "event type %d: the user shards hold %lld rows, n_users_total is %lld", d, (long long)rows, (long long)n_users);
  if (fp && fp->on) {
    fp->bounds.assign((size_t)W + 1, 0);
    fp->cpb.assign((size_t)W * (size_t)(W + 1), 0);
    fp->wire16 = true;
    for (int p = 0; p < W; ++p) {
      const int64_t* rec = recs.data() + (size_t)R * (size_t)p;
      for (int q = 0; q <= W; ++q) {
        fp->cpb[(size_t)p * (size_t)(W + 1) + (size_t)q] = rec[q];
        if (p == 0) fp->bounds[(size_t)q] = (int32_t)rec[W + 1 + q];
        else if (fp->bounds[(size_t)q] != (int32_t)rec[W + 1 + q]) return fail(URCCO_INTERNAL, "ranks 0 and %d disagree on the item range bounds", p);
      }
      if (rec[0] != 0 || rec[W] != sizes[(size_t)XS * p + 1]) return fail(URCCO_INTERNAL, "rank %d: the fragment record does not cover its shard", p);
      fp->wire16 = fp->wire16 && rec[2 * W + 2] == 0;
    }
    c->h_bounds = fp->bounds;
  }
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    URC(E.f_deg.ensure((size_t)rows + 1));
    URC(E.f_deg16.ensure((size_t)rows + 8));
    URC(E.f_rp.ensure((size_t)rows + 2));
    int64_t nnz_in = nnz, nnz_out = 0;
    if (filt) {
      nnz_in = 0;
      for (int p = 0; p < W; ++p) {
        if (T[(size_t)p * W + D.rank] < 0 || T[(size_t)D.rank * W + p] < 0) return fail(URCCO_INTERNAL, "event type %d: negative exchange size", d);
        nnz_in += T[(size_t)p * W + D.rank];
        nnz_out += T[(size_t)D.rank * W + p];
      }
      URC(E.pack.ensure((size_t)nnz_out + 4));
      if (deg16) {
        URC(E.mlen16.ensure((size_t)W * (size_t)sh[(size_t)d][(size_t)(&D - c->devs.data())].n_rows + 8));
        URC(E.mlen_bad.ensure(1));
      }
    }
    URC(E.f_ci.ensure((size_t)nnz_in + 4));
    URC(E.scan_tmp.ensure((size_t)(rows / urcco::SCAN_TILE + 4)));
    if (fp && fp->on) {
      D.item_lo = fp->bounds[(size_t)D.rank];
      D.item_hi = fp->bounds[(size_t)D.rank + 1];
      int64_t ents = 0;
      for (int p = 0; p < W; ++p) ents += fp->cpb[(size_t)p * (size_t)(W + 1) + (size_t)D.rank + 1] - fp->cpb[(size_t)p * (size_t)(W + 1) + (size_t)D.rank];
      URC(D.f_len.ensure((size_t)W * (size_t)(D.item_hi - D.item_lo) * 4 + 64));
      URC(D.f_ent.ensure((size_t)ents + 4));
    }
  }
  // offsets / counts of the fragment all-to-all-v: one set per local device, alive until group_end (a collectives callback may
  // keep the pointers until its group ends -- include/urcco.h)
  struct A2A { std::vector<int64_t> so, sc, ro, rc, eso, esc, ero, erc, lso, lsc, cso, csc, cro, crc; };
  std::vector<A2A> a2a(c->devs.size());
  if (filt) {  // pack this shard's rows per destination (and narrow the masked lengths to their wire width)
    URC(c->workers->run([&](size_t g) -> int {
      DevState& D = c->devs[g];
      URC(set_dev(D));
      EvState& E = D.ev[(size_t)d];
      const Shard& s = sh[(size_t)d][g];
      E.s->begin(URCCO_STAGE_EXCHANGE);
      HIPC(urcco::launch_pack_rows(E.s->stream, D.n_cu, s.n_rows, E.s_rp.p, packed_ok ? E.s_pk.p : E.s_ci.p, D.need.p, W, E.moff.p, E.pack.p));
      if (deg16 && s.n_rows > 0) HIPC(urcco::launch_narrow_counts(E.s->stream, D.n_cu, E.mlen.p, (int64_t)W * s.n_rows, E.mlen16.p, E.mlen_bad.p));
      E.s->end();
      return URCCO_OK;
    }));
  }
  URC(c->group_start());
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    if (filt) {
      // to rank q: the lengths of ALL of this shard's rows as q may see them (0 where q's range does not touch the row) and the
      // rows it may see, packed; from rank p: the same for p's shard, one shard behind the other in rank order -- the layout the
      // all-gather produced, so the rebuild below does not change
      A2A& x = a2a[(size_t)(&D - c->devs.data())];
      const int64_t lb = deg16 ? 2 : 4, n_loc = sh[(size_t)d][(size_t)(&D - c->devs.data())].n_rows;
      for (std::vector<int64_t>* v : {&x.lso, &x.lsc, &x.cso, &x.csc, &x.cro, &x.crc}) v->assign((size_t)W, 0);
      int64_t s_at = 0, r_at = 0;
      for (int q = 0; q < W; ++q) {
        x.lso[(size_t)q] = (int64_t)q * n_loc * lb;
        x.lsc[(size_t)q] = n_loc * lb;
        x.cso[(size_t)q] = s_at * 4;
        x.csc[(size_t)q] = T[(size_t)D.rank * W + q] * 4;
        s_at += T[(size_t)D.rank * W + q];
        x.cro[(size_t)q] = r_at * 4;
        x.crc[(size_t)q] = T[(size_t)q * W + D.rank] * 4;
        r_at += T[(size_t)q * W + D.rank];
      }
      URC(c->all_to_all_v(D, deg16 ? (const void*)E.mlen16.p : (const void*)E.mlen.p, x.lso.data(), x.lsc.data(), deg16 ? (void*)E.f_deg16.p : (void*)E.f_deg.p, off_r.data(),
                          cnt_r.data(), E.s->stream));
      URC(c->all_to_all_v(D, E.pack.p, x.cso.data(), x.csc.data(), E.f_ci.p, x.cro.data(), x.crc.data(), E.s->stream));
    } else {
      if (deg16) URC(c->all_gather_v(D, E.deg16.p, E.f_deg16.p, off_r.data(), cnt_r.data(), E.s->stream));
      else URC(c->all_gather_v(D, E.deg.p, E.f_deg.p, off_r.data(), cnt_r.data(), E.s->stream));
      URC(c->all_gather_v(D, packed_ok ? E.s_pk.p : E.s_ci.p, E.f_ci.p, off_c.data(), cnt_c.data(), E.s->stream));
    }
    if (fp && fp->on) {
      // to rank q: the column lengths and the entries of q's item range, as they lie in this shard's CSC; from rank p: the same
      // for this GPU's range, one fragment behind the other in rank order
      const int64_t lb = fp->wire16 ? 2 : 4;
      const int64_t n_range = D.item_hi - D.item_lo;
      A2A& x = a2a[(size_t)(&D - c->devs.data())];
      std::vector<int64_t>&so = x.so, &sc = x.sc, &ro = x.ro, &rc = x.rc, &eso = x.eso, &esc = x.esc, &ero = x.ero, &erc = x.erc;
      for (std::vector<int64_t>* v : {&so, &sc, &ro, &rc, &eso, &esc, &ero, &erc}) v->assign((size_t)W, 0);
      int64_t e_at = 0;
      const int64_t* mine = fp->cpb.data() + (size_t)D.rank * (size_t)(W + 1);
      for (int q = 0; q < W; ++q) {
        so[(size_t)q] = (int64_t)fp->bounds[(size_t)q] * lb;
        sc[(size_t)q] = (int64_t)(fp->bounds[(size_t)q + 1] - fp->bounds[(size_t)q]) * lb;
        ro[(size_t)q] = (int64_t)q * n_range * lb;
        rc[(size_t)q] = n_range * lb;
        eso[(size_t)q] = mine[q] * 4;
        esc[(size_t)q] = (mine[q + 1] - mine[q]) * 4;
        const int64_t* theirs = fp->cpb.data() + (size_t)q * (size_t)(W + 1);
        ero[(size_t)q] = e_at * 4;
        erc[(size_t)q] = (theirs[D.rank + 1] - theirs[D.rank]) * 4;
        e_at += theirs[D.rank + 1] - theirs[D.rank];
      }
      const void* lens = fp->wire16 ? (const void*)D.len16.p : (const void*)D.l_cnt.p;
      URC(c->all_to_all_v(D, lens, so.data(), sc.data(), D.f_len.p, ro.data(), rc.data(), E.s->stream));
      URC(c->all_to_all_v(D, D.l_ri.p, eso.data(), esc.data(), D.f_ent.p, ero.data(), erc.data(), E.s->stream));
    }
  }
  URC(c->group_end());
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    EvState& E = D.ev[(size_t)d];
    E.s->begin(URCCO_STAGE_EXCHANGE);
    if (deg16) HIPC(urcco::launch_scan_u16(E.s->stream, E.f_deg16.p, rows, E.f_rp.p, E.scan_tmp.p));
    else HIPC(urcco::launch_scan_i32(E.s->stream, E.f_deg.p, rows, E.f_rp.p, E.scan_tmp.p));
    E.s->end();
    E.b_rp = E.f_rp.p;
    E.b_ci = E.f_ci.p;
    E.b_pk_known = packed_ok;
    E.b_col_mask = col_mask;
    E.b_rows = rows;
    E.b_nnz_bound = nnz;
    if (filt) {
      int64_t nnz_in = 0;
      for (int p = 0; p < W; ++p) nnz_in += T[(size_t)p * W + D.rank];
      E.b_nnz_bound = nnz_in;
    }
  }
  return URCCO_OK;
}

int build_sharded(urcco_context* c, const std::vector<std::vector<Shard>>& sh, const std::vector<DsParams>& ps, int64_t n_users, int32_t seed) {
  const int n_ds = (int)ps.size();
  const int W = c->world;
  const int R = 2 * W + 3;
  const int32_t n_items_a = (int32_t)ps[0].n_cols;
  FragPlan fp;
  fp.on = fragments(c);
  // ---- primary: input phase, then the balance key.  The key is the A'A row work added up from the user shards (the
  // work of A'B_d sums the same users' B_d row lengths and follows it closely), so the ranges are fixed before any
  // whole-matrix work and the one blocking host read comes right after the primary's short chain.  With fragments every
  // rank also transposes its own shard here (1 / W of the entries each), under the all-reduce of the key.
  URC(input_phase(c, 0, sh, ps, seed));
  URC(c->workers->run([&](size_t g) -> int {
    DevState& D = c->devs[g];
    URC(set_dev(D));
    EvState& A = D.ev[0];
    const Shard& s = sh[0][g];
    URC(D.work.ensure((size_t)n_items_a + 1));
    return urcco_dev_row_work_csr(A.s, s.n_rows, A.s_rp.p, A.s_ci.p, s.nnz, A.s_rp.p, n_items_a, D.work.p);
  }));
  URC(c->group_start());
  for (DevState& D : c->devs) {
    URC(set_dev(D));
    URC(c->all_reduce(D, D.work.p, n_items_a, 1, D.ev[0].s->stream));
  }
  URC(c->group_end());
  std::vector<int64_t> sizes;
  if (fp.on) {
    URC(c->workers->run([&](size_t g) -> int {
      DevState& D = c->devs[g];
      URC(set_dev(D));
      EvState& A = D.ev[0];
      const Shard& s = sh[0][g];
      URC(D.l_cp.ensure((size_t)n_items_a + 2));
      URC(D.l_ri.ensure((size_t)s.nnz + 4));
      URC(D.len16.ensure((size_t)n_items_a + 8));
      URC(D.len_bad.ensure(1));
      URC(D.bounds.ensure((size_t)W + 1));
      URC(D.rec.ensure((size_t)R * (size_t)W));
      URC(urcco_dev_transpose(A.s, s.n_rows, A.s_rp.p, A.s_ci.p, s.nnz, n_items_a, D.l_cnt.p, 0, n_items_a, D.l_cp.p, D.l_ri.p));
      A.s->begin(URCCO_STAGE_EXCHANGE);
      HIPC(urcco::launch_narrow_counts(A.s->stream, D.n_cu, D.l_cnt.p, n_items_a, D.len16.p, D.len_bad.p));
      A.s->end();
      // identical bounds on every rank: the same scan + split of the same summed key; they stay on the device
      URC(urcco_detail::partition_dev(A.s, n_items_a, D.work.p, W, D.bounds.p, nullptr));
      A.s->begin(URCCO_STAGE_EXCHANGE);
      HIPC(urcco::launch_frag_record(A.s->stream, W, D.bounds.p, D.l_cp.p, D.len_bad.p, D.rec.p + (size_t)R * (size_t)D.rank));
      A.s->end();
      return URCCO_OK;
    }));
    std::vector<int64_t> off((size_t)W), cnt((size_t)W, 8 * (int64_t)R);
    for (int r = 0; r < W; ++r) off[(size_t)r] = 8 * (int64_t)R * r;
    URC(c->group_start());
    for (DevState& D : c->devs) {
      URC(set_dev(D));
      URC(c->all_gather_v(D, D.rec.p + (size_t)R * (size_t)D.rank, D.rec.p, off.data(), cnt.data(), D.ev[0].s->stream));
    }
    URC(c->group_end());
    if (filtered(c)) {  // who needs which user: masks from the shard's rows of A' and the bounds; then what the primary's rows cost per destination
      URC(c->workers->run([&](size_t g) -> int {
        DevState& D = c->devs[g];
        URC(set_dev(D));
        EvState& A = D.ev[0];
        const Shard& s = sh[0][g];
        URC(D.need.ensure((size_t)s.n_rows + 1));
        // D.need is ONE buffer per GPU: the previous build's masked_lengths / pack_rows of the other event types' streams may still read
        // it when builds are enqueued back to back -- the primary's stream waits for their chains (ev_done is recorded at the
        // end of every event type's chain; an event never recorded is complete)
        for (EvState& Eo : D.ev)
          if (&Eo != &A && Eo.ev_done) HIPC(hipStreamWaitEvent(A.s->stream, Eo.ev_done, 0));
        A.s->begin(URCCO_STAGE_EXCHANGE);
        HIPC(urcco::launch_need_mask(A.s->stream, D.n_cu, s.n_rows, A.s_rp.p, A.s_ci.p, D.bounds.p, W, D.need.p));
        A.s->end();
        HIPC(hipEventRecord(D.need_ready, A.s->stream));
        return filtered_sizes(c, D, 0, s);
      }));
      URC(gather_filtered_sizes(c, 0));
    }
    URC(exchange_phase(c, 0, sh, ps, n_users, sizes, &fp));  // the build's one blocking read for the primary: shard sizes + fragment records
  } else {
    c->h_bounds.assign((size_t)W + 1, 0);
    for (DevState& D : c->devs) {
      URC(set_dev(D));
      std::vector<int32_t> b((size_t)W + 1);
      URC(urcco_dev_partition(D.ev[0].s, n_items_a, D.work.p, W, b.data()));  // synchronises the primary's stream
      c->h_bounds = b;
      D.item_lo = b[(size_t)D.rank];
      D.item_hi = b[(size_t)D.rank + 1];
    }
    URC(exchange_phase(c, 0, sh, ps, n_users, sizes));
  }
  c->h_sizes.assign((size_t)n_ds, 0);
  int64_t a_nnz = 0;
  for (int r = 0; r < W; ++r) a_nnz += sizes[(size_t)XS * r + 1];
  c->h_sizes[0] = a_nnz;
  URC(c->workers->run([&](size_t g) -> int {
    DevState& D = c->devs[g];
    URC(set_dev(D));
    EvState& A = D.ev[0];
    URC(D.a_cp[D.par].ensure((size_t)n_items_a + 2));
    if (fp.on) {
      int64_t ents = 0;
      for (int p = 0; p < W; ++p) ents += fp.cpb[(size_t)p * (size_t)(W + 1) + (size_t)D.rank + 1] - fp.cpb[(size_t)p * (size_t)(W + 1) + (size_t)D.rank];
      D.a_ents = ents;
      URC(D.a_ri[D.par].ensure((size_t)ents + 4));
      URC(urcco_dev_merge_fragments(A.s, W, D.item_lo, D.item_hi, n_items_a, D.f_len.p, fp.wire16 ? 1 : 0, D.f_ent.p, ents, A.sizes.p, post_of(D, 0).p,
                                    D.a_cp[D.par].p, D.a_ri[D.par].p));
    } else {
      D.a_ents = a_nnz;
      URC(D.a_ri[D.par].ensure((size_t)a_nnz + 4));
      URC(urcco_dev_transpose(A.s, n_users, A.f_rp.p, A.f_ci.p, a_nnz, n_items_a, post_of(D, 0).p, D.item_lo, D.item_hi, D.a_cp[D.par].p, D.a_ri[D.par].p));
    }
    HIPC(hipEventRecord(D.a_ready, A.s->stream));
    return stage_rows(D, A, A, 0, ps[0], ps[0], n_users, D.a_ents);
  }));
  // ---- secondaries: every input phase is enqueued on its own stream (they run under A'A); then per event type the shard
  // sizes are read (the host waits for that stream's sampling only), the exchange is issued and A'B_d runs behind it
  for (int d = 1; d < n_ds; ++d) URC(input_phase(c, d, sh, ps, seed));
  // With two or more secondaries their expand preparation is FUSED as in the one-rank build (one pass over the rank's CSC slice of A'
  // serves every secondary: one scattered sector per CSC entry instead of one line per entry AND event type).  It needs every
  // secondary's exchanged row_ptr, so all exchanges are issued first (each behind its own stream's sampling; they travel together) and
  // the A'B_d follow the fused pass.  Round 3 prepared every event type on its own in this path: 7.7 against 4.05 ms of row work on config 4.
  std::vector<int64_t> b_nnz((size_t)n_ds, 0);
  for (int d = 1; d < n_ds; ++d) {
    URC(exchange_phase(c, d, sh, ps, n_users, sizes));
    for (int r = 0; r < W; ++r) b_nnz[(size_t)d] += sizes[(size_t)XS * r + 1];
    c->h_sizes[(size_t)d] = b_nnz[(size_t)d];
  }
  bool fuse = n_ds >= 3 && n_ds - 1 <= urcco::EXPAND_MULTI_MAX && !(c->debug & 4096);
  for (int d = 1; d < n_ds; ++d) fuse = fuse && b_nnz[(size_t)d] < ((int64_t)1 << 32);
  URC(c->workers->run([&](size_t g) -> int {
    DevState& D = c->devs[g];
    URC(set_dev(D));
    if (fuse && D.a_ents > 0) {
      EvState& L = D.ev[1];
      if (L.s != D.ev[0].s) HIPC(hipStreamWaitEvent(L.s->stream, D.a_ready, 0));
      std::vector<const int64_t*> rp((size_t)n_ds - 1);
      std::vector<int64_t*> pst((size_t)n_ds - 1);
      std::vector<int32_t*> pl((size_t)n_ds - 1);
      std::vector<int64_t*> ts((size_t)n_ds - 1);
      for (int d = 1; d < n_ds; ++d) {
        EvState& E = D.ev[(size_t)d];
        HIPC(hipEventRecord(E.ev_sampled, E.s->stream));  // behind the scan that rebuilt the whole matrix's row_ptr
        if (E.s != L.s) HIPC(hipStreamWaitEvent(L.s->stream, E.ev_sampled, 0));
        URC(E.pre_pstart.ensure((size_t)D.a_ents + 1));
        URC(E.pre_plen.ensure((size_t)D.a_ents + 1));
        URC(E.pre_tsum.ensure(expand_tile_words(D.a_ents)));
        ts[(size_t)d - 1] = E.pre_tsum.p;
        rp[(size_t)d - 1] = E.b_rp;
        pst[(size_t)d - 1] = E.pre_pstart.p;
        pl[(size_t)d - 1] = E.pre_plen.p;
      }
      URC(expand_multi(L.s, n_ds - 1, D.a_cp[D.par].p, n_items_a, D.a_ri[D.par].p, D.a_ents, rp.data(), n_users, pst.data(), pl.data(), ts.data()));
      HIPC(hipEventRecord(D.b_expanded, L.s->stream));
    }
    for (int d = 1; d < n_ds; ++d) {
      EvState& E = D.ev[(size_t)d];
      if (E.s != D.ev[0].s) HIPC(hipStreamWaitEvent(E.s->stream, D.a_ready, 0));
      if (fuse && D.a_ents > 0 && E.s != D.ev[1].s) HIPC(hipStreamWaitEvent(E.s->stream, D.b_expanded, 0));
      URC(stage_rows(D, E, D.ev[0], d, ps[0], ps[(size_t)d], n_users, D.a_ents, fuse && D.a_ents > 0));
    }
    return URCCO_OK;
  }));
  return URCCO_OK;
}

int check_params(const DsParams& p, int d) {
  if (p.n_cols < 0 || p.n_cols > 0x7ffffff0ll) return fail(URCCO_BAD_ARG, "dataset %d: bad column count", d);
  if (p.max_rows <= 0 || p.k <= 0) return fail(URCCO_BAD_ARG, "dataset %d: maxElementsPerRow / maxInterestingElements must be positive", d);
  return URCCO_OK;
}

int run_build(urcco_context* c, const std::vector<std::vector<Shard>>& sh, const std::vector<DsParams>& ps, int64_t n_users, int32_t seed,
              hipStream_t input_stream, InputGate* gate = nullptr) {
  const int n_ds = (int)ps.size();
  for (DevState& D : c->devs) {
    if (!gate) URC(ensure_events(c, D, n_ds));  // with a gate the caller has done it (its staging thread is using the streams)
    // this build's set of the primary's shared buffers was last read by the A'B_d of the build before the previous one
    D.par = (D.par + 1) % A_SETS;
    for (size_t d = 1; d < D.ev.size(); ++d)
      if (D.ev[d].cons_valid[D.par] && D.ev[d].s != D.ev[0].s) HIPC(hipStreamWaitEvent(D.ev[0].s->stream, D.ev[d].ev_cons[D.par], 0));
    if (input_stream && c->devs.size() == 1) {
      HIPC(hipEventRecord(D.in_ready, input_stream));
      for (int d = 0; d < n_ds; ++d) HIPC(hipStreamWaitEvent(D.ev[(size_t)d].s->stream, D.in_ready, 0));
    }
  }
  if (!c->exchange()) return build_single(c, c->devs[0], [&] {
    std::vector<Shard> one((size_t)n_ds);
    for (int d = 0; d < n_ds; ++d) one[(size_t)d] = sh[(size_t)d][0];
    return one;
  }(), ps, n_users, seed, gate);
  return build_sharded(c, sh, ps, n_users, seed);
}

// ---------------------------------------------------------------------------------------------------------
// host level: staging
// ---------------------------------------------------------------------------------------------------------
PendingBuild::~PendingBuild() {
  if (gate)
    for (size_t d = 0; d < gate->staged.size(); ++d)
      if (!gate->released[d]) gate->release((int)d, URCCO_INTERNAL);  // a builder still waiting must not wait forever
  if (builder.joinable()) builder.join();
  for (urcco_indicators& o : res)  // results finish never handed out
    for (void* p : {(void*)o.row_ptr, (void*)o.col_idx, (void*)o.llr})
      if (p) (void)pinned_pool().put(p);
  if (stats_block) (void)pinned_pool().put(stats_block);
}

// One event type's indicator matrix device -> pinned host memory, by the thread that enqueued its chain (single-GPU host level): the
// slice's row_ptr and the statistics first, then -- its stream drained, the size known -- exactly nnz entries.
int download_event(urcco_context* c, PendingBuild* pb, int d) {
  DevState& D = c->devs[0];
  URC(set_dev(D));
  EvState& E = D.ev[(size_t)d];
  urcco_indicators& o = pb->res[(size_t)d];
  const int32_t n = D.item_hi - D.item_lo;  // one rank: every item row
  int64_t* h_stats = static_cast<int64_t*>(pb->stats_block) + (size_t)d * URCCO_STATS_LEN;
  int64_t* h_sampled = static_cast<int64_t*>(pb->stats_block) + (size_t)pb->n_ds * URCCO_STATS_LEN + (size_t)d;
  o.n_rows = n;
  o.n_cols = pb->ps[(size_t)d].n_cols;
  o.row_ptr = (int64_t*)pinned_pool().get(sizeof(int64_t) * ((size_t)n + 1));
  if (!o.row_ptr) return fail(URCCO_OOM_HOST, "pinned indicator row_ptr");
  o.row_ptr[0] = 0;
  if (n > 0) HIPC(hipMemcpyAsync(o.row_ptr + 1, E.c_rp.p + 1, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, E.s->stream));
  HIPC(hipMemcpyAsync(h_stats, E.stats.p, sizeof(int64_t) * URCCO_STATS_LEN, hipMemcpyDeviceToHost, E.s->stream));
  HIPC(hipMemcpyAsync(h_sampled, E.s_rp.p + pb->n_users, sizeof(int64_t), hipMemcpyDeviceToHost, E.s->stream));
  HIPC(hipStreamSynchronize(E.s->stream));
  pb->trace.mark("indicator row_ptr on the host", d);
  o.nnz = n > 0 ? o.row_ptr[n] : 0;
  o.col_idx = (int32_t*)pinned_pool().get(sizeof(int32_t) * (size_t)(o.nnz ? o.nnz : 1));
  o.llr = (double*)pinned_pool().get(sizeof(double) * (size_t)(o.nnz ? o.nnz : 1));
  if (!o.col_idx || !o.llr) return fail(URCCO_OOM_HOST, "pinned indicator arrays");
  if (o.nnz > 0) {
    // (round 6 A/B, profiles/r06_host_level_ab.log: the same bytes moved by the GPU's own stores into the mapped block instead of the copy engine
    // slowed the concurrent uploads down -- 158-172 ms per call against 129-131)
    // (also measured and not kept: the two arrays in 32 MB / 8 MB pieces so that upload chunks could slip in between -- 139-142 ms per call against 123-126:
    // uploads and downloads slow each other down whatever the granularity -- 7 GB cross the link in ~125 ms, ~56 GB/s in BOTH directions together)
    HIPC(hipMemcpyAsync(o.col_idx, E.c_idx.p, sizeof(int32_t) * (size_t)o.nnz, hipMemcpyDeviceToHost, E.s->stream));
    HIPC(hipMemcpyAsync(o.llr, E.c_llr.p, sizeof(double) * (size_t)o.nnz, hipMemcpyDeviceToHost, E.s->stream));
    HIPC(hipStreamSynchronize(E.s->stream));
  }
  pb->trace.mark("indicator entries on the host", d);
  pb->res_done[(size_t)d] = 1;
  return URCCO_OK;
}

// pageable host memory -> device through the GPU's pinned ring, `max_threads` copy threads; every chunk's H2D is enqueued
// on `st` as soon as the chunk sits in pinned memory, so the link is busy while later chunks are still being copied.
// Returns once the caller's bytes have all been READ (every chunk sits in pinned memory, its H2D enqueued).
int stage_copy(StageRing& ring, hipStream_t st, void* dst, const void* src, size_t bytes, int max_threads) {
  if (bytes == 0) return URCCO_OK;
  const size_t n_chunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
  const long long base = ring.next_chunk.fetch_add((long long)n_chunks);
  const size_t n_slots = ring.slots.size();
  std::atomic<size_t> take{0};
  std::atomic<int> status{URCCO_OK};
  auto worker = [&]() {
    if (hipSetDevice(ring.device) != hipSuccess) { status = URCCO_HIP_ERROR; return; }
    for (;;) {
      const size_t k = take.fetch_add(1);
      if (k >= n_chunks) break;
      const long long q = base + (long long)k;
      StageRing::Slot& s = *ring.slots[(size_t)(q % (long long)n_slots)];
      const long long gen = q / (long long)n_slots;
      while (s.gen.load(std::memory_order_acquire) != gen) std::this_thread::yield();  // the slot's previous user has recorded its event
      if (gen > 0 && hipEventSynchronize(s.ev) != hipSuccess) status = URCCO_HIP_ERROR;  // ... and its H2D has left the slot
      const size_t o = k * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - o);
      char* pin = ring.base + (size_t)(q % (long long)n_slots) * STAGE_CHUNK;
      memcpy(pin, (const char*)src + o, len);
      if (hipMemcpyAsync((char*)dst + o, pin, len, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(s.ev, st) != hipSuccess) status = URCCO_HIP_ERROR;
      s.gen.store(gen + 1, std::memory_order_release);
    }
  };
  const int nt = (int)std::min<size_t>((size_t)(max_threads > 0 ? max_threads : 1), n_chunks);
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(worker);
  worker();
  for (auto& t : th) t.join();
  if (status != URCCO_OK) return fail(URCCO_HIP_ERROR, "host -> device staging failed");
  return URCCO_OK;
}

std::mutex g_default_mu;
// The one-shot entry points share ONE process-wide context: a build occupies it from its stage to its finish.  A second thread's
// stage waits here for the first thread's finish (round 3 dropped the lock between the two halves: the second stage then failed with
// "previous staged build has not been finished" or, with different options, destroyed the context under the pending build -- ADVICE r03).
std::condition_variable g_default_cv;
bool g_default_busy = false;
std::thread::id g_default_owner;  // the thread whose staged build occupies the context
urcco_context* g_default_ctx = nullptr;
int g_default_n_gpus = -1, g_default_device = -1, g_default_mode = -1, g_default_flags = 0;

}  // namespace

extern "C" {

int urcco_comm_unique_id(void* out) {
  if (!out) return fail(URCCO_BAD_ARG, "urcco_comm_unique_id: out is NULL");
  Rccl* r = Rccl::get();
  if (!r) return fail(URCCO_RCCL_ERROR, "librccl could not be loaded: %s", dlerror() ? dlerror() : "not found");
  Rccl::UniqueId id;
  RCCLC(r, r->GetUniqueId(&id));
  memcpy(out, &id, URCCO_UNIQUE_ID_BYTES);
  return URCCO_OK;
}

void urcco_context_destroy(urcco_context* c) {
  if (!c) return;
  CallerDevice restore;
  c->pending.reset();  // joins a builder thread of a staged build nobody finished
  for (DevState& D : c->devs) {
    (void)hipSetDevice(D.device);
    for (urcco_session* s : D.sessions) (void)hipStreamSynchronize(s->stream);
    if (D.comm && c->rccl) (void)c->rccl->CommDestroy(D.comm);
    for (EvState& E : D.ev) E.release();
    for (int q = 0; q < A_SETS; ++q) { D.a_cp[q].release(); D.a_ri[q].release(); D.a_post[q].release(); }
    D.work.release(); D.bounds.release();
    D.l_cnt.release(); D.l_ri.release(); D.len_bad.release(); D.f_ent.release(); D.l_cp.release(); D.rec.release(); D.len16.release(); D.f_len.release();
    if (D.a_ready) (void)hipEventDestroy(D.a_ready);
    if (D.in_ready) (void)hipEventDestroy(D.in_ready);
    if (D.b_expanded) (void)hipEventDestroy(D.b_expanded);
    if (D.need_ready) (void)hipEventDestroy(D.need_ready);
    D.need.release();
    for (urcco_session* s : D.sessions) urcco_session_destroy(s);
  }
  if (c->emu_stream) (void)hipStreamDestroy(c->emu_stream);
  c->rings.clear();
  delete c;
}

int urcco_context_create(const urcco_options* options, const urcco_comm_config* comm, urcco_context** out) {
  CallerDevice restore;
  return guarded([&]() -> int {
    if (!out) return fail(URCCO_BAD_ARG, "urcco_context_create: out is NULL");
    *out = nullptr;
    const int n_dev = urcco_device_count();
    if (n_dev <= 0) return fail(URCCO_NO_DEVICE, "no HIP device visible (liburcco has no CPU fallback)");
    const int first = options ? options->device : 0;
    int n_local = options ? options->n_gpus : 0;
    if (first < 0 || first >= n_dev) return fail(URCCO_BAD_ARG, "device %d out of range [0,%d)", first, n_dev);
    const bool emulate = options && (options->flags & URCCO_FLAG_EMULATE_RANKS);
    if (emulate) {  // measurement only: n_gpus ranks on the one device `first`, see include/urcco.h
      if (n_local < 1) return fail(URCCO_BAD_ARG, "URCCO_FLAG_EMULATE_RANKS: n_gpus must name the number of ranks to emulate");
      if (!comm || !comm->collectives) return fail(URCCO_BAD_ARG, "URCCO_FLAG_EMULATE_RANKS needs caller-supplied collectives (RCCL cannot put two ranks on one device)");
    } else if (n_local < 0 || first + n_local > n_dev) {
      return fail(URCCO_BAD_ARG, "n_gpus %d from device %d: only %d device(s) visible", n_local, first, n_dev);
    }
    if (n_local == 0) n_local = n_dev - first;
    const int mode = options ? options->row_rate_mode : URCCO_ROW_RATE_MAHOUT_INT_DIV;
    if ((mode & ~URCCO_RNG_MIX32) != URCCO_ROW_RATE_MAHOUT_INT_DIV && (mode & ~URCCO_RNG_MIX32) != URCCO_ROW_RATE_FRACTIONAL) return fail(URCCO_BAD_ARG, "unknown row_rate_mode %d", mode);
    std::unique_ptr<urcco_context, void (*)(urcco_context*)> c(new urcco_context(), urcco_context_destroy);
    c->row_rate_mode = mode;
    c->flags = options ? options->flags : 0;
    c->world = (comm && comm->world_size > 0) ? comm->world_size : n_local;
    c->first_rank = comm ? comm->first_rank : 0;
    if (c->first_rank < 0 || c->first_rank + n_local > c->world) return fail(URCCO_BAD_ARG, "ranks [%d, %d) outside a world of %d", c->first_rank, c->first_rank + n_local, c->world);
    if (comm && comm->collectives) {
      const urcco_collectives* k = comm->collectives;
      // a host built against an older header hands in a shorter struct: reading its missing tail would yield a garbage function pointer
      if (k->struct_size < sizeof(urcco_collectives))
        return fail(URCCO_BAD_ARG, "urcco_collectives.struct_size is %zu, this library's struct has %zu bytes (ABI %d): rebuild the host against include/urcco.h", k->struct_size,
                    sizeof(urcco_collectives), URCCO_VERSION);
      if (!k->group_start || !k->group_end || !k->all_reduce_sum || !k->all_gather_v) return fail(URCCO_BAD_ARG, "urcco_collectives: every callback must be set");
      c->cb = *k;
      c->have_cb = true;
    }
    c->devs.resize((size_t)n_local);
    for (int g = 0; g < n_local; ++g) {
      DevState& D = c->devs[(size_t)g];
      D.device = emulate ? first : first + g;
      D.rank = c->first_rank + g;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, D.device) == hipSuccess && prop.multiProcessorCount > 0) D.n_cu = prop.multiProcessorCount;
    }
    if (emulate) {
      HIPC(hipSetDevice(first));
      HIPC(hipStreamCreate(&c->emu_stream));
      c->workers->sequential = true;
    }
    c->workers->start((size_t)n_local);
    const unsigned hc = std::thread::hardware_concurrency();
    c->copy_threads = hc >= 32 ? 8 : (hc >= 8 ? 4 : 2);
    if (const char* e = getenv("URCCO_COPY_THREADS")) c->copy_threads = std::max(1, std::min(64, atoi(e)));
    if (!c->have_cb && (c->world > 1 || (c->flags & URCCO_FLAG_FORCE_EXCHANGE))) {
      c->rccl = Rccl::get();
      if (!c->rccl) return fail(URCCO_RCCL_ERROR, "a multi-rank build needs librccl, which could not be loaded");
      if (n_local == c->world) {
        std::vector<int> ids((size_t)n_local);
        std::vector<Rccl::Comm> comms((size_t)n_local, nullptr);
        for (int g = 0; g < n_local; ++g) ids[(size_t)g] = first + g;
        RCCLC(c->rccl, c->rccl->CommInitAll(comms.data(), n_local, ids.data()));
        for (int g = 0; g < n_local; ++g) c->devs[(size_t)g].comm = comms[(size_t)g];
      } else {
        if (!comm || !comm->nccl_unique_id) return fail(URCCO_BAD_ARG, "ranks spread over processes need comm->nccl_unique_id");
        Rccl::UniqueId id;
        memcpy(&id, comm->nccl_unique_id, URCCO_UNIQUE_ID_BYTES);
        RCCLC(c->rccl, c->rccl->GroupStart());
        for (int g = 0; g < n_local; ++g) {
          HIPC(hipSetDevice(first + g));
          RCCLC(c->rccl, c->rccl->CommInitRank(&c->devs[(size_t)g].comm, c->world, id, c->first_rank + g));
        }
        RCCLC(c->rccl, c->rccl->GroupEnd());
      }
    }
    *out = c.release();
    return URCCO_OK;
  });
}

int32_t urcco_context_local_gpus(const urcco_context* c) { return c ? (int32_t)c->devs.size() : 0; }

int urcco_context_set_flags(urcco_context* c, int32_t flags) {
  if (!c) return fail(URCCO_BAD_ARG, "context is NULL");
  if (!c->have_cb && !c->rccl && c->world == 1 && (flags & URCCO_FLAG_FORCE_EXCHANGE))
    return fail(URCCO_BAD_ARG, "URCCO_FLAG_FORCE_EXCHANGE must be given to urcco_context_create (the communicator is created there)");
  if ((flags ^ c->flags) & URCCO_FLAG_EMULATE_RANKS) return fail(URCCO_BAD_ARG, "URCCO_FLAG_EMULATE_RANKS must be given to urcco_context_create");
  c->flags = flags;
  return URCCO_OK;
}

int urcco_context_set_debug(urcco_context* c, int32_t flags) {
  if (!c) return fail(URCCO_BAD_ARG, "context is NULL");
  c->debug = flags;
  for (DevState& D : c->devs)
    for (urcco_session* s : D.sessions) s->debug = flags;
  return URCCO_OK;
}

int urcco_context_set_timing(urcco_context* c, int32_t enable) {
  CallerDevice restore;
  if (!c) return fail(URCCO_BAD_ARG, "context is NULL");
  c->timing = enable != 0;
  for (DevState& D : c->devs) {
    HIPC(hipSetDevice(D.device));
    for (urcco_session* s : D.sessions) URC(urcco_session_set_timing(s, enable));
  }
  return URCCO_OK;
}

int urcco_context_get_timings(urcco_context* c, double* ms, int64_t* launches) {
  CallerDevice restore;
  if (!c || !ms || !launches) return fail(URCCO_BAD_ARG, "urcco_context_get_timings: bad argument");
  for (int i = 0; i < URCCO_N_STAGES; ++i) { ms[i] = 0; launches[i] = 0; }
  for (DevState& D : c->devs) {
    HIPC(hipSetDevice(D.device));
    for (urcco_session* s : D.sessions) {
      double m[URCCO_N_STAGES];
      int64_t n[URCCO_N_STAGES];
      URC(urcco_session_get_timings(s, m, n));
      for (int i = 0; i < URCCO_N_STAGES; ++i) { ms[i] += m[i]; launches[i] += n[i]; }
    }
  }
  return URCCO_OK;
}

int urcco_context_get_timings_gpu(urcco_context* c, int32_t g, double* ms, int64_t* launches) {
  CallerDevice restore;
  if (!c || !ms || !launches || g < 0 || (size_t)g >= c->devs.size()) return fail(URCCO_BAD_ARG, "urcco_context_get_timings_gpu: bad argument");
  for (int i = 0; i < URCCO_N_STAGES; ++i) { ms[i] = 0; launches[i] = 0; }
  DevState& D = c->devs[(size_t)g];
  HIPC(hipSetDevice(D.device));
  for (urcco_session* s : D.sessions) {
    double m[URCCO_N_STAGES];
    int64_t n[URCCO_N_STAGES];
    URC(urcco_session_get_timings(s, m, n));
    for (int i = 0; i < URCCO_N_STAGES; ++i) { ms[i] += m[i]; launches[i] += n[i]; }
  }
  return URCCO_OK;
}

int urcco_context_synchronize(urcco_context* c) {
  CallerDevice restore;
  if (!c) return fail(URCCO_BAD_ARG, "context is NULL");
  for (DevState& D : c->devs) {
    HIPC(hipSetDevice(D.device));
    for (urcco_session* s : D.sessions) HIPC(hipStreamSynchronize(s->stream));
  }
  return URCCO_OK;
}

int urcco_context_wait_stream(urcco_context* c, void* stream) {
  CallerDevice restore;
  if (!c) return fail(URCCO_BAD_ARG, "context is NULL");
  if (c->devs.size() != 1) return fail(URCCO_BAD_ARG, "urcco_context_wait_stream: single-GPU contexts only");
  DevState& D = c->devs[0];
  HIPC(hipSetDevice(D.device));
  for (EvState& E : D.ev)
    if (E.ev_done) HIPC(hipStreamWaitEvent((hipStream_t)stream, E.ev_done, 0));
  return URCCO_OK;
}

int urcco_context_build_device(urcco_context* c, const urcco_dev_dataset* datasets, int32_t n_ds, int64_t n_users, int32_t seed, void* input_stream,
                               urcco_dev_result* out) {
  CallerDevice restore;
  return guarded([&]() -> int {
    err_buf()[0] = 0;
    if (!c || !datasets || n_ds <= 0 || !out || n_users < 0) return fail(URCCO_BAD_ARG, "urcco_context_build_device: bad argument");
    const size_t L = c->devs.size();
    std::vector<DsParams> ps((size_t)n_ds);
    std::vector<std::vector<Shard>> sh((size_t)n_ds, std::vector<Shard>(L));
    for (int d = 0; d < n_ds; ++d) {
      const urcco_dev_dataset& ds = datasets[d];
      DsParams& p = ps[(size_t)d];
      p.n_cols = ds.n_cols; p.max_rows = ds.max_elements_per_row; p.k = ds.max_interesting_elements; p.has_min_llr = ds.has_min_llr; p.min_llr = ds.min_llr;
      URC(check_params(p, d));
      if (!ds.shards) return fail(URCCO_BAD_ARG, "dataset %d: shards is NULL", d);
      for (size_t g = 0; g < L; ++g) {
        const urcco_dev_shard& s = ds.shards[g];
        if (s.n_rows < 0 || s.nnz < 0 || s.row_base < 0 || !s.row_ptr || (s.nnz > 0 && !s.col_idx)) return fail(URCCO_BAD_ARG, "dataset %d shard %zu: bad shard", d, g);
        if (s.n_rows != datasets[0].shards[g].n_rows || s.row_base != datasets[0].shards[g].row_base)
          return fail(URCCO_BAD_ARG, "dataset %d shard %zu: every event type shards the users the same way", d, g);
        sh[(size_t)d][g] = Shard{s.n_rows, s.row_base, s.nnz, s.row_ptr, s.col_idx};
      }
    }
    if (!c->exchange() && sh[0][0].n_rows != n_users) return fail(URCCO_BAD_ARG, "one rank holds %lld rows, n_users_total is %lld", (long long)sh[0][0].n_rows, (long long)n_users);
    URC(run_build(c, sh, ps, n_users, seed, (hipStream_t)input_stream));
    for (int d = 0; d < n_ds; ++d)
      for (size_t g = 0; g < L; ++g) {
        DevState& D = c->devs[g];
        EvState& E = D.ev[(size_t)d];
        urcco_dev_result& r = out[(size_t)d * L + g];
        r.item_lo = D.item_lo; r.item_hi = D.item_hi;
        r.row_ptr = E.c_rp.p; r.col_idx = E.c_idx.p; r.llr = E.c_llr.p; r.stats = E.stats.p;
        r.sampled_row_ptr = E.b_rp; r.sampled_col_idx = E.b_ci; r.sampled_rows = E.b_rows;
        r.sampled_col_mask = (int32_t)E.b_col_mask;
        r.sampled_nnz_total = c->exchange() && (size_t)d < c->h_sizes.size() ? c->h_sizes[(size_t)d] : -1;
      }
    return URCCO_OK;
  });
}

// ---- host level, split: stage (reads the caller's arrays) / finish (waits for the build, hands out the results) ----
int urcco_context_stage(urcco_context* c, const urcco_dataset* datasets, int32_t n_ds, int32_t seed) {
  CallerDevice restore;
  return guarded([&]() -> int {
    err_buf()[0] = 0;
    if (!c || !datasets || n_ds <= 0) return fail(URCCO_BAD_ARG, "datasets is NULL or n_datasets <= 0");
    if (c->pending) return fail(URCCO_BAD_ARG, "urcco_context_stage: the previous staged build has not been finished");
    const size_t L = c->devs.size();
    if ((int)L != c->world) return fail(URCCO_BAD_ARG, "the host-level build needs every rank in this process");
    std::unique_ptr<PendingBuild> P(new PendingBuild());
    HostTrace& trace = P->trace;
    P->n_ds = n_ds;
    P->seed = seed;
    P->ps.resize((size_t)n_ds);
    P->nnz_raw.resize((size_t)n_ds);
    std::vector<DsParams>& ps = P->ps;
    const int64_t n_users = datasets[0].matrix.n_rows;
    P->n_users = n_users;
    for (int d = 0; d < n_ds; ++d) {
      const urcco_csr& m = datasets[d].matrix;
      if (m.n_rows < 0 || m.n_cols < 0 || m.n_rows > 0x7fffffffll) return fail(URCCO_BAD_ARG, "dataset %d: bad shape", d);
      if (!m.row_ptr) return fail(URCCO_BAD_ARG, "dataset %d: row_ptr is NULL", d);
      if (m.row_ptr[0] != 0) return fail(URCCO_BAD_ARG, "dataset %d: row_ptr[0] != 0", d);
      if (m.n_rows != n_users)
        return fail(URCCO_BAD_ARG, "dataset %d has %lld rows, the primary has %lld: all matrices share the user dictionary", d, (long long)m.n_rows, (long long)n_users);
      const int64_t nnz = m.row_ptr[m.n_rows];
      if (nnz < 0 || (nnz > 0 && !m.col_idx)) return fail(URCCO_BAD_ARG, "dataset %d: bad nnz / col_idx", d);
      P->nnz_raw[(size_t)d] = nnz;
      DsParams& p = ps[(size_t)d];
      p.n_cols = m.n_cols; p.max_rows = datasets[d].max_elements_per_row; p.k = datasets[d].max_interesting_elements;
      p.has_min_llr = datasets[d].has_min_llr; p.min_llr = datasets[d].min_llr;
      URC(check_params(p, d));
    }
    // ---- user ranges of the GPUs: contiguous, ~equal interactions summed over the event types
    std::vector<int64_t> cut(L + 1, 0);
    cut[L] = n_users;
    if (L > 1) {
      auto total_at = [&](int64_t u) { int64_t t = 0; for (int d = 0; d < n_ds; ++d) t += datasets[d].matrix.row_ptr[u]; return t; };
      const int64_t total = total_at(n_users);
      for (size_t g = 1; g < L; ++g) {
        const int64_t target = total / (int64_t)L * (int64_t)g;
        int64_t lo = cut[g - 1], hi = n_users;
        while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (total_at(mid) >= target) hi = mid; else lo = mid + 1; }
        cut[g] = lo;
      }
      // the copy ranges below come from the caller's row_ptr at the cuts, BEFORE the device-side check has seen it: they must
      // at least be ordered and inside the arrays (a non-monotone row_ptr would otherwise become an out-of-bounds host read)
      for (int d = 0; d < n_ds; ++d) {
        const int64_t* rp = datasets[d].matrix.row_ptr;
        for (size_t g = 0; g < L; ++g)
          if (rp[cut[g]] < 0 || rp[cut[g]] > rp[cut[g + 1]] || rp[cut[g + 1]] > rp[n_users]) return fail(URCCO_BAD_ARG, "dataset %d: row_ptr not monotone", d);
      }
    }
    for (DevState& D : c->devs) URC(ensure_events(c, D, n_ds));
    // ---- stage + validate every shard on its event stream (copy threads feed the pinned ring; the link is busy while
    // the next chunks are copied), heaviest transfers last so that the primary starts first
    trace.mark("arguments checked, streams ready");
    while (c->rings.size() < L) c->rings.emplace_back(new StageRing());
    for (size_t g = 0; g < L; ++g) URC(c->rings[g]->ensure(c->devs[g].device, L > 1 ? 12 : 24));
    P->sh.assign((size_t)n_ds, std::vector<Shard>(L));
    std::vector<std::vector<Shard>>& sh = P->sh;
    const int threads_each = L > 1 ? std::max(1, c->copy_threads / 2) : c->copy_threads;  // every GPU has its own link: all stage at once
    // buffers and shard descriptors first (everything but the bytes is known from the caller's row_ptr)
    for (int d = 0; d < n_ds; ++d) {
      const urcco_csr& m = datasets[d].matrix;
      for (size_t g = 0; g < L; ++g) {
        DevState& D = c->devs[g];
        URC(set_dev(D));
        EvState& E = D.ev[(size_t)d];
        const int64_t u0 = cut[g], u1 = cut[g + 1], rows = u1 - u0;
        const int64_t e0 = m.row_ptr[u0], e1 = m.row_ptr[u1], nnz = e1 - e0;
        if (nnz < 0) return fail(URCCO_BAD_ARG, "dataset %d: row_ptr not monotone", d);
        URC(E.in_rp.ensure((size_t)rows + 1));
        URC(E.in_ci.ensure((size_t)nnz + 4));
        URC(E.verr.ensure(1));
        URC(E.ensure_h_verr());
        sh[(size_t)d][g] = Shard{rows, u0, nnz, E.in_rp.p, E.in_ci.p};
      }
    }
    auto stage_event = [&](int d) -> int {  // copies + boundary check of event type d, enqueued on its stream on every GPU
      const urcco_csr& m = datasets[d].matrix;
      return c->workers->run([&](size_t g) -> int {
        DevState& D = c->devs[g];
        URC(set_dev(D));
        EvState& E = D.ev[(size_t)d];
        const Shard& x = sh[(size_t)d][g];
        const int64_t e0 = m.row_ptr[x.row_base];
        URC(stage_copy(*c->rings[g], E.s->stream, E.in_rp.p, m.row_ptr + x.row_base, sizeof(int64_t) * ((size_t)x.n_rows + 1), threads_each));
        URC(stage_copy(*c->rings[g], E.s->stream, E.in_ci.p, m.col_idx + e0, sizeof(int32_t) * (size_t)x.nnz, threads_each));
        HIPC(hipMemsetAsync(E.verr.p, 0, sizeof(unsigned long long), E.s->stream));
        int gl = x.n_rows > 0 ? ceil_log2_i64((x.nnz + x.n_rows - 1) / x.n_rows) : 1;
        gl = gl < 1 ? 1 : (gl > 6 ? 6 : gl);
        HIPC(urcco::launch_validate_csr(E.s->stream, D.n_cu, x.n_rows, E.in_rp.p, E.in_ci.p, x.nnz, (int32_t)m.n_cols, gl, e0, E.verr.p));
        HIPC(urcco::launch_rebase_i64(E.s->stream, D.n_cu, E.in_rp.p, x.n_rows + 1, e0));
        HIPC(urcco::launch_publish_word(E.s->stream, E.verr.p, E.h_verr_dev));
        return URCCO_OK;
      });
    };
    if (L == 1 && !c->exchange()) {
      // one GPU: the build is enqueued by its own thread(s) while this thread stages; every event type starts the moment its
      // own matrices have landed and passed the boundary check (InputGate)
      P->gate.reset(new InputGate(n_ds));
      P->gate->trace = &trace;
      PendingBuild* pb = P.get();
      pb->res.assign((size_t)n_ds, urcco_indicators{});
      pb->res_done.assign((size_t)n_ds, 0);
      pb->stats_block = pinned_pool().get(sizeof(int64_t) * (URCCO_STATS_LEN + 1) * (size_t)n_ds);
      if (!pb->stats_block) return fail(URCCO_OOM_HOST, "pinned statistics block");
      memset(pb->stats_block, 0, sizeof(int64_t) * (URCCO_STATS_LEN + 1) * (size_t)n_ds);
      pb->gate->after_chain = [c, pb](int d) { return download_event(c, pb, d); };
      {  // (fixed before the builder starts: it reads gate->order)
        std::vector<int> ord((size_t)n_ds);
        for (int d = 0; d < n_ds; ++d) ord[(size_t)d] = d;
        std::stable_sort(ord.begin() + 1, ord.end(), [&](int a, int b) { return pb->nnz_raw[(size_t)a] > pb->nnz_raw[(size_t)b]; });
        pb->gate->order = ord;
      }
      pb->builder = std::thread([c, pb] {
        pb->build_st = guarded([&] { return run_build(c, pb->sh, pb->ps, pb->n_users, pb->seed, nullptr, pb->gate.get()); });
        if (pb->build_st != URCCO_OK) pb->build_msg = err_buf();
      });
      int stage_st = URCCO_OK;
      std::string stage_msg;
      // upload order: the primary, then the secondaries from the largest down -- the heaviest A'B_d starts first and runs under the
      // remaining uploads, and what is still to come when the link goes quiet is the smallest chain and its few results (round 3
      // staged in index order: on config 4 the largest secondary's SpGEMM and 2.2 GB of finished results waited behind the uploads)
      const std::vector<int>& order = P->gate->order;
      for (int d : order) {
        if (stage_st == URCCO_OK) {
          stage_st = guarded([&] { return stage_event(d); });
          if (stage_st != URCCO_OK) stage_msg = err_buf();
        }
        P->gate->release(d, stage_st);  // also on failure: the builder must not wait forever
        trace.mark("staged (copies enqueued)", d);
      }
      if (stage_st != URCCO_OK) {
        P.reset();  // joins the builder
        return fail(stage_st, "%s", stage_msg.c_str());
      }
    } else {
      // several GPUs / the exchange path: stage + validate everything; the boundary check must have passed before any kernel
      // consumes the matrices, and the build itself (blocking host reads between its collectives) is issued by finish
      for (int d = 0; d < n_ds; ++d) URC(stage_event(d));
      P->build_deferred = true;
    }
    trace.mark("caller arrays released");
    c->pending = std::move(P);
    return URCCO_OK;
  });
}

int urcco_context_finish(urcco_context* c, urcco_indicators* out, urcco_dataset_stats* stats) {
  CallerDevice restore;
  if (out && c && c->pending)
    for (int d = 0; d < c->pending->n_ds; ++d) memset(&out[d], 0, sizeof(urcco_indicators));
  int n_ds = 0;
  const int status = guarded([&]() -> int {
    err_buf()[0] = 0;
    if (!c || !out) return fail(URCCO_BAD_ARG, "context / out is NULL");
    if (!c->pending) return fail(URCCO_BAD_ARG, "urcco_context_finish: nothing staged");
    std::unique_ptr<PendingBuild> P = std::move(c->pending);
    HostTrace& trace = P->trace;
    n_ds = P->n_ds;
    const size_t L = c->devs.size();
    const std::vector<DsParams>& ps = P->ps;
    const int64_t n_users = P->n_users;
    if (stats)
      for (int d = 0; d < n_ds; ++d) memset(&stats[d], 0, sizeof(urcco_dataset_stats));
    if (P->build_deferred) {
      for (int d = 0; d < n_ds; ++d)
        for (size_t g = 0; g < L; ++g) {
          DevState& D = c->devs[g];
          URC(set_dev(D));
          EvState& E = D.ev[(size_t)d];
          unsigned long long bad = 0;
          HIPC(hipMemcpyAsync(&bad, E.verr.p, sizeof(bad), hipMemcpyDeviceToHost, E.s->stream));
          HIPC(hipStreamSynchronize(E.s->stream));
          if (bad) return fail(URCCO_BAD_ARG, "dataset %d: %llu invalid entries (row_ptr not monotone, or col_idx out of [0, n_cols) / not strictly increasing inside a row)", d, bad);
        }
      URC(run_build(c, P->sh, ps, n_users, P->seed, nullptr));
    } else {
      if (P->builder.joinable()) P->builder.join();
      trace.mark("build enqueued");
      if (P->build_st != URCCO_OK) return fail(P->build_st, "%s", P->build_msg.c_str());
      if (!P->res.empty()) {  // one GPU: every event type's thread has already brought its results over (download_event)
        const int64_t* hs = static_cast<const int64_t*>(P->stats_block);
        for (int d = 0; d < n_ds; ++d) {
          if (!P->res_done[(size_t)d]) return fail(URCCO_INTERNAL, "event type %d: results were not downloaded", d);
          out[d] = P->res[(size_t)d];
          P->res[(size_t)d] = urcco_indicators{};  // the blocks now belong to the caller
          if (stats) {
            urcco_dataset_stats& st = stats[d];
            const int64_t* h = hs + (size_t)d * URCCO_STATS_LEN;
            st.nnz_raw = P->nnz_raw[(size_t)d];
            st.nnz_out = out[d].nnz;
            st.pairs = h[0];
            for (int b = 0; b < URCCO_N_BINS; ++b) st.rows_by_bin[b] = h[1 + (size_t)b];
            st.nnz_sampled = hs[(size_t)n_ds * URCCO_STATS_LEN + (size_t)d];
          }
        }
        trace.mark("return");
        return URCCO_OK;
      }
    }
    // ---- results: row_ptr of every GPU's slice first (small), then exactly nnz entries each
    const int32_t n_items_a = (int32_t)ps[0].n_cols;
    // per-build statistics land in PINNED memory: an "async" copy into pageable memory blocks the calling thread until the stream
    // has drained, which serialised the result phase behind the slowest event type
    struct PinnedBlock {
      void* p = nullptr;
      ~PinnedBlock() { if (p) (void)pinned_pool().put(p); }
    } stats_block;
    stats_block.p = pinned_pool().get(sizeof(int64_t) * (URCCO_STATS_LEN + 1) * (size_t)n_ds * L);
    if (!stats_block.p) return fail(URCCO_OOM_HOST, "pinned statistics block");
    int64_t* h_stats = static_cast<int64_t*>(stats_block.p);
    memset(h_stats, 0, sizeof(int64_t) * (URCCO_STATS_LEN + 1) * (size_t)n_ds * L);
    int64_t* h_sampled = h_stats + (size_t)URCCO_STATS_LEN * (size_t)n_ds * L;  // [n_ds]: nnz' of the single-rank build
    for (int d = 0; d < n_ds; ++d) {
      urcco_indicators& o = out[d];
      o.n_rows = n_items_a;
      o.n_cols = ps[(size_t)d].n_cols;
      o.row_ptr = (int64_t*)pinned_pool().get(sizeof(int64_t) * ((size_t)n_items_a + 1));
      if (!o.row_ptr) return fail(URCCO_OOM_HOST, "pinned indicator row_ptr");
      o.row_ptr[0] = 0;
      for (size_t g = 0; g < L; ++g) {
        DevState& D = c->devs[g];
        URC(set_dev(D));
        EvState& E = D.ev[(size_t)d];
        const int32_t n = D.item_hi - D.item_lo;
        // slice row_ptr[1..n] lands at out.row_ptr[item_lo + 1 ..]; re-based below once the slices' sizes are known
        if (n > 0) HIPC(hipMemcpyAsync(o.row_ptr + D.item_lo + 1, E.c_rp.p + 1, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, E.s->stream));
        HIPC(hipMemcpyAsync(h_stats + ((size_t)d * L + g) * URCCO_STATS_LEN, E.stats.p, sizeof(int64_t) * URCCO_STATS_LEN, hipMemcpyDeviceToHost, E.s->stream));
        if (stats && !c->exchange() && g == 0)
          HIPC(hipMemcpyAsync(h_sampled + d, E.s_rp.p + n_users, sizeof(int64_t), hipMemcpyDeviceToHost, E.s->stream));
        HIPC(hipEventRecord(E.ev_rp, E.s->stream));
      }
    }
    // The entries of an event type leave the moment ITS row_ptr has arrived, in the order the event types finish -- not in
    // index order: round 3 waited for event 1 (the heaviest: `view`) before it even asked for the rows of events 2.., which had been
    // ready for tens of milliseconds on config 4 (their 2.2 GB now cross the link under view's SpGEMM).
    std::vector<char> fetched((size_t)n_ds, 0);
    for (int done = 0; done < n_ds; ++done) {
      int d = -1;
      for (unsigned spin = 0; d < 0; ++spin) {
        for (int e = 0; e < n_ds && d < 0; ++e) {
          if (fetched[(size_t)e]) continue;
          bool ready = true;
          for (size_t g = 0; g < L && ready; ++g) {
            URC(set_dev(c->devs[g]));
            const hipError_t q = hipEventQuery(c->devs[g].ev[(size_t)e].ev_rp);
            if (q == hipErrorNotReady) ready = false;
            else if (q != hipSuccess) return hip_fail(q, "hipEventQuery(ev_rp)");
          }
          if (ready) d = e;
        }
        if (d < 0) {
          if (spin < 64) std::this_thread::yield();
          else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
      }
      fetched[(size_t)d] = 1;
      urcco_indicators& o = out[d];
      std::vector<int64_t> base(L + 1, 0);
      for (size_t g = 0; g < L; ++g) {
        DevState& D = c->devs[g];
        URC(set_dev(D));
        HIPC(hipEventSynchronize(D.ev[(size_t)d].ev_rp));
        trace.mark("indicator row_ptr on the host", d);
        const int32_t n = D.item_hi - D.item_lo;
        base[g + 1] = base[g] + (n > 0 ? o.row_ptr[D.item_hi] : 0);
      }
      o.nnz = base[L];
      o.col_idx = (int32_t*)pinned_pool().get(sizeof(int32_t) * (size_t)(o.nnz ? o.nnz : 1));
      o.llr = (double*)pinned_pool().get(sizeof(double) * (size_t)(o.nnz ? o.nnz : 1));
      if (!o.col_idx || !o.llr) return fail(URCCO_OOM_HOST, "pinned indicator arrays");
      for (size_t g = 0; g < L; ++g) {
        DevState& D = c->devs[g];
        URC(set_dev(D));
        EvState& E = D.ev[(size_t)d];
        const int64_t nz = base[g + 1] - base[g];
        if (nz > 0) {
          HIPC(hipMemcpyAsync(o.col_idx + base[g], E.c_idx.p, sizeof(int32_t) * (size_t)nz, hipMemcpyDeviceToHost, E.s->stream));
          HIPC(hipMemcpyAsync(o.llr + base[g], E.c_llr.p, sizeof(double) * (size_t)nz, hipMemcpyDeviceToHost, E.s->stream));
        }
        if (base[g] != 0)
          for (int32_t i = D.item_lo + 1; i <= D.item_hi; ++i) o.row_ptr[i] += base[g];
      }
    }
    URC(urcco_context_synchronize(c));
    trace.mark("all indicator entries on the host");
    if (stats)
      for (int d = 0; d < n_ds; ++d) {
        urcco_dataset_stats& st = stats[d];
        st.nnz_raw = P->nnz_raw[(size_t)d];
        st.nnz_out = out[d].nnz;
        for (size_t g = 0; g < L; ++g) {
          const int64_t* h = h_stats + ((size_t)d * L + g) * URCCO_STATS_LEN;
          st.pairs += h[0];
          for (int b = 0; b < URCCO_N_BINS; ++b) st.rows_by_bin[b] += h[1 + (size_t)b];
        }
        st.nnz_sampled = c->exchange() ? c->h_sizes[(size_t)d] : h_sampled[d];
      }
    trace.mark("return");
    return URCCO_OK;
  });
  if (status != URCCO_OK && out && n_ds > 0) urcco_free_indicators(out, n_ds);  // out[0..n_ds) was zeroed above: only library blocks are released
  return status;
}

int urcco_context_cross_occurrence(urcco_context* c, const urcco_dataset* datasets, int32_t n_ds, int32_t seed, urcco_indicators* out,
                                   urcco_dataset_stats* stats) {
  if (out && n_ds > 0) memset(out, 0, sizeof(urcco_indicators) * (size_t)n_ds);  // before any fallible step: a failure frees nothing of the caller's
  if (!out) return fail(URCCO_BAD_ARG, "out is NULL");
  URC(urcco_context_stage(c, datasets, n_ds, seed));
  return urcco_context_finish(c, out, stats);
}

// a staged build nobody will take: finished into blocks that are released at once (the pending state is consumed whatever the outcome)
static void discard_pending(urcco_context* c) {
  if (!c || !c->pending) return;
  const int n = c->pending->n_ds;
  std::vector<urcco_indicators> tmp((size_t)n);
  memset(tmp.data(), 0, sizeof(urcco_indicators) * (size_t)n);
  (void)urcco_context_finish(c, tmp.data(), nullptr);
  urcco_free_indicators(tmp.data(), n);
}

// ---- process-wide default context + the Mahout-shaped one-shot entry points ---------------------------------
int urcco_shutdown(void) {
  std::lock_guard<std::mutex> g(g_default_mu);
  if (g_default_ctx) urcco_context_destroy(g_default_ctx);
  g_default_ctx = nullptr;
  g_default_busy = false;  // a staged build nobody finished went with the context
  g_default_cv.notify_all();
  pinned_pool().trim();
  return URCCO_OK;
}

void urcco_free_indicators(urcco_indicators* ind, int32_t n) {
  if (!ind) return;
  for (int32_t d = 0; d < n; ++d) {
    for (void* p : {(void*)ind[d].row_ptr, (void*)ind[d].col_idx, (void*)ind[d].llr})
      if (p && !pinned_pool().put(p)) free(p);
    memset(&ind[d], 0, sizeof(urcco_indicators));
  }
}

// the process-wide context, (re)created when the options that shape it change: first device, number of GPUs, row-rate mode
// AND the flags (a flag change on a warm context must not be dropped silently)
static int default_context(const urcco_options* options, urcco_context** out) {
  const int want_dev = options ? options->device : 0, want_n = options ? options->n_gpus : 0;
  const int want_mode = options ? options->row_rate_mode : URCCO_ROW_RATE_MAHOUT_INT_DIV;
  const int want_flags = options ? options->flags : 0;
  if (g_default_ctx && (g_default_device != want_dev || g_default_n_gpus != want_n || g_default_mode != want_mode || g_default_flags != want_flags)) {
    urcco_context_destroy(g_default_ctx);
    g_default_ctx = nullptr;
  }
  if (!g_default_ctx) {
    URC(urcco_context_create(options, nullptr, &g_default_ctx));
    g_default_device = want_dev; g_default_n_gpus = want_n; g_default_mode = want_mode; g_default_flags = want_flags;
  }
  *out = g_default_ctx;
  return URCCO_OK;
}

int urcco_cross_occurrence_stage(const urcco_dataset* datasets, int32_t n_datasets, int32_t random_seed, const urcco_options* options) {
  return guarded([&]() -> int {
    err_buf()[0] = 0;
    std::unique_lock<std::mutex> g(g_default_mu);
    // the same thread staging twice is a protocol error (waiting for itself would never end); another thread's build is waited for
    if (g_default_busy && g_default_owner == std::this_thread::get_id())
      return fail(URCCO_BAD_ARG, "urcco_cross_occurrence_stage: the previous staged build has not been finished");
    // (bounded: a thread that staged and then died or walked away must not hang every other caller for ever -- ADVICE r04.  The build
    // itself never takes that long; URCCO_STAGE_WAIT_S overrides the 600 s)
    const char* we = getenv("URCCO_STAGE_WAIT_S");
    const long wait_s = we && *we ? atol(we) : 600;
    if (!g_default_cv.wait_for(g, std::chrono::seconds(wait_s > 0 ? wait_s : 1), [] { return !g_default_busy; }))
      return fail(URCCO_BUSY, "urcco_cross_occurrence_stage: another thread's staged build was not finished within %ld s (its owner finishes or cancels it; urcco_cross_occurrence_cancel_any discards it for an owner that is gone)", wait_s);
    urcco_context* c = nullptr;
    URC(default_context(options, &c));
    const int st = urcco_context_stage(c, datasets, n_datasets, random_seed);
    g_default_busy = st == URCCO_OK;  // released by the matching urcco_cross_occurrence_finish
    if (g_default_busy) g_default_owner = std::this_thread::get_id();
    return st;
  });
}

int urcco_cross_occurrence_finish(urcco_indicators* out, int32_t n_datasets, urcco_dataset_stats* stats) {
  if (out && n_datasets > 0) memset(out, 0, sizeof(urcco_indicators) * (size_t)n_datasets);
  return guarded([&]() -> int {
    err_buf()[0] = 0;
    std::lock_guard<std::mutex> g(g_default_mu);
    if (!g_default_ctx || !g_default_ctx->pending) {
      // nothing to hand out; if the flag is still up (the context went away under a staged build) the waiting threads must not wait for ever
      if (g_default_busy) { g_default_busy = false; g_default_cv.notify_all(); }
      return fail(URCCO_BAD_ARG, "urcco_cross_occurrence_finish: nothing staged");
    }
    if (g_default_ctx->pending->n_ds != n_datasets) {
      // the caller cannot take what was staged: the build is DISCARDED (finished into scratch blocks that go straight back to the pool) and
      // the default context is free again -- an early return here used to leave it occupied for good
      const int staged = g_default_ctx->pending->n_ds;
      discard_pending(g_default_ctx);
      g_default_busy = false;
      g_default_cv.notify_all();
      return fail(URCCO_BAD_ARG, "urcco_cross_occurrence_finish: %d datasets were staged, out holds %d (the staged build was discarded)", staged, n_datasets);
    }
    const int st = urcco_context_finish(g_default_ctx, out, stats);  // consumes the pending build whatever its outcome
    g_default_busy = false;
    g_default_cv.notify_all();
    return st;
  });
}

static int cancel_staged(bool any) {
  return guarded([&]() -> int {
    err_buf()[0] = 0;
    std::lock_guard<std::mutex> g(g_default_mu);
    // ownership: only the thread that staged may discard with _cancel -- e.g. a thread whose own _stage just timed out would
    // otherwise throw away the owner's build, and the owner's _finish would fail with "nothing staged"
    if (!any && g_default_busy && g_default_owner != std::this_thread::get_id())
      return fail(URCCO_BUSY, "urcco_cross_occurrence_cancel: the staged build belongs to another thread (urcco_cross_occurrence_cancel_any discards it regardless)");
    if (g_default_ctx && g_default_ctx->pending) discard_pending(g_default_ctx);
    g_default_busy = false;
    g_default_cv.notify_all();
    return URCCO_OK;
  });
}
int urcco_cross_occurrence_cancel(void) { return cancel_staged(false); }
int urcco_cross_occurrence_cancel_any(void) { return cancel_staged(true); }

int urcco_cross_occurrence_downsampled(const urcco_dataset* datasets, int32_t n_datasets, int32_t random_seed, const urcco_options* options,
                                       urcco_indicators* out, urcco_dataset_stats* stats) {
  // out[] is zeroed before any fallible step (context creation, argument checks): on failure only library-owned blocks are
  // ever released, whatever the caller's array held on entry
  if (out && n_datasets > 0) memset(out, 0, sizeof(urcco_indicators) * (size_t)n_datasets);
  if (!out) return fail(URCCO_BAD_ARG, "out is NULL");
  URC(urcco_cross_occurrence_stage(datasets, n_datasets, random_seed, options));
  return urcco_cross_occurrence_finish(out, n_datasets, stats);
}

int urcco_cooccurrences_idss(const urcco_csr* datasets, int32_t n_datasets, int32_t random_seed, int32_t max_interesting_items_per_thing,
                             int32_t max_num_interactions, const urcco_options* options, urcco_indicators* out, urcco_dataset_stats* stats) {
  if (out && n_datasets > 0) memset(out, 0, sizeof(urcco_indicators) * (size_t)n_datasets);
  return guarded([&]() -> int {
    err_buf()[0] = 0;
    if (!datasets || n_datasets <= 0) return fail(URCCO_BAD_ARG, "datasets is NULL or n_datasets <= 0");
    std::vector<urcco_dataset> ds((size_t)n_datasets);
    for (int d = 0; d < n_datasets; ++d) {
      ds[(size_t)d].matrix = datasets[d];
      ds[(size_t)d].max_elements_per_row = max_num_interactions;
      ds[(size_t)d].max_interesting_elements = max_interesting_items_per_thing;
      ds[(size_t)d].min_llr = 0.0;
      ds[(size_t)d].has_min_llr = 0;
      ds[(size_t)d].reserved = 0;
    }
    return urcco_cross_occurrence_downsampled(ds.data(), n_datasets, random_seed, options, out, stats);
  });
}

}  // extern "C"
