// TEST-ONLY host simulator runtime (see include/hip/hip_runtime.h in this directory).  Never linked into the
// product library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <dlfcn.h>
#include <sys/mman.h>
#include <mutex>
#include <unordered_map>

#if !defined(__x86_64__)
#error "hostsim context switch is written for x86-64"
#endif

extern "C" void hipsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipsim_switch
.type hipsim_switch,@function
hipsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipsim_switch,.-hipsim_switch
)");

namespace hipsim {

thread_local Idx tIdx, bIdx, bDim, gDim;
thread_local int cur_device = 0;
thread_local int last_error = 0;

// ---- guard-page allocations (HIPSIM_GUARD=1) ------------------------------------------------------------
bool guard_on() {
  static const bool on = [] { const char* e = getenv("HIPSIM_GUARD"); return e && *e && *e != '0'; }();
  return on;
}
namespace {
std::mutex g_guard_mu;
std::unordered_map<void*, std::pair<void*, size_t>> g_guard;  // user pointer -> (mapping, length)
}  // namespace
namespace {
size_t guard_align() {  // HIPSIM_GUARD=2: buffers end to 4 bytes at the guard page (their starts are then 4-aligned only: scalar paths)
  static const size_t a = [] { const char* e = getenv("HIPSIM_GUARD"); return (e && *e == '2') ? (size_t)4 : (size_t)16; }();
  return a;
}
}  // namespace
namespace {
bool guard_before() {  // HIPSIM_GUARD=3: buffers START at a page boundary behind a PROT_NONE page (buffer[-1], tile_sums[t - 1] at t = 0, ...)
  static const bool b = [] { const char* e = getenv("HIPSIM_GUARD"); return e && *e == '3'; }();
  return b;
}
}  // namespace
void* guard_alloc(size_t n) {
  if (guard_before()) {
    const size_t pages = (n + GUARD_PAGE - 1) & ~(GUARD_PAGE - 1);
    char* m = static_cast<char*>(mmap(nullptr, pages + GUARD_PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (m == MAP_FAILED) return nullptr;
    mprotect(m, GUARD_PAGE, PROT_NONE);
    memset(m + GUARD_PAGE, 0x7f, pages);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[m + GUARD_PAGE] = {m, pages + GUARD_PAGE};
    return m + GUARD_PAGE;
  }
  const size_t body = (n + guard_align() - 1) & ~(guard_align() - 1);
  const size_t pages = (body + GUARD_PAGE - 1) & ~(GUARD_PAGE - 1);
  char* m = static_cast<char*>(mmap(nullptr, pages + GUARD_PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  if (m == MAP_FAILED) return nullptr;
  mprotect(m + pages, GUARD_PAGE, PROT_NONE);
  char* p = m + pages - body;
  memset(m, 0xA5, pages - body);  // whatever precedes the buffer is not zero either
  // fresh device memory is NOT zero on hardware (it may hold whatever this or another process freed): ints of 0x7f7f7f7f index far
  // out of any buffer, so a kernel that uses unwritten memory as an index or a length faults here instead of once in a while there
  memset(p, 0x7f, body);
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guard[p] = {m, pages + GUARD_PAGE};
  return p;
}
bool guard_free(void* p) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  auto it = g_guard.find(p);
  if (it == g_guard.end()) return false;
  munmap(it->second.first, it->second.second);
  g_guard.erase(it);
  return true;
}
void arena_unguard(void* base, size_t cap) {
  if (!base || !cap) return;
  char* b = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + GUARD_PAGE - 1) & ~(uintptr_t)(GUARD_PAGE - 1));
  char* e = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + cap) & ~(uintptr_t)(GUARD_PAGE - 1));
  if (e > b) mprotect(b, (size_t)(e - b), PROT_READ | PROT_WRITE);
}
char* arena_place(char* base, size_t off, size_t bytes, size_t* new_off) {
  if (guard_before()) {
    const uintptr_t guard = (reinterpret_cast<uintptr_t>(base) + off + GUARD_PAGE - 1) & ~(uintptr_t)(GUARD_PAGE - 1);
    mprotect(reinterpret_cast<void*>(guard), GUARD_PAGE, PROT_NONE);
    char* p = reinterpret_cast<char*>(guard + GUARD_PAGE);
    const size_t rounded = (bytes + 15) & ~(size_t)15;
    memset(p, 0x7f, rounded);
    *new_off = (size_t)(guard + GUARD_PAGE + rounded - reinterpret_cast<uintptr_t>(base));
    return p;
  }
  const size_t body = (bytes + guard_align() - 1) & ~(guard_align() - 1);
  const uintptr_t lo = reinterpret_cast<uintptr_t>(base) + off;
  const uintptr_t end = (lo + body + GUARD_PAGE - 1) & ~(uintptr_t)(GUARD_PAGE - 1);
  mprotect(reinterpret_cast<void*>(end), GUARD_PAGE, PROT_NONE);
  *new_off = (size_t)(end + GUARD_PAGE - reinterpret_cast<uintptr_t>(base));
  memset(reinterpret_cast<char*>(end - body), 0x7f, body);  // scratch handed out again: poisoned like fresh memory (see guard_alloc)
  return reinterpret_cast<char*>(end - body);
}

namespace {

enum State { RUNNABLE = 0, AT_BARRIER = 1, AT_WAVE = 2, DONE = 3 };
constexpr size_t STACK_BYTES = 256 * 1024;

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  int state = DONE;
  int op = 0;
  int arg = 0;
  uint64_t payload = 0;
  uint64_t result = 0;
  const void* site = nullptr;
};

struct BlockCtx {
  std::vector<Fiber> fibers;
  void* sched_sp = nullptr;
  int cur = -1;
  int nthreads = 0;
  const std::function<void()>* body = nullptr;
};

thread_local BlockCtx* g_ctx = nullptr;

void fiber_entry() {
  BlockCtx* c = g_ctx;
  (*c->body)();
  c = g_ctx;
  Fiber& f = c->fibers[(size_t)c->cur];
  f.state = DONE;
  hipsim_switch(&f.sp, c->sched_sp);
  abort();  // never resumed
}

void park(int state) {
  BlockCtx* c = g_ctx;
  Fiber& f = c->fibers[(size_t)c->cur];
  f.state = state;
  hipsim_switch(&f.sp, c->sched_sp);
}

void prepare(Fiber& f) {
  if (!f.stack) f.stack = (char*)aligned_alloc(64, STACK_BYTES);
  uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                 // fake return address of fiber_entry (keeps the ABI alignment)
  *--sp = (void*)&fiber_entry;     // `ret` target of the first switch
  for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
  f.sp = (void*)sp;
  f.state = RUNNABLE;
}

void resolve_wave(BlockCtx* c, int w0, int w1) {
  const void* site = nullptr;
  int op = 0;
  bool any = false;
  for (int t = w0; t < w1; ++t) {
    Fiber& f = c->fibers[(size_t)t];
    if (f.state != AT_WAVE) continue;
    if (!any) { site = f.site; op = f.op; any = true; }
    else if (f.site != site || f.op != op) {
      fprintf(stderr, "hipsim: lanes of one wave sit at different wave ops (divergent __shfl/__ballot) in block (%u) of %u threads, grid %u: lane %d at %p (op %d) vs %p (op %d)\n",
              bIdx.x, bDim.x, gDim.x, t - w0, f.site, f.op, site, op);
      for (int q = w0; q < w1; ++q) fprintf(stderr, " %d:%d:%p", q - w0, c->fibers[(size_t)q].state, c->fibers[(size_t)q].state == AT_WAVE ? c->fibers[(size_t)q].site : nullptr);
      Dl_info di;
      if (dladdr(site, &di)) fprintf(stderr, "\n offsets: %lx %lx in %s", (unsigned long)((const char*)site - (const char*)di.dli_fbase), (unsigned long)((const char*)f.site - (const char*)di.dli_fbase), di.dli_fname);
      fprintf(stderr, "\n");
      abort();
    }
  }
  if (!any) return;
  if (op == OP_BALLOT) {
    uint64_t mask = 0;
    for (int t = w0; t < w1; ++t)
      if (c->fibers[(size_t)t].state == AT_WAVE && c->fibers[(size_t)t].payload) mask |= 1ull << (t - w0);
    for (int t = w0; t < w1; ++t)
      if (c->fibers[(size_t)t].state == AT_WAVE) c->fibers[(size_t)t].result = mask;
  } else {
    for (int t = w0; t < w1; ++t) {
      Fiber& f = c->fibers[(size_t)t];
      if (f.state != AT_WAVE) continue;
      const int src = w0 + f.arg;
      if (src >= w0 && src < w1 && c->fibers[(size_t)src].state == AT_WAVE) f.result = c->fibers[(size_t)src].payload;
      else f.result = 0xDEADBEEFDEADBEEFull;  // reading an inactive lane: undefined on hardware
    }
  }
  for (int t = w0; t < w1; ++t)
    if (c->fibers[(size_t)t].state == AT_WAVE) c->fibers[(size_t)t].state = RUNNABLE;
}

void run_block(BlockCtx* c, dim3 block) {
  const int T = (int)(block.x * block.y * block.z);
  c->nthreads = T;
  if ((int)c->fibers.size() < T) c->fibers.resize((size_t)T);
  for (int t = 0; t < T; ++t) prepare(c->fibers[(size_t)t]);
  int live = T;
  // HIPSIM_ORDER: the order in which the WAVES of a block get their turn between two rendezvous -- any order is a schedule the hardware
  // may produce.  The default (ascending) hides a missing __syncthreads() whenever the producer is the lower wave, which is also the
  // order hardware mostly happens to run in: exactly the bug that survives ordinary runs.  "reverse" = descending, "rotate" = a
  // different first wave every round.  Lanes inside a wave keep their order (they run in lock step on hardware).
  static const int order_mode = [] {
    const char* e = getenv("HIPSIM_ORDER");
    return !e ? 0 : (!strcmp(e, "reverse") ? 1 : (!strcmp(e, "rotate") ? 2 : 0));
  }();
  const int n_waves = (T + 63) / 64;
  unsigned round = 0;
  while (live > 0) {
    for (int wi = 0; wi < n_waves; ++wi) {
      const int w = order_mode == 1 ? n_waves - 1 - wi : (order_mode == 2 ? (int)((wi + round * 7 + 1) % (unsigned)n_waves) : wi);
      for (int t = w * 64; t < T && t < (w + 1) * 64; ++t) {
        Fiber& f = c->fibers[(size_t)t];
        if (f.state != RUNNABLE) continue;
        c->cur = t;
        tIdx.x = (unsigned)t % block.x;
        tIdx.y = ((unsigned)t / block.x) % block.y;
        tIdx.z = (unsigned)t / (block.x * block.y);
        hipsim_switch(&c->sched_sp, f.sp);
        if (f.state == DONE) --live;
      }
    }
    ++round;
    if (live == 0) break;
    bool resolved = false;
    for (int w0 = 0; w0 < T; w0 += 64) {
      const int w1 = w0 + 64 < T ? w0 + 64 : T;
      bool any_wave = false;
      for (int t = w0; t < w1; ++t) any_wave |= c->fibers[(size_t)t].state == AT_WAVE;
      if (any_wave) {
        resolve_wave(c, w0, w1);
        resolved = true;
      }
    }
    if (resolved) continue;
    // nobody at a wave op: every live fiber is at the block barrier
    for (int t = 0; t < T; ++t)
      if (c->fibers[(size_t)t].state == AT_BARRIER) c->fibers[(size_t)t].state = RUNNABLE;
  }
}

}  // namespace

void sync_threads() { park(AT_BARRIER); }

int lane_id() { return g_ctx->cur & 63; }

uint64_t wave_op(int op, uint64_t payload, int arg, const void* site) {
  BlockCtx* c = g_ctx;
  Fiber& f = c->fibers[(size_t)c->cur];
  f.op = op;
  f.payload = payload;
  f.arg = arg;
  f.site = site;
  park(AT_WAVE);
  return g_ctx->fibers[(size_t)g_ctx->cur].result;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const long long nblocks = (long long)grid.x * grid.y * grid.z;
#pragma omp parallel
  {
    static thread_local BlockCtx ctx;
    g_ctx = &ctx;
    ctx.body = &body;
    bDim = Idx{block.x, block.y, block.z};
    gDim = Idx{grid.x, grid.y, grid.z};
#pragma omp for schedule(dynamic, 1)
    for (long long b = 0; b < nblocks; ++b) {
      bIdx.x = (unsigned)(b % grid.x);
      bIdx.y = (unsigned)((b / grid.x) % grid.y);
      bIdx.z = (unsigned)(b / ((long long)grid.x * grid.y));
      run_block(&ctx, block);
    }
  }
}

}  // namespace hipsim

// test-side allocations (tests/conftest.py wraps them as tensors for kernel inputs and outputs)
extern "C" void* hipsim_guard_malloc(size_t n) { return hipsim::guard_alloc(n ? n : 1); }
extern "C" void hipsim_guard_release(void* p) { if (p) (void)hipsim::guard_free(p); }
