// TEST-ONLY host simulator runtime (see include/hip/hip_runtime.h in this directory).  Never linked into the
// product library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <dlfcn.h>

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
  while (live > 0) {
    for (int t = 0; t < T; ++t) {
      Fiber& f = c->fibers[(size_t)t];
      if (f.state != RUNNABLE) continue;
      c->cur = t;
      tIdx.x = (unsigned)t % block.x;
      tIdx.y = ((unsigned)t / block.x) % block.y;
      tIdx.z = (unsigned)t / (block.x * block.y);
      hipsim_switch(&c->sched_sp, f.sp);
      if (f.state == DONE) --live;
    }
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
