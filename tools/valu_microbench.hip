// VALU issue-rate microbenchmark for gfx950 (MI355X): how many cycles a wave64 vector instruction occupies its SIMD.
// VERDICT r02 asked for a MEASURED issue ceiling (DESIGN assumed 4 cycles per wave64 VALU instruction, the microarchitecture
// guide says 2 on CDNA4's SIMD-32) before any SpGEMM class is called issue-bound.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_microbench.hip -o gpurun_out/valu_microbench && gpurun_out/valu_microbench > out.json
//
// Every kernel runs ITER x UNROLL copies of one instruction per lane in inline asm (the compiler cannot fold them); the grid
// puts W waves on every SIMD of every CU (blocks of 256 threads = one wave per SIMD, W blocks per CU).  Reported per
// instruction kind and W: wave-instructions per second (chip), and cycles per wave-instruction per SIMD at the clock the
// run sustained (measured with s_memtime-free wall clock: cycles = SIMDs x clock x time / instructions).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string_view>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2048;
constexpr int UNROLL = 32;

// dependent chain: every instruction reads the previous result (latency-bound with one wave, hidden with more)
#define KERNEL_DEP(name, ASM)                                                                   \
  __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed) {                   \
    unsigned x = threadIdx.x + seed, y = seed | 1u;                                             \
    for (int i = 0; i < ITER; ++i) {                                                            \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) asm volatile(ASM : "+v"(x) : "v"(y)); \
    }                                                                                           \
    if (x == 0x12345u) out[0] = x;                                                              \
  }
// four independent chains
#define KERNEL_ILP(name, ASM)                                                                   \
  __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed) {                   \
    unsigned x0 = threadIdx.x + seed, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = seed | 1u;     \
    for (int i = 0; i < ITER; ++i) {                                                            \
      _Pragma("unroll") for (int u = 0; u < UNROLL / 4; ++u) {                                  \
        asm volatile(ASM : "+v"(x0) : "v"(y));                                                  \
        asm volatile(ASM : "+v"(x1) : "v"(y));                                                  \
        asm volatile(ASM : "+v"(x2) : "v"(y));                                                  \
        asm volatile(ASM : "+v"(x3) : "v"(y));                                                  \
      }                                                                                         \
    }                                                                                           \
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;                                           \
  }

KERNEL_DEP(k_add_dep, "v_add_u32 %0, %0, %1")
KERNEL_ILP(k_add_ilp, "v_add_u32 %0, %0, %1")
KERNEL_ILP(k_xor_ilp, "v_xor_b32 %0, %0, %1")
KERNEL_ILP(k_lshl_add_ilp, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL_ILP(k_mul_lo_ilp, "v_mul_lo_u32 %0, %0, %1")
KERNEL_ILP(k_mul_hi_ilp, "v_mul_hi_u32 %0, %0, %1")
KERNEL_ILP(k_alignbit_ilp, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL_ILP(k_bfe_ilp, "v_bfe_u32 %0, %0, 3, 20")
KERNEL_ILP(k_cndmask_ilp, "v_cndmask_b32 %0, %0, %1, vcc")


// --- selects, compares and carries: what compiled control flow is made of
#define KERNEL_RAW(name, SETUP, BODY4, ...)                                                  \
  __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed) {                   \
    unsigned x0 = threadIdx.x + seed, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = seed | 1u;     \
    SETUP;                                                                                      \
    for (int i = 0; i < ITER; ++i) {                                                            \
      _Pragma("unroll") for (int u = 0; u < UNROLL / 4; ++u) {                                  \
        asm volatile(BODY4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y) : __VA_ARGS__);    \
      }                                                                                         \
    }                                                                                           \
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;                                           \
  }
KERNEL_RAW(k_cndmask_e32_vcc, asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(x0), "v"(y) : "vcc"),
           "v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc", "memory")
KERNEL_RAW(k_cndmask_e64_sgpr, asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21"),
           "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]", "s20", "s21")
KERNEL_RAW(k_cmp_cnd_pair, (void)0,
           "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32_e32 %1, %1, %4, vcc", "vcc")
KERNEL_RAW(k_cmp_u32_sgpr, (void)0,
           "v_cmp_lt_u32 s[20:21], %0, %4\n v_cmp_lt_u32 s[22:23], %1, %4\n v_cmp_lt_u32 s[24:25], %2, %4\n v_cmp_lt_u32 s[26:27], %3, %4", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
KERNEL_RAW(k_addc_chain, (void)0,
           "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc", "vcc")
KERNEL_RAW(k_and_or_select, (void)0,   // branch-free select by mask arithmetic: x ^ ((x ^ y) & m)
           "v_xor_b32 %1, %0, %4\n v_and_b32 %1, %1, %2\n v_xor_b32 %0, %0, %1\n v_xor_b32 %3, %3, %4", "memory")
KERNEL_RAW(k_bfi_select, (void)0,      // v_bfi_b32: one-instruction bitwise select
           "v_bfi_b32 %0, %2, %4, %0\n v_bfi_b32 %1, %2, %4, %1\n v_bfi_b32 %3, %2, %4, %3\n v_bfi_b32 %0, %2, %4, %0", "memory")
KERNEL_RAW(k_mbcnt, (void)0,
           "v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_hi_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_hi_u32_b32 %3, -1, %3", "memory")
KERNEL_RAW(k_dpp_row_shr, (void)0,
           "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
           " v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf", "memory")
KERNEL_RAW(k_readfirstlane, (void)0,
           "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3", "s20", "s21", "s22", "s23")
KERNEL_RAW(k_salu_add, (void)0,
           "s_add_u32 s20, s20, s21\n s_add_u32 s22, s22, s21\n s_add_u32 s24, s24, s21\n s_add_u32 s26, s26, s21", "s20", "s21", "s22", "s24", "s26", "scc")

// --- LDS: plain reads / writes / atomics, conflict-free (lane-indexed) and all lanes on one address
__global__ __launch_bounds__(256) void k_lds_read(unsigned* out, unsigned seed) {
  __shared__ unsigned s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i + seed;
  __syncthreads();
  unsigned a = threadIdx.x * 4, acc = 0;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(u * 1024 % 8192));
      acc += v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  if (acc == 0x12345u) out[0] = acc;
}
#define KERNEL_LDS_ATOMIC(name, ASM, SAME)                                                       \
  __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed) {                   \
    __shared__ unsigned s[4096];                                                                \
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0;                                     \
    __syncthreads();                                                                            \
    unsigned a = (SAME) ? (threadIdx.x / 64) * 256 : threadIdx.x * 4, one = 1u, acc = 0;        \
    for (int i = 0; i < ITER / 4; ++i) {                                                        \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                      \
        unsigned v;                                                                             \
        asm volatile(ASM : "=v"(v) : "v"(a), "v"(one));                                         \
        acc += v;                                                                               \
      }                                                                                         \
    }                                                                                           \
    if (acc == 0x12345u) out[0] = acc;                                                          \
  }
KERNEL_LDS_ATOMIC(k_lds_add_rtn, "ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)", false)
KERNEL_LDS_ATOMIC(k_lds_add_rtn_same_addr, "ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)", true)
__global__ __launch_bounds__(256) void k_lds_add_nortn(unsigned* out, unsigned seed) {
  __shared__ unsigned s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0;
  __syncthreads();
  unsigned a = threadIdx.x * 4, one = 1u;
  for (int i = 0; i < ITER / 4; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(one));
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  __syncthreads();
  if (s[threadIdx.x] == 0x12345u) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_lds_cas_rtn(unsigned* out, unsigned seed) {
  __shared__ unsigned s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0;
  __syncthreads();
  unsigned a = threadIdx.x * 4, zero = 0u, val = 7u, acc = 0;
  for (int i = 0; i < ITER / 4; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      unsigned v;
      asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "v"(zero), "v"(val));
      acc += v;
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}

// 64-bit integer multiply-add: v_mad_u64_u32 (what a 64 x 64 -> 64 multiply of the down-sampling hash is made of)
__global__ __launch_bounds__(256) void k_mad64_ilp(unsigned* out, unsigned seed) {
  unsigned long long a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  unsigned y = seed | 1u;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a0) : "v"(y) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a1) : "v"(y) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a2) : "v"(y) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a3) : "v"(y) : "vcc");
    }
  }
  if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345ull) out[0] = (unsigned)a0;
}
// fp64 fused multiply-add (the LLR arithmetic)
__global__ __launch_bounds__(256) void k_fma64_ilp(unsigned* out, unsigned seed) {
  double a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, y = 1.0000001;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(y));
      asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a1) : "v"(y));
      asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a2) : "v"(y));
      asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a3) : "v"(y));
    }
  }
  if (a0 + a1 + a2 + a3 == 0.12345) out[0] = 1;
}
// the down-sampling hash itself (cco_device.h hash53), four per iteration: its measured cost in SIMD cycles per entry
__device__ __forceinline__ unsigned long long hash53(unsigned seed, unsigned row, unsigned col) {
  unsigned long long x = ((unsigned long long)row << 32) | (unsigned long long)col;
  x ^= (unsigned long long)seed * 0x9E3779B97F4A7C15ull;
  x += 0x9E3779B97F4A7C15ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x >> 11;
}
constexpr int HASH_ITER = 4096;
__global__ __launch_bounds__(256) void k_hash53(unsigned* out, unsigned seed) {
  unsigned c = threadIdx.x * 2654435761u + seed, r = blockIdx.x;
  unsigned kept = 0;
  for (int i = 0; i < HASH_ITER; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      kept += hash53(seed, r, c) <= 0x000fffffffffffffull ? 1u : 0u;
      c += 0x9E3779B9u;
    }
  }
  if (kept == 0xffffffffu) out[0] = kept;
}

template <typename K>
static double run(K kern, int blocks, unsigned* d_out) {
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3.0 * 1e-3;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double clock_hz = prop.clockRate * 1e3;  // kHz -> Hz (the peak engine clock; the sustained clock may be lower)
  unsigned* d_out;
  CHECK(hipMalloc(&d_out, 64));
  struct Row { const char* name; double inst_per_thread; double (*fn)(int, unsigned*); };
#define ROW(k, n) Row{#k, (double)(n), [](int b, unsigned* o) { return run(k, b, o); }}
  std::vector<Row> rows = {ROW(k_add_dep, (double)ITER * UNROLL), ROW(k_add_ilp, (double)ITER * UNROLL), ROW(k_xor_ilp, (double)ITER * UNROLL),
                           ROW(k_lshl_add_ilp, (double)ITER * UNROLL), ROW(k_alignbit_ilp, (double)ITER * UNROLL), ROW(k_bfe_ilp, (double)ITER * UNROLL),
                           ROW(k_cndmask_ilp, (double)ITER * UNROLL), ROW(k_mul_lo_ilp, (double)ITER * UNROLL), ROW(k_mul_hi_ilp, (double)ITER * UNROLL),
                           ROW(k_cndmask_e32_vcc, (double)ITER * UNROLL), ROW(k_cndmask_e64_sgpr, (double)ITER * UNROLL), ROW(k_cmp_cnd_pair, (double)ITER * UNROLL),
                           ROW(k_cmp_u32_sgpr, (double)ITER * UNROLL), ROW(k_addc_chain, (double)ITER * UNROLL), ROW(k_and_or_select, (double)ITER * UNROLL),
                           ROW(k_bfi_select, (double)ITER * UNROLL), ROW(k_mbcnt, (double)ITER * UNROLL), ROW(k_dpp_row_shr, (double)ITER * UNROLL),
                           ROW(k_readfirstlane, (double)ITER * UNROLL), ROW(k_salu_add, (double)ITER * UNROLL),
                           ROW(k_lds_read, (double)ITER * UNROLL), ROW(k_lds_add_rtn, (double)ITER / 4 * UNROLL), ROW(k_lds_add_rtn_same_addr, (double)ITER / 4 * UNROLL),
                           ROW(k_lds_add_nortn, (double)ITER / 4 * UNROLL), ROW(k_lds_cas_rtn, (double)ITER / 4 * UNROLL),
                           ROW(k_mad64_ilp, (double)ITER * UNROLL), ROW(k_fma64_ilp, (double)ITER * UNROLL), ROW(k_hash53, (double)HASH_ITER * 4)};
  printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"n_cu\": %d, \"clock_MHz\": %.0f, \"iter_x_unroll\": %d,\n \"rows\": [\n", prop.name, prop.gcnArchName, n_cu, clock_hz / 1e6,
         ITER * UNROLL);
  double best_add = 0;
  bool first = true;
  for (const Row& r : rows)
    for (int w : {1, 2, 4, 8}) {
      const int blocks = n_cu * w;  // 256 threads = 4 waves = one wave per SIMD; w blocks per CU
      const double s = r.fn(blocks, d_out);
      const double wave_inst = (double)blocks * 4.0 * r.inst_per_thread;
      const double rate = wave_inst / s;
      const double cyc = (double)n_cu * 4.0 * clock_hz * s / wave_inst;
      if (std::string_view(r.name) == "k_add_ilp" && rate > best_add) best_add = rate;
      printf("%s  {\"kernel\": \"%s\", \"waves_per_simd\": %d, \"G_wave_ops_per_s\": %.1f, \"simd_cycles_per_wave_op_at_peak_clock\": %.2f}", first ? "" : ",\n", r.name, w,
             rate / 1e9, cyc);
      first = false;
    }
  printf("\n ],\n \"wave_valu_instructions_per_s_G\": %.1f, \"cycles_per_wave64_valu\": %.2f,\n", best_add / 1e9, (double)n_cu * 4.0 * clock_hz / best_add);
  printf(" \"note\": \"ops = one instruction per lane of a wave64 (k_hash53: one 64-bit splitmix hash + compare per lane); cycles at the PEAK clock of hipDeviceProp (a lower sustained clock shows as more cycles)\"}\n");
  return 0;
}
