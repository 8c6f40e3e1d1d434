// Scattered-access microbenchmarks behind the round-4 designs of the CSR row scan and the SpGEMM:
//   gather_u8 / gather_u32   per-lane random loads from a table in global memory, with a fraction of the lanes active per
//                            load instruction: is the cost of a gather per INSTRUCTION or per ACTIVE LANE?
//   lds_u8 / lds_u32         the same lookups against a table in LDS
//   rows48                   random B'-row reads (48-byte rows of a 1 GiB array): a known byte count to calibrate FETCH_SIZE on
//                            (rocprofv3 --pmc FETCH_SIZE -- gather_microbench rows48), MI355X_MICROARCH.md "HBM"
// hipcc --offload-arch=gfx950 -O3 -o tools/_build/gather_microbench tools/gather_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// Every thread: ITERS rounds of K loads; load q of a round is issued only by lanes whose coin for (round, q) is below `active256`.
// the same with every lane of a wave inside a window of `window` consecutive entries (window = 1: one address for the whole wave):
// what the SpGEMM's score phase does when it reads xLogX at k11 (a handful of small integers) and at cA - k11
template <typename T, int K>
__global__ __launch_bounds__(256) void window_kernel(const T* __restrict__ tab, unsigned mask, int iters, unsigned window, unsigned long long* __restrict__ out) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  unsigned s = mix(gid * 2654435761u + 12345u);
  unsigned sw = mix((gid >> 6) * 40503u + 7u);  // per wave
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
      s = s * 1664525u + 1013904223u;
      sw = sw * 1664525u + 1013904223u;
      v[q] = (unsigned)tab[((mix(sw) & mask) & ~(window - 1u)) + (mix(s) & (window - 1u))];
    }
#pragma unroll
    for (int q = 0; q < K; ++q) acc += v[q];
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}

template <typename T, int K>
__global__ __launch_bounds__(256) void gather_kernel(const T* __restrict__ tab, unsigned mask, int iters, unsigned active256, unsigned long long* __restrict__ out) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  unsigned s = mix(gid * 2654435761u + 12345u);
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned idx[K];
    bool on[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = mix(s);
      idx[q] = r & mask;
      on[q] = ((r >> 24) & 255u) < active256;
    }
    unsigned v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = on[q] ? (unsigned)tab[idx[q]] : 0u;
#pragma unroll
    for (int q = 0; q < K; ++q) acc += v[q];
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}

// The same against LDS: the block first copies `lds_bytes` of the table into LDS.
template <typename T, int K>
__global__ __launch_bounds__(1024) void lds_kernel(const T* __restrict__ tab, int lds_elems, int iters, unsigned long long* __restrict__ out) {
  extern __shared__ unsigned char s_raw[];
  T* s_tab = reinterpret_cast<T*>(s_raw);
  for (int i = threadIdx.x; i < lds_elems; i += blockDim.x) s_tab[i] = tab[i];
  __syncthreads();
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = mix(gid * 2654435761u + 12345u);
  const unsigned mask = (unsigned)lds_elems - 1u;
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
      s = s * 1664525u + 1013904223u;
      v[q] = (unsigned)s_tab[mix(s) & mask];
    }
#pragma unroll
    for (int q = 0; q < K; ++q) acc += v[q];
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}

// Control: the index arithmetic alone.
template <int K>
__global__ __launch_bounds__(256) void alu_kernel(int iters, unsigned long long* __restrict__ out) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  unsigned s = mix(gid * 2654435761u + 12345u);
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < K; ++q) {
      s = s * 1664525u + 1013904223u;
      acc += mix(s) & 0xffffu;
    }
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}

// rows48: `lanes_per_row` lanes read one row of `row_words` consecutive 4-byte words starting at a random row of the array
// (rows are row_words * 4 bytes apart: a 48-byte row straddles a 128-byte line 3 times in 8).
__global__ __launch_bounds__(256) void rows_kernel(const int* __restrict__ arr, unsigned n_rows, int row_words, int lanes_log2, int rows_per_group, unsigned long long* __restrict__ out) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  const unsigned grp = gid >> lanes_log2, gl = gid & ((1u << lanes_log2) - 1u);
  unsigned long long acc = 0;
  for (int r = 0; r < rows_per_group; ++r) {
    const unsigned row = mix(grp * 2654435761u + (unsigned)r * 40503u + 7u) % n_rows;
    const int* p = arr + (size_t)row * (size_t)row_words;
    for (int w = (int)gl; w < row_words; w += 1 << lanes_log2) acc += (unsigned)p[w];
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}

// rows_dep: as rows_kernel with one lane per row, but every load's address depends on the value of the previous one (the array
// holds zeros): the row is walked one word after the other, each load issued when its predecessor has returned -- the SpGEMM's
// expand loop (load a column index, insert it, load the next).  rows_kernel issues its 12 loads back to back: they all miss.
__global__ __launch_bounds__(256) void rows_dep_kernel(const int* __restrict__ arr, unsigned n_rows, int row_words, int rows_per_lane, unsigned long long* __restrict__ out) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  unsigned long long acc = 0;
  for (int r = 0; r < rows_per_lane; ++r) {
    const unsigned row = mix(gid * 2654435761u + (unsigned)r * 40503u + 7u) % n_rows;
    const int* p = arr + (size_t)row * (size_t)row_words;
    int v = 0;
    for (int w = 0; w < row_words; ++w) {
      v = p[w + v];
      acc += (unsigned)v;
    }
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}

// rowsfit: rows_kernel over three placements of the SAME number of 48-byte rows -- dense (a row straddles a 128-byte line 11 times
// in 32), no line straddle (every row inside a 2x slack region, moved up to the next line start when it would cross one: what a
// "loose" B' layout computable from the ordinary prefix sum would give), and 64-byte slots (no straddle of 64-byte sectors either).
// Is the unit of the random-row cost the 128-byte line, the 64-byte sector, or the request?
__global__ __launch_bounds__(256) void rows_layout_kernel(const int* __restrict__ arr, unsigned n_rows, int row_words, int lanes_log2, int rows_per_group, int layout, unsigned long long* __restrict__ out) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  const unsigned grp = gid >> lanes_log2, gl = gid & ((1u << lanes_log2) - 1u);
  unsigned long long acc = 0;
  for (int r = 0; r < rows_per_group; ++r) {
    const unsigned row = mix(grp * 2654435761u + (unsigned)r * 40503u + 7u) % n_rows;
    size_t start;
    if (layout == 0) {
      start = (size_t)row * (size_t)row_words;
    } else if (layout == 1) {
      const size_t s2 = 2 * (size_t)row * (size_t)row_words;
      start = ((s2 & 31) + (size_t)row_words <= 32) ? s2 : ((s2 + 31) & ~(size_t)31);
    } else {
      start = (size_t)row * 16;
    }
    const int* p = arr + start;
    for (int w = (int)gl; w < row_words; w += 1 << lanes_log2) acc += (unsigned)p[w];
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}

// stream: the wide coalesced read (16 B per lane) the guide's FETCH_SIZE correction is stated for -- the reference point of the calibration
__global__ __launch_bounds__(256) void stream_kernel(const int4* __restrict__ arr, size_t n_vec, unsigned long long* __restrict__ out) {
  unsigned acc = 0;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * 256) {
    const int4 x = arr[v];
    acc += (unsigned)(x.x ^ x.y ^ x.z ^ x.w);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv) {
  const std::string only = argc > 1 ? argv[1] : "all";
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  unsigned long long* out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = n_cu * 8 * 4;  // 8 blocks of 256 per CU resident, four rounds
  constexpr int K = 8;
  const int iters = 64;
  const double loads = (double)blocks * 256 * iters * K;

  if (only == "all" || only == "alu") {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(alu_kernel<K>, dim3(blocks), dim3(256), 0, 0, iters, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      if (rep) printf("{\"test\": \"alu_only\", \"ms\": %.4f, \"G_slots_per_s\": %.1f}\n", time_ms(e0, e1), loads / time_ms(e0, e1) / 1e6);
    }
  }
  if (only == "all" || only == "gather") {
    for (size_t tab_bytes : {(size_t)256 << 10, (size_t)2 << 20, (size_t)16 << 20}) {
      unsigned char* tab;
      CK(hipMalloc(&tab, tab_bytes));
      CK(hipMemset(tab, 1, tab_bytes));
      for (int width : {1, 4}) {
        for (unsigned act : {256u, 128u, 64u, 16u, 4u}) {
          float ms = 0;
          for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (width == 1)
              hipLaunchKernelGGL((gather_kernel<unsigned char, K>), dim3(blocks), dim3(256), 0, 0, tab, (unsigned)(tab_bytes - 1), iters, act, out);
            else
              hipLaunchKernelGGL((gather_kernel<unsigned, K>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<unsigned*>(tab), (unsigned)(tab_bytes / 4 - 1), iters, act, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            ms = time_ms(e0, e1);
          }
          printf("{\"test\": \"gather_u%d\", \"table_KB\": %zu, \"active_frac\": %.4f, \"ms\": %.4f, \"G_slots_per_s\": %.1f, \"G_active_lookups_per_s\": %.1f, \"cycles_per_wave_instr_per_cu\": %.1f}\n",
                 width * 8, tab_bytes >> 10, act / 256.0, ms, loads / ms / 1e6, loads * (act / 256.0) / ms / 1e6, ms * 1e-3 * 2.4e9 * n_cu / (loads / 64));
        }
      }
      CK(hipFree(tab));
    }
  }
  if (only == "all" || only == "window") {
    double* tab;
    const size_t n = 4096;  // the xLogX table: 32 KB of doubles
    CK(hipMalloc(&tab, n * 8));
    CK(hipMemset(tab, 0, n * 8));
    for (unsigned window : {1u, 4u, 16u, 64u, 512u, 4096u}) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((window_kernel<double, K>), dim3(blocks), dim3(256), 0, 0, tab, (unsigned)(n - 1), iters, window, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        ms = time_ms(e0, e1);
      }
      printf("{\"test\": \"window_f64_32KB_table\", \"lanes_within_entries\": %u, \"ms\": %.4f, \"G_lookups_per_s\": %.1f, \"cycles_per_wave_instr_per_cu\": %.1f}\n", window, ms,
             loads / ms / 1e6, ms * 1e-3 * 2.4e9 * n_cu / (loads / 64));
    }
    CK(hipFree(tab));
  }
  if (only == "all" || only == "lds") {
    unsigned char* tab;
    CK(hipMalloc(&tab, 128 << 10));
    CK(hipMemset(tab, 1, 128 << 10));
    for (int lds_kb : {32, 64, 128}) {
      for (int width : {1, 4}) {
        const int threads = 1024;
        const int lblocks = n_cu * 4;
        const int liters = 256;
        const double lloads = (double)lblocks * threads * liters * K;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          if (width == 1) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_kernel<unsigned char, K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb << 10));
            hipLaunchKernelGGL((lds_kernel<unsigned char, K>), dim3(lblocks), dim3(threads), (size_t)lds_kb << 10, 0, tab, lds_kb << 10, liters, out);
          } else {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_kernel<unsigned, K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb << 10));
            hipLaunchKernelGGL((lds_kernel<unsigned, K>), dim3(lblocks), dim3(threads), (size_t)lds_kb << 10, 0, reinterpret_cast<unsigned*>(tab), (lds_kb << 10) / 4, liters, out);
          }
          CK(hipGetLastError());
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          ms = time_ms(e0, e1);
        }
        printf("{\"test\": \"lds_u%d\", \"lds_KB\": %d, \"threads\": %d, \"ms\": %.4f, \"G_lookups_per_s\": %.1f, \"cycles_per_wave_instr_per_cu\": %.1f}\n", width * 8, lds_kb, threads, ms,
               lloads / ms / 1e6, ms * 1e-3 * 2.4e9 * n_cu / (lloads / 64));
      }
    }
    CK(hipFree(tab));
  }
  if (only == "all" || only == "stream") {
    const size_t bytes = (size_t)1 << 30;
    int4* arr;
    CK(hipMalloc(&arr, bytes));
    CK(hipMemset(arr, 1, bytes));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(stream_kernel, dim3(n_cu * 16), dim3(256), 0, 0, arr, bytes / 16, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      ms = time_ms(e0, e1);
    }
    printf("{\"test\": \"stream\", \"algorithmic_bytes_per_launch\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", bytes, ms, bytes / ms / 1e6);
    CK(hipFree(arr));
  }
  if (only == "all" || only == "rows48dep") {
    const size_t bytes = (size_t)1 << 30;
    int* arr;
    CK(hipMalloc(&arr, bytes));
    CK(hipMemset(arr, 0, bytes));
    const int row_words = 12, rows_per_lane = 16;
    const unsigned n_rows = (unsigned)(bytes / (size_t)(row_words * 4));
    const int rblocks = n_cu * 8 * 8;
    const double rows = (double)rblocks * 256 * rows_per_lane;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(rows_dep_kernel, dim3(rblocks), dim3(256), 0, 0, arr, n_rows, row_words, rows_per_lane, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      ms = time_ms(e0, e1);
    }
    printf("{\"test\": \"rows_dep\", \"row_bytes\": 48, \"lanes_per_row\": 1, \"rows\": %.0f, \"algorithmic_bytes_per_launch\": %.0f, \"ms\": %.4f, \"M_rows_per_s\": %.1f, \"alg_GBps\": %.1f}\n",
           rows, rows * row_words * 4, ms, rows / ms / 1e3, rows * row_words * 4 / ms / 1e6);
    CK(hipFree(arr));
  }
  if (only == "all" || only == "rows48" || only == "rows64" || only == "rows128") {
    const size_t bytes = (size_t)1 << 30;
    int* arr;
    CK(hipMalloc(&arr, bytes));
    CK(hipMemset(arr, 1, bytes));
    for (int row_words : {12, 16, 32}) {
      if (only == "rows48" && row_words != 12) continue;
      if (only == "rows64" && row_words != 16) continue;
      if (only == "rows128" && row_words != 32) continue;
      for (int lanes_log2 : {0, 2, 4}) {
        if (only != "all" && lanes_log2 != 0) continue;
        const unsigned n_rows = (unsigned)(bytes / (size_t)(row_words * 4));
        const int rows_per_group = 16;
        const int rblocks = n_cu * 8 * 8;
        const double rows = (double)rblocks * 256 / (1 << lanes_log2) * rows_per_group;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(rows_kernel, dim3(rblocks), dim3(256), 0, 0, arr, n_rows, row_words, lanes_log2, rows_per_group, out);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          ms = time_ms(e0, e1);
        }
        printf("{\"test\": \"rows\", \"row_bytes\": %d, \"lanes_per_row\": %d, \"rows\": %.0f, \"algorithmic_bytes_per_launch\": %.0f, \"ms\": %.4f, \"M_rows_per_s\": %.1f, \"alg_GBps\": %.1f}\n",
               row_words * 4, 1 << lanes_log2, rows, rows * row_words * 4, ms, rows / ms / 1e3, rows * row_words * 4 / ms / 1e6);
      }
    }
    CK(hipFree(arr));
  }
  if (only == "all" || only == "rowsfit") {
    const size_t bytes = (size_t)2 << 30;
    int* arr;
    CK(hipMalloc(&arr, bytes));
    CK(hipMemset(arr, 1, bytes));
    const int row_words = 12;
    const unsigned n_rows = (unsigned)(((size_t)1 << 30) / (size_t)(row_words * 4));
    for (int lanes_log2 : {0, 2}) {
      for (int layout : {0, 1, 2}) {
        const int rows_per_group = 16;
        const int rblocks = n_cu * 8 * (lanes_log2 ? 32 : 8);
        const double rows = (double)rblocks * 256 / (1 << lanes_log2) * rows_per_group;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(rows_layout_kernel, dim3(rblocks), dim3(256), 0, 0, arr, n_rows, row_words, lanes_log2, rows_per_group, layout, out);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          ms = time_ms(e0, e1);
        }
        printf("{\"test\": \"rowsfit\", \"row_bytes\": 48, \"layout\": \"%s\", \"lanes_per_row\": %d, \"rows\": %.0f, \"ms\": %.4f, \"M_rows_per_s\": %.1f, \"alg_GBps\": %.1f}\n",
               layout == 0 ? "dense" : (layout == 1 ? "no_line_straddle_2x_slack" : "64B_slots"), 1 << lanes_log2, rows, ms, rows / ms / 1e3, rows * row_words * 4 / ms / 1e6);
      }
    }
    CK(hipFree(arr));
  }
  return 0;
}
