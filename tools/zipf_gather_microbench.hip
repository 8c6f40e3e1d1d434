// What would an LDS-resident threshold cache save the CSR row scan?  
// The flags kernel gathers one threshold byte per interaction from a table of n_cols bytes; the columns follow a Zipf law, so the
// lanes of a wave are NOT uniformly scattered: the hot head repeats (same cache lines, L1 hits, merged lanes).  This benchmark
// draws a Zipf-1.0 column stream over 2^21 items (ids scattered by a bijective hash, as a random permutation would), remembers
// every entry's popularity RANK beside it, and times the flags kernel's access pattern (two runs of eight consecutive entries
// per thread, 16-byte loads) with the byte gather issued only for ranks in [lo, hi):
//     all             every entry gathers (today's kernel)
//     rank >= H       the top-H columns are served elsewhere (an LDS cache of H entries)
//     rank >= n_hot   only the cold columns (count <= 500) gather
//     none            no gather (the floor)
// hipcc --offload-arch=gfx950 -O3 -o tools/_build/zipf_gather_microbench tools/zipf_gather_microbench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LOG_ITEMS = 21;
constexpr unsigned N_ITEMS = 1u << LOG_ITEMS;

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned perm21(unsigned r) {  // bijection on [0, 2^21)
  r = (r * 0x9E3779B1u) & (N_ITEMS - 1);
  r ^= r >> 11;
  r = (r * 0x85EBCA6Bu) & (N_ITEMS - 1);
  r ^= r >> 9;
  r = (r * 0xC2B2AE35u) & (N_ITEMS - 1);
  return r;
}

__global__ void gen_kernel(const double* __restrict__ cdf, long long n, int* __restrict__ col, int* __restrict__ rank) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const unsigned a = mix((unsigned)e * 2654435761u + 99u), b = mix(a ^ (unsigned)(e >> 32) ^ 0x5bd1e995u);
    const double u = ((double)a * 4294967296.0 + (double)b) * (1.0 / 18446744073709551616.0);
    unsigned lo = 0, hi = N_ITEMS - 1;
    while (lo < hi) {
      const unsigned mid = (lo + hi) >> 1;
      if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    rank[e] = (int)lo;
    col[e] = (int)perm21(lo);
  }
}

// flags-kernel access pattern: tiles of 4096 entries, 256 threads, two runs of eight entries per thread
__global__ __launch_bounds__(256) void scan_kernel(const int* __restrict__ col, const int* __restrict__ rank, const unsigned char* __restrict__ thr8, int lo, int hi,
                                                   int with_rank, unsigned long long* __restrict__ out) {
  const long long e0 = (long long)blockIdx.x * 4096;
  unsigned acc = 0;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const long long e = e0 + (long long)(g * 256 + (int)threadIdx.x) * 8;
    const int4 x = *reinterpret_cast<const int4*>(col + e), y = *reinterpret_cast<const int4*>(col + e + 4);
    const int c[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
    int r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (with_rank) {
      const int4 p = *reinterpret_cast<const int4*>(rank + e), q = *reinterpret_cast<const int4*>(rank + e + 4);
      r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = p.w; r[4] = q.x; r[5] = q.y; r[6] = q.z; r[7] = q.w;
    }
    unsigned v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (r[k] >= lo && r[k] < hi) ? (unsigned)thr8[c[k]] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k] + (unsigned)c[k];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const long long n = 256ll << 20;  // 1 GiB of column indices
  std::vector<double> cdf(N_ITEMS);
  double s = 0;
  for (unsigned i = 0; i < N_ITEMS; ++i) { s += 1.0 / (double)(i + 1); cdf[i] = s; }
  for (unsigned i = 0; i < N_ITEMS; ++i) cdf[i] /= s;
  double* d_cdf; int *col, *rank; unsigned char* thr8; unsigned long long* out;
  CK(hipMalloc(&d_cdf, sizeof(double) * N_ITEMS));
  CK(hipMemcpy(d_cdf, cdf.data(), sizeof(double) * N_ITEMS, hipMemcpyHostToDevice));
  CK(hipMalloc(&col, n * 4)); CK(hipMalloc(&rank, n * 4)); CK(hipMalloc(&thr8, N_ITEMS)); CK(hipMalloc(&out, 64));
  CK(hipMemset(thr8, 7, N_ITEMS));
  hipLaunchKernelGGL(gen_kernel, dim3(256 * 32), dim3(256), 0, 0, d_cdf, n, col, rank);
  CK(hipDeviceSynchronize());
  // hot = count > 500 under this stream: count(rank r) = n / (H_N (r + 1))
  const int n_hot = (int)((double)n / (s * 500.0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { const char* name; int lo, hi, with_rank; };
  const Case cases[] = {{"none (floor, no rank stream)", 1, 0, 0}, {"none (floor, with rank stream)", 1 << 30, 1 << 30, 1}, {"all", 0, 1 << 30, 1}, {"rank >= 1K", 1024, 1 << 30, 1},
                        {"rank >= 4K", 4096, 1 << 30, 1}, {"rank >= 16K", 16384, 1 << 30, 1}, {"rank >= n_hot (cold only)", n_hot, 1 << 30, 1},
                        {"rank in [4K, n_hot) (hot but not cached)", 4096, n_hot, 1}, {"rank in [16K, n_hot)", 16384, n_hot, 1}, {"rank < 4K (head only)", 0, 4096, 1}};
  printf("{\"entries\": %lld, \"n_items\": %u, \"n_hot\": %d}\n", n, N_ITEMS, n_hot);
  for (const Case& c : cases) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(scan_kernel, dim3((unsigned)(n / 4096)), dim3(256), 0, 0, col, rank, thr8, c.lo, c.hi, c.with_rank, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    // fraction of entries that gather: sum over ranks in [lo, hi) of 1/(r+1) / H_N
    double f = 0;
    if (c.hi > c.lo) {
      const int hi = c.hi > (int)N_ITEMS ? (int)N_ITEMS : c.hi;
      f = (cdf[hi - 1] - (c.lo > 0 ? cdf[c.lo - 1] : 0.0));
    }
    printf("{\"case\": \"%s\", \"ms\": %.4f, \"gathering_fraction\": %.4f, \"G_entries_per_s\": %.1f}\n", c.name, ms, f, (double)n / ms / 1e6);
  }
  return 0;
}
