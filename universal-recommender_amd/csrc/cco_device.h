// Device-side arithmetic shared by the CCO kernels: the down-sampling RNG and the log-likelihood ratio.
// gfx950 (CDNA4) only; compiled by hipcc with -ffp-contract=off so every fp64 operation is a single IEEE
// operation (no FMA contraction) and the LLR is reproducible bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urcco {

// Stateless down-sampling RNG (decision D10): uniform [0,1) double keyed by (seed,row,col).
// splitmix64 finaliser over the packed key; top 53 bits.
__device__ __forceinline__ unsigned long long hash53(uint32_t seed, uint32_t row, uint32_t col) {
  unsigned long long x = ((unsigned long long)row << 32) | (unsigned long long)col;
  x ^= (unsigned long long)seed * 0x9E3779B97F4A7C15ull;
  x += 0x9E3779B97F4A7C15ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x >> 11;
}
__device__ __forceinline__ double u01_hash(uint32_t seed, uint32_t row, uint32_t col) {
  return (double)hash53(seed, row, col) * (1.0 / 9007199254740992.0);
}

// The second form of decision D10 (URCCO_RNG_MIX32; the reference pins neither: Mahout draws from a per-partition java.util.Random):
// a 32-bit uniform keyed by (seed,row,col).  The row and the seed enter through one multiply-add each -- a key the row scan forms
// once per tile plus one multiply per entry --, the column by xor, and the two-round multiply-xorshift finaliser ("lowbias32", bias
// measured by its author at the level of the murmur3 finaliser) mixes the sum: 2 + 8 vector instructions per interaction where the
// 64-bit splitmix finaliser above costs ~25 on a machine without a 64-bit integer multiplier.  u01 = h * 2^-32.
// Limitation (documented in include/urcco.h): seed and row enter the key linearly, so another seed is the same stream with the rows
// shifted by a constant -- a relabeling, not an independent sample.  A non-linear row key would cost a multiply-xorshift round per ENTRY (the
// scan forms key0 + t * MIX32_ROW incrementally inside a tile), which is the saving this mode exists for; the default RNG has no such structure.
constexpr uint32_t MIX32_ROW = 0x9E3779B1u, MIX32_SEED = 0x85EBCA77u, MIX32_ADD = 0xC2B2AE3Du;
__device__ __forceinline__ uint32_t mix32_row_key(uint32_t seed, uint32_t row) { return row * MIX32_ROW + (seed * MIX32_SEED + MIX32_ADD); }
__device__ __forceinline__ uint32_t mix32_finish(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t mix32(uint32_t seed, uint32_t row, uint32_t col) { return mix32_finish(col ^ mix32_row_key(seed, row)); }
__device__ __forceinline__ double u01_mix32(uint32_t seed, uint32_t row, uint32_t col) { return (double)mix32(seed, row, col) * (1.0 / 4294967296.0); }

// Natural log of a positive, normal double (the path only ever feeds it positive integers < 2^53).
// Classic argument-reduction + degree-14 odd polynomial in s = f/(2+f) (the algorithm of the freely
// distributable Sun fdlibm e_log.c, error < 1 ulp); written with explicit single operations so that the
// value does not depend on the math library or on FMA contraction.
__device__ __forceinline__ double log_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int hx = __double2hiint(x);
  const int lx = __double2loint(x);
  int k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int i = (hx + 0x95f64) & 0x100000;
  x = __hiloint2double(hx | (i ^ 0x3ff00000), lx); /* normalise x or x/2 */
  k += (i >> 20);
  const double f = x - 1.0;
  const double dk = (double)k;
  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2**-20 */
    if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  i = hx - 0x6147a;
  const double w = z * z;
  const int j = 0x6b851 - hx;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  const double R = t2 + t1;
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// LogLikelihood.xLogX
__device__ __forceinline__ double x_log_x(long long x) { return x == 0 ? 0.0 : (double)x * log_pos((double)x); }

// LogLikelihood.entropy(a, b) = xLogX(a+b) - xLogX(a) - xLogX(b), left to right.
__device__ __forceinline__ double entropy2(long long a, long long b) { return (x_log_x(a + b) - x_log_x(a)) - x_log_x(b); }

// xLogX through a table for small arguments.  xlx_tab[x] = x_log_x(x) was produced by the same device function, so the
// value is bit-identical to evaluating it in place.  After the interaction cut k11, k12 and k21 are almost always
// below the table size, which leaves one real logarithm (k22) per candidate.
constexpr int XLX_TABLE = 4096;
__device__ __forceinline__ double x_log_x_tab(long long x, const double* __restrict__ xlx_tab) {
  return x < (long long)XLX_TABLE ? xlx_tab[x] : (double)x * log_pos((double)x);
}
// k22 = N - cA - cB + k11 sits within a few hundred of N once the interaction cut has capped cA and cB, so its xLogX comes
// from a second table, xlx_hi[d] = x_log_x(N - d) (filled by the same device function for the N of the build): with it the
// common case evaluates no logarithm at all -- four table reads -- and the value stays bit-identical.
__device__ __forceinline__ double x_log_x_hi(long long x, long long n_users, const double* __restrict__ xlx_hi, const double* __restrict__ xlx_tab) {
  const long long d = n_users - x;
  return (d >= 0 && d < (long long)XLX_TABLE) ? xlx_hi[d] : x_log_x_tab(x, xlx_tab);
}
// columnEntropy = entropy(cB, N - cB) = (xLogX(N) - xLogX(cB)) - xLogX(N - cB) from the two tables: the same three values in the
// same order as entropy2(cB, N - cB), so bit-identical to the per-item array it replaces -- without the 8-byte gather per candidate
// (a 2M-item catalogue's entropy array is 16 MB: four times an XCD's L2).
__device__ __forceinline__ double column_entropy_tab(long long cb, double xlx_n, long long n_users, const double* __restrict__ xlx_tab,
                                                     const double* __restrict__ xlx_hi) {
  return (xlx_n - x_log_x_tab(cb, xlx_tab)) - x_log_x_hi(n_users - cb, n_users, xlx_hi, xlx_tab);
}
// ... or, for the counts the interaction cut leaves (below the table size), from the per-build table of that very expression
__device__ __forceinline__ double column_entropy_of(long long cb, double xlx_n, long long n_users, const double* __restrict__ xlx_tab,
                                                    const double* __restrict__ xlx_hi, const double* __restrict__ col_ent) {
  return (col_ent != nullptr && cb < (long long)XLX_TABLE) ? col_ent[cb] : column_entropy_tab(cb, xlx_n, n_users, xlx_tab, xlx_hi);
}
__device__ __forceinline__ double llr_from_entropies_tab(double row_entropy, double column_entropy, double xlx_n, long long k11, long long k12,
                                                         long long k21, long long k22, const double* __restrict__ xlx_tab,
                                                         long long n_users, const double* __restrict__ xlx_hi) {
  const double matrix_entropy =
      (((xlx_n - x_log_x_tab(k11, xlx_tab)) - x_log_x_tab(k12, xlx_tab)) - x_log_x_tab(k21, xlx_tab)) - x_log_x_hi(k22, n_users, xlx_hi, xlx_tab);
  const double s = row_entropy + column_entropy;
  if (s < matrix_entropy) return 0.0; /* round off error */
  return 2.0 * (s - matrix_entropy);
}

// LogLikelihood.logLikelihoodRatio with the row / column entropies supplied (they are per-item constants:
// rowEntropy = entropy(cA[i], N - cA[i]), columnEntropy = entropy(cB[j], N - cB[j])); xlx_n = xLogX(N).
// Same operations in the same order as the Java, so the value equals the un-hoisted formula bit for bit.
__device__ __forceinline__ double llr_from_entropies(double row_entropy, double column_entropy, double xlx_n, long long k11,
                                                     long long k12, long long k21, long long k22) {
  const double matrix_entropy = (((xlx_n - x_log_x(k11)) - x_log_x(k12)) - x_log_x(k21)) - x_log_x(k22);
  const double s = row_entropy + column_entropy;
  if (s < matrix_entropy) return 0.0; /* round off error */
  return 2.0 * (s - matrix_entropy);
}

// SimilarityAnalysis.logLikelihoodRatio(numInteractionsWithA, ..WithB, ..WithAandB, numInteractions)
__device__ __forceinline__ double llr_full(long long with_a, long long with_b, long long with_ab, long long n) {
  const long long k11 = with_ab, k12 = with_a - with_ab, k21 = with_b - with_ab, k22 = n - with_a - with_b + with_ab;
  const double row_entropy = entropy2(k11 + k12, k21 + k22);
  const double column_entropy = entropy2(k11 + k21, k12 + k22);
  return llr_from_entropies(row_entropy, column_entropy, x_log_x(k11 + k12 + k21 + k22), k11, k12, k21, k22);
}

}  // namespace urcco
