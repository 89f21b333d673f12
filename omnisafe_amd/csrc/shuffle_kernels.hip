// libomnisafe_amd -- the minibatch shuffles of an update (gfx950).
//
// Reference: every pass of `_update` iterates a `DataLoader(TensorDataset(...), batch_size, shuffle=True)`
// (algorithms/on_policy/base/policy_gradient.py:357-377, natural_pg.py:196-223): one uniform random permutation
// of the M rows per pass, drawn by torch's RandomSampler (`torch.randperm`).  Rounds 1-3 drew all passes'
// permutations in one batched `argsort` of random 62-bit keys -- torch / rocPRIM segmented merge sorts, ~45 launches
// and 0.3 ms per epoch of eight passes over 65 536 rows, the largest block of not-hand-written kernels inside the timed
// region.  A permutation does not need a sort: a keyed BIJECTION of [0, 2^k) evaluated at i, "cycle-walked" back
// into [0, M) when M is not a power of two, IS row i of the permutation -- one pass, no memory traffic but the
// result, every element independent (the construction of Mitchell et al., "Bandwidth-optimal random shuffling for
// GPUs", 2021, which cub / thrust's shuffle also follow; the cipher below is this file's own).
//
// The bijection: k = ceil(log2 M) bits split into a high part of ceil(k / 2) bits and a low part of floor(k / 2);
// OSA_SHUF_ROUNDS alternating Feistel steps  hi ^= F_r(lo) & mask_hi,  lo ^= F_r(hi) & mask_lo  -- each step is an
// involution on one half given the other, so the composition is a bijection whatever F is.  F_r is a 32-bit
// multiply-xorshift mixer keyed per (row, round); the round keys come from the row's 64-bit seed (drawn by torch's
// device generator: the same `seed` gives the same shuffles) through a splitmix64 sequence.
// Statistics: tests/test_shuffle_gpu.py (bijectivity for M in 1 .. 2^20, position x value chi-square, neighbour
// correlation, independence of rows).
#include "osa_common.h"

#define OSA_SHUF_ROUNDS 24

__device__ __forceinline__ unsigned long long osa_splitmix64(unsigned long long& s) {
  s += 0x9E3779B97F4A7C15ull;
  unsigned long long z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// 32-bit mixer (two odd multiplies, three xor-shifts), keyed by XOR on the way in and an add between the multiplies
__device__ __forceinline__ unsigned osa_shuf_f(unsigned x, unsigned k0, unsigned k1) {
  x ^= k0;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x += k1;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// grid (ceil(M / (256 * OSA_SHUF_PER)), rows), 256 threads
#define OSA_SHUF_PER 8
__global__ __launch_bounds__(256) void osa_shuffle_rows_kernel(const long long* __restrict__ row_seeds, long M,
                                                               int bits, long long* __restrict__ perm) {
  __shared__ unsigned keys[2 * OSA_SHUF_ROUNDS];
  const int row = blockIdx.y;
  if (threadIdx.x < OSA_SHUF_ROUNDS) {
    unsigned long long s = (unsigned long long)row_seeds[row] ^ (0xD1B54A32D192ED03ull * (unsigned long long)(threadIdx.x + 1));
    const unsigned long long z = osa_splitmix64(s);
    keys[2 * threadIdx.x] = (unsigned)z;
    keys[2 * threadIdx.x + 1] = (unsigned)(z >> 32);
  }
  __syncthreads();
  const int lo_bits = bits >> 1, hi_bits = bits - lo_bits;
  const unsigned mask_lo = lo_bits ? (0xFFFFFFFFu >> (32 - lo_bits)) : 0u;
  const unsigned mask_hi = hi_bits ? (0xFFFFFFFFu >> (32 - hi_bits)) : 0u;
  long long* __restrict__ out = perm + (long)row * M;
  const long base = ((long)blockIdx.x * OSA_SHUF_PER) * 256 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < OSA_SHUF_PER; ++u) {
    const long i = base + (long)u * 256;
    if (i >= M) break;
    unsigned long long v = (unsigned long long)i;
    do {  // cycle walking: the bijection of [0, 2^bits) restricted to the orbit's first point inside [0, M)
      unsigned hi = (unsigned)(v >> lo_bits) & mask_hi, lo = (unsigned)v & mask_lo;
#pragma unroll
      for (int r = 0; r < OSA_SHUF_ROUNDS; r += 2) {
        hi ^= osa_shuf_f(lo, keys[2 * r], keys[2 * r + 1]) & mask_hi;
        lo ^= osa_shuf_f(hi, keys[2 * r + 2], keys[2 * r + 3]) & mask_lo;
      }
      v = ((unsigned long long)hi << lo_bits) | lo;
    } while ((long)v >= M);
    out[i] = (long long)v;
  }
}

extern "C" {

// perm[row][0 .. M) = a pseudo-random permutation of 0 .. M-1 per row, keyed by row_seeds[row] (device array)
int osa_shuffle_rows(const long long* row_seeds, int rows, long M, long long* perm, void* stream) {
  OSA_REQUIRE(row_seeds && perm && rows > 0 && M > 0);
  if (M > (1L << 40)) return OSA_EUNSUPPORTED;
  int bits = 1;  // (one bit at least: M = 1 walks 1 -> 0)
  while ((1L << bits) < M) ++bits;
  if (bits > 62 || (bits - (bits >> 1)) > 32) return OSA_EUNSUPPORTED;
  const long per_block = 256L * OSA_SHUF_PER;
  const long gx = (M + per_block - 1) / per_block;
  if (gx > 2147483647L || rows > 65535) return OSA_EUNSUPPORTED;
  hipLaunchKernelGGL(osa_shuffle_rows_kernel, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, osa_stream(stream),
                     row_seeds, M, bits, perm);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

}  // extern "C"
