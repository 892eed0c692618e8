// Exact row matching on the device: which rows of X1 also occur in X2?
//
// The reference's data manager answers this with a broadcast comparison,
//     torch.all(x1.unsqueeze(1) == x2.unsqueeze(0), dim=-1).any(-1)      (FidelityFusion_Models/MF_data.py:196-199,234-237)
// i.e. an N1 x N2 x D boolean temporary (4.3 GB at N = 16384, D = 16) and O(N1 N2 D) work.  Here: an open-addressing
// hash table of X2's rows (64-bit mix of the canonicalised bit patterns, linear probing, one atomicCAS per insert),
// then one probe per row of X1 with an exact comparison on every candidate -- O((N1 + N2) D) bytes of HBM traffic.
// Equality follows IEEE `==` as torch's does: -0.0 equals +0.0 (canonicalised before hashing), NaN equals nothing
// (rows holding a NaN are neither inserted nor matched).
#include "ffgp_internal.h"

__device__ __forceinline__ unsigned long long ffgp_mix64(unsigned long long h, unsigned long long v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 32;
  return h;
}

// hash of one row; returns false if the row holds a NaN
__device__ __forceinline__ bool ffgp_row_hash(const double* __restrict__ x, int D, unsigned long long& out) {
  unsigned long long h = 0x243f6a8885a308d3ull;
  bool ok = true;
  for (int k = 0; k < D; ++k) {
    double v = x[k];
    if (v != v) ok = false;
    if (v == 0.0) v = 0.0;   // -0.0 -> +0.0
    h = ffgp_mix64(h, (unsigned long long)__double_as_longlong(v));
  }
  out = h;
  return ok;
}

// table[slot] = row index + 1 (0 = empty)
__global__ __launch_bounds__(256) void ffgp_join_build(const double* __restrict__ X2, int n2, int D, int* __restrict__ table,
                                                       unsigned mask) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n2) return;
  unsigned long long h;
  if (!ffgp_row_hash(X2 + (size_t)j * D, D, h)) return;
  unsigned slot = (unsigned)h & mask;
  while (atomicCAS(&table[slot], 0, j + 1) != 0) slot = (slot + 1) & mask;   // table is at least twice n2: always terminates
}

__global__ __launch_bounds__(256) void ffgp_join_probe(const double* __restrict__ X1, int n1, const double* __restrict__ X2,
                                                       int D, const int* __restrict__ table, unsigned mask,
                                                       unsigned char* __restrict__ found) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n1) return;
  const double* a = X1 + (size_t)i * D;
  unsigned long long h;
  unsigned char hit = 0;
  if (ffgp_row_hash(a, D, h)) {
    unsigned slot = (unsigned)h & mask;
    for (;;) {
      const int e = table[slot];
      if (e == 0) break;
      const double* b = X2 + (size_t)(e - 1) * D;
      bool eq = true;
      for (int k = 0; k < D && eq; ++k) eq = (a[k] == b[k]);
      if (eq) {
        hit = 1;
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
  found[i] = hit;
}

int ffgp_rows_in_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, unsigned char* found) {
  if (n1 <= 0) return FFGP_OK;
  if (!X1 || !found || D <= 0 || (n2 > 0 && !X2)) return FFGP_ERR_ARG;
  if (n2 <= 0) {
    FFGP_HIP(hipMemsetAsync(found, 0, (size_t)n1, h->stream));
    return FFGP_OK;
  }
  unsigned cap = 1024;
  while (cap < 2u * (unsigned)n2) cap <<= 1;
  FFGP_CHECK(ffgp_ensure_ws(h, (size_t)cap * sizeof(int)));
  int* table = reinterpret_cast<int*>(h->ws);
  FFGP_HIP(hipMemsetAsync(table, 0, (size_t)cap * sizeof(int), h->stream));
  hipLaunchKernelGGL(ffgp_join_build, dim3((n2 + 255) / 256), dim3(256), 0, h->stream, X2, n2, D, table, cap - 1);
  hipLaunchKernelGGL(ffgp_join_probe, dim3((n1 + 255) / 256), dim3(256), 0, h->stream, X1, n1, X2, D, table, cap - 1, found);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
