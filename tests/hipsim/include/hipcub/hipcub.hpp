// TEST INFRASTRUCTURE (hipsim): sequential host stand-ins for the few hipcub device primitives the library calls, with hipcub's
// two-phase protocol (d_temp == nullptr -> size query).  Stable, like the device versions.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace hipcub {
template <class T> struct CountingInputIterator {
  T base;
  explicit CountingInputIterator(T b = T()) : base(b) {}
  T operator[](size_t i) const { return base + (T)i; }
};
struct DeviceSelect {
  template <class In, class Flag, class Out, class Count>
  static hipError_t Flagged(void* tmp, size_t& bytes, In in, Flag flags, Out out, Count num_selected, int n, hipStream_t = nullptr) {
    if (!tmp) { bytes = 256; return hipSuccess; }
    hipsim::count(hipsim::C_LIBCALL);
    int k = 0;
    for (int i = 0; i < n; i++)
      if (flags[i]) out[k++] = in[i];
    *num_selected = k;
    return hipSuccess;
  }
  template <class In, class Out, class Count>
  static hipError_t Unique(void* tmp, size_t& bytes, In in, Out out, Count num_selected, int n, hipStream_t = nullptr) {
    if (!tmp) { bytes = 256; return hipSuccess; }
    hipsim::count(hipsim::C_LIBCALL);
    int k = 0;
    for (int i = 0; i < n; i++)
      if (i == 0 || !(in[i] == in[i - 1])) out[k++] = in[i];
    *num_selected = k;
    return hipSuccess;
  }
};
namespace detail {
template <class K> static inline unsigned long long key_bits(K k, int begin_bit, int end_bit) {
  unsigned long long u = 0;
  memcpy(&u, &k, sizeof(K));
  if (std::is_floating_point<K>::value) {  // radix order of IEEE keys
    const unsigned long long sign = 1ull << (8 * sizeof(K) - 1);
    u = (u & sign) ? ~u : (u | sign);
    if (sizeof(K) < 8) u &= (1ull << (8 * sizeof(K))) - 1;
  } else if (std::is_signed<K>::value) {
    u ^= 1ull << (8 * sizeof(K) - 1);
  }
  u >>= begin_bit;
  const int w = end_bit - begin_bit;
  return w >= 64 ? u : (u & ((1ull << w) - 1));
}
template <class K, class V> static void sort_pairs(const K* kin, K* kout, const V* vin, V* vout, size_t n, int b, int e, bool descending) {
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), (size_t)0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t c) {
    const unsigned long long ka = key_bits(kin[a], b, e), kc = key_bits(kin[c], b, e);
    return descending ? ka > kc : ka < kc;
  });
  std::vector<K> ks(n);
  std::vector<V> vs(n);
  for (size_t i = 0; i < n; i++) { ks[i] = kin[idx[i]]; vs[i] = vin[idx[i]]; }
  for (size_t i = 0; i < n; i++) { kout[i] = ks[i]; vout[i] = vs[i]; }
}
}  // namespace detail
struct DeviceRadixSort {
  template <class K, class V>
  static hipError_t SortPairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, int n, int b = 0, int e = 8 * sizeof(K), hipStream_t = nullptr) {
    if (!tmp) { bytes = 256; return hipSuccess; }
    hipsim::count(hipsim::C_LIBCALL);
    detail::sort_pairs(kin, kout, vin, vout, (size_t)n, b, e, false);
    return hipSuccess;
  }
  template <class K, class V>
  static hipError_t SortPairsDescending(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, int n, int b = 0, int e = 8 * sizeof(K), hipStream_t = nullptr) {
    if (!tmp) { bytes = 256; return hipSuccess; }
    hipsim::count(hipsim::C_LIBCALL);
    detail::sort_pairs(kin, kout, vin, vout, (size_t)n, b, e, true);
    return hipSuccess;
  }
};
}  // namespace hipcub

namespace rocprim {
struct default_config {};
template <class A, class B, class C, unsigned Limit> struct radix_sort_config {};
template <class Config = default_config, class K, class V>
static inline hipError_t radix_sort_pairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned b = 0, unsigned e = 8 * sizeof(K),
                                          hipStream_t = nullptr, bool = false) {
  if (!tmp) { bytes = 256; return hipSuccess; }
    hipsim::count(hipsim::C_LIBCALL);
  hipcub::detail::sort_pairs(kin, kout, vin, vout, n, (int)b, (int)e, false);
  return hipSuccess;
}
template <class Config = default_config, class K>
static inline hipError_t radix_sort_keys(void* tmp, size_t& bytes, const K* kin, K* kout, size_t n, unsigned b = 0, unsigned e = 8 * sizeof(K), hipStream_t = nullptr,
                                         bool = false) {
  if (!tmp) { bytes = 256; return hipSuccess; }
  hipsim::count(hipsim::C_LIBCALL);
  std::vector<unsigned char> dummy_in(n), dummy_out(n);
  hipcub::detail::sort_pairs(kin, kout, dummy_in.data(), dummy_out.data(), n, (int)b, (int)e, false);  // stable on the bits [b, e)
  return hipSuccess;
}
}  // namespace rocprim
