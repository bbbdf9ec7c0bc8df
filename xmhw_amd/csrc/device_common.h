// device_common.h -- device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xmhw {

// Order-preserving integer keys.  key(a) < key(b)  <=>  a < b for non-NaN
// floats (-0.0 sorts just below +0.0); key 0 is reserved for "invalid" (NaN
// sample, sample outside [0,T), padding) and sorts below every real value
// (the smallest real key is key(-inf) = 0x007FFFFF).
__device__ __forceinline__ uint32_t f32_key(float x) {
    const uint32_t b = __float_as_uint(x);
    const uint32_t k = b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u);
    return (x != x) ? 0u : k;
}
__device__ __forceinline__ float key_f32(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(b);
}
__device__ __forceinline__ uint64_t f64_key(double x) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(x));
    const uint64_t k = b ^ (static_cast<uint64_t>(static_cast<int64_t>(b) >> 63) | 0x8000000000000000ull);
    return (x != x) ? 0ull : k;
}
__device__ __forceinline__ double key_f64(uint64_t k) {
    const uint64_t b = (k & 0x8000000000000000ull) ? (k ^ 0x8000000000000000ull) : ~k;
    return __longlong_as_double(static_cast<long long>(b));
}

template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    using type = uint32_t;
    static constexpr int bits = 32;
    __device__ static __forceinline__ type key(float x) { return f32_key(x); }
    __device__ static __forceinline__ double value(type k) { return static_cast<double>(key_f32(k)); }
};
template <> struct KeyOf<double> {
    using type = uint64_t;
    static constexpr int bits = 64;
    __device__ static __forceinline__ type key(double x) { return f64_key(x); }
    __device__ static __forceinline__ double value(type k) { return key_f64(k); }
};

// numpy.quantile(method="linear") interpolation, numpy/lib/_function_base_impl.py
// _lerp (numpy 2.2.6): a + (b-a)*g, replaced by b - (b-a)*(1-g) when g >= 0.5.
// Compiled with -ffp-contract=off so that no FMA changes the rounding.
__device__ __forceinline__ double numpy_lerp(double a, double b, double g) {
    const double diff = b - a;
    double r = a + diff * g;
    if (g >= 0.5) r = b - diff * (1.0 - g);
    return r;
}

__device__ __forceinline__ double make_nan() { return __longlong_as_double(0x7FF8000000000000ll); }

// ---- where a kernel's samples come from ---------------------------------------------------------------------------
// PlainSrc<T>: a float32 / float64 series.  PackedSrc: int16 codes + the CF recipe (kernels.h: PackedI16).  at(i) = the
// sample a kernel keys and sums (before the kernel's own negation for cold spells), value(v) = the output value of a
// keyed sample, mean(m) = the output value of the mean of the keyed samples, guess(y) = a sample near output value y.
template <typename T>
struct PlainSrc {
    using sample = T;
    const T* p;
    __device__ __forceinline__ T at(int64_t i) const { return p[i]; }
    __device__ __forceinline__ double value(T v) const { return static_cast<double>(v); }
    __device__ __forceinline__ double mean(double m) const { return m; }
    __device__ __forceinline__ T guess(double y) const { return static_cast<T>(y); }
};


// splitmix64 finaliser: counter-based generator for the synthetic SST
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace xmhw
