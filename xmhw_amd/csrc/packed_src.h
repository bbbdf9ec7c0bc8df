// packed_src.h -- int16-packed input read in place (kernels.h: PackedI16; device_common.h: PlainSrc for the interface).
// The arithmetic is xmhw_decode()'s (kernels_ingest.hip: decode_slab), compiled with -ffp-contract=off like it: a
// multiplication and an addition, each rounded.
#pragma once
#include "device_common.h"
#include "kernels.h"

namespace xmhw {

__device__ __forceinline__ int32_t packed_code(const PackedI16& pk, int16_t raw) {
    uint16_t b;
    __builtin_memcpy(&b, &raw, 2);
    if (pk.swap) b = static_cast<uint16_t>((b >> 8) | (b << 8));
    return static_cast<int32_t>(static_cast<int16_t>(b));
}
// the sample the kernels key and sum: NaN for the fill code
__device__ __forceinline__ float packed_sample(const PackedI16& pk, int32_t code) {
    float f = static_cast<float>(code);
    if (pk.mode == 1) {
        f = f * pk.sf;
        f = f + pk.of;
    }
    return code == pk.fill ? __uint_as_float(0x7FC00000u) : f;
}
// the output value of a keyed sample / of the mean of the keyed samples (v as the kernel holds it: negated if key_neg)
__device__ __forceinline__ double packed_value(const PackedI16& pk, double v) {
    if (pk.mode != 2) return v;
    const double c = pk.key_neg ? -v : v;
    double y = c * pk.s;
    y = y + pk.o;
    return pk.val_neg ? -y : y;
}

struct PackedSrc {
    using sample = float;
    const int16_t* p;
    PackedI16 pk;
    __device__ __forceinline__ float at(int64_t i) const { return packed_sample(pk, packed_code(pk, p[i])); }
    __device__ __forceinline__ double value(float v) const { return packed_value(pk, static_cast<double>(v)); }
    __device__ __forceinline__ double mean(double m) const { return packed_value(pk, m); }
    __device__ __forceinline__ float guess(double y) const {
        if (pk.mode != 2) return static_cast<float>(y);
        const double c = ((pk.val_neg ? -y : y) - pk.o) / pk.s;
        return static_cast<float>(pk.key_neg ? -c : c);
    }
};

}  // namespace xmhw
