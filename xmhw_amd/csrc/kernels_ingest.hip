// kernels_ingest.hip -- file bytes -> samples, on the device.
//
// The reference leaves reading to xarray (docs/gettingstarted.rst:30-33), whose CF decoding turns a
// packed variable into floats on the host: `raw * scale_factor + add_offset`, `_FillValue` -> NaN,
// big-endian (netCDF classic) -> native.  Here the RAW bytes of a column slab go over PCIe (half the
// bytes for int16-packed archives) and this kernel decodes them in HBM: lanes run along the cell
// axis, one read and one write per sample, HBM-bound.
//   float32 result: (float)raw * (float)scale + (float)offset, two float32 roundings -- what xarray
//   computes for float32 attributes; float64 result: the same in double (float64 attributes).
#include "device_common.h"
#include "kernels.h"

namespace xmhw {
namespace {

__device__ __forceinline__ uint16_t bswap(uint16_t v) { return static_cast<uint16_t>((v << 8) | (v >> 8)); }
__device__ __forceinline__ uint32_t bswap(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ uint64_t bswap(uint64_t v) { return __builtin_bswap64(v); }

template <typename OUT>
__device__ __forceinline__ OUT out_nan();
template <> __device__ __forceinline__ float out_nan<float>() { return __uint_as_float(0x7FC00000u); }
template <> __device__ __forceinline__ double out_nan<double>() { return make_nan(); }

// RAW: int16_t (packed), float, double; SWAP: the file is big-endian
template <typename RAW, typename OUT, bool SWAP>
__global__ __launch_bounds__(256) void decode_slab(const RAW* __restrict__ in, int64_t rows, int64_t cols, int64_t ld_in,
                                                   OUT* __restrict__ out, int64_t ld_out, OUT scale, OUT offset,
                                                   int has_scale, int has_fill, RAW fill) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        RAW v = in[r * ld_in + c];
        if constexpr (SWAP) {
            if constexpr (sizeof(RAW) == 2) {
                uint16_t b; __builtin_memcpy(&b, &v, 2); b = bswap(b); __builtin_memcpy(&v, &b, 2);
            } else if constexpr (sizeof(RAW) == 4) {
                uint32_t b; __builtin_memcpy(&b, &v, 4); b = bswap(b); __builtin_memcpy(&v, &b, 4);
            } else {
                uint64_t b; __builtin_memcpy(&b, &v, 8); b = bswap(b); __builtin_memcpy(&v, &b, 8);
            }
        }
        OUT x = static_cast<OUT>(v);
        if (has_scale) {
            x = x * scale;        // -ffp-contract=off: two roundings, as numpy
            x = x + offset;
        }
        if (has_fill && v == fill) x = out_nan<OUT>();     // a NaN fill value never compares equal: NaN stays NaN anyway
        out[r * ld_out + c] = x;
    }
}

template <typename RAW, typename OUT>
hipError_t launch(const void* in, int swap, int64_t rows, int64_t cols, int64_t ld_in, void* out, int64_t ld_out,
                  double scale, double offset, int has_scale, int has_fill, double fill, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    dim3 grid(static_cast<unsigned>((cols + 255) / 256), static_cast<unsigned>(rows < 2048 ? rows : 2048));
    if (swap)
        hipLaunchKernelGGL((decode_slab<RAW, OUT, true>), grid, dim3(256), 0, stream, static_cast<const RAW*>(in), rows, cols,
                           ld_in, static_cast<OUT*>(out), ld_out, static_cast<OUT>(scale), static_cast<OUT>(offset),
                           has_scale, has_fill, static_cast<RAW>(fill));
    else
        hipLaunchKernelGGL((decode_slab<RAW, OUT, false>), grid, dim3(256), 0, stream, static_cast<const RAW*>(in), rows, cols,
                           ld_in, static_cast<OUT*>(out), ld_out, static_cast<OUT>(scale), static_cast<OUT>(offset),
                           has_scale, has_fill, static_cast<RAW>(fill));
    return hipGetLastError();
}


// The inverse for float32 -> int16: code = rint((x - add_offset) / scale_factor) in float64, clamped to the int16 range
// without its lowest code; NaN -> fill_code.  What writing a packed archive does (xarray's CF encoding); bench.py and
// the tests make packed input with it.
__global__ __launch_bounds__(256) void encode_i16(const float* __restrict__ in, int64_t rows, int64_t cols, int64_t ld_in,
                                                  int16_t* __restrict__ out, int64_t ld_out, double scale, double offset,
                                                  int32_t fill) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const float x = in[r * ld_in + c];
        double k = rint((static_cast<double>(x) - offset) / scale);
        k = k < -32767.0 ? -32767.0 : (k > 32767.0 ? 32767.0 : k);
        out[r * ld_out + c] = static_cast<int16_t>(x == x ? static_cast<int32_t>(k) : fill);
    }
}
// ---------------------------------------------------------------------------
// pad_gaps: ts.interpolate_na(dim=tdim, max_gap=maxPadLength) of the reference (xmhw/xmhw.py:159-160,
// :409-410), i.e. xarray's linear interpolate_na with use_coordinate=True on the device copy of the
// compacted series, in place.  One thread per cell walks its column (lanes along the cell axis, rows
// loaded kAhead at a time).  A run of NaN strictly between two valid samples at steps a < b is filled
// iff x[b] - x[a] <= max_gap (x = the numeric time coordinate), with numpy.interp's arithmetic in
// double -- slope = (y[b] - y[a]) / (x[b] - x[a]); slope * (x[u] - x[a]) + y[a]; the two NaN fallbacks
// of numpy's compiled_base.c -- rounded to the sample type on store.  Leading and trailing runs and
// all-NaN cells stay as they are.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pad_gaps(T* __restrict__ ts, int64_t Tn, int64_t C, int64_t ld,
                                                const double* __restrict__ x, double max_gap) {
    constexpr int kAhead = 8;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    T* col = ts + c;
    int64_t a = -1;
    double ya = 0.0;
    for (int64_t t0 = 0; t0 < Tn; t0 += kAhead) {
        T v[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int64_t t = t0 + u < Tn ? t0 + u : Tn - 1;
            v[u] = col[t * ld];
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int64_t t = t0 + u;
            if (t >= Tn || !(v[u] == v[u])) continue;
            const double yb = static_cast<double>(v[u]);
            if (a >= 0 && t - a > 1) {
                const double xa = x[a], xb = x[t];
                if (xb - xa <= max_gap) {
                    const double slope = (yb - ya) / (xb - xa);
                    for (int64_t k = a + 1; k < t; ++k) {
                        double r = slope * (x[k] - xa) + ya;
                        if (r != r) {
                            r = slope * (x[k] - xb) + yb;
                            if (r != r && ya == yb) r = ya;
                        }
                        col[k * ld] = static_cast<T>(r);
                    }
                }
            }
            a = t;
            ya = yb;
        }
    }
}

}  // namespace

hipError_t launch_encode_i16(const float* in, int64_t rows, int64_t cols, int64_t ld_in, int16_t* out, int64_t ld_out,
                             double scale, double offset, int32_t fill, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    dim3 grid(static_cast<unsigned>((cols + 255) / 256), static_cast<unsigned>(rows < 2048 ? rows : 2048));
    hipLaunchKernelGGL(encode_i16, grid, dim3(256), 0, stream, in, rows, cols, ld_in, out, ld_out, scale, offset, fill);
    return hipGetLastError();
}

hipError_t launch_decode(const void* in, int raw_type, int swap, int64_t rows, int64_t cols, int64_t ld_in, void* out,
                         int out_itemsize, int64_t ld_out, double scale, double offset, int has_scale, int has_fill,
                         double fill, hipStream_t stream) {
    // raw_type: 2 = int16, 4 = float32, 8 = float64 (the item size)
    if (raw_type == 2 && out_itemsize == 4)
        return launch<int16_t, float>(in, swap, rows, cols, ld_in, out, ld_out, scale, offset, has_scale, has_fill, fill, stream);
    if (raw_type == 2 && out_itemsize == 8)
        return launch<int16_t, double>(in, swap, rows, cols, ld_in, out, ld_out, scale, offset, has_scale, has_fill, fill, stream);
    if (raw_type == 4 && out_itemsize == 4)
        return launch<float, float>(in, swap, rows, cols, ld_in, out, ld_out, scale, offset, has_scale, has_fill, fill, stream);
    if (raw_type == 8 && out_itemsize == 8)
        return launch<double, double>(in, swap, rows, cols, ld_in, out, ld_out, scale, offset, has_scale, has_fill, fill, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_pad_gaps(void* ts, int itemsize, int64_t Tn, int64_t C, int64_t ld, const double* x, double max_gap,
                           hipStream_t stream) {
    if (Tn <= 0 || C <= 0) return hipSuccess;
    dim3 grid(static_cast<unsigned>((C + 255) / 256));
    if (itemsize == 4)
        hipLaunchKernelGGL(pad_gaps<float>, grid, dim3(256), 0, stream, static_cast<float*>(ts), Tn, C, ld, x, max_gap);
    else if (itemsize == 8)
        hipLaunchKernelGGL(pad_gaps<double>, grid, dim3(256), 0, stream, static_cast<double*>(ts), Tn, C, ld, x, max_gap);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace xmhw
