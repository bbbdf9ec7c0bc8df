// kernels.h -- host-callable launchers (defined in the .hip files).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xmhw {

struct DevChunk { int32_t warm_start, begin, end; };
// int16-packed input (CF packing: value = code * scale_factor + add_offset, _FillValue -> NaN; what xmhw_decode() would
// write out, see kernels_ingest.hip) read by the kernels directly: the sorted-list kernel, its recomputation and the
// leftover kernel take the codes and this recipe instead of a decoded copy of the series (capi.cpp: xmhw_clim_raw_i16).
//   mode 1  float32 decode: a sample is float(code) * sf + of, two float32 roundings -- the float32 series xarray (and
//           xmhw_decode) would hand over; the kernels key and sum exactly those values;
//   mode 2  float64 decode (float64 packing attributes): code -> value is monotone, so the kernels key float(code)
//           (exact) and decode only the two selected codes and the mean of the codes: double(code) * s + o;
//   mode 3  no packing attributes: a sample is float(code).
struct PackedI16 {
    int32_t mode = 0;
    int32_t fill = 0x7FFFFFFF;   // the code that means "missing" (0x7FFFFFFF: none)
    int32_t swap = 0;            // the codes are big-endian
    int32_t key_neg = 0;         // the kernel keys the NEGATED sample (cold spells; mode 2: XOR a negative scale_factor)
    int32_t val_neg = 0;         // mode 2: the outputs are those of the negated series (cold spells)
    float sf = 1.0f, of = 0.0f;
    double s = 1.0, o = 0.0;
};
// a chunk of the sorted-list kernel: its table / flag rows start at trow0 (row of step warm_start)
struct DevSortedChunk { int32_t warm_start, begin, end, trow0; };

// generic kernel (any plan): thread per (cell, row)
template <typename T>
hipError_t launch_generic(const T* ts, int64_t Tn, int64_t C, int64_t ld, const int32_t* row_ptr,
                          const int32_t* centres, int32_t D, int32_t w, double q, int negate,
                          double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                          const uint32_t* run_flag = nullptr);   // non-NULL: return at once unless *run_flag != 0

// ring kernel (fast path).  Returns hipErrorInvalidValue if (w, yps) is not instantiated.
bool ring_supported(int32_t w, int32_t yps, int32_t subs, int elem_bytes);
// tracks per lane (0 if none) and lanes per cell (8, or 16 for records of 49..96 tracks) of the
// float32 ring kernel that covers ntracks tracks
int32_t ring_pick(int32_t w, int32_t ntracks, int elem_bytes, int32_t* subs_out);
hipError_t launch_ring_f32(const float* ts, int64_t C, int64_t ld, const uint32_t* table,
                           int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                           int32_t w, int32_t yps, int32_t subs, double q, int negate, double* thresh,
                           double* seas, int64_t ldo, hipStream_t stream,
                           unsigned long long* stats = nullptr);

// second-generation float32 ring kernel (kernels_ring2.hip): 8 lanes per cell, tracks dealt y-major;
// variant bit 0 = 8-bit SAD probes, bit 1 = extraction skips empty ring positions
int32_t ring2_pick_yps(int32_t w, int32_t ntracks, int32_t variant);   // tracks per lane, 0 if not instantiated
bool ring2_narrowing_supported(int32_t w, int32_t yps, int32_t variant);   // float64 -> float32 narrowing instantiation exists
hipError_t launch_narrow_probe(const double* ts, int64_t Tn, int64_t C, int64_t ld, uint32_t* narrow_flag, hipStream_t stream);
hipError_t launch_ring2_f32_narrowing(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                                      const uint32_t* sflags, int32_t step_min, const DevChunk* chunks,
                                      int32_t nchunks, int32_t w, int32_t yps, int32_t ntracks, int32_t variant,
                                      double q, int negate, double* thresh, double* seas, int64_t ldo,
                                      hipStream_t stream, uint32_t* narrow_flag);
bool ring2_f32_supported(int32_t w, int32_t yps, int32_t variant);         // float32 instantiation exists
bool ring2_x64_supported(int32_t w, int32_t yps, int32_t variant);         // 64-bit (high / low key word) instantiation exists
hipError_t launch_ring2_f64(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t ntracks, int32_t variant, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream, const uint32_t* run_flag);
bool ring_stats_built();                                                // the counter twins exist (-DXMHW_RING_STATS)
int32_t ring2_subs(int32_t variant);                                    // lanes per cell of that variant
hipError_t launch_ring2_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t ntracks, int32_t variant, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                            unsigned long long* stats = nullptr);

// third-generation float32 ring kernel (kernels_ring3.hip): per-cell histogram in LDS + band compaction;
// 8 or 4 lanes per cell (ring2 variants 20 and 21), w = 5
int32_t ring3_pick_yps(int32_t w, int32_t ntracks, int32_t subs);
bool ring3_supported(int32_t w, int32_t yps, int32_t subs);
hipError_t launch_ring3_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                            unsigned long long* stats = nullptr);

// fifth generation (kernels_sorted.hip): sorted row-lists in LDS + a parallel merge-select; 2 lanes per cell, w = 5,
// float32 (or int16 codes read in place), on its own chunks and table rows (plan.h: sorted_plan).  A cell-row the select
// cannot settle (a row-list too short for it) is recomputed exactly inside the kernel, by the whole wave, from the samples.
int32_t sorted_pick_yps(int32_t w, int32_t ntracks);     // tracks per lane, 0 if not instantiated
int32_t sorted_pick_k(int32_t w, int32_t ntracks);       // keys kept per row-list, 0 if not instantiated
int32_t sorted_lds_bytes(int32_t w, int32_t ntracks);    // LDS of a wave (32 cells): 11 lists x the ranks kept in LDS, in 1,280-byte pieces
// the device behaviour the kernel's rank-major lists rely on (an LDS read outside the allocation returns 0): *d_bad = 0 if it holds
hipError_t sorted_lds_probe(uint32_t* d_bad, hipStream_t stream);
hipError_t launch_sorted_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                             const uint32_t* sflags, const DevSortedChunk* chunks, int32_t nchunks,
                             int32_t w, int32_t yps, int32_t ntracks, double q, int negate, double* thresh, double* seas,
                             int64_t ldo, hipStream_t stream, unsigned long long* stats = nullptr);
// the sorted-list kernel on int16 codes (instantiated for the same records as launch_sorted_f32)
hipError_t launch_sorted_i16(const int16_t* codes, const PackedI16& pk, int64_t C, int64_t ld, int64_t Tn,
                             const uint32_t* table, const uint32_t* sflags, const DevSortedChunk* chunks, int32_t nchunks,
                             int32_t w, int32_t yps, int32_t ntracks, double q, int negate, double* thresh, double* seas,
                             int64_t ldo, hipStream_t stream);

// fourth-generation float32 ring kernel (kernels_ring4.hip): a windowed key store in LDS instead of histogram + band
// compaction; same lane layouts and step tables as the third generation (ring2 variants 30 / 31 / 32 = 8 / 4 / 2 lanes)
int32_t ring4_pick_yps(int32_t w, int32_t ntracks, int32_t subs);
bool ring4_supported(int32_t w, int32_t yps, int32_t subs);
bool ring4_stats_built();
hipError_t launch_ring4_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                            unsigned long long* stats = nullptr);

// genuinely float64 samples: the 64-bit mode (high / low key words) of the third-generation kernel, 8 lanes per cell
bool ring3_x64_supported(int32_t w, int32_t yps, int32_t subs);
hipError_t launch_ring3_f64(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q, int negate, double* thresh,
                            double* seas, int64_t ldo, hipStream_t stream, const uint32_t* run_flag);
// the same for float64 input whose samples are float32-representable (see launch_ring2_f32_narrowing)
bool ring3_narrowing_supported(int32_t w, int32_t yps, int32_t subs);
hipError_t launch_ring3_f32_narrowing(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                                      const uint32_t* sflags, int32_t step_min, const DevChunk* chunks,
                                      int32_t nchunks, int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q,
                                      int negate, double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                                      uint32_t* narrow_flag);

// float64 input through the float32 ring kernel: zeroes narrow_flag, probes the series, runs the
// kernel with samples narrowed on load; narrow_flag != 0 afterwards: some sample is not float32-
// representable and the outputs are garbage (queue launch_ring2_f64(..., run_flag = narrow_flag) behind)
hipError_t launch_ring_f32_narrowing(const double* ts, int64_t Tn, int64_t C, int64_t ld, const uint32_t* table,
                                     int32_t step_min, const DevChunk* chunks, int32_t nchunks, int32_t w,
                                     int32_t yps, int32_t subs, double q, int negate, double* thresh, double* seas,
                                     int64_t ldo, hipStream_t stream, unsigned long long* stats,
                                     uint32_t* narrow_flag);

// Feb-29 substitution + circular running mean, per cell over present groups
hipError_t launch_finish(const double* th_in, const double* se_in, int64_t C, int64_t ldo, int32_t D,
                         int32_t i59, int32_t i60, int32_t i61, int feb29_fix, int smooth,
                         int32_t width, double* th_out, double* se_out, hipStream_t stream,
                         uint8_t* flags = nullptr);      // C bytes of scratch: enables the one-pass kernel (width 31)

template <typename T>
hipError_t launch_land_mask(const T* ts, int64_t Tn, int64_t C, int64_t ld, int anynans,
                            uint8_t* keep, hipStream_t stream);

hipError_t launch_land_mask_i16(const int16_t* codes, int64_t Tn, int64_t C, int64_t ld, int16_t fill_raw, int anynans,
                                uint8_t* keep, hipStream_t stream);

template <typename T>
hipError_t launch_gather_cells(const T* in, int64_t rows, int64_t ld_in, const int64_t* index, int64_t n,
                               T* out, int64_t ld_out, hipStream_t stream);
hipError_t launch_scatter_cells(const double* in, int64_t rows, int64_t ld_in, const int64_t* index,
                                int64_t n, double* out, int64_t ld_out, int64_t ncols_out,
                                hipStream_t stream);

// detect() front end: exceedance + event filter + gap joining, thread per cell
template <typename T>
hipError_t launch_detect(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                         const int32_t* row_of_t, int32_t min_duration, int32_t join_gaps, int32_t max_gap,
                         int32_t negate, int32_t* events, int32_t* start, int32_t* end, uint8_t* bthresh,
                         int64_t ldo, int32_t* nevents, hipStream_t stream);

hipError_t launch_count_events(const int32_t* start, int64_t Tn, int64_t C, int64_t ldo, int32_t* nevents,
                               hipStream_t stream);

// per-event statistics into a compact table (kEventColumns doubles per event)
constexpr int kEventColumns = 31;
template <typename T>
hipError_t launch_event_stats(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* seas,
                              const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                              const int32_t* events, int64_t ldo, const int64_t* offsets, double* table,
                              hipStream_t stream);

// table-only define_events() (kernels_events.hip): exceedance bits, run walk, per-event statistics
hipError_t launch_floor_to_f32(const double* th, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo,
                               hipStream_t stream);
template <typename T, typename TH>
hipError_t launch_exceed_bits(const T* ts, int64_t Tn, int64_t C, int64_t ld, const TH* thresh, int64_t ldt,
                              const int32_t* row_of_t, int32_t negate, uint64_t* bits, int64_t ldb,
                              hipStream_t stream);
template <typename T, typename TH, int TILE>
hipError_t launch_exceed_bits_tiled(const T* ts, int64_t C, int64_t ld, const TH* thresh, int64_t ldt, int64_t D,
                                    const int32_t* tile_begin, int32_t ntiles, const int32_t* chunk_t0,
                                    const int32_t* chunk_i0, const int32_t* chunk_n, int32_t negate, uint64_t* bits,
                                    int64_t ldb, hipStream_t stream);
hipError_t launch_events_from_bits(const uint64_t* bits, int64_t Tn, int64_t C, int64_t ldb, int32_t min_duration,
                                   int32_t join_gaps, int32_t max_gap, const int64_t* offsets, int32_t* nevents,
                                   double* table, hipStream_t stream);
template <typename T>
hipError_t launch_event_stats_sparse(const T* ts, int64_t Tn, int64_t ld, const double* seas, const double* thresh,
                                     int64_t ldc, const int32_t* row_of_t, int32_t negate, int64_t n_events,
                                     double* table, hipStream_t stream);

// exclusive prefix sum of per-cell event counts into int64 table offsets [n+1]; block_sums: scratch of
// (n + 1023) / 1024 + 1 int64
hipError_t launch_offsets_from_counts(const int32_t* counts, int64_t n, int64_t* offsets, int64_t* block_sums,
                                      hipStream_t stream);

// per-step columns of mhw_df(): out [8][T][ldv] f64, dur [4][T][ldv] u8
template <typename T>
hipError_t launch_event_intermediate(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* seas,
                                     const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                                     const int32_t* events, int64_t ldo, double* out, int64_t ldv, uint8_t* dur,
                                     hipStream_t stream);

// block_average() (kernels_stats.hip): segmented reductions keyed by (cell, year bin); out[stat][bin][cell]
constexpr int kBlockEventStats = 15;
hipError_t launch_block_events(const double* table, const int64_t* offsets, int64_t C, const int32_t* bin_of_t, int64_t Tn,
                               int32_t nbins, int32_t mtime_col, double* out, int64_t ldo, hipStream_t stream);
template <typename T>
hipError_t launch_block_time(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* cats, int64_t ldcat,
                             const int32_t* bin_of_t, int32_t nbins, double* out, int64_t ldo, hipStream_t stream);

// file bytes -> samples (kernels_ingest.hip): raw_type = item size of the stored type (2 int16, 4 float32,
// 8 float64), swap = the file is big-endian, optional scale/offset (CF packing) and fill value -> NaN
hipError_t launch_encode_i16(const float* in, int64_t rows, int64_t cols, int64_t ld_in, int16_t* out, int64_t ld_out,
                             double scale, double offset, int32_t fill, hipStream_t stream);
hipError_t launch_decode(const void* in, int raw_type, int swap, int64_t rows, int64_t cols, int64_t ld_in, void* out,
                         int out_itemsize, int64_t ld_out, double scale, double offset, int has_scale, int has_fill,
                         double fill, hipStream_t stream);

// ts.interpolate_na(dim=tdim, max_gap=...) on the device copy of the series, in place (kernels_ingest.hip)
hipError_t launch_pad_gaps(void* ts, int itemsize, int64_t Tn, int64_t C, int64_t ld, const double* x, double max_gap,
                           hipStream_t stream);

template <typename T>
hipError_t launch_synth(T* ts, int64_t Tn, int64_t C, int64_t ld, int64_t cell0, uint64_t seed,
                        double nan_frac, hipStream_t stream);

template <typename T>
hipError_t launch_synth_ex(T* ts, int64_t Tn, int64_t C, int64_t ld, int64_t cell0, uint64_t seed, double nan_frac,
                           double quant, double ice_frac, double rho, int64_t ice_patch, hipStream_t stream);

}  // namespace xmhw
