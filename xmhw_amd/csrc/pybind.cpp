// pybind.cpp -- thin pybind11 layer over the C ABI (include/xmhw_amd.h).
// No logic lives here: argument unpacking, error code -> exception.
#include <pybind11/numpy.h>
#include <algorithm>
#include <pybind11/pybind11.h>

#include <cstdint>
#include <stdexcept>
#include <string>

#include "../../include/xmhw_amd.h"

namespace py = pybind11;

namespace {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct InvalidError : std::runtime_error { using std::runtime_error::runtime_error; };
struct CommError : std::runtime_error { using std::runtime_error::runtime_error; };

void check(int rc) {
    if (rc == XMHW_OK) return;
    std::string msg = xmhw_last_error();
    if (rc == XMHW_ERR_INVALID) throw InvalidError(msg);
    if (rc == XMHW_ERR_NOMEM) throw std::bad_alloc();
    if (rc == XMHW_ERR_COMM) throw CommError(msg);
    throw HipError(msg + " (code " + std::to_string(rc) + ")");
}

inline void* vp(uintptr_t p) { return reinterpret_cast<void*>(p); }
inline xmhw_plan* pp(uintptr_t p) { return reinterpret_cast<xmhw_plan*>(p); }

using i32arr = py::array_t<int32_t, py::array::c_style | py::array::forcecast>;

}  // namespace

PYBIND11_MODULE(_xmhw_hip, m) {
    m.doc() = "C-ABI bindings of the gfx950 xmhw threshold() path";
    py::register_exception<HipError>(m, "HipError");
    py::register_exception<InvalidError>(m, "InvalidArgument");
    py::register_exception<CommError>(m, "CommError");

    m.attr("KERNEL_AUTO") = XMHW_KERNEL_AUTO;
    m.attr("KERNEL_RING") = XMHW_KERNEL_RING;
    m.attr("KERNEL_GENERIC") = XMHW_KERNEL_GENERIC;

    m.def("version", &xmhw_version);
    m.def("arch", []() { return std::string(xmhw_arch()); });
    m.def("device_count", []() { int n = 0; check(xmhw_device_count(&n)); return n; });
    m.def("set_device", [](int d) { check(xmhw_set_device(d)); });
    m.def("get_device", []() { int d = 0; check(xmhw_get_device(&d)); return d; });
    m.def("device_info", [](int d) {
        char name[256] = {0};
        int cus = 0;
        uint64_t bytes = 0;
        check(xmhw_device_info(d, name, sizeof(name), &cus, &bytes));
        py::dict r;
        r["name"] = std::string(name);
        r["compute_units"] = cus;
        r["hbm_bytes"] = bytes;
        return r;
    });

    m.def("malloc", [](size_t n) { void* p = nullptr; check(xmhw_malloc(&p, n)); return reinterpret_cast<uintptr_t>(p); });
    m.def("free", [](uintptr_t p) { check(xmhw_free(vp(p))); });
    m.def("memcpy_h2d", [](uintptr_t dst, py::buffer src, uintptr_t stream) {
        py::buffer_info bi = src.request();
        check(xmhw_memcpy_h2d(vp(dst), bi.ptr, static_cast<size_t>(bi.size) * bi.itemsize, vp(stream)));
    }, py::arg("dst"), py::arg("src"), py::arg("stream") = 0);
    m.def("memcpy2d_h2d", [](uintptr_t dst, py::buffer src, int64_t col0, int64_t ncols, uintptr_t stream) {
        // columns [col0, col0 + ncols) of a C-contiguous 2-D host array -> dense (rows, ncols) device array
        py::buffer_info bi = src.request();
        // rows must be contiguous; the row pitch is free (record variables of a netCDF classic file are
        // interleaved with the other record variables: rows are one record apart)
        if (bi.ndim != 2 || bi.strides[1] != bi.itemsize || bi.strides[0] < bi.itemsize * bi.shape[1])
            throw InvalidError("memcpy2d_h2d needs a 2-D array with contiguous rows");
        if (col0 < 0 || ncols < 0 || col0 + ncols > bi.shape[1]) throw InvalidError("column range outside the array");
        const size_t isz = static_cast<size_t>(bi.itemsize);
        const size_t spitch = static_cast<size_t>(bi.strides[0]);
        py::gil_scoped_release r;
        check(xmhw_memcpy2d_h2d(vp(dst), isz * static_cast<size_t>(ncols), static_cast<const char*>(bi.ptr) + isz * col0,
                                spitch, isz * static_cast<size_t>(ncols), static_cast<size_t>(bi.shape[0]), vp(stream)));
    }, py::arg("dst"), py::arg("src"), py::arg("col0"), py::arg("ncols"), py::arg("stream") = 0);
    m.def("memcpy_d2h", [](py::buffer dst, uintptr_t src, uintptr_t stream) {
        py::buffer_info bi = dst.request(true);
        py::gil_scoped_release r;        // a large copy must not stall the thread that feeds the next slab
        check(xmhw_memcpy_d2h(bi.ptr, vp(src), static_cast<size_t>(bi.size) * bi.itemsize, vp(stream)));
    }, py::arg("dst"), py::arg("src"), py::arg("stream") = 0);
    m.def("memcpy_d2h_bytes", [](py::buffer dst, uintptr_t src, size_t nbytes, uintptr_t stream) {
        py::buffer_info bi = dst.request(true);
        if (nbytes > static_cast<size_t>(bi.size) * bi.itemsize) throw InvalidError("destination too small");
        check(xmhw_memcpy_d2h(bi.ptr, vp(src), nbytes, vp(stream)));
    }, py::arg("dst"), py::arg("src"), py::arg("nbytes"), py::arg("stream") = 0);
    m.def("memset", [](uintptr_t dst, int v, size_t n, uintptr_t stream) { check(xmhw_memset(vp(dst), v, n, vp(stream))); },
          py::arg("dst"), py::arg("value"), py::arg("nbytes"), py::arg("stream") = 0);
    m.def("stream_create", []() { void* s = nullptr; check(xmhw_stream_create(&s)); return reinterpret_cast<uintptr_t>(s); });
    m.def("stream_destroy", [](uintptr_t s) { check(xmhw_stream_destroy(vp(s))); });
    m.def("stream_sync", [](uintptr_t s) { py::gil_scoped_release r; check(xmhw_stream_sync(vp(s))); }, py::arg("stream") = 0);
    m.def("event_create", []() { void* e = nullptr; check(xmhw_event_create(&e)); return reinterpret_cast<uintptr_t>(e); });
    m.def("event_destroy", [](uintptr_t e) { check(xmhw_event_destroy(vp(e))); });
    m.def("event_record", [](uintptr_t e, uintptr_t s) { check(xmhw_event_record(vp(e), vp(s))); }, py::arg("event"), py::arg("stream") = 0);
    m.def("stream_wait_event", [](uintptr_t s, uintptr_t e) { check(xmhw_stream_wait_event(vp(s), vp(e))); });
    m.def("event_elapsed_ms", [](uintptr_t a, uintptr_t b) { float ms = 0; check(xmhw_event_elapsed_ms(vp(a), vp(b), &ms)); return ms; });

    m.def("plan_create", [](i32arr doy, int w) {
        xmhw_plan* p = nullptr;
        check(xmhw_plan_create(doy.data(), doy.size(), w, &p));
        return reinterpret_cast<uintptr_t>(p);
    });
    m.def("plan_destroy", [](uintptr_t p) { check(xmhw_plan_destroy(pp(p))); });
    m.def("plan_info", [](uintptr_t p) {
        int32_t D, nt, k, ns, smin;
        check(xmhw_plan_info(pp(p), &D, &nt, &k, &ns, &smin));
        py::dict r;
        r["D"] = D; r["ntracks"] = nt; r["kernel"] = k; r["nsteps"] = ns; r["step_min"] = smin;
        return r;
    });
    m.def("plan_doys", [](uintptr_t p) {
        int32_t D;
        check(xmhw_plan_info(pp(p), &D, nullptr, nullptr, nullptr, nullptr));
        py::array_t<int32_t> out(D);
        check(xmhw_plan_doys(pp(p), out.mutable_data()));
        return out;
    });
    m.def("plan_set_kernel", [](uintptr_t p, int k) { check(xmhw_plan_set_kernel(pp(p), k)); });
    m.def("plan_set_narrowing", [](uintptr_t p, int on) { check(xmhw_plan_set_narrowing(pp(p), on)); });
    m.def("plan_narrowed", [](uintptr_t p) { int32_t n = 0; check(xmhw_plan_narrowed(pp(p), &n)); return n != 0; });
    m.def("plan_set_chunks", [](uintptr_t p, int n) { check(xmhw_plan_set_chunks(pp(p), n)); });
    m.def("plan_table", [](uintptr_t p, int yps) {
        int32_t ns, ntp;
        check(xmhw_plan_info(pp(p), nullptr, nullptr, nullptr, &ns, nullptr));
        check(xmhw_plan_table(pp(p), yps, nullptr, &ntp));
        py::array_t<uint32_t> out({static_cast<py::ssize_t>(ns), static_cast<py::ssize_t>(ntp)});
        check(xmhw_plan_table(pp(p), yps, out.mutable_data(), &ntp));
        return out;
    });

    m.def("plan_set_timing", [](uintptr_t p, int on) { check(xmhw_plan_set_timing(pp(p), on)); });
    m.def("plan_kernel_ms", [](uintptr_t p, int back) { float ms = 0; check(xmhw_plan_kernel_ms(pp(p), back, &ms)); return ms; });
    m.def("plan_sorted_info", [](uintptr_t p, int64_t C) {
        int32_t k = 0, lds = 0, pieces = 0;
        check(xmhw_plan_sorted_info(pp(p), C, &k, &lds, &pieces));
        return py::make_tuple(k, lds, pieces);
    });
    m.def("plan_sorted_table", [](uintptr_t p, int pieces) {
        int32_t nc = 0, nr = 0, ntp = 0;
        check(xmhw_plan_sorted_table(pp(p), pieces, &nc, &nr, &ntp, nullptr, nullptr, nullptr));
        py::array_t<int32_t> chunks({static_cast<py::ssize_t>(nc), static_cast<py::ssize_t>(4)});
        py::array_t<uint32_t> table({static_cast<py::ssize_t>(nr), static_cast<py::ssize_t>(ntp)});
        py::array_t<uint32_t> flags(static_cast<py::ssize_t>(nr));
        check(xmhw_plan_sorted_table(pp(p), pieces, &nc, &nr, &ntp, chunks.mutable_data(), table.mutable_data(),
                                     flags.mutable_data()));
        return py::make_tuple(chunks, table, flags);
    });

    m.def("plan_debug_stats", [](uintptr_t p, int enable, bool read) {
        py::array_t<uint64_t> out(16);
        std::fill(out.mutable_data(), out.mutable_data() + 16, uint64_t(0));
        check(xmhw_plan_debug_stats_n(pp(p), enable, read ? out.mutable_data() : nullptr, 16));
        return out;
    });
    m.def("plan_chunks_in_use", [](uintptr_t p, int64_t C) { int32_t v = 0; check(xmhw_plan_chunks_in_use(pp(p), C, &v)); return v; });
    m.def("debug_stats_available", []() { return xmhw_debug_stats_available() != 0; });
    m.def("plan_set_layout", [](uintptr_t p, int layout) { check(xmhw_plan_set_layout(pp(p), layout)); });
    m.def("plan_layout_in_use", [](uintptr_t p) { int32_t v = -1; check(xmhw_plan_layout_in_use(pp(p), &v)); return v; });
    m.def("sorted_device_ok", []() { int32_t v = 0; check(xmhw_sorted_device_ok(&v)); return v != 0; });
    m.def("plan_ring2_in_use", [](uintptr_t p) { int32_t v = -1; check(xmhw_plan_ring2_in_use(pp(p), &v)); return v; });
    m.def("plan_f64_mode", [](uintptr_t p) { int32_t v = -1; check(xmhw_plan_f64_mode(pp(p), &v)); return v; });
    m.def("plan_set_ring2", [](uintptr_t p, int variant) { check(xmhw_plan_set_ring2(pp(p), variant)); });

    m.def("clim_raw", [](uintptr_t plan, uintptr_t ts, int itemsize, int64_t C, int64_t ld, double q, int negate,
                         uintptr_t th, uintptr_t se, int64_t ldo, uintptr_t stream) {
        if (itemsize == 4)
            check(xmhw_clim_raw_f32(pp(plan), static_cast<const float*>(vp(ts)), C, ld, q, negate,
                                    static_cast<double*>(vp(th)), static_cast<double*>(vp(se)), ldo, vp(stream)));
        else if (itemsize == 8)
            check(xmhw_clim_raw_f64(pp(plan), static_cast<const double*>(vp(ts)), C, ld, q, negate,
                                    static_cast<double*>(vp(th)), static_cast<double*>(vp(se)), ldo, vp(stream)));
        else throw InvalidError("itemsize must be 4 or 8");
    }, py::arg("plan"), py::arg("ts"), py::arg("itemsize"), py::arg("C"), py::arg("ld"), py::arg("q"),
       py::arg("negate"), py::arg("thresh"), py::arg("seas"), py::arg("ldo"), py::arg("stream") = 0);

    m.def("clim_raw_i16", [](uintptr_t plan, uintptr_t codes, int64_t C, int64_t ld, int big_endian, int has_scale,
                             double scale_factor, double add_offset, int has_fill, int32_t fill_code, int decoded_itemsize,
                             double q, int negate, uintptr_t th, uintptr_t se, int64_t ldo, uintptr_t stream) {
        check(xmhw_clim_raw_i16(pp(plan), static_cast<const int16_t*>(vp(codes)), C, ld, big_endian, has_scale, scale_factor,
                                add_offset, has_fill, fill_code, decoded_itemsize, q, negate, static_cast<double*>(vp(th)),
                                static_cast<double*>(vp(se)), ldo, vp(stream)));
    }, py::arg("plan"), py::arg("codes"), py::arg("C"), py::arg("ld"), py::arg("big_endian"), py::arg("has_scale"),
       py::arg("scale_factor"), py::arg("add_offset"), py::arg("has_fill"), py::arg("fill_code"), py::arg("decoded_itemsize"),
       py::arg("q"), py::arg("negate"), py::arg("thresh"), py::arg("seas"), py::arg("ldo"), py::arg("stream") = 0);

    m.def("clim_finish", [](uintptr_t plan, uintptr_t th_in, uintptr_t se_in, int64_t C, int64_t ldo, int feb29_fix,
                            int smooth, int width, uintptr_t th_out, uintptr_t se_out, uintptr_t stream) {
        check(xmhw_clim_finish(pp(plan), static_cast<const double*>(vp(th_in)), static_cast<const double*>(vp(se_in)),
                               C, ldo, feb29_fix, smooth, width, static_cast<double*>(vp(th_out)),
                               static_cast<double*>(vp(se_out)), vp(stream)));
    }, py::arg("plan"), py::arg("thresh_in"), py::arg("seas_in"), py::arg("C"), py::arg("ldo"),
       py::arg("feb29_fix"), py::arg("smooth"), py::arg("width"), py::arg("thresh_out"), py::arg("seas_out"),
       py::arg("stream") = 0);

    m.def("clim_dev", [](uintptr_t ts, int itemsize, i32arr doy, int64_t C, int D, int w, double q, int smooth,
                         int smooth_w, int feb29_fix, int negate, uintptr_t th, uintptr_t se, uintptr_t stream) {
        py::gil_scoped_release r;
        if (itemsize == 4)
            check(xmhw_clim_f32(static_cast<const float*>(vp(ts)), doy.data(), doy.size(), C, D, w, q, smooth, smooth_w,
                                feb29_fix, negate, static_cast<double*>(vp(th)), static_cast<double*>(vp(se)), vp(stream)));
        else if (itemsize == 8)
            check(xmhw_clim_f64(static_cast<const double*>(vp(ts)), doy.data(), doy.size(), C, D, w, q, smooth, smooth_w,
                                feb29_fix, negate, static_cast<double*>(vp(th)), static_cast<double*>(vp(se)), vp(stream)));
        else throw InvalidError("itemsize must be 4 or 8");
    });

    m.def("clim_host", [](py::array ts, i32arr doy, int D, int w, double q, int smooth, int smooth_w, int feb29_fix,
                          int negate) {
        if (ts.ndim() != 2) throw InvalidError("ts must be (T, C)");
        if (!(ts.flags() & py::array::c_style)) throw InvalidError("ts must be C-contiguous");
        if (ts.shape(0) != doy.size()) throw InvalidError("doy length must equal T");
        const int64_t T = ts.shape(0), C = ts.shape(1);
        py::array_t<double> th({static_cast<py::ssize_t>(D), static_cast<py::ssize_t>(C)});
        py::array_t<double> se({static_cast<py::ssize_t>(D), static_cast<py::ssize_t>(C)});
        int rc;
        if (ts.dtype().is(py::dtype::of<float>())) {
            py::gil_scoped_release r;
            rc = xmhw_clim_host_f32(static_cast<const float*>(ts.data()), doy.data(), T, C, D, w, q, smooth, smooth_w,
                                    feb29_fix, negate, th.mutable_data(), se.mutable_data());
        } else if (ts.dtype().is(py::dtype::of<double>())) {
            py::gil_scoped_release r;
            rc = xmhw_clim_host_f64(static_cast<const double*>(ts.data()), doy.data(), T, C, D, w, q, smooth, smooth_w,
                                    feb29_fix, negate, th.mutable_data(), se.mutable_data());
        } else {
            throw InvalidError("ts dtype must be float32 or float64");
        }
        check(rc);
        return py::make_tuple(th, se);
    });

    m.def("land_mask", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, int anynans, uintptr_t keep,
                          uintptr_t stream) {
        if (itemsize == 4)
            check(xmhw_land_mask_f32(static_cast<const float*>(vp(ts)), T, C, ld, anynans, static_cast<uint8_t*>(vp(keep)), vp(stream)));
        else if (itemsize == 8)
            check(xmhw_land_mask_f64(static_cast<const double*>(vp(ts)), T, C, ld, anynans, static_cast<uint8_t*>(vp(keep)), vp(stream)));
        else throw InvalidError("itemsize must be 4 or 8");
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("anynans"),
       py::arg("keep"), py::arg("stream") = 0);
    m.def("land_mask_i16", [](uintptr_t codes, int64_t T, int64_t C, int64_t ld, int big_endian, int has_fill, int32_t fill_code,
                              int anynans, uintptr_t keep, uintptr_t stream) {
        check(xmhw_land_mask_i16(static_cast<const int16_t*>(vp(codes)), T, C, ld, big_endian, has_fill, fill_code, anynans,
                                 static_cast<uint8_t*>(vp(keep)), vp(stream)));
    }, py::arg("codes"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("big_endian"), py::arg("has_fill"),
       py::arg("fill_code"), py::arg("anynans"), py::arg("keep"), py::arg("stream") = 0);

    m.def("gather_cells", [](uintptr_t in, int itemsize, int64_t rows, int64_t ld_in, uintptr_t index, int64_t n,
                             uintptr_t out, int64_t ld_out, uintptr_t stream) {
        if (itemsize == 4)
            check(xmhw_gather_cells_f32(static_cast<const float*>(vp(in)), rows, ld_in, static_cast<const int64_t*>(vp(index)),
                                        n, static_cast<float*>(vp(out)), ld_out, vp(stream)));
        else if (itemsize == 8)
            check(xmhw_gather_cells_f64(static_cast<const double*>(vp(in)), rows, ld_in, static_cast<const int64_t*>(vp(index)),
                                        n, static_cast<double*>(vp(out)), ld_out, vp(stream)));
        else if (itemsize == 2)
            check(xmhw_gather_cells_i16(static_cast<const int16_t*>(vp(in)), rows, ld_in, static_cast<const int64_t*>(vp(index)),
                                        n, static_cast<int16_t*>(vp(out)), ld_out, vp(stream)));
        else throw InvalidError("itemsize must be 2, 4 or 8");
    }, py::arg("in"), py::arg("itemsize"), py::arg("rows"), py::arg("ld_in"), py::arg("index"), py::arg("n"),
       py::arg("out"), py::arg("ld_out"), py::arg("stream") = 0);
    m.def("scatter_cells", [](uintptr_t in, int64_t rows, int64_t ld_in, uintptr_t index, int64_t n, uintptr_t out,
                              int64_t ld_out, uintptr_t stream) {
        check(xmhw_scatter_cells_f64(static_cast<const double*>(vp(in)), rows, ld_in, static_cast<const int64_t*>(vp(index)),
                                     n, static_cast<double*>(vp(out)), ld_out, vp(stream)));
    }, py::arg("in"), py::arg("rows"), py::arg("ld_in"), py::arg("index"), py::arg("n"), py::arg("out"),
       py::arg("ld_out"), py::arg("stream") = 0);

    m.def("detect_events", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, uintptr_t thresh, int64_t ldt,
                              i32arr row_of_t, int min_duration, int join_gaps, int max_gap, int negate,
                              uintptr_t events, uintptr_t start, uintptr_t end, uintptr_t bthresh, int64_t ldo,
                              uintptr_t nevents, uintptr_t stream) {
        if (row_of_t.size() != T) throw InvalidError("row_of_t length must equal T");
        py::gil_scoped_release r;
        int rc;
        if (itemsize == 4)
            rc = xmhw_detect_events_f32(static_cast<const float*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(thresh)), ldt,
                                        row_of_t.data(), min_duration, join_gaps, max_gap, negate,
                                        static_cast<int32_t*>(vp(events)), static_cast<int32_t*>(vp(start)),
                                        static_cast<int32_t*>(vp(end)), static_cast<uint8_t*>(vp(bthresh)), ldo, static_cast<int32_t*>(vp(nevents)), vp(stream));
        else if (itemsize == 8)
            rc = xmhw_detect_events_f64(static_cast<const double*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(thresh)), ldt,
                                        row_of_t.data(), min_duration, join_gaps, max_gap, negate,
                                        static_cast<int32_t*>(vp(events)), static_cast<int32_t*>(vp(start)),
                                        static_cast<int32_t*>(vp(end)), static_cast<uint8_t*>(vp(bthresh)), ldo, static_cast<int32_t*>(vp(nevents)), vp(stream));
        else
            rc = -1;
        if (rc == -1) throw InvalidError("itemsize must be 4 or 8");
        check(rc);
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("thresh"), py::arg("ldt"),
       py::arg("row_of_t"), py::arg("min_duration"), py::arg("join_gaps"), py::arg("max_gap"), py::arg("negate"),
       py::arg("events"), py::arg("start"), py::arg("end"), py::arg("bthresh") = 0, py::arg("ldo"), py::arg("nevents") = 0,
       py::arg("stream") = 0);

    m.def("count_events", [](uintptr_t start, int64_t T, int64_t C, int64_t ldo, uintptr_t nevents, uintptr_t stream) {
        check(xmhw_count_events(static_cast<const int32_t*>(vp(start)), T, C, ldo, static_cast<int32_t*>(vp(nevents)), vp(stream)));
    }, py::arg("start"), py::arg("T"), py::arg("C"), py::arg("ldo"), py::arg("nevents"), py::arg("stream") = 0);
    m.attr("EVENT_COLUMNS") = XMHW_EVENT_COLUMNS;
    m.def("event_stats", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, uintptr_t seas, uintptr_t thresh,
                            int64_t ldc, i32arr row_of_t, int negate, uintptr_t events, int64_t ldo, uintptr_t offsets,
                            uintptr_t table, uintptr_t stream) {
        if (row_of_t.size() != T) throw InvalidError("row_of_t length must equal T");
        py::gil_scoped_release r;
        int rc = -1;
        if (itemsize == 4)
            rc = xmhw_event_stats_f32(static_cast<const float*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(seas)),
                                      static_cast<const double*>(vp(thresh)), ldc, row_of_t.data(), negate,
                                      static_cast<const int32_t*>(vp(events)), ldo, static_cast<const int64_t*>(vp(offsets)),
                                      static_cast<double*>(vp(table)), vp(stream));
        else if (itemsize == 8)
            rc = xmhw_event_stats_f64(static_cast<const double*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(seas)),
                                      static_cast<const double*>(vp(thresh)), ldc, row_of_t.data(), negate,
                                      static_cast<const int32_t*>(vp(events)), ldo, static_cast<const int64_t*>(vp(offsets)),
                                      static_cast<double*>(vp(table)), vp(stream));
        if (rc == -1) throw InvalidError("itemsize must be 4 or 8");
        check(rc);
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("seas"), py::arg("thresh"),
       py::arg("ldc"), py::arg("row_of_t"), py::arg("negate"), py::arg("events"), py::arg("ldo"), py::arg("offsets"),
       py::arg("table"), py::arg("stream") = 0);

    m.def("set_exceed_kernel", [](int mode) { check(xmhw_set_exceed_kernel(mode)); }, py::arg("mode"));
    m.def("exceed_bits", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, uintptr_t thresh, int64_t ldt,
                            int64_t D, i32arr row_of_t, int negate, uintptr_t bits, int64_t ldb, uintptr_t stream) {
        if (row_of_t.size() != T) throw InvalidError("row_of_t length must equal T");
        py::gil_scoped_release r;
        int rc = -1;
        if (itemsize == 4)
            rc = xmhw_exceed_bits_f32(static_cast<const float*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(thresh)), ldt,
                                      D, row_of_t.data(), negate, static_cast<uint64_t*>(vp(bits)), ldb, vp(stream));
        else if (itemsize == 8)
            rc = xmhw_exceed_bits_f64(static_cast<const double*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(thresh)), ldt,
                                      D, row_of_t.data(), negate, static_cast<uint64_t*>(vp(bits)), ldb, vp(stream));
        if (rc == -1) throw InvalidError("itemsize must be 4 or 8");
        check(rc);
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("thresh"), py::arg("ldt"),
       py::arg("D"), py::arg("row_of_t"), py::arg("negate"), py::arg("bits"), py::arg("ldb"), py::arg("stream") = 0);
    m.def("events_from_bits", [](uintptr_t bits, int64_t T, int64_t C, int64_t ldb, int min_duration, int join_gaps,
                                 int max_gap, uintptr_t offsets, uintptr_t nevents, uintptr_t table, uintptr_t stream) {
        check(xmhw_events_from_bits(static_cast<const uint64_t*>(vp(bits)), T, C, ldb, min_duration, join_gaps, max_gap,
                                    static_cast<const int64_t*>(vp(offsets)), static_cast<int32_t*>(vp(nevents)),
                                    static_cast<double*>(vp(table)), vp(stream)));
    }, py::arg("bits"), py::arg("T"), py::arg("C"), py::arg("ldb"), py::arg("min_duration"), py::arg("join_gaps"),
       py::arg("max_gap"), py::arg("offsets") = 0, py::arg("nevents") = 0, py::arg("table") = 0, py::arg("stream") = 0);
    m.def("event_stats_sparse", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, uintptr_t seas,
                                   uintptr_t thresh, int64_t ldc, i32arr row_of_t, int negate, int64_t n_events,
                                   uintptr_t table, uintptr_t stream) {
        if (row_of_t.size() != T) throw InvalidError("row_of_t length must equal T");
        py::gil_scoped_release r;
        int rc = -1;
        if (itemsize == 4)
            rc = xmhw_event_stats_sparse_f32(static_cast<const float*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(seas)),
                                             static_cast<const double*>(vp(thresh)), ldc, row_of_t.data(), negate, n_events,
                                             static_cast<double*>(vp(table)), vp(stream));
        else if (itemsize == 8)
            rc = xmhw_event_stats_sparse_f64(static_cast<const double*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(seas)),
                                             static_cast<const double*>(vp(thresh)), ldc, row_of_t.data(), negate, n_events,
                                             static_cast<double*>(vp(table)), vp(stream));
        if (rc == -1) throw InvalidError("itemsize must be 4 or 8");
        check(rc);
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("seas"), py::arg("thresh"),
       py::arg("ldc"), py::arg("row_of_t"), py::arg("negate"), py::arg("n_events"), py::arg("table"), py::arg("stream") = 0);

    m.def("event_intermediate", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, uintptr_t seas,
                                   uintptr_t thresh, int64_t ldc, i32arr row_of_t, int negate, uintptr_t events,
                                   int64_t ldo, uintptr_t out, int64_t ldv, uintptr_t dur, uintptr_t stream) {
        if (row_of_t.size() != T) throw InvalidError("row_of_t length must equal T");
        py::gil_scoped_release r;
        int rc = -1;
        if (itemsize == 4)
            rc = xmhw_event_intermediate_f32(static_cast<const float*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(seas)),
                                             static_cast<const double*>(vp(thresh)), ldc, row_of_t.data(), negate,
                                             static_cast<const int32_t*>(vp(events)), ldo, static_cast<double*>(vp(out)), ldv,
                                             static_cast<uint8_t*>(vp(dur)), vp(stream));
        else if (itemsize == 8)
            rc = xmhw_event_intermediate_f64(static_cast<const double*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(seas)),
                                             static_cast<const double*>(vp(thresh)), ldc, row_of_t.data(), negate,
                                             static_cast<const int32_t*>(vp(events)), ldo, static_cast<double*>(vp(out)), ldv,
                                             static_cast<uint8_t*>(vp(dur)), vp(stream));
        if (rc == -1) throw InvalidError("itemsize must be 4 or 8");
        check(rc);
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("seas"), py::arg("thresh"),
       py::arg("ldc"), py::arg("row_of_t"), py::arg("negate"), py::arg("events"), py::arg("ldo"), py::arg("out"),
       py::arg("ldv"), py::arg("dur"), py::arg("stream") = 0);

    // ---- sharded path: RCCL communicator + the one gather ---------------------------------
    m.def("comm_unique_id", []() {
        std::string id(XMHW_UNIQUE_ID_BYTES, '\0');
        check(xmhw_comm_unique_id(&id[0]));
        return py::bytes(id);
    });
    m.def("comm_create", [](int rank, int nranks, py::bytes id) {
        std::string sid = id;
        if (sid.size() != XMHW_UNIQUE_ID_BYTES) throw InvalidError("unique id must be 128 bytes");
        xmhw_comm* c = nullptr;
        {
            py::gil_scoped_release r;
            check(xmhw_comm_create(rank, nranks, sid.data(), &c));
        }
        return reinterpret_cast<uintptr_t>(c);
    });
    m.def("comm_destroy", [](uintptr_t c) { check(xmhw_comm_destroy(reinterpret_cast<xmhw_comm*>(c))); });
    m.def("comm_allgather_i64_begin", [](uintptr_t c, int64_t value, uintptr_t stream) {
        check(xmhw_comm_allgather_i64_begin(reinterpret_cast<xmhw_comm*>(c), value, vp(stream)));
    }, py::arg("comm"), py::arg("value"), py::arg("stream") = 0);
    m.def("comm_allgather_i64_end", [](uintptr_t c) {
        int rank = 0, n = 1;
        check(xmhw_comm_info(reinterpret_cast<xmhw_comm*>(c), &rank, &n));
        py::array_t<int64_t> out(n);
        {
            py::gil_scoped_release r;
            int rc = xmhw_comm_allgather_i64_end(reinterpret_cast<xmhw_comm*>(c), out.mutable_data());
            py::gil_scoped_acquire a;
            check(rc);
        }
        return out;
    });
    m.def("comm_allgather_i64", [](uintptr_t c, int64_t value, uintptr_t stream) {
        xmhw_comm* cc = reinterpret_cast<xmhw_comm*>(c);
        int rank = 0, n = 0;
        check(xmhw_comm_info(cc, &rank, &n));
        py::array_t<int64_t> out(n);
        {
            py::gil_scoped_release r;
            check(xmhw_comm_allgather_i64(cc, value, out.mutable_data(), vp(stream)));
        }
        return out;
    }, py::arg("comm"), py::arg("value"), py::arg("stream") = 0);
    m.def("comm_allgather_bytes", [](uintptr_t c, uintptr_t send, uintptr_t recv, size_t n, uintptr_t stream) {
        check(xmhw_comm_allgather_bytes(reinterpret_cast<xmhw_comm*>(c), vp(send), vp(recv), n, vp(stream)));
    }, py::arg("comm"), py::arg("send"), py::arg("recv"), py::arg("bytes_per_rank"), py::arg("stream") = 0);
    m.def("gather_blocks", [](uintptr_t c, uintptr_t send, int64_t rows, int64_t cols, uintptr_t recv,
                              py::array_t<int64_t, py::array::c_style | py::array::forcecast> cols_of_rank, int root,
                              uintptr_t stream) {
        check(xmhw_gather_blocks(reinterpret_cast<xmhw_comm*>(c), static_cast<const double*>(vp(send)), rows, cols,
                                 static_cast<double*>(vp(recv)), cols_of_rank.size() ? cols_of_rank.data() : nullptr, root,
                                 vp(stream)));
    }, py::arg("comm"), py::arg("send"), py::arg("rows"), py::arg("cols"), py::arg("recv"), py::arg("cols_of_rank"),
       py::arg("root") = 0, py::arg("stream") = 0);
    m.def("memcpy2d_d2h", [](py::buffer dst, int64_t col0, int64_t ncols, uintptr_t src, uintptr_t stream) {
        // dense (rows, ncols) device array -> columns [col0, col0 + ncols) of a C-contiguous 2-D host array
        py::buffer_info bi = dst.request(true);
        if (bi.ndim != 2 || bi.strides[1] != bi.itemsize || bi.strides[0] != bi.itemsize * bi.shape[1])
            throw InvalidError("memcpy2d_d2h needs a C-contiguous 2-D array");
        if (col0 < 0 || ncols < 0 || col0 + ncols > bi.shape[1]) throw InvalidError("column range outside the array");
        const size_t isz = static_cast<size_t>(bi.itemsize);
        py::gil_scoped_release r;
        check(xmhw_memcpy2d_d2h(static_cast<char*>(bi.ptr) + isz * col0, isz * static_cast<size_t>(bi.shape[1]), vp(src),
                                isz * static_cast<size_t>(ncols), isz * static_cast<size_t>(ncols),
                                static_cast<size_t>(bi.shape[0]), vp(stream)));
    }, py::arg("dst"), py::arg("col0"), py::arg("ncols"), py::arg("src"), py::arg("stream") = 0);
    m.def("memcpy2d_d2h_async", [](py::buffer dst, int64_t col0, int64_t ncols, uintptr_t src, uintptr_t stream) {
        // dense (rows, ncols) device array -> columns [col0, col0 + ncols) of a C-contiguous 2-D host array
        py::buffer_info bi = dst.request(true);
        if (bi.ndim != 2 || bi.strides[1] != bi.itemsize || bi.strides[0] != bi.itemsize * bi.shape[1])
            throw InvalidError("memcpy2d_d2h_async needs a C-contiguous 2-D array");
        if (col0 < 0 || ncols < 0 || col0 + ncols > bi.shape[1]) throw InvalidError("column range outside the array");
        const size_t isz = static_cast<size_t>(bi.itemsize);
        py::gil_scoped_release r;
        check(xmhw_memcpy2d_d2h_async(static_cast<char*>(bi.ptr) + isz * col0, isz * static_cast<size_t>(bi.shape[1]), vp(src),
                                isz * static_cast<size_t>(ncols), isz * static_cast<size_t>(ncols),
                                static_cast<size_t>(bi.shape[0]), vp(stream)));
    }, py::arg("dst"), py::arg("col0"), py::arg("ncols"), py::arg("src"), py::arg("stream") = 0);

    m.def("decode", [](uintptr_t raw, int raw_itemsize, int big_endian, int64_t rows, int64_t cols, int64_t ld_raw,
                       uintptr_t out, int out_itemsize, int64_t ld_out, bool has_scale, double scale, double offset,
                       bool has_fill, double fill, uintptr_t stream) {
        check(xmhw_decode(vp(raw), raw_itemsize, big_endian, rows, cols, ld_raw, vp(out), out_itemsize, ld_out, has_scale,
                          scale, offset, has_fill, fill, vp(stream)));
    }, py::arg("raw"), py::arg("raw_itemsize"), py::arg("big_endian"), py::arg("rows"), py::arg("cols"), py::arg("ld_raw"),
       py::arg("out"), py::arg("out_itemsize"), py::arg("ld_out"), py::arg("has_scale"), py::arg("scale"),
       py::arg("offset"), py::arg("has_fill"), py::arg("fill"), py::arg("stream") = 0);
    m.def("encode_i16", [](uintptr_t in, int64_t rows, int64_t cols, int64_t ld_in, uintptr_t out, int64_t ld_out, double scale,
                           double offset, int32_t fill, uintptr_t stream) {
        check(xmhw_encode_i16(static_cast<const float*>(vp(in)), rows, cols, ld_in, static_cast<int16_t*>(vp(out)), ld_out, scale,
                              offset, fill, vp(stream)));
    }, py::arg("in"), py::arg("rows"), py::arg("cols"), py::arg("ld_in"), py::arg("out"), py::arg("ld_out"), py::arg("scale"),
       py::arg("offset"), py::arg("fill"), py::arg("stream") = 0);
    m.def("read_rows", [](int fd, int64_t off, int64_t pitch, int64_t row_bytes, int64_t rows, uintptr_t dst) {
        py::gil_scoped_release r;
        int rc = xmhw_read_rows(fd, off, pitch, row_bytes, rows, vp(dst));
        py::gil_scoped_acquire a;
        check(rc);
    }, py::arg("fd"), py::arg("file_offset"), py::arg("row_pitch"), py::arg("row_bytes"), py::arg("rows"), py::arg("dst"));
    m.def("pad_gaps", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, uintptr_t x, double max_gap,
                         uintptr_t stream) {
        check(xmhw_pad_gaps(vp(ts), itemsize, T, C, ld, static_cast<const double*>(vp(x)), max_gap, vp(stream)));
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("x"), py::arg("max_gap"),
       py::arg("stream") = 0);
    m.def("host_alloc", [](size_t n) { void* p = nullptr; check(xmhw_host_alloc(&p, n)); return reinterpret_cast<uintptr_t>(p); });
    m.def("host_free", [](uintptr_t p) { check(xmhw_host_free(vp(p))); });
    m.def("host_view", [](uintptr_t p, size_t nbytes) {
        // a numpy uint8 view of page-locked memory obtained from host_alloc (the caller keeps it alive)
        return py::array_t<uint8_t>({nbytes}, {1}, static_cast<uint8_t*>(vp(p)), py::none());
    });
    m.def("memcpy2d_h2d_async", [](uintptr_t dst, size_t dpitch, uintptr_t src, size_t spitch, size_t width, size_t height,
                                   uintptr_t stream) {
        check(xmhw_memcpy2d_h2d_async(vp(dst), dpitch, vp(src), spitch, width, height, vp(stream)));
    });
    m.def("memcpy_h2d_async", [](uintptr_t dst, uintptr_t src, size_t bytes, uintptr_t stream) {
        check(xmhw_memcpy_h2d_async(vp(dst), vp(src), bytes, vp(stream)));
    });
    m.def("event_sync", [](uintptr_t e) { py::gil_scoped_release r; check(xmhw_event_sync(vp(e))); });
    m.def("memcpy_d2h_async", [](uintptr_t dst, uintptr_t src, size_t bytes, uintptr_t stream) {
        check(xmhw_memcpy_d2h_async(vp(dst), vp(src), bytes, vp(stream)));
    });

    m.def("block_events", [](uintptr_t table, uintptr_t offsets, int64_t C, uintptr_t bin_of_t, int64_t T, int nbins,
                             int mtime_col, uintptr_t out, int64_t ldo, uintptr_t stream) {
        check(xmhw_block_events(static_cast<const double*>(vp(table)), static_cast<const int64_t*>(vp(offsets)), C,
                                static_cast<const int32_t*>(vp(bin_of_t)), T, nbins, mtime_col, static_cast<double*>(vp(out)),
                                ldo, vp(stream)));
    }, py::arg("table"), py::arg("offsets"), py::arg("C"), py::arg("bin_of_t"), py::arg("T"), py::arg("nbins"),
       py::arg("mtime_col"), py::arg("out"), py::arg("ldo"), py::arg("stream") = 0);
    m.def("block_time", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, uintptr_t cats, int64_t ldcat,
                           uintptr_t bin_of_t, int nbins, uintptr_t out, int64_t ldo, uintptr_t stream) {
        if (itemsize == 4)
            check(xmhw_block_time_f32(static_cast<const float*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(cats)), ldcat,
                                      static_cast<const int32_t*>(vp(bin_of_t)), nbins, static_cast<double*>(vp(out)), ldo, vp(stream)));
        else if (itemsize == 8)
            check(xmhw_block_time_f64(static_cast<const double*>(vp(ts)), T, C, ld, static_cast<const double*>(vp(cats)), ldcat,
                                      static_cast<const int32_t*>(vp(bin_of_t)), nbins, static_cast<double*>(vp(out)), ldo, vp(stream)));
        else throw InvalidError("itemsize must be 4 or 8");
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("cats"), py::arg("ldcat"),
       py::arg("bin_of_t"), py::arg("nbins"), py::arg("out"), py::arg("ldo"), py::arg("stream") = 0);

    m.def("release_cached_tables", []() { check(xmhw_release_cached_tables()); });
    m.def("offsets_from_counts", [](uintptr_t counts, int64_t n, uintptr_t offsets, uintptr_t stream) {
        check(xmhw_offsets_from_counts(static_cast<const int32_t*>(vp(counts)), n, static_cast<int64_t*>(vp(offsets)), vp(stream)));
    }, py::arg("counts"), py::arg("n"), py::arg("offsets"), py::arg("stream") = 0);

    m.def("synth_sst_ex", [](uintptr_t ts, int64_t T, int64_t C, int64_t ld, int64_t cell0, uint64_t seed, double nan_frac,
                             double quant, double ice_frac, double rho, int64_t ice_patch, uintptr_t stream) {
        check(xmhw_synth_sst_ex_f32(static_cast<float*>(vp(ts)), T, C, ld, cell0, seed, nan_frac, quant, ice_frac, rho, ice_patch, vp(stream)));
    }, py::arg("ts"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("cell0"), py::arg("seed"), py::arg("nan_frac") = 0.0,
       py::arg("quant") = 0.0, py::arg("ice_frac") = 0.0, py::arg("rho") = 0.0, py::arg("ice_patch") = 0, py::arg("stream") = 0);
    m.def("synth_sst", [](uintptr_t ts, int itemsize, int64_t T, int64_t C, int64_t ld, int64_t cell0, uint64_t seed,
                          double nan_frac, uintptr_t stream) {
        if (itemsize == 4)
            check(xmhw_synth_sst_f32(static_cast<float*>(vp(ts)), T, C, ld, cell0, seed, nan_frac, vp(stream)));
        else if (itemsize == 8)
            check(xmhw_synth_sst_f64(static_cast<double*>(vp(ts)), T, C, ld, cell0, seed, nan_frac, vp(stream)));
        else throw InvalidError("itemsize must be 4 or 8");
    }, py::arg("ts"), py::arg("itemsize"), py::arg("T"), py::arg("C"), py::arg("ld"), py::arg("cell0"),
       py::arg("seed"), py::arg("nan_frac") = 0.0, py::arg("stream") = 0);
}
