// comm.cpp -- the multi-GPU part of the C ABI (include/xmhw_amd.h, "sharded path"): an RCCL
// communicator and the one collective the path needs.
//
// The reference's "collect" is dask.compute(climls) followed by xr.concat(dim='cell')
// (xmhw/xmhw.py:197, :210-211).  Here every rank computes a contiguous block of cells and the
// (rows, cols_r) float64 result blocks travel ONCE, device to device, to the root over xGMI:
// grouped ncclSend / ncclRecv on the kernels' own output buffers (no host bounce, no padding).
//
// RCCL is loaded with dlopen at the first use, so that libxmhw_amd.so has no link-time dependency
// on it: single-GPU users never load it, and a process that already holds an RCCL (for instance
// the one bundled with PyTorch) keeps using that one.
#include "../../include/xmhw_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

int xmhw_set_error_(int code, const std::string& msg);   // capi.cpp

namespace {

// the handful of RCCL entry points and constants used (rccl.h, RCCL 2.x; values are ABI-stable)
typedef struct { char internal[128]; } ncclUniqueId_;
typedef void* ncclComm_;
typedef int ncclResult_;
constexpr int kNcclUint8 = 1, kNcclInt64 = 4, kNcclFloat64 = 8;

struct Rccl {
    void* lib = nullptr;
    ncclResult_ (*GetUniqueId)(ncclUniqueId_*) = nullptr;
    ncclResult_ (*CommInitRank)(ncclComm_*, int, ncclUniqueId_, int) = nullptr;
    ncclResult_ (*CommDestroy)(ncclComm_) = nullptr;
    ncclResult_ (*Send)(const void*, size_t, int, int, ncclComm_, hipStream_t) = nullptr;
    ncclResult_ (*Recv)(void*, size_t, int, int, ncclComm_, hipStream_t) = nullptr;
    ncclResult_ (*AllGather)(const void*, void*, size_t, int, ncclComm_, hipStream_t) = nullptr;
    ncclResult_ (*GroupStart)() = nullptr;
    ncclResult_ (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.error = std::string("cannot load RCCL: ") + dlerror();
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(r.lib, name);
            if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int rccl_fail(ncclResult_ e, const char* what) {
    Rccl* r = rccl();
    return xmhw_set_error_(XMHW_ERR_COMM, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}
#define RCCL_TRY(expr)                                 \
    do {                                               \
        ncclResult_ _e = (expr);                       \
        if (_e != 0) return rccl_fail(_e, #expr);      \
    } while (0)
#define HIPC_TRY(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) return xmhw_set_error_(XMHW_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

}  // namespace

struct xmhw_comm {
    ncclComm_ comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    int64_t* d_scratch = nullptr;     // nranks int64 for the metadata all-gather (+ 1: this rank's value)
    int64_t* h_pinned = nullptr;      // pinned mirror of d_scratch for the begin / end form
    hipEvent_t ev = nullptr;          // completion of a begun all-gather
    bool pending = false;
};

extern "C" {

int xmhw_comm_unique_id(void* id_out) {
    if (!id_out) return xmhw_set_error_(XMHW_ERR_INVALID, "id_out is NULL");
    Rccl* r = rccl();
    if (!r->error.empty()) return xmhw_set_error_(XMHW_ERR_COMM, r->error);
    ncclUniqueId_ id;
    RCCL_TRY(r->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, XMHW_UNIQUE_ID_BYTES);
    return XMHW_OK;
}

int xmhw_comm_create(int rank, int nranks, const void* id, xmhw_comm** comm) {
    if (!comm) return xmhw_set_error_(XMHW_ERR_INVALID, "comm is NULL");
    *comm = nullptr;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return xmhw_set_error_(XMHW_ERR_INVALID, "bad rank/nranks/id");
    Rccl* r = rccl();
    if (!r->error.empty()) return xmhw_set_error_(XMHW_ERR_COMM, r->error);
    xmhw_comm* c = new (std::nothrow) xmhw_comm();
    if (!c) return xmhw_set_error_(XMHW_ERR_NOMEM, "out of host memory");
    c->rank = rank;
    c->nranks = nranks;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return xmhw_set_error_(XMHW_ERR_HIP, "hipGetDevice failed"); }
    ncclUniqueId_ uid;
    std::memcpy(uid.internal, id, XMHW_UNIQUE_ID_BYTES);
    ncclResult_ e = r->CommInitRank(&c->comm, nranks, uid, rank);
    if (e != 0) { delete c; return rccl_fail(e, "ncclCommInitRank"); }
    if (hipMalloc(&c->d_scratch, sizeof(int64_t) * static_cast<size_t>(nranks + 1)) != hipSuccess) {
        r->CommDestroy(c->comm);
        delete c;
        return xmhw_set_error_(XMHW_ERR_NOMEM, "hipMalloc of the communicator scratch failed");
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), sizeof(int64_t) * static_cast<size_t>(nranks + 1),
                      hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) {
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        (void)hipFree(c->d_scratch);
        r->CommDestroy(c->comm);
        delete c;
        return xmhw_set_error_(XMHW_ERR_NOMEM, "pinned scratch / event of the communicator failed");
    }
    *comm = c;
    return XMHW_OK;
}

int xmhw_comm_destroy(xmhw_comm* comm) {
    if (!comm) return XMHW_OK;
    Rccl* r = rccl();
    if (comm->ev) (void)hipEventDestroy(comm->ev);
    if (comm->h_pinned) (void)hipHostFree(comm->h_pinned);
    if (comm->d_scratch) (void)hipFree(comm->d_scratch);
    if (comm->comm && r->CommDestroy) (void)r->CommDestroy(comm->comm);
    delete comm;
    return XMHW_OK;
}

int xmhw_comm_info(const xmhw_comm* comm, int* rank, int* nranks) {
    if (!comm) return xmhw_set_error_(XMHW_ERR_INVALID, "comm is NULL");
    if (rank) *rank = comm->rank;
    if (nranks) *nranks = comm->nranks;
    return XMHW_OK;
}

int xmhw_comm_allgather_i64(xmhw_comm* comm, int64_t value, int64_t* out_host, void* stream) {
    if (!comm || !out_host) return xmhw_set_error_(XMHW_ERR_INVALID, "NULL argument");
    Rccl* r = rccl();
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t* mine = comm->d_scratch + comm->nranks;
    HIPC_TRY(hipMemcpyAsync(mine, &value, sizeof(int64_t), hipMemcpyHostToDevice, st));
    RCCL_TRY(r->AllGather(mine, comm->d_scratch, 1, kNcclInt64, comm->comm, st));
    HIPC_TRY(hipMemcpyAsync(out_host, comm->d_scratch, sizeof(int64_t) * static_cast<size_t>(comm->nranks),
                            hipMemcpyDeviceToHost, st));
    HIPC_TRY(hipStreamSynchronize(st));
    return XMHW_OK;
}

// the same all-gather in two halves: begin() queues the copy in, the collective and the copy out (pinned memory on
// both ends, nothing waits) and returns; end() waits for that work only and hands the values over.  Host work --
// or kernels on other streams -- can sit between the two.
int xmhw_comm_allgather_i64_begin(xmhw_comm* comm, int64_t value, void* stream) {
    if (!comm) return xmhw_set_error_(XMHW_ERR_INVALID, "comm is NULL");
    if (comm->pending) return xmhw_set_error_(XMHW_ERR_INVALID, "an all-gather is already in flight on this communicator");
    Rccl* r = rccl();
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t* mine = comm->d_scratch + comm->nranks;
    comm->h_pinned[comm->nranks] = value;
    HIPC_TRY(hipMemcpyAsync(mine, comm->h_pinned + comm->nranks, sizeof(int64_t), hipMemcpyHostToDevice, st));
    RCCL_TRY(r->AllGather(mine, comm->d_scratch, 1, kNcclInt64, comm->comm, st));
    HIPC_TRY(hipMemcpyAsync(comm->h_pinned, comm->d_scratch, sizeof(int64_t) * static_cast<size_t>(comm->nranks),
                            hipMemcpyDeviceToHost, st));
    HIPC_TRY(hipEventRecord(comm->ev, st));
    comm->pending = true;
    return XMHW_OK;
}

int xmhw_comm_allgather_i64_end(xmhw_comm* comm, int64_t* out_host) {
    if (!comm || !out_host) return xmhw_set_error_(XMHW_ERR_INVALID, "NULL argument");
    if (!comm->pending) return xmhw_set_error_(XMHW_ERR_INVALID, "no all-gather in flight on this communicator");
    HIPC_TRY(hipEventSynchronize(comm->ev));
    comm->pending = false;
    std::memcpy(out_host, comm->h_pinned, sizeof(int64_t) * static_cast<size_t>(comm->nranks));
    return XMHW_OK;
}

int xmhw_comm_allgather_bytes(xmhw_comm* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    if (!comm) return xmhw_set_error_(XMHW_ERR_INVALID, "comm is NULL");
    if (bytes_per_rank == 0) return XMHW_OK;
    if (!send_dev || !recv_dev) return xmhw_set_error_(XMHW_ERR_INVALID, "NULL device buffer");
    Rccl* r = rccl();
    RCCL_TRY(r->AllGather(send_dev, recv_dev, bytes_per_rank, kNcclUint8, comm->comm, static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}

int xmhw_gather_blocks(xmhw_comm* comm, const double* send_dev, int64_t rows, int64_t cols, double* recv_dev,
                       const int64_t* cols_of_rank, int root, void* stream) {
    if (!comm) return xmhw_set_error_(XMHW_ERR_INVALID, "comm is NULL");
    if (rows < 0 || cols < 0 || root < 0 || root >= comm->nranks) return xmhw_set_error_(XMHW_ERR_INVALID, "bad rows/cols/root");
    if (rows * cols > 0 && !send_dev) return xmhw_set_error_(XMHW_ERR_INVALID, "send_dev is NULL");
    Rccl* r = rccl();
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (comm->rank != root) {
        // one message per rank; an empty block sends nothing (the root posts no receive for it)
        if (rows * cols > 0)
            RCCL_TRY(r->Send(send_dev, static_cast<size_t>(rows * cols), kNcclFloat64, root, comm->comm, st));
        return XMHW_OK;
    }
    if (!cols_of_rank) return xmhw_set_error_(XMHW_ERR_INVALID, "cols_of_rank is NULL on the root");
    if (cols_of_rank[root] != cols) return xmhw_set_error_(XMHW_ERR_INVALID, "cols_of_rank[root] != cols");
    int64_t total = 0;
    for (int p = 0; p < comm->nranks; ++p) {
        if (cols_of_rank[p] < 0) return xmhw_set_error_(XMHW_ERR_INVALID, "negative block width");
        total += cols_of_rank[p];
    }
    if (rows * total > 0 && !recv_dev) return xmhw_set_error_(XMHW_ERR_INVALID, "recv_dev is NULL on the root");
    // blocks land one after the other in rank order, each (rows, cols_r) contiguous
    RCCL_TRY(r->GroupStart());
    int64_t off = 0;
    ncclResult_ err = 0;
    for (int p = 0; p < comm->nranks; ++p) {
        const int64_t n = rows * cols_of_rank[p];
        if (p != root && n > 0 && err == 0)
            err = r->Recv(recv_dev + off, static_cast<size_t>(n), kNcclFloat64, p, comm->comm, st);
        off += n;
    }
    ncclResult_ e2 = r->GroupEnd();
    if (err != 0) return rccl_fail(err, "ncclRecv");
    if (e2 != 0) return rccl_fail(e2, "ncclGroupEnd");
    // the root's own block: a device-to-device copy on the same stream
    off = 0;
    for (int p = 0; p < root; ++p) off += rows * cols_of_rank[p];
    if (rows * cols > 0 && recv_dev + off != send_dev)
        HIPC_TRY(hipMemcpyAsync(recv_dev + off, send_dev, sizeof(double) * static_cast<size_t>(rows * cols),
                                hipMemcpyDeviceToDevice, st));
    return XMHW_OK;
}

int xmhw_memcpy2d_d2h(void* dst, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t height,
                      void* stream) {
    if (width == 0 || height == 0) return XMHW_OK;
    if (!dst || !src_dev || dpitch < width || spitch < width) return xmhw_set_error_(XMHW_ERR_INVALID, "bad pointer/pitch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPC_TRY(hipMemcpy2DAsync(dst, dpitch, src_dev, spitch, width, height, hipMemcpyDeviceToHost, st));
    HIPC_TRY(hipStreamSynchronize(st));
    return XMHW_OK;
}

// the same copy without the wait: the root drains a gathered buffer with one of these per rank and ONE
// xmhw_stream_sync at the end (dst should be pinned for the copies to overlap each other)
int xmhw_memcpy2d_d2h_async(void* dst, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t height,
                            void* stream) {
    if (width == 0 || height == 0) return XMHW_OK;
    if (!dst || !src_dev || dpitch < width || spitch < width) return xmhw_set_error_(XMHW_ERR_INVALID, "bad pointer/pitch");
    HIPC_TRY(hipMemcpy2DAsync(dst, dpitch, src_dev, spitch, width, height, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}

}  // extern "C"
