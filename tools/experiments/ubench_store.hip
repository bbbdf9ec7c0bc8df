// ubench_store.hip -- round 4: what the key store's push costs on gfx950, by where the lanes whose key is OUTSIDE the
// window (85 % of them) send their ds_add_rtn + ds_write: (a) a dummy word pair per lane inside the cell's block (the
// first build of kernels_ring4.hip: 64 lanes on 16 banks), (b) dummy words laid out lane-major (one bank per lane of a
// half-wave), (c) an address beyond the workgroup's LDS allocation.  (c) first needs its semantics checked: does a DS
// atomic beyond the allocation return 0, is a DS write there dropped, and does neither touch the NEXT workgroup's LDS?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o tools/ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
constexpr int kWords = 9472;       // 37,888 bytes: four workgroups per CU, as kernels_ring4.hip

// ---- semantics: every workgroup fills its LDS with a pattern, then every lane does atomics and writes beyond the
// allocation (just past it, 64 KB past it, at 0xFFFFFF00); afterwards the pattern must be intact in EVERY workgroup
__global__ __launch_bounds__(128) void k_oob(uint32_t* out, uint32_t* bad) {
    __shared__ uint32_t lds[kWords];
    for (int i = threadIdx.x; i < kWords; i += 128) lds[i] = 0xA5000000u + i;
    __syncthreads();
    const uint32_t offs[4] = {kWords * 4u, kWords * 4u + 4096u, kWords * 4u + 65536u, 0xFFFFFF00u};
    uint32_t got = 0;
    for (int rep = 0; rep < 64; ++rep)
        for (int k = 0; k < 4; ++k) {
            uint32_t addr = offs[k] + threadIdx.x * 4, add = 0x10001u, r;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(add) : "memory");
            got |= r;
            asm volatile("ds_write_b32 %0, %1 offset:4\n s_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(0xDEADBEEFu) : "memory");
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
            got |= r;
        }
    __syncthreads();
    // (stay resident a while so that the neighbours' stores would have landed)
    for (int i = 0; i < 2000; ++i) asm volatile("s_sleep 8");
    __syncthreads();
    uint32_t nbad = 0;
    for (int i = threadIdx.x; i < kWords; i += 128) nbad += lds[i] != 0xA5000000u + i;
    if (nbad) atomicAdd(bad, nbad);
    if (got) atomicOr(bad + 1, got);
    out[blockIdx.x * 128 + threadIdx.x] = got;
}

// ---- timing: 10 x (ds_add_rtn) then 10 x (ds_write at the slot handed out), 16 cells of 296 words per wave
template <int MODE>   // 0: dummy pair per lane inside the cell block, 1: lane-major dummy words, 2: beyond the allocation, 3: every lane in a row
__global__ __launch_bounds__(128) void k_push(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[kWords];
    for (int i = threadIdx.x; i < kWords; i += 128) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 3, cw = lane >> 2;
    const uint32_t cell = ((wave * 16 + cw) * 296) * 4;
    uint32_t dump;
    if (MODE == 0) dump = cell + (288 + 2 * sub) * 4;
    else if (MODE == 1) dump = (32 * 296 - 0) * 4 - 1024 + threadIdx.x * 8;     // (inside the array: last kilobyte, lane-major pairs)
    else dump = 0xFFFF0000u + threadIdx.x * 8;
    uint32_t r = threadIdx.x * 2654435761u + seed, acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t a[10], w[10];
#pragma unroll
        for (int y = 0; y < 10; ++y) {
            r = r * 1664525u + 1013904223u;
            const bool in = MODE == 3 || (r >> 24) < 38;                 // 15 % of the keys are inside the window
            const uint32_t row = (r >> 8) & 7;
            a[y] = in ? cell + 12 + row * 144 : dump;
            uint32_t add = in ? 0x10001u : 0u;
            asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(w[y]) : "v"(a[y]), "v"(add) : "memory");
        }
#pragma unroll
        for (int y = 0; y < 10; ++y) {
            r = r * 1664525u + 1013904223u;
            const bool in = MODE == 3 || (r >> 24) < 38;
            const uint32_t row = (r >> 8) & 7;
            uint32_t ad = in ? cell + 12 + row * 144 : dump, add = in ? 0xFFFFFFFFu : 0u;
            asm volatile("ds_add_u32 %0, %1" ::"v"(ad), "v"(add) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int y = 0; y < 10; ++y) {
            const uint32_t ea = a[y] + (((w[y] >> 16) & 31u) << 2);
            asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(ea), "v"(r) : "memory");
            acc += w[y];
        }
    }
    __syncthreads();
    out[blockIdx.x * 128 + threadIdx.x] = acc + lds[threadIdx.x];
}

struct Ent { const char* name; void (*k)(uint32_t*, int, uint32_t); };

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    uint32_t *out, *bad; hipMalloc(&out, sizeof(uint32_t) * 128 * cus * 16); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k_oob, dim3(cus * 4), dim3(128), 0, 0, out, bad);
    hipDeviceSynchronize();
    uint32_t hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("beyond the allocation (4 workgroups of 37,888 B per CU): words of ANY workgroup's LDS changed: %u; OR of everything "
           "the atomics and reads there returned: 0x%08x  (%s)\n", hb[0], hb[1],
           hb[0] == 0 && hb[1] == 0 ? "atomics return 0, stores are dropped" : "NOT SAFE");
    std::vector<Ent> ks = {{"10 pushes + 10 evictions, dummy pair per lane in the cell block", k_push<0>},
                           {"  ... dummy words lane-major", k_push<1>},
                           {"  ... dummy beyond the allocation", k_push<2>},
                           {"  ... every lane inside a row (no dummy)", k_push<3>}};
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("cycles per row-step (10 ds_add_rtn + 10 ds_add + 10 ds_write) per workgroup of 2 waves, 4 workgroups per CU\n");
    for (auto& e : ks) {
        printf("%-64s", e.name);
        for (int wps : {4}) {
            dim3 grid(cus * wps);
            hipLaunchKernelGGL(e.k, grid, dim3(128), 0, 0, out, 10, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, grid, dim3(128), 0, 0, out, iters, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" %8.1f", ms * 1e-3 * clk / double(iters));
        }
        printf("\n");
    }
    return 0;
}
