// ubench_lds.hip -- what the pieces of the round-3 band compaction cost on gfx950 (cycles per unit per
// SIMD, one block of 256 threads = one wave per SIMD, w blocks per CU): masked ds_write_b32, ds_add_u32 with
// linear / clustered / uniformly random addresses, a taken s_cbranch, and the compare + EXEC pattern
// without its LDS part.   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
constexpr int kCells = 64, kHS = 257;

__global__ __launch_bounds__(256) void k_ds_write_full(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[kCells * kHS];
    uint32_t addr = threadIdx.x * 4, v = seed;
    for (int i = 0; i < iters; ++i) { REP8(asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");) }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_ds_write_masked(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[kCells * kHS];
    uint32_t addr = threadIdx.x * 4, v = seed;
    unsigned long long m = 0x0000010000000100ull;      // two lanes
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("s_mov_b64 exec, %2\n ds_write_b32 %0, %1\n s_mov_b64 exec, -1" ::"v"(addr), "v"(v), "s"(m) : "memory");)
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_ds_write_masked_add(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[kCells * kHS];
    uint32_t addr = threadIdx.x * 4, v = seed;
    unsigned long long m = 0x0000010000000100ull;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("s_mov_b64 exec, %2\n ds_write_b32 %0, %1\n v_xor_b32 %0, 4, %0\n s_mov_b64 exec, -1" : "+v"(addr) : "v"(v), "s"(m) : "memory");)
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
template <int MODE>   // 0 linear, 1 clustered around bucket 128 of the lane's cell (4 lanes per cell), 2 uniform over the cell's 256 buckets
__global__ __launch_bounds__(256) void k_ds_add(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[kCells * kHS];
    for (int i = threadIdx.x; i < kCells * kHS; i += 256) lds[i] = 0;
    __syncthreads();
    const uint32_t cellbase = (threadIdx.x >> 2) * kHS * 4;
    uint32_t r = threadIdx.x * 2654435761u + seed, one = 1;
    for (int i = 0; i < iters; ++i) {
        REP8({
            r = r * 1664525u + 1013904223u;
            uint32_t tag = MODE == 0 ? 0u : MODE == 1 ? 108u + ((r >> 20) % 40u) : (r >> 24);
            uint32_t addr = MODE == 0 ? threadIdx.x * 4 : cellbase + tag * 4;
            asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory");
        })
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
// the same address arithmetic without the atomic (to subtract)
template <int MODE>
__global__ __launch_bounds__(256) void k_addr_only(uint32_t* out, int iters, uint32_t seed) {
    const uint32_t cellbase = (threadIdx.x >> 2) * kHS * 4;
    uint32_t r = threadIdx.x * 2654435761u + seed, acc = 0;
    for (int i = 0; i < iters; ++i) {
        REP8({
            r = r * 1664525u + 1013904223u;
            uint32_t tag = MODE == 0 ? 0u : MODE == 1 ? 108u + ((r >> 20) % 40u) : (r >> 24);
            uint32_t addr = MODE == 0 ? threadIdx.x * 4 : cellbase + tag * 4;
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(addr));
        })
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_branch_taken(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed;
    unsigned long long z = 0, sv;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, %1, %2\n s_cbranch_scc0 1f\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 3, %0\n1:\n s_mov_b64 exec, %1" : "+v"(a), "=&s"(sv) : "s"(z) : "scc");)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_branch_not_taken(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed;
    unsigned long long z = 0x100, sv;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, %1, %2\n s_cbranch_scc0 1f\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 3, %0\n1:\n s_mov_b64 exec, %1" : "+v"(a), "=&s"(sv) : "s"(z) : "scc");)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_cmp_exec_add(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed, k = a * 77u, e0 = 5, w = 3, t;
    unsigned long long m, sv;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("s_mov_b64 %3, exec\n v_sub_u32 %1, %4, %5\n v_cmp_lt_u32_e64 %2, %1, %6\n s_and_b64 exec, %3, %2\n v_add_u32 %0, 4, %0\n s_mov_b64 exec, %3"
                          : "+v"(a), "=&v"(t), "=&s"(m), "=&s"(sv) : "v"(k), "v"(e0), "v"(w) : "scc");)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

typedef void (*K)(uint32_t*, int, uint32_t);
struct Ent { const char* name; K k; };
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz; cycles per unit per SIMD (one wave per SIMD per block, w blocks per CU)\n", prop.name, cus, clk / 1e6);
    uint32_t* out; hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 16);
    std::vector<Ent> ks = {{"ds_write_b32 all lanes", k_ds_write_full}, {"ds_write_b32 2 lanes (exec set+restore)", k_ds_write_masked},
                           {"  same + masked v_xor", k_ds_write_masked_add},
                           {"ds_add_u32 linear", k_ds_add<0>}, {"ds_add_u32 clustered (40 buckets)", k_ds_add<1>},
                           {"ds_add_u32 uniform (256 buckets)", k_ds_add<2>},
                           {"  address arithmetic only, linear", k_addr_only<0>}, {"  address arithmetic only, clustered", k_addr_only<1>},
                           {"  address arithmetic only, uniform", k_addr_only<2>},
                           {"s_and exec + s_cbranch taken (skips 2 VALU)", k_branch_taken},
                           {"s_and exec + s_cbranch not taken + 2 VALU", k_branch_not_taken},
                           {"v_sub + v_cmp + s_and exec + v_add + restore", k_cmp_exec_add}};
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-48s %8s %8s %8s\n", "unit", "w=1", "w=2", "w=3");
    for (auto& e : ks) {
        printf("%-48s", e.name);
        for (int wps : {1, 2, 3}) {
            dim3 grid(cus * wps);
            hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, 10, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, iters, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" %8.2f", ms * 1e-3 * clk / (double(iters) * 8 * wps));
        }
        printf("\n");
    }
    return 0;
}
