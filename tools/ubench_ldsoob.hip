// ubench_ldsoob.hip -- what does a ds_read_b32 return for an address outside the workgroup's LDS allocation (gfx950)?
// The rank-major list layout of kernels_sorted.hip (round 6) relies on it: a window that reaches past the last rank of a
// list (or above rank 0: the address wraps) must read 0.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsoob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_ld(uint32_t a) { return *reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(a)); }
template <int BYTES>
__global__ __launch_bounds__(64) void probe(const uint32_t* addrs, int n, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[BYTES / 4];
    for (int i = threadIdx.x; i < BYTES / 4; i += 64) lds[i] = 0xA0000000u + blockIdx.x * 0x100000u + i;
    __syncthreads();
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u32*)lds));
    for (int i = 0; i < n; ++i) {
        const uint32_t a = base + addrs[i] + threadIdx.x * 4u;
        out[(blockIdx.x * n + i) * 64 + threadIdx.x] = lds_ld(a);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * n * 64] = base;
}
int main() {
    std::vector<uint32_t> addrs = {0u, 22528u - 256u, 22528u, 22528u + 128u, 22528u + 512u, 23040u, 23552u, 32768u, 65536u, 131072u,
                                   163840u, 0xFFFFFF00u, 0xFFFFF000u, 0xFFFF0000u, 0x80000000u, 22528u - 128u};
    const int n = static_cast<int>(addrs.size()), blocks = 2048;   // several workgroups per CU: a neighbour's LDS is behind ours
    uint32_t *d_a, *d_o;
    hipMalloc(&d_a, n * 4); hipMalloc(&d_o, (blocks * n * 64 + 1) * 4);
    hipMemcpy(d_a, addrs.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> o(blocks * n * 64 + 1);
    for (int variant = 0; variant < 2; ++variant) {
        hipMemset(d_o, 0xEE, o.size() * 4);
        if (variant == 0) hipLaunchKernelGGL(probe<22528>, dim3(blocks), dim3(64), 0, 0, d_a, n, d_o);
        else hipLaunchKernelGGL(probe<22656>, dim3(blocks), dim3(64), 0, 0, d_a, n, d_o);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
        printf("allocation %d bytes, lds base %u\n", variant == 0 ? 22528 : 22656, o.back());
        for (int i = 0; i < n; ++i) {
            // over all workgroups and lanes: how many reads returned 0, how many the own pattern, how many something else
            long zero = 0, own = 0, other = 0; uint32_t sample = 0;
            for (int b = 0; b < blocks; ++b)
                for (int l = 0; l < 64; ++l) {
                    const uint32_t v = o[(b * n + i) * 64 + l];
                    const uint32_t want = 0xA0000000u + b * 0x100000u + (addrs[i] / 4 + l);
                    if (v == 0) ++zero; else if (v == want) ++own; else { ++other; sample = v; }
                }
            printf("  offset %10u (0x%08x): zero %6ld own %6ld other %6ld%s", addrs[i], addrs[i], zero, own, other, other ? "  e.g. " : "\n");
            if (other) printf("0x%08x\n", sample);
        }
    }
    return 0;
}
