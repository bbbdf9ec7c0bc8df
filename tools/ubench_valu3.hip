// ubench_valu3.hip -- third issue-rate sweep (gfx950, round 6): the instructions the sorted-list kernel's ADDRESS and KEY paths
// could be rebuilt from -- 64-bit adds and multiply-adds, the float comparators (a sort on raw float bits would save the key
// conversion of the samples that are not kept), moves through DPP.  Same harness as ubench_valu2.hip: cycles per wave64
// instruction and SIMD at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu3 tools/ubench_valu3.hip ; ./tools/ubench_valu3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define KERNEL(NAME, BODY, NINST)                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t seed) { \
        uint32_t a0 = threadIdx.x ^ seed, a1 = a0 * 3u + 1, a2 = a0 * 5u + 2, a3 = a0 * 7u + 3; \
        uint32_t b0 = a0 + 11, b1 = a1 + 13, b2 = a2 + 17, b3 = a3 + 19;              \
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;                                       \
        unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, e0 = b0, e1 = b1, e2 = b2, e3 = b3; \
        for (int i = 0; i < iters; ++i) { REP8(BODY) }                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + c0 + c1 + c2 + c3 + \
            (uint32_t)(d0 + d1 + d2 + d3);                                            \
    }                                                                                  \
    static const int NAME##_n = 8 * (NINST);
KERNEL(k_max_u32,
    asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_lshl_add_u64,
    asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %5\n v_lshl_add_u64 %2, %2, 0, %6\n v_lshl_add_u64 %3, %3, 0, %7\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3));, 4)
KERNEL(k_mad_u64_u32,
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");, 4)
KERNEL(k_mov_b64,
    asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %6\n v_mov_b64 %3, %7\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3));, 4)
KERNEL(k_max_f32,
    asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_max_f32 %2, %2, %6\n v_max_f32 %3, %3, %7\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_max3_f32,
    asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %5, %6\n v_max3_f32 %2, %2, %6, %7\n v_max3_f32 %3, %3, %7, %4\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_max3_u32,
    asm volatile("v_max3_u32 %0, %0, %4, %5\n v_max3_u32 %1, %1, %5, %6\n v_max3_u32 %2, %2, %6, %7\n v_max3_u32 %3, %3, %7, %4\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_cvt_f64_f32,
    asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_mov_dpp,
    asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_max_u32_dpp,
    asm volatile("v_max_u32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %1, %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_cndmask,
    asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");, 4)
KERNEL(k_xad_u32,
    asm volatile("v_xad_u32 %0, %0, %4, %5\n v_xad_u32 %1, %1, %5, %6\n v_xad_u32 %2, %2, %6, %7\n v_xad_u32 %3, %3, %7, %4\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_bitop3,
    asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x36\n v_bitop3_b32 %1, %1, %5, %6 bitop3:0x36\n v_bitop3_b32 %2, %2, %6, %7 bitop3:0x36\n v_bitop3_b32 %3, %3, %7, %4 bitop3:0x36\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_accvgpr_rw,
    asm volatile("v_accvgpr_write_b32 a0, %4\n v_accvgpr_read_b32 %0, a0\n v_accvgpr_write_b32 a1, %5\n v_accvgpr_read_b32 %1, a1\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "a0", "a1");, 4)
typedef void (*K)(uint32_t*, int, uint32_t);
struct Ent { const char* name; K k; int n; };
#define E(x) {#x, x, x##_n}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    uint32_t* out; hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 16);
    std::vector<Ent> ks = {E(k_max_u32), E(k_lshl_add_u64), E(k_mad_u64_u32), E(k_mov_b64), E(k_max_f32), E(k_max3_f32), E(k_max3_u32),
                           E(k_cvt_f64_f32), E(k_mov_dpp), E(k_max_u32_dpp), E(k_cndmask), E(k_xad_u32), E(k_bitop3), E(k_accvgpr_rw)};
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-24s %8s %8s %8s\n", "op", "w=1", "w=2", "w=4");
    for (auto& e : ks) {
        printf("%-24s", e.name);
        for (int wps : {1, 2, 4}) {
            dim3 grid(cus * wps);
            hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, 10, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, iters, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" %8.2f", ms * 1e-3 * clk / (double(iters) * e.n * wps));
        }
        printf("\n");
    }
    return 0;
}
