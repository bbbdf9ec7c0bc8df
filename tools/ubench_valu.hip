// ubench_valu.hip -- issue-rate microbenchmark for the VALU ops the selection
// passes are built from (gfx950).  Prints cycles per wave-instruction per SIMD
// at 1/2/4/8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(NAME, BODY, NINST)                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t seed) { \
        uint32_t a0 = threadIdx.x ^ seed, a1 = a0 * 3u + 1, a2 = a0 * 5u + 2, a3 = a0 * 7u + 3; \
        uint32_t b0 = a0 + 11, b1 = a1 + 13, b2 = a2 + 17, b3 = a3 + 19;              \
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;                                       \
        uint32_t p = seed * 2654435761u;                                               \
        for (int i = 0; i < iters; ++i) { REP8(BODY) }                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + c0 + c1 + c2 + c3; \
    }                                                                                  \
    static const int NAME##_n = 8 * (NINST);

// each BODY is a group of independent instructions
KERNEL(k_cmp_addc,
    asm volatile("v_cmp_le_u32 vcc, %4, %8\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
                 "v_cmp_le_u32 vcc, %5, %8\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                 "v_cmp_le_u32 vcc, %6, %8\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n"
                 "v_cmp_le_u32 vcc, %7, %8\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(p) : "vcc");, 8)
KERNEL(k_cmp_sgpr_addc,
    asm volatile("v_cmp_le_u32 s[20:21], %4, %8\n v_cmp_le_u32 s[22:23], %5, %8\n"
                 "v_cmp_le_u32 s[24:25], %6, %8\n v_cmp_le_u32 s[26:27], %7, %8\n"
                 "v_addc_co_u32 %0, s[28:29], 0, %0, s[20:21]\n v_addc_co_u32 %1, s[28:29], 0, %1, s[22:23]\n"
                 "v_addc_co_u32 %2, s[28:29], 0, %2, s[24:25]\n v_addc_co_u32 %3, s[28:29], 0, %3, s[26:27]\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(p)
                 : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29");, 8)
KERNEL(k_sad_u32,
    asm volatile("v_sad_u32 %0, %4, %8, %0\n v_sad_u32 %1, %5, %8, %1\n v_sad_u32 %2, %6, %8, %2\n v_sad_u32 %3, %7, %8, %3\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(p));, 4)
KERNEL(k_sad_u16,
    asm volatile("v_sad_u16 %0, %4, %8, %0\n v_sad_u16 %1, %5, %8, %1\n v_sad_u16 %2, %6, %8, %2\n v_sad_u16 %3, %7, %8, %3\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(p));, 4)
KERNEL(k_sad_u8,
    asm volatile("v_sad_u8 %0, %4, %8, %0\n v_sad_u8 %1, %5, %8, %1\n v_sad_u8 %2, %6, %8, %2\n v_sad_u8 %3, %7, %8, %3\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(p));, 4)
KERNEL(k_sub_med3_min,
    asm volatile("v_sub_u32 %0, %4, %6\n v_med3_u32 %2, %1, %2, %0\n v_min_u32 %1, %1, %0\n"
                 "v_sub_u32 %3, %5, %6\n v_med3_u32 %8, %7, %8, %3\n v_min_u32 %7, %7, %3\n"
                 : "+v"(c0), "+v"(b0), "+v"(b1), "+v"(c1) : "v"(a0), "v"(a1), "v"(p), "v"(b2), "v"(b3));, 6)
KERNEL(k_min_u32,
    asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %5\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %7\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_min3_u32,
    asm volatile("v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %5, %6\n v_min3_u32 %2, %2, %6, %7\n v_min3_u32 %3, %3, %7, %4\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_add_u32,
    asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_add3_u32,
    asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %6\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %7, %4\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_fma_f32,
    asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %5, %6, %1\n v_fma_f32 %2, %6, %7, %2\n v_fma_f32 %3, %7, %4, %3\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_max_f32,
    asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_max_f32 %2, %2, %6\n v_max_f32 %3, %3, %7\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_pk_add_u16,
    asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %6\n v_pk_add_u16 %3, %3, %7\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_pk_sub_u16_clamp,
    asm volatile("v_pk_sub_u16 %0, %4, %5 clamp\n v_pk_sub_u16 %1, %5, %6 clamp\n v_pk_sub_u16 %2, %6, %7 clamp\n v_pk_sub_u16 %3, %7, %4 clamp\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_pk_min_u16,
    asm volatile("v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %5\n v_pk_min_u16 %2, %2, %6\n v_pk_min_u16 %3, %3, %7\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_cndmask,
    asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");, 4)
KERNEL(k_add_dpp,
    asm volatile("v_add_u32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %5, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                 "v_add_u32_dpp %2, %6, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %7, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));, 4)
KERNEL(k_bpermute,
    asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0));, 4)
KERNEL(k_swizzle,
    asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"pp0pp\")\n ds_swizzle_b32 %1, %1 offset:swizzle(BITMASK_PERM, \"pp0pp\")\n"
                 "ds_swizzle_b32 %2, %2 offset:swizzle(BITMASK_PERM, \"pp0pp\")\n ds_swizzle_b32 %3, %3 offset:swizzle(BITMASK_PERM, \"pp0pp\")\n s_waitcnt lgkmcnt(0)\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));, 4)
KERNEL(k_permlane32_swap,
    asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));, 2)
KERNEL(k_add_f64,
    asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n" : "+v"(*(double*)&c0), "+v"(*(double*)&c2) : "v"(*(double*)&a0));, 2)

typedef void (*K)(uint32_t*, int, uint32_t);
struct Ent { const char* name; K k; int n; };
#define E(x) {#x, x, x##_n}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;   // Hz
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    uint32_t* out; hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 16);
    std::vector<Ent> ks = {E(k_cmp_addc), E(k_cmp_sgpr_addc), E(k_sad_u32), E(k_sad_u16), E(k_sad_u8), E(k_sub_med3_min),
        E(k_min_u32), E(k_min3_u32), E(k_add_u32), E(k_add3_u32), E(k_fma_f32), E(k_max_f32), E(k_pk_add_u16),
        E(k_pk_sub_u16_clamp), E(k_pk_min_u16), E(k_cndmask), E(k_add_dpp), E(k_bpermute), E(k_swizzle),
        E(k_permlane32_swap), E(k_add_f64)};
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD at w waves/SIMD)\n", "op", "w=1", "w=2", "w=4", "w=8");
    for (auto& e : ks) {
        printf("%-22s", e.name);
        for (int wps : {1, 2, 4, 8}) {
            dim3 grid(cus * wps);   // 256-thread blocks: 1 wave per SIMD per block
            hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, 10, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, iters, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double inst_per_simd = double(iters) * e.n * wps;
            printf(" %8.2f", ms * 1e-3 * clk / inst_per_simd);
        }
        printf("\n");
    }
    return 0;
}
