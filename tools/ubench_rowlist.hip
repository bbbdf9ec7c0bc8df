// ubench_rowlist.hip -- what does the LOAD PATTERN of the sorted-list kernel cost by itself, and beside vector work?
// A wave owns 32 cells and walks `rows` steps; per step it reads one 128-byte line (32 cells x float32) from each of 40
// tracks (tracks one "year" = 365 steps apart, a step = C * 4 bytes), requested one step ahead -- exactly what
// clim_sorted_f32<20, 16> does on configs[2] (C = 1,036,800, T = 14,610).  Variants of WHICH LANE loads WHAT:
//   0  lane = 2 * cell + sub, sub holds tracks 2y + sub: 20 global_load_dword per lane, every instruction touches two
//      lines with the lanes alternating between them (the product kernel's pattern)
//   1  lane = sub * 32 + cell: every half-wave reads one contiguous line (needs a cross-half exchange in a real kernel)
//   2  quad layout: lane 4q + i -> cell 2q + (i & 1), sub = i >> 1 (partner = lane ^ 2, still a quad_perm)
//   3  10 global_load_dwordx2 per lane: lane 4q + i reads cells (2q, 2q + 1) of track 4y + i (8 contiguous bytes per lane,
//      a quad reads four tracks), to be redistributed inside the quad by DPP
// `work` = dependent-free v_max_u32 instructions per step and wave (0: loads only; ~1000: the product kernel's row).
// LDS per workgroup is set so that 7 (or 8) waves fit a CU, as in the product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_rowlist tools/ubench_rowlist.hip ; ./tools/ubench_rowlist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int VAR, int LDSB>
__global__ __launch_bounds__(64, 2) void walk(const float* __restrict__ ts, long C, int rows, int work, uint32_t* __restrict__ out) {
    __shared__ uint32_t lds[LDSB / 4];
    const int lane = threadIdx.x;
    lds[lane] = lane;
    int cw, sub;
    if (VAR == 0) { cw = lane >> 1; sub = lane & 1; }
    else if (VAR == 1) { cw = lane & 31; sub = lane >> 5; }
    else { cw = 2 * (lane >> 2) + (lane & 1); sub = (lane >> 1) & 1; }
    const long cell0 = (long)blockIdx.x * 32;
    const long year = 365L * C;
    uint32_t acc = lds[(lane * 7) & 63];
    uint32_t a0 = lane, a1 = lane * 3 + 1, a2 = lane * 5 + 2, a3 = lane * 7 + 3;
    if (VAR == 4) {
        for (int s = 0; s < rows; ++s) {
            for (int i = 0; i < work; i += 16) {
                asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a1), "v"(a2), "v"(a3), "v"(a0));
            }
        }
    } else if (VAR != 3) {
        const float* p[20];
#pragma unroll
        for (int y = 0; y < 20; ++y) p[y] = ts + (long)(2 * y + sub) * year + cell0 + cw;
        float x[20];
#pragma unroll
        for (int y = 0; y < 20; ++y) x[y] = *p[y];
        for (int s = 0; s < rows; ++s) {
#pragma unroll
            for (int y = 0; y < 20; ++y) acc += __float_as_uint(x[y]);
#pragma unroll
            for (int y = 0; y < 20; ++y) { p[y] += C; x[y] = *p[y]; }
            for (int i = 0; i < work; i += 16) {
                asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a1), "v"(a2), "v"(a3), "v"(a0));
            }
        }
    } else {
        const int q = lane >> 2, i4 = lane & 3;
        const float2* p[10];
#pragma unroll
        for (int y = 0; y < 10; ++y) p[y] = reinterpret_cast<const float2*>(ts + (long)(4 * y + i4) * year + cell0 + 2 * q);
        float2 x[10];
#pragma unroll
        for (int y = 0; y < 10; ++y) x[y] = *p[y];
        for (int s = 0; s < rows; ++s) {
#pragma unroll
            for (int y = 0; y < 10; ++y) acc += __float_as_uint(x[y].x) ^ __float_as_uint(x[y].y);
#pragma unroll
            for (int y = 0; y < 10; ++y) { p[y] += C / 2; x[y] = *p[y]; }
            for (int i = 0; i < work; i += 16) {
                asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7\n"
                             "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a1), "v"(a2), "v"(a3), "v"(a0));
            }
        }
    }
    out[blockIdx.x * 64 + lane] = acc + a0 + a1 + a2 + a3;
}

typedef void (*K)(const float*, long, int, int, uint32_t*);

int main(int argc, char** argv) {
    const long C = 1036800, T = 14610;
    float* ts;
    if (hipMalloc(&ts, sizeof(float) * C * T) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMemset(ts, 0, sizeof(float) * C * T);
    uint32_t* out;
    hipMalloc(&out, 4 * 64 * (C / 32));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rows = 330;          // 40 tracks x 365 steps: step 0 .. 364 - a margin
    struct V { const char* name; K k7, k8; };
    V vs[] = {{"0 lane=2c+sub (product)", walk<0, 23040>, walk<0, 20480>}, {"1 half-wave per track", walk<1, 23040>, walk<1, 20480>},
              {"2 quad layout", walk<2, 23040>, walk<2, 20480>}, {"3 dwordx2, quad reads 4 tracks", walk<3, 23040>, walk<3, 20480>},
              {"4 no loads: the vector work alone", walk<4, 23040>, walk<4, 20480>}};
    printf("%-34s %6s %5s %10s %10s\n", "variant", "waves", "work", "ms", "TB/s");
    for (int work : {0, 800, 1008, 1200}) {
        for (auto& v : vs) {
            for (int w8 = 0; w8 < 2; ++w8) {
                K k = w8 ? v.k8 : v.k7;
                hipLaunchKernelGGL(k, dim3(C / 32), dim3(64), 0, 0, ts, C, 8, work, out);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(C / 32), dim3(64), 0, 0, ts, C, rows, work, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)C * 40 * 4 * rows;
                printf("%-34s %6d %5d %10.3f %10.3f\n", v.name, w8 ? 8 : 7, work, ms, bytes / ms / 1e9);
            }
        }
    }
    return 0;
}
