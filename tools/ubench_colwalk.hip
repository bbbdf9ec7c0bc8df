// What does the access pattern of the finish kernel cost by itself?  (D x C) float64 in, (D x C) out, two arrays:
//   flat     : a plain copy, every thread 16 bytes, consecutive threads consecutive addresses (the ceiling)
//   walk<A>  : thread = cell, walking down the D rows of its column with A loads in flight (what clim_finish_stream
//              does with A = 1: 512 contiguous bytes per wave and row, the next row 8 * C bytes further on)
//   walk2d<A,P>: the same with the rows cut into P parts (more waves, each walking D / P rows)
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_colwalk tools/ubench_colwalk.hip ; ./tools/ubench_colwalk [C] [D]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void flat(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}

template <int A>
__global__ __launch_bounds__(256) void walk(const double* __restrict__ in0, const double* __restrict__ in1,
                                            double* __restrict__ out0, double* __restrict__ out1, long C, int D,
                                            int rpp) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double* in = (blockIdx.z ? in1 : in0) + c;
    double* out = (blockIdx.z ? out1 : out0) + c;
    const int d0 = blockIdx.y * rpp, d1 = d0 + rpp < D ? d0 + rpp : D;
    double pre[A];
#pragma unroll
    for (int k = 0; k < A; ++k) pre[k] = d0 + k < d1 ? in[(long)(d0 + k) * C] : 0.0;
    for (int base = d0; base < d1; base += A) {
#pragma unroll
        for (int k = 0; k < A; ++k) {
            const int d = base + k;
            if (d < d1) {
                const double v = pre[k];
                if (d + A < d1) pre[k] = in[(long)(d + A) * C];
                out[(long)d * C] = v * 1.0000001;
            }
        }
    }
}

template <int A>
float run_walk(const double* i0, const double* i1, double* o0, double* o1, long C, int D, int parts) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rpp = (D + parts - 1) / parts;
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)parts, 2);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(walk<A>, grid, dim3(256), 0, 0, i0, i1, o0, o1, C, D, rpp);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const long C = argc > 1 ? atol(argv[1]) : 1036800;
    const int D = argc > 2 ? atoi(argv[2]) : 366;
    const size_t n = (size_t)C * D;
    double *i0, *i1, *o0, *o1;
    hipMalloc(&i0, n * 8); hipMalloc(&i1, n * 8); hipMalloc(&o0, n * 8); hipMalloc(&o1, n * 8);
    hipMemset(i0, 0, n * 8); hipMemset(i1, 0, n * 8);
    const double gb = 4.0 * n * 8 / 1e9;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(flat, dim3(256 * 32), dim3(256), 0, 0, (const double2*)i0, (double2*)o0, n / 2);
        hipLaunchKernelGGL(flat, dim3(256 * 32), dim3(256), 0, 0, (const double2*)i1, (double2*)o1, n / 2);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("C=%ld D=%d  %.2f GB moved\n", C, D, gb);
    printf("flat copy          %7.3f ms  %6.0f GB/s\n", best, gb / best * 1e3);
    for (int parts : {1, 2, 4, 8}) {
        float a1 = run_walk<1>(i0, i1, o0, o1, C, D, parts);
        float a2 = run_walk<2>(i0, i1, o0, o1, C, D, parts);
        float a4 = run_walk<4>(i0, i1, o0, o1, C, D, parts);
        float a8 = run_walk<8>(i0, i1, o0, o1, C, D, parts);
        printf("walk parts=%d  A=1 %7.3f ms %5.0f GB/s | A=2 %7.3f %5.0f | A=4 %7.3f %5.0f | A=8 %7.3f %5.0f\n", parts, a1,
               gb / a1 * 1e3, a2, gb / a2 * 1e3, a4, gb / a4 * 1e3, a8, gb / a8 * 1e3);
    }
    return 0;
}
