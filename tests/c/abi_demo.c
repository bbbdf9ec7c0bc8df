/* A C99 caller of the C ABI (include/xmhw_amd.h), with no Python and no C++ in sight: the climatology
 * of a small synthetic grid through the resident-data entry points, checked against the one-shot host
 * entry point.  Built and run by tests/test_c_abi_program.py.
 *
 *   gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude tests/c/abi_demo.c -Lxmhw_amd -lxmhw_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xmhw_amd.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != 0) {                                                               \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, xmhw_last_error());  \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static int is_leap(int y) { return (y % 4 == 0 && y % 100 != 0) || y % 400 == 0; }

int main(void) {
    /* 12 years of daily data on the reference's 366-slot calendar (add_doy, identify.py:73-76) */
    const int y0 = 2001, ny = 12;
    const int64_t C = 100;
    int64_t T = 0;
    int y, d;
    for (y = y0; y < y0 + ny; ++y) T += is_leap(y) ? 366 : 365;
    int32_t *doy = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    float *ts = (float *)malloc(sizeof(float) * (size_t)(T * C));
    int64_t t = 0, c;
    unsigned long long state = 12345;
    if (!doy || !ts) return 2;
    for (y = y0; y < y0 + ny; ++y) {
        const int n = is_leap(y) ? 366 : 365;
        for (d = 1; d <= n; ++d, ++t) {
            doy[t] = (!is_leap(y) && d >= 60) ? d + 1 : d;
            for (c = 0; c < C; ++c) {
                state = state * 6364136223846793005ULL + 1442695040888963407ULL;
                ts[t * C + c] = 15.0f + 5.0f * (float)sin(6.283185307 * (double)(t - 3 * c) / 365.25) +
                                (float)((double)(state >> 40) / 16777216.0 - 0.5);
            }
        }
    }
    int ndev = 0;
    CHECK(xmhw_device_count(&ndev));
    printf("xmhw_amd %d  arch %s  devices %d\n", xmhw_version(), xmhw_arch(), ndev);

    const int32_t D = 366;
    double *th_a = (double *)malloc(sizeof(double) * (size_t)(D * C)), *se_a = (double *)malloc(sizeof(double) * (size_t)(D * C));
    double *th_b = (double *)malloc(sizeof(double) * (size_t)(D * C)), *se_b = (double *)malloc(sizeof(double) * (size_t)(D * C));
    if (!th_a || !se_a || !th_b || !se_b) return 2;

    /* one-shot: host buffers in, host buffers out */
    CHECK(xmhw_clim_host_f32(ts, doy, T, C, D, 5, 0.9, 1, 31, 1, 0, th_a, se_a));

    /* resident data: plan once, raw selection + finish on device buffers */
    xmhw_plan *plan = NULL;
    void *d_ts = NULL, *d_rt = NULL, *d_rs = NULL, *d_t = NULL, *d_s = NULL;
    int32_t Dp = 0, ntracks = 0, kernel = 0, nsteps = 0, step_min = 0;
    CHECK(xmhw_plan_create(doy, T, 5, &plan));
    CHECK(xmhw_plan_info(plan, &Dp, &ntracks, &kernel, &nsteps, &step_min));
    if (Dp != D) { fprintf(stderr, "D = %d\n", Dp); return 1; }
    CHECK(xmhw_malloc(&d_ts, sizeof(float) * (size_t)(T * C)));
    CHECK(xmhw_malloc(&d_rt, sizeof(double) * (size_t)(D * C)));
    CHECK(xmhw_malloc(&d_rs, sizeof(double) * (size_t)(D * C)));
    CHECK(xmhw_malloc(&d_t, sizeof(double) * (size_t)(D * C)));
    CHECK(xmhw_malloc(&d_s, sizeof(double) * (size_t)(D * C)));
    CHECK(xmhw_memcpy_h2d(d_ts, ts, sizeof(float) * (size_t)(T * C), NULL));
    CHECK(xmhw_clim_raw_f32(plan, (const float *)d_ts, C, C, 0.9, 0, (double *)d_rt, (double *)d_rs, C, NULL));
    CHECK(xmhw_clim_finish(plan, (const double *)d_rt, (const double *)d_rs, C, C, 1, 1, 31, (double *)d_t,
                           (double *)d_s, NULL));
    CHECK(xmhw_memcpy_d2h(th_b, d_t, sizeof(double) * (size_t)(D * C), NULL));
    CHECK(xmhw_memcpy_d2h(se_b, d_s, sizeof(double) * (size_t)(D * C), NULL));

    if (memcmp(th_a, th_b, sizeof(double) * (size_t)(D * C)) != 0 || memcmp(se_a, se_b, sizeof(double) * (size_t)(D * C)) != 0) {
        fprintf(stderr, "one-shot and resident results differ\n");
        return 1;
    }
    printf("ntracks %d kernel %d: thresh[0][0] = %.6f seas[0][0] = %.6f thresh[59][7] = %.6f\n", ntracks, kernel,
           th_a[0], se_a[0], th_a[59 * C + 7]);

    /* argument errors come back as status codes with a message, never as exceptions */
    if (xmhw_clim_host_f32(ts, doy, T, C, D, 5, 0.9, 1, 30, 1, 0, th_a, se_a) == 0) {
        fprintf(stderr, "an even smoothing width must be refused\n");
        return 1;
    }
    printf("refused: %s\n", xmhw_last_error());

    xmhw_free(d_ts); xmhw_free(d_rt); xmhw_free(d_rs); xmhw_free(d_t); xmhw_free(d_s);
    xmhw_plan_destroy(plan);
    free(doy); free(ts); free(th_a); free(se_a); free(th_b); free(se_b);
    printf("ok\n");
    return 0;
}
