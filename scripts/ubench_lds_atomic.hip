// Cost of ds_add_f64 on gfx950 in units of FP64 FMA issue slots, for several lane/address patterns.
// Each wave runs ITER iterations of (NF dependent-free FMAs + 6 LDS atomics); the no-atomic run is the
// baseline.  256 workgroups x 8 waves, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int iters) {
    __shared__ double lds[8 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *m = lds + wave * 1024;
    for (int i = lane; i < 1024; i += 64) m[i] = 0.0;
    __syncthreads();
    double a0 = 1.0 + lane * 1e-9, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    const double f = 1.0000001;
    int addr;
    bool act = true;
    if (MODE == 1) addr = lane * 8;                         // 64 lanes, 64 distinct addresses
    if (MODE == 2) addr = (lane & 7) * 8;                   // 64 lanes, 8 addresses (8-way conflict)
    if (MODE == 3) addr = 0;                                // 64 lanes, one address
    if (MODE == 4) { addr = lane * 8; act = lane < 20; }    // 20 lanes, distinct
    if (MODE == 5) { addr = lane * 8; act = lane < 4; }     // 4 lanes
    if (MODE == 6) addr = (lane < 45 ? (lane & 7) : lane) * 8;   // 45 lanes on 8 addresses, 19 distinct
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            a0 = fma(a0, f, 1e-9); a1 = fma(a1, f, 1e-9); a2 = fma(a2, f, 1e-9); a3 = fma(a3, f, 1e-9);
            a4 = fma(a4, f, 1e-9); a5 = fma(a5, f, 1e-9); a6 = fma(a6, f, 1e-9); a7 = fma(a7, f, 1e-9);
        }
        if (MODE && act) {
#pragma unroll
            for (int q = 0; q < 6; ++q) atomicAdd(m + addr + q, a0);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m[lane];
}

template <int MODE>
float run(double *d, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    double *d; CHECK(hipMalloc(&d, 256 * 512 * sizeof(double)));
    const int iters = 20000;
    const float b = run<0>(d, iters);
    const double per_fma = b / (iters * 64.0);          // ms per FMA instruction slot (per wave pair schedule)
    printf("baseline: %.2f ms for %d x 64 FMAs per wave\n", b, iters);
    const char *names[] = {"", "64 lanes, distinct addresses", "64 lanes on 8 addresses", "64 lanes on 1 address",
                           "20 lanes, distinct", "4 lanes", "45 lanes on 8 addresses + 19 distinct"};
    float t[7];
    t[1] = run<1>(d, iters); t[2] = run<2>(d, iters); t[3] = run<3>(d, iters);
    t[4] = run<4>(d, iters); t[5] = run<5>(d, iters); t[6] = run<6>(d, iters);
    for (int m = 1; m <= 6; ++m)
        printf("%-40s %.2f ms  -> %.1f FMA slots per ds_add_f64\n", names[m], t[m], (t[m] - b) / (iters * 6.0) / per_fma);
    return 0;
}
