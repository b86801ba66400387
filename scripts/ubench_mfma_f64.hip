// v_mfma_f64_16x16x4_f64 on gfx950: operand layout check and issue rates, alone and beside FP64 VALU work.
// The scan kernel's near-field step is a small contraction: for 4 sites, 16 test sites j and 64 grid pairs p,
//     prod_u (1 + F_j v_u(p)) - 1 = sum_{k=1..4} F_j^k e_k(p)            (e_k: elementary symmetric polynomials)
// i.e. D[16 x 16] = A[16 x 4] B[4 x 16] per 16 pairs -- exactly one v_mfma_f64_16x16x4_f64.  This program measures
// what that costs against the 5-instruction Horner form per (j, lane) the kernel used in round 1.
//   make -C scripts && ./scripts/ubench_mfma_f64
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- layout
__global__ void layout_kernel(const double *A, const double *B, double *D) {
    // A[16][4] row-major, B[4][16] row-major, D[16][16] row-major
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];      // lane l: A[i = l & 15][k = l >> 4]
    const double b = B[(l >> 4) * 16 + (l & 15)];     // lane l: B[k = l >> 4][n = l & 15]
    d4 c = {0., 0., 0., 0.};
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];   // row = (l >> 4) + 4 r, col = l & 15
}

// 4x4 transpose of (register index, 16-lane row) by permlane swaps: out[m] row g = in[g] row m
__device__ __forceinline__ void transpose4(double (&e)[4]) {
    unsigned lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lo[k] = (unsigned)__double2loint(e[k]); hi[k] = (unsigned)__double2hiint(e[k]); }
    auto tr = [](unsigned (&x)[4]) {
        u2 r;
        r = __builtin_amdgcn_permlane16_swap(x[0], x[1], false, false); x[0] = r.x; x[1] = r.y;
        r = __builtin_amdgcn_permlane16_swap(x[2], x[3], false, false); x[2] = r.x; x[3] = r.y;
        r = __builtin_amdgcn_permlane32_swap(x[0], x[2], false, false); x[0] = r.x; x[2] = r.y;
        r = __builtin_amdgcn_permlane32_swap(x[1], x[3], false, false); x[1] = r.x; x[3] = r.y;
    };
    tr(lo);
    tr(hi);
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = __hiloint2double((int)hi[k], (int)lo[k]);
}

__global__ void transpose_kernel(const double *in, double *out) {   // in[4][64] -> out[4][64]
    const int l = threadIdx.x;
    double e[4];
    for (int k = 0; k < 4; ++k) e[k] = in[k * 64 + l];
    transpose4(e);
    for (int k = 0; k < 4; ++k) out[k * 64 + l] = e[k];
}

// ---------------------------------------------------------------- rates
// MODE 0: Horner block (round-1 form): 4 v, 10 e, 16 x (4 fma + 1 mul)           = 94 VALU per 4 sites
// MODE 1: MFMA block: 4 v, 10 e, 8 swaps, 4 MFMA, 16 fma                          = 38 VALU + 4 MFMA
// MODE 2: bare MFMA stream, 4 independent accumulators
// MODE 3: MFMA block without the transposes (cost of the swaps)
// MODE 4: one MFMA block + one Horner block per iteration (both pipes from one wave)
template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(double *out, int iters, double seed) {
    const int lane = threadIdx.x & 63;
    double acc[16], accm[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = 1.0; accm[j] = 1.0; }
    double F[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) F[j] = __builtin_amdgcn_readfirstlane((int)(seed * 1000)) * 1e-6 + 0.4 + 0.01 * j;   // wave-uniform
    const double fl = 0.4 + 0.01 * (lane & 15);
    double fp = fl;
    for (int k = 0; k < (lane >> 4); ++k) fp *= fl;          // lane: F_{l&15}^{(l>>4)+1}
    double R0 = -0.5 + 1e-3 * lane, E0 = 0.3;
    for (int it = 0; it < iters; ++it) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (E0 + 1e-4 * u) * R0;
        E0 = E0 * 0.9999;
        if (MODE == 2) {
            d4 c0 = {accm[0], accm[1], accm[2], accm[3]}, c1 = {accm[4], accm[5], accm[6], accm[7]},
               c2 = {accm[8], accm[9], accm[10], accm[11]}, c3 = {accm[12], accm[13], accm[14], accm[15]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fp, v[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fp, v[1], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fp, v[2], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(fp, v[3], c3, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { accm[r] = c0[r]; accm[4 + r] = c1[r]; accm[8 + r] = c2[r]; accm[12 + r] = c3[r]; }
            continue;
        }
        const double s01 = v[0] + v[1], q01 = v[0] * v[1];
        const double s23 = v[2] + v[3], q23 = v[2] * v[3];
        double e[4];
        e[0] = s01 + s23;
        e[1] = fma(s01, s23, q01 + q23);
        e[2] = fma(q01, s23, q23 * s01);
        e[3] = q01 * q23;
        if (MODE == 0 || MODE == 4) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double t = fma(F[j], e[3], e[2]);
                t = fma(F[j], t, e[1]);
                t = fma(F[j], t, e[0]);
                acc[j] *= fma(F[j], t, 1.0);
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 1 || MODE == 3 || MODE == 4) {
            double b[4] = {e[0], e[1], e[2], e[3]};
            if (MODE != 3) transpose4(b);
            const d4 z = {0., 0., 0., 0.};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(fp, b[m], z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) accm[4 * m + r] = fma(accm[4 * m + r], d[r], accm[4 * m + r]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j] + accm[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(double *d, int blocks, int threads, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    // ---- layout
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i + 0.01 * i * i; hB[i] = 0.5 - 0.11 * i + 0.003 * i * i; }
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
        double s = 0; for (int k = 0; k < 4; ++k) s = fma(hA[i * 4 + k], hB[k * 16 + n], s);
        ref[i * 16 + n] = s;
    }
    double *dA, *dB, *dD;
    CHECK(hipMalloc(&dA, sizeof hA)); CHECK(hipMalloc(&dB, sizeof hB)); CHECK(hipMalloc(&dD, sizeof hD));
    CHECK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CHECK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
    int bad = 0, bitexact = 0;
    for (int i = 0; i < 256; ++i) { if (fabs(hD[i] - ref[i]) > 1e-12 * fabs(ref[i])) bad++; if (hD[i] == ref[i]) bitexact++; }
    printf("layout: %d / 256 wrong; %d / 256 bit-identical to a k-ordered fma chain from C\n", bad, bitexact);
    // ---- transpose
    double hin[256], hout[256];
    for (int k = 0; k < 4; ++k) for (int l = 0; l < 64; ++l) hin[k * 64 + l] = k * 1000 + l;
    double *din, *dout;
    CHECK(hipMalloc(&din, sizeof hin)); CHECK(hipMalloc(&dout, sizeof hout));
    CHECK(hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(transpose_kernel, dim3(1), dim3(64), 0, 0, din, dout);
    CHECK(hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost));
    bad = 0;
    for (int m = 0; m < 4; ++m) for (int g = 0; g < 4; ++g) for (int q = 0; q < 16; ++q)
        if (hout[m * 64 + 16 * g + q] != hin[g * 64 + 16 * m + q]) bad++;
    printf("transpose4 (out[m] row g = in[g] row m): %d / 256 wrong\n", bad);
    // ---- rates
    double *d; CHECK(hipMalloc(&d, (size_t)1024 * 512 * sizeof(double)));
    const int iters = 20000;
    const char *names[] = {"Horner block (94 VALU / 4 sites)", "MFMA block (38 VALU + 4 MFMA)", "bare MFMA x16 per iteration",
                           "MFMA block without swaps", "one MFMA block + one Horner block"};
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps;
        float t[5];
        t[0] = run<0>(d, 256, threads, iters); t[1] = run<1>(d, 256, threads, iters); t[2] = run<2>(d, 256, threads, iters);
        t[3] = run<3>(d, 256, threads, iters); t[4] = run<4>(d, 256, threads, iters);
        for (int m = 0; m < 5; ++m) {
            const double per = t[m] * 1e6 / iters;            // ns per iteration per wave... per SIMD: / wps waves sharing it
            const double blocks_per_iter = m == 4 ? 2.0 : 1.0;
            printf("%d wave/SIMD  %-36s %8.3f ms -> %7.1f ns per iteration per wave = %6.0f cycles @2.4GHz; per SIMD per 4-site block %6.0f cycles%s\n",
                   wps, names[m], t[m], per, per * 2.4, per * 2.4 / wps / blocks_per_iter,
                   m == 2 ? "  (/16 = per MFMA)" : "");
        }
    }
    return 0;
}
