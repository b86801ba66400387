// Microbenchmark: FP64 VALU issue rates on gfx950 (diagnostic; establishes the empirical ceiling
// that bench.py's roofline is compared with).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, const double *in, int iters) {
    double acc[8];
    int ex[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double c1 = in[0], c2 = in[1];          // uniform -> SGPR
    const double f0 = in[2], f1 = in[3], f2 = in[4], f3 = in[5], f4 = in[6], f5 = in[7], f6 = in[8], f7 = in[9];
    double s = in[10] + threadIdx.x * 1e-9, q = in[11] + threadIdx.x * 1e-9;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 1.0 + j * 1e-3 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 8 independent FMA (VGPR operands + 2 SGPR)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fma(acc[j], c1, c2);
        } else if (MODE == 1) {   // 8 independent MUL
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = acc[j] * c1;
        } else if (MODE == 2) {   // the pair-form body: 2 FMA + 1 MUL per j, F_j in SGPRs
            acc[0] *= fma(f0, fma(f0, q, s), 1.0);
            acc[1] *= fma(f1, fma(f1, q, s), 1.0);
            acc[2] *= fma(f2, fma(f2, q, s), 1.0);
            acc[3] *= fma(f3, fma(f3, q, s), 1.0);
            acc[4] *= fma(f4, fma(f4, q, s), 1.0);
            acc[5] *= fma(f5, fma(f5, q, s), 1.0);
            acc[6] *= fma(f6, fma(f6, q, s), 1.0);
            acc[7] *= fma(f7, fma(f7, q, s), 1.0);
            s += 1e-12; q -= 1e-12;   // 2 more DP ops, keeps the compiler from hoisting
        } else if (MODE == 4) {   // exponent extraction as the scan kernels did it up to round 3: integer ops on the high word
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] *= c1;
                unsigned hi = (unsigned)__double2hiint(acc[j]);
                int e = (int)((hi >> 20) & 0x7ffu);
                ex[j] = (e == 0) ? -(1 << 28) : ex[j] + (e - 1023);
                hi = (hi & 0x800fffffu) | 0x3ff00000u;
                acc[j] = __hiloint2double((int)hi, __double2loint(acc[j]));
            }
        } else if (MODE == 5) {   // ... with v_frexp_exp_i32_f64 / v_frexp_mant_f64
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] *= c1;
                ex[j] += __builtin_amdgcn_frexp_exp(acc[j]);
                acc[j] = __builtin_amdgcn_frexp_mant(acc[j]);
            }
        } else if (MODE == 3) {   // FP32 FMA for reference
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (float)acc[j];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fmaf(a[j], (float)c1, (float)c2);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = a[j];
        }
    }
    double r = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += acc[j] + ex[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    double h[12] = {1.0000001, 1e-9, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 1e-3, 1e-7};
    double *din, *dout;
    CK(hipMalloc(&din, sizeof(h)));
    CK(hipMalloc(&dout, 2048 * 256 * 8));
    CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200000;
    const double per_iter[6] = {8, 8, 26, 32, 56, 32};       // VALU instructions per iteration per wave
    const char *name[6] = {"v_fma_f64 x8", "v_mul_f64 x8", "pair body (16 fma + 8 mul + 2 add)", "v_fma_f32 x32 (+cvt)",
                           "mul + integer renorm x8 (7 VALU each)", "mul + frexp renorm x8 (4 VALU each)"};
    for (int mode = 0; mode < 6; ++mode)
        for (int blocks : {256, 512, 1024}) {       // 1, 2, 4 waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, dout, din, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, dout, din, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, dout, din, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, dout, din, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, dout, din, iters);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, dout, din, iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            double waves_per_simd = blocks / 256.0;
            double inst = per_iter[mode] * iters * waves_per_simd;       // per SIMD
            printf("%-38s %4d blocks (%.0f wave/SIMD): %8.3f ms  -> %.3f ns per wave-instruction per SIMD (%.2f cyc @2.4GHz)\n",
                   name[mode], blocks, waves_per_simd, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
        }
    return 0;
}
