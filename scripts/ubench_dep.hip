// Microbenchmark: what a dependent FP64 instruction costs on gfx950 -- v_fma_f64 / v_mul_f64 streams in which every instruction
// reads the result written D instructions earlier (D = 1: one chain; D = 8: eight chains round-robin), at 1 and 2 waves per SIMD.
// Diagnostic: tells whether chain-major schedules (Horner chains, power ladders) in the scan kernels leave issue slots empty.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int D, int OP>
__global__ __launch_bounds__(256) void k(double *out, const double *in, int iters) {
    double a[8];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double c1 = in[0] + threadIdx.x * 1e-12, c2 = in[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + j * 1e-3 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / D; ++r)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c2));
            }
    }
    double r = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) {      // core clocks and 100 MHz ticks of this wave's lifetime
        out[1024 * 256] = (double)(__builtin_amdgcn_s_memtime() - t0);
        out[1024 * 256 + 1] = (double)(__builtin_amdgcn_s_memrealtime() - r0);
    }
}

template <int D, int OP>
static void run(double *dout, const double *din, const char *name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int blocks : {256, 512, 1024}) {       // 1, 2, 4 waves per SIMD
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<D, OP>), dim3(blocks), dim3(256), 0, 0, dout, din, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        double clk[2];
        CK(hipMemcpy(clk, dout + 1024 * 256, sizeof(clk), hipMemcpyDeviceToHost));
        const double ghz = clk[0] / clk[1] * 0.1;                 // core clocks per 100 MHz tick
        const double w = blocks / 256.0, inst = 64.0 * iters * w;
        printf("%s distance %d, %.0f wave/SIMD: %8.3f ms, clock %.3f GHz -> %.2f cycles per instruction per SIMD\n", name, D, w, ms, ghz,
               ms * 1e6 / inst * ghz);
    }
}

int main() {
    double h[2] = {1.0000001, 1e-9};
    double *din, *dout;
    CK(hipMalloc(&din, sizeof(h)));
    CK(hipMalloc(&dout, (1024 * 256 + 2) * 8));
    CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    run<1, 0>(dout, din, "v_fma_f64"); run<2, 0>(dout, din, "v_fma_f64"); run<4, 0>(dout, din, "v_fma_f64"); run<8, 0>(dout, din, "v_fma_f64");
    run<1, 1>(dout, din, "v_mul_f64"); run<2, 1>(dout, din, "v_mul_f64"); run<4, 1>(dout, din, "v_mul_f64"); run<8, 1>(dout, din, "v_mul_f64");
    run<1, 2>(dout, din, "v_add_f64"); run<2, 2>(dout, din, "v_add_f64"); run<8, 2>(dout, din, "v_add_f64");
    return 0;
}
