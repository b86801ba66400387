// Diagnostic: run pieces of bmx_math.h on the device one by one (mode = argv[1]).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../ballermixplus_amd/csrc/bmx_math.h"

__global__ void probe(int mode, const double *in, double *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = in[i];
    double r = 0;
    if (mode == 0) r = bmx::crlog(x);
    else if (mode == 1) r = bmx::cephes::lgam_pos(x);
    else if (mode == 2) r = bmx::cephes::gamma_pos(x);
    else if (mode == 3) r = bmx::cephes::rgamma_pos(x);
    else if (mode == 4) r = bmx::cephes::lbeta_pos(x, 2.0 * x + 1.0);
    else if (mode == 5) r = bmx::betabinom_pmf(i % 51, 50, x, x / 0.3 - x);
    out[i] = r;
}

int main(int argc, char **argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int n = 64;
    double h[n], o[n];
    for (int i = 0; i < n; i++) h[i] = (mode == 3) ? 0.05 * (i + 1) : (i < 20 ? 0.3 * (i + 1) : (i < 40 ? 7.5 * (i - 18) : 1e3 * (i - 38) * (i - 38) * (i - 38)));
    double *di, *dout;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, di, dout, n);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("mode %d: %s\n", mode, hipGetErrorString(e));
    for (int i = 0; i < n; i += 7) printf("  f(%g) = %.17g\n", h[i], o[i]);
    return 0;
}
