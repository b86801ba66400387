// Diagnostic: launch bb_lut_kernel directly with a tiny model and growing sizes.
#include "../ballermixplus_amd/csrc/bmxscan.hip"
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 5;
    int nx = argc > 2 ? atoi(argv[2]) : 1;
    int nab = argc > 3 ? atoi(argv[3]) : 1;
    int stat = argc > 4 ? atoi(argv[4]) : 0;
    int rows = n + 1;
    std::vector<int32_t> sizes{n}, row_off{0, rows};
    std::vector<double> g(rows, 0.1), prop{1.0}, x(nx), ab(nab);
    for (int i = 0; i < nx; i++) x[i] = 0.05 * (i + 1);
    double abv[] = {0.001, 1, 5, 100, 1e3, 1e6, 1e9};
    for (int i = 0; i < nab; i++) ab[i] = abv[i % 7];
    LutParams P;
    P.stat = stat; P.min_count = 1; P.n_sizes = 1; P.rows = rows; P.nx = nx; P.nab = nab; P.NP = (nx * nab + 63) / 64 * 64;
    int32_t *ds, *dr; double *dg, *dp, *dx, *da, *ps, *R, *Rt;
    CK(hipMalloc(&ds, 4)); CK(hipMalloc(&dr, 8)); CK(hipMalloc(&dg, rows * 8)); CK(hipMalloc(&dp, 8));
    CK(hipMalloc(&dx, nx * 8)); CK(hipMalloc(&da, nab * 8));
    size_t tab = (size_t)nx * nab * rows;
    CK(hipMalloc(&ps, tab * 8)); CK(hipMalloc(&R, tab * 8)); CK(hipMalloc(&Rt, (size_t)rows * P.NP * 8));
    CK(hipMemcpy(ds, sizes.data(), 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, row_off.data(), 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, g.data(), rows * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, prop.data(), 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), nx * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da, ab.data(), nab * 8, hipMemcpyHostToDevice));
    P.sizes = ds; P.row_off = dr; P.g = dg; P.prop = dp; P.x = dx; P.abeta = da; P.psel = ps; P.R = R; P.Rt = Rt;
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(bb_lut_kernel, dim3((unsigned)((tab + 127) / 128)), dim3(128), 0, 0, P);
    CK(hipDeviceSynchronize());
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> h(tab);
    CK(hipMemcpy(h.data(), ps, tab * 8, hipMemcpyDeviceToHost));
    double s = 0; for (double v : h) s += v;
    printf("n=%d nx=%d nab=%d stat=%d: %.2f ms, sum psel=%.15g (expect ~%d)\n", n, nx, nab, stat, ms, s, nx * nab);
    return 0;
}
