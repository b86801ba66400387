// Diagnostic: bisect bb_lut_kernel's body.
#include "../ballermixplus_amd/csrc/bmxscan.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int V>
__global__ void kv(LutParams P) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int npairs = P.nx * P.nab;
    if (gid >= npairs * P.rows) return;
    int p = gid / P.rows, r = gid % P.rows;
    int ix = p / P.nab, ia = p % P.nab;
    int j = 0;
    if (V >= 1) while (j + 1 < P.n_sizes && r >= P.row_off[j + 1]) j++;
    int n = P.sizes[j], k = r - P.row_off[j];
    double x = P.x[ix], a = P.abeta[ia];
    double xm = 1. - x;
    double b1 = a / x - a, b2 = a / xm - a;
    double raw = 1.0;
    if (V >= 2) raw = 0.5 * (raw_prob(P.stat, k, n, a, b1) + raw_prob(P.stat, k, n, a, b2));
    double base = 1.0;
    if (V >= 3) {
        int m = P.min_count, stat = P.stat;
        int nex = m;
        if (stat == BMX_STAT_B2MAF) nex += (m - 1 > 0 ? m - 1 : 0);
        if (stat == BMX_STAT_B0) nex += 1;
        if (stat == BMX_STAT_B0MAF) nex += m;
        auto excl = [&](int i) -> double {
            int c;
            if (i < m) c = i; else if (stat == BMX_STAT_B0) c = n; else c = n - m + 1 + (i - m);
            return 0.5 * (bmx::betabinom_pmf(c, n, a, b1) + bmx::betabinom_pmf(c, n, a, b2));
        };
        base = 1. - np_sum_gen(nex, excl);
    }
    P.psel[((size_t)ix * P.nab + ia) * P.rows + r] = raw / base;
}
int main(int argc, char **argv) {
    int v = atoi(argv[1]);
    int n = 5, nx = 1, nab = 1, rows = n + 1;
    std::vector<int32_t> sizes{n}, row_off{0, rows};
    std::vector<double> g(rows, 0.1), prop{1.0}, x{0.3}, ab{2.0};
    LutParams P;
    P.stat = 0; P.min_count = 1; P.n_sizes = 1; P.rows = rows; P.nx = nx; P.nab = nab; P.NP = 64;
    int32_t *ds, *dr; double *dg, *dp, *dx, *da, *ps;
    CK(hipMalloc(&ds, 4)); CK(hipMalloc(&dr, 8)); CK(hipMalloc(&dg, rows * 8)); CK(hipMalloc(&dp, 8));
    CK(hipMalloc(&dx, 8)); CK(hipMalloc(&da, 8)); CK(hipMalloc(&ps, rows * 8));
    CK(hipMemcpy(ds, sizes.data(), 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, row_off.data(), 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, g.data(), rows * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, prop.data(), 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da, ab.data(), 8, hipMemcpyHostToDevice));
    P.sizes = ds; P.row_off = dr; P.g = dg; P.prop = dp; P.x = dx; P.abeta = da; P.psel = ps; P.R = ps; P.Rt = ps;
    if (v == 0) hipLaunchKernelGGL(kv<0>, dim3(1), dim3(128), 0, 0, P);
    if (v == 1) hipLaunchKernelGGL(kv<1>, dim3(1), dim3(128), 0, 0, P);
    if (v == 2) hipLaunchKernelGGL(kv<2>, dim3(1), dim3(128), 0, 0, P);
    if (v == 3) hipLaunchKernelGGL(kv<3>, dim3(1), dim3(128), 0, 0, P);
    CK(hipDeviceSynchronize());
    double h[6]; CK(hipMemcpy(h, ps, 48, hipMemcpyDeviceToHost));
    printf("v%d ok: %g %g %g\n", v, h[0], h[1], h[5]);
    return 0;
}
