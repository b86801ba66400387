/* CPU ORACLE (plain C) -- test infrastructure, NOT product code.
 *
 * Restates, on the host with libm, the arithmetic of the reference hot path:
 *   - scipy.stats.betabinom.pmf as the reference calls it (BalLeRMix+_v1.py:308,369,371,382).
 *     scipy is a third-party dependency that is not vendored in /root/reference
 *     (requirements.txt:1 "scipy>=1.5.0"; this image pins scipy 1.15.3, whose
 *     special functions are the published Cephes algorithms: lgam / Gamma / stirf /
 *     rgamma / lbeta, shipped as headers in scipy/special/xsf/cephes/).  The
 *     functions below restate those published algorithms, operation order included,
 *     because the reference's results at alpha_beta = 1e6..1e9 depend on the
 *     rounding noise of lgam(a)+lgam(b)-lgam(a+b) (hard part 1).
 *   - NormalizedBetaBinom (v1:310-433) as a (k,n)-indexed table.
 *   - calcBaller (v1:436-507) in the algebraically equal log1p form.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Pinned by tests/test_oracle_golden.py against fixtures generated
 * by running the reference (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: scipy's x86-64 wheels are built without FMA.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- published Cephes constants ------------------------------------------------- */
static const double GAMMA_P[] = {1.60119522476751861407E-4, 1.19135147006586384913E-3, 1.04213797561761569935E-2,
                                 4.76367800457137231464E-2, 2.07448227648435975150E-1, 4.94214826801497100753E-1,
                                 9.99999999999999996796E-1};
static const double GAMMA_Q[] = {-2.31581873324120129819E-5, 5.39605580493303397842E-4, -4.45641913851797240494E-3,
                                 1.18139785222060435552E-2, 3.58236398605498653373E-2, -2.34591795718243348568E-1,
                                 7.14304917030273074085E-2, 1.00000000000000000320E0};
static const double GAMMA_STIR[5] = {7.87311395793093628397E-4, -2.29549961613378126380E-4,
                                     -2.68132617805781232825E-3, 3.47222221605458667310E-3,
                                     8.33333333333482257126E-2};
static const double LGAM_A[] = {8.11614167470508450300E-4, -5.95061904284301438324E-4, 7.93650340457716943945E-4,
                                -2.77777777730099687205E-3, 8.33333333333331927722E-2};
static const double LGAM_B[] = {-1.37825152569120859100E3, -3.88016315134637840924E4, -3.31612992738871184744E5,
                                -1.16237097492762307383E6, -1.72173700820839662146E6, -8.53555664245765465627E5};
static const double LGAM_C[] = {-3.51815701436523470549E2, -1.70642106651881159223E4, -2.20528590553854454839E5,
                                -1.13933444367982507207E6, -2.53252307177582951285E6, -2.01889141433532773231E6};
static const double RGAMMA_R[] = {
    3.13173458231230000000E-17, -6.70718606477908000000E-16, 2.20039078172259550000E-15,
    2.47691630348254132600E-13, -6.60074100411295197440E-12, 5.13850186324226978840E-11,
    1.08965386454418662084E-9,  -3.33964630686836942556E-8,  2.68975996440595483619E-7,
    2.96001177518801696639E-6,  -8.04814124978471142852E-5,  4.16609138709688864714E-4,
    5.06579864028608725080E-3,  -6.41925436109158228810E-2,  -4.98558728684003594785E-3,
    1.27546015610523951063E-1};
#define MAXGAM 171.624376956302725
#define MAXSTIR 143.01608
#define SQRTPI 2.50662827463100050242E0
#define LS2PI 0.91893853320467274178
#define MAXLGM 2.556348e305
#define ASYMP_FACTOR 1e6

static double polevl(double x, const double *c, int n) {
    double ans = c[0];
    for (int i = 1; i <= n; i++) ans = ans * x + c[i];
    return ans;
}
static double p1evl(double x, const double *c, int n) {
    double ans = x + c[0];
    for (int i = 1; i < n; i++) ans = ans * x + c[i];
    return ans;
}
static double chbevl(double x, const double *c, int n) {
    double b0 = c[0], b1 = 0.0, b2 = 0.0;
    for (int i = 1; i < n; i++) {
        b2 = b1;
        b1 = b0;
        b0 = x * b1 - b2 + c[i];
    }
    return 0.5 * (b0 - b2);
}

/* Gamma by Stirling's formula, 33 < x <= 171.6 */
static double stirf(double x) {
    if (x >= MAXGAM) return INFINITY;
    double w = 1.0 / x;
    w = 1.0 + w * polevl(w, GAMMA_STIR, 4);
    double y = exp(x);
    if (x > MAXSTIR) {
        double v = pow(x, 0.5 * x - 0.25);
        y = v * (v / y);
    } else {
        y = pow(x, x - 0.5) / y;
    }
    return SQRTPI * y * w;
}

/* Gamma(x) for x > 0 (the reference only reaches positive arguments) */
double orc_gamma(double x) {
    if (!isfinite(x)) return x > 0 ? x : NAN;
    if (x == 0) return copysign(INFINITY, x);
    if (x > 33.0) return stirf(x);
    if (x < 0.0) return NAN; /* never reached from betabinom with a,b > 0 */
    double z = 1.0;
    while (x >= 3.0) {
        x -= 1.0;
        z *= x;
    }
    while (x < 2.0) {
        if (x < 1.e-9) return z / ((1.0 + 0.5772156649015329 * x) * x);
        z /= x;
        x += 1.0;
    }
    if (x == 2.0) return z;
    x -= 2.0;
    return z * polevl(x, GAMMA_P, 6) / polevl(x, GAMMA_Q, 7);
}

/* 1/Gamma(x), x > 0 */
static double rgamma_pos(double x) {
    if (x == 0) return x;
    if (fabs(x) > 4.0) return 1.0 / orc_gamma(x);
    double z = 1.0, w = x;
    while (w > 1.0) {
        w -= 1.0;
        z *= w;
    }
    if (w == 0.0) return 0.0;
    if (w == 1.0) return 1.0 / z;
    return w * (1.0 + chbevl(4.0 * w - 2.0, RGAMMA_R, 16)) / z;
}

/* log|Gamma(x)|, x > 0 */
double orc_lgam(double x) {
    if (!isfinite(x)) return x;
    if (x < 13.0) {
        double z = 1.0, p = 0.0, u = x;
        while (u >= 3.0) {
            p -= 1.0;
            u = x + p;
            z *= u;
        }
        while (u < 2.0) {
            if (u == 0.0) return INFINITY;
            z /= u;
            p += 1.0;
            u = x + p;
        }
        if (z < 0.0) z = -z;
        if (u == 2.0) return log(z);
        p -= 2.0;
        x = x + p;
        p = x * polevl(x, LGAM_B, 5) / p1evl(x, LGAM_C, 6);
        return log(z) + p;
    }
    if (x > MAXLGM) return INFINITY;
    if (x >= 1000.0) {
        double q = (x - 0.5) * log(x) - x + LS2PI;
        if (x > 1.0e8) return q;
        double p = 1.0 / (x * x);
        p = ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p + 0.0833333333333333333333) / x;
        return q + p;
    }
    double q = (x - 0.5) * log(x) - x + LS2PI;
    double p = 1.0 / (x * x);
    return q + polevl(p, LGAM_A, 4) / x;
}

static double lbeta_asymp(double a, double b) {
    double r = orc_lgam(b);
    r -= b * log(a);
    r += b * (1 - b) / (2 * a);
    r += b * (1 - b) * (1 - 2 * b) / (12 * a * a);
    r += -b * b * (1 - b) * (1 - b) / (12 * a * a * a);
    return r;
}

/* log|B(a,b)|, a,b > 0 */
double orc_lbeta(double a, double b) {
    double y;
    if (fabs(a) < fabs(b)) {
        y = a;
        a = b;
        b = y;
    }
    if (fabs(a) > ASYMP_FACTOR * fabs(b) && a > ASYMP_FACTOR) return lbeta_asymp(a, b);
    y = a + b;
    if (fabs(y) > MAXGAM || fabs(a) > MAXGAM || fabs(b) > MAXGAM) {
        y = orc_lgam(y);
        y = orc_lgam(b) - y;
        y = orc_lgam(a) + y;
        return y;
    }
    y = rgamma_pos(y);
    a = orc_gamma(a);
    b = orc_gamma(b);
    if (isinf(y)) return INFINITY;
    if (fabs(fabs(a * y) - 1.0) > fabs(fabs(b * y) - 1.0)) {
        y = b * y;
        y *= a;
    } else {
        y = a * y;
        y *= b;
    }
    if (y < 0) y = -y;
    return log(y);
}

/* scipy.stats.betabinom(n,a,b).pmf(k): exp(-log(n+1) - betaln(n-k+1,k+1) + betaln(k+a,n-k+b)
 * - betaln(a,b)), zero outside 0..n, clipped to [0,1] (rv_discrete.pmf). */
double orc_betabinom_pmf(int k, int n, double a, double b) {
    if (k < 0 || k > n) return 0.0;
    double combiln = -log((double)(n + 1)) - orc_lbeta((double)(n - k + 1), (double)(k + 1));
    double lp = combiln + orc_lbeta(k + a, n - k + b) - orc_lbeta(a, b);
    double p = exp(lp);
    if (p < 0.0) p = 0.0;
    if (p > 1.0) p = 1.0;
    return p;
}

/* numpy's pairwise summation for n <= 128 (what np.sum does on a short contiguous array) */
static double np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

enum { ST_B2 = 0, ST_B2MAF = 1, ST_B0 = 2, ST_B0MAF = 3, ST_B1 = 4 };

/* raw (un-mirrored) probabilities for all counts, v1:375-396 */
static void raw_probs(int stat, int n, double a, double b, double *out) {
    if (stat == ST_B1) {
        double pn = orc_betabinom_pmf(n, n, a, b);
        out[0] = pn;
        out[1] = 1. - pn - pn;
        return;
    }
    for (int k = 0; k <= n; k++) {
        double p = orc_betabinom_pmf(k, n, a, b);
        if (stat == ST_B2MAF || stat == ST_B0MAF) {
            p = p + orc_betabinom_pmf(n - k, n, a, b);
            if (n % 2 == 0 && k == n / 2) p = p / 2;
        }
        out[k] = p;
    }
}

static int excluded(int stat, int n, int m, int *ex) {
    int c = 0;
    for (int j = 0; j < m; j++) ex[c++] = j;
    if (stat == ST_B2MAF)
        for (int j = n - m + 1; j < n; j++) ex[c++] = j;
    else if (stat == ST_B0)
        ex[c++] = n;
    else if (stat == ST_B0MAF)
        for (int j = n - m + 1; j <= n; j++) ex[c++] = j;
    return c;
}

/* NormalizedBetaBinom as a table: out[ix][ia][k], k = 0..n (2 entries for B1). v1:319-433 */
void orc_sel_table(int stat, int n, int min_count, int nx, const double *xs, int nab, const double *abetas,
                   double *out) {
    int rows = stat == ST_B1 ? 2 : n + 1;
    double *r1 = (double *)malloc(sizeof(double) * (n + 2));
    double *r2 = (double *)malloc(sizeof(double) * (n + 2));
    int *ex = (int *)malloc(sizeof(int) * (2 * min_count + 4));
    double *ep = (double *)malloc(sizeof(double) * (2 * min_count + 4));
    int nex = excluded(stat, n, min_count, ex);
    for (int ix = 0; ix < nx; ix++)
        for (int ia = 0; ia < nab; ia++) {
            double x = xs[ix], a = abetas[ia];
            double xm = 1. - x;
            double b1 = a / x - a, b2 = a / xm - a; /* v1:316 */
            raw_probs(stat, n, a, b1, r1);
            raw_probs(stat, n, a, b2, r2);
            for (int j = 0; j < nex; j++)
                ep[j] = 0.5 * (orc_betabinom_pmf(ex[j], n, a, b1) + orc_betabinom_pmf(ex[j], n, a, b2));
            double base = 1. - np_sum(ep, nex);
            double *o = out + ((size_t)ix * nab + ia) * rows;
            for (int k = 0; k < rows; k++) o[k] = (0.5 * (r1[k] + r2[k])) / base;
        }
    free(r1);
    free(r2);
    free(ex);
    free(ep);
}

static int64_t lower_bound(const double *g, int64_t N, double v) {
    int64_t a = 0, b = N;
    while (a < b) {
        int64_t m = (a + b) / 2;
        if (g[m] < v) a = m + 1; else b = m;
    }
    return a;
}

/* calcBaller in log1p form over a (k,n)-indexed table R[ix][ia][row] = P_sel*prop/g - 1:
 *   T(A,x,a) = 2 * sum_{i in win(A)} log1p(alpha_i * R[x][a][row_i]),  alpha_i = exp(-A*|g_i - t|)
 *   win(A)   = { i in [win_lo,win_hi] : exp(-A*d_i) >= 1e-8 and g_i != t }            (v1:455-457)
 * argmax in the order A outer, x, a inner with strict '>' from Tmax = 0 (v1:451,501).
 * Outputs per test site: clr, ix, ia, iA (=-1 when nothing beat 0), nsites. */
int orc_scan(int nx, int nab, int rows, const double *R, const double *A, int nA, int64_t N,
             const double *genpos, const int32_t *row, int64_t M, const double *test_gen,
             const int64_t *win_lo, const int64_t *win_hi, double *clr, int32_t *oix, int32_t *oia,
             int32_t *oiA, int32_t *onsites) {
    int np_ = nx * nab;
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * np_);
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < M; t++) {
            double tg = test_gen[t];
            double best = 0.0;
            int bix = -1, bia = -1, biA = -1, bns = 0;
            int64_t lo = win_lo[t], hi = win_hi[t];
            if (lo < 0) lo = 0;
            if (hi > N - 1) hi = N - 1;
            for (int iA = 0; iA < nA; iA++) {
                for (int p = 0; p < np_; p++) acc[p] = 0.0;
                int ns = 0;
                /* genpos is non-decreasing: only sites within 19/A can pass exp(-A*d) >= 1e-8 */
                double rad = 19.0 / A[iA];
                int64_t i0 = lower_bound(genpos, N, tg - rad), i1 = lower_bound(genpos, N, tg + rad);
                if (i0 > 0) i0--;
                if (i1 > N - 1) i1 = N - 1;
                if (i0 < lo) i0 = lo;
                if (i1 > hi) i1 = hi;
                for (int64_t i = i0; i <= i1; i++) {
                    double d = fabs(genpos[i] - tg);
                    double al = exp(-A[iA] * d);
                    if (!(al >= 1e-8) || genpos[i] == tg) continue;
                    ns++;
                    const double *Rr = R + row[i];
                    for (int p = 0; p < np_; p++) acc[p] += log1p(al * Rr[(size_t)p * rows]);
                }
                if (ns == 0) continue;
                for (int p = 0; p < np_; p++) {
                    double T = 2.0 * acc[p];
                    if (T > best) {
                        best = T;
                        bix = p / nab;
                        bia = p % nab;
                        biA = iA;
                        bns = ns;
                    }
                }
            }
            clr[t] = best;
            oix[t] = bix;
            oia[t] = bia;
            oiA[t] = biA;
            onsites[t] = bns;
        }
        free(acc);
    }
    return 0;
}
