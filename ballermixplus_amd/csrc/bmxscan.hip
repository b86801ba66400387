// bmxscan.hip -- libbmxscan.so: gfx950 kernels + C ABI (include/bmxscan.h).
//
// Hot path being replaced (reference BalLeRMix+_v1.py, "v1:LINE"):
//   K1  bb_lut_kernel              <- NormalizedBetaBinom.__init__/get_raw_probs/get_*_normBase  v1:319-433
//   K2  prep_kernel + clr_scan_prepared_kernel    <- calcBaller, J = 16 / 8 test sites per wave-group (what ships for dense
//                                     test sites; round 3)                                                  v1:436-507
//       prep_solo_kernel + clr_scan_solo_kernel    <- calcBaller, one test site per wave (sparse or unsorted test sites)
//       clr_scan_grouped_kernel, clr_scan_kernel  round 2's single-kernel forms: variants 12 / 2 (cross-checks in the
//                                     tests), and the path for tables of 4 GiB and more
//       locate_kernel / finalize_kernel: test-site positions, per-slice argmax merge + nSites
//       surface_kernel             <- the full T[A,x,alpha] surface of one site (v1:449-450 wish)
//
// The shipped library reads ONE environment variable, BMX_TRACE (stage messages on stderr, no effect on results).
// Tuning knobs for A/B runs (BMX_LDS_PAD, BMX_DENSE_GAP, BMX_SOLO_GAP, BMX_FORCE_J, BMX_FAR_EPS, BMX_MOM_SLOTS, BMX_SPB,
// BMX_ROWMAX_GLOBAL) exist only in the diagnostic builds (-DBMX_DIAG: `make diag|prof|count`); the scan variant is chosen with
// bmx_ctx_set_variant().
//
// K2 formulation.  For a test site t and linkage value A the reference sums, over the sites
// i of the window with alpha_i = exp(-A*|g_i - t|) >= 1e-8 and g_i != t (v1:454-457),
//     log(alpha_i*S_i + (1-alpha_i)*g_i) - log(g_i)              (v1:494-499)
// which equals log(1 + alpha_i*R[x,a][row_i]) with R = S*prop/g - 1 tabulated per (k,n) row.
// The scan keeps, per lane, the running PRODUCT  prod_i (1 + alpha_i*R)  (one FMA and one
// MUL per site and grid pair, no transcendental), pulls the binary exponent out of the product
// every few sites so it cannot over/underflow, and takes a single log per (t, A, pair).
// Lanes run over the (x, alpha_beta) pairs, so nothing is reduced across lanes until the
// final argmax.  Far from the test sites (alpha*|R| small: three quarters of a window) not even that: the sites'
// alpha^k are added to per-row moments and the product picks up exp(sum_k +-w_k F^k sum_rows R^k M_k)
// -- the log1p series with economised coefficients (order 12 on |x| <= 0.15 in the prepared kernels, order 8 on
// |x| <= 0.05 in the solo and round-2 kernels).  Since round 3 everything that does not depend on the pair -- positions,
// exp(-A d), the near / far decision, the moments, where each window ends -- is computed ONCE per group of test sites by a
// lanes-over-sites kernel (prep_kernel) and streamed through HBM to the pair-parallel waves (see "K2, prepared" below).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>
#include <errno.h>
#include <rccl/rccl.h>     // types only: the library is opened with dlopen when a gather is first asked for

#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/bmxscan.h"
#include "bmx_math.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(BMX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

// BMX_TRACE=1 in the environment prints one line per stage to stderr (diagnostics only)
// tuning knobs: environment variables in diagnostic builds, always absent in the shipped library
const char *diag_env(const char *name) {
#ifdef BMX_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

bool trace_on() {
    static int on = -1;
    if (on < 0) on = getenv("BMX_TRACE") ? 1 : 0;
    return on == 1;
}
#define TRACE(...)                                   \
    do {                                             \
        if (trace_on()) {                            \
            fprintf(stderr, "[bmx] " __VA_ARGS__);   \
            fprintf(stderr, "\n");                   \
            fflush(stderr);                          \
        }                                            \
    } while (0)

constexpr int WAVE = 64;
constexpr int SCAN_THREADS = 256;             // 4 waves per workgroup (default)
constexpr int SCAN_THREADS_MAX = 512;         // 8 waves when one R slice fills most of a CU's LDS
constexpr int SCAN_THREADS_J8 = 768;          // 12 waves = 3 per SIMD: the prepared kernel's 8-test-site form (168 registers)
constexpr int SITE_THREADS = 1024;            // per-site kernel: 16 waves per workgroup, two workgroups per CU = 8 waves per SIMD
                                              // (62 VGPRs; the kernel is latency-bound per pass and lives on occupancy)
constexpr int LDS_LIMIT_BYTES = 160 * 1024;   // gfx950: 160 KiB per CU
constexpr int MOM_SLOTS = 254;                // grouped kernel: rows whose far-field sites can be summed as moments
constexpr int MOM_SLOTS_LDS = 64;             // ... of which this many are used while the R slice occupies the LDS
#ifndef BMX_PAIR_PREFETCH
#define BMX_PAIR_PREFETCH (!USE_LDS)      // pair blocks one block ahead: pays for R from global memory only (LDS: 64.22 vs 64.03 ms)
#endif
#ifndef BMX_PPAIR_PREFETCH
#define BMX_PPAIR_PREFETCH 1              // prepared kernel, pair blocks one block ahead: round 4 (no spills, 13 spare VGPRs): +0.4 % with the table in LDS too
#endif
#ifndef BMX_FOLD_RPRE
#define BMX_FOLD_RPRE (!USE_LDS)          // rows of a fold batch requested with its moments: global memory only (LDS: 64.51 vs 64.03 ms)
#endif
#ifndef BMX_FOLD_EARLY
#define BMX_FOLD_EARLY (!USE_LDS)         // prepared kernel's fold: heads and R of the next two slots requested before this step's powers
#endif
#ifndef BMX_MIDTRI
#define BMX_MIDTRI 1
#endif
#ifndef BMX_XCD_MAP
#define BMX_XCD_MAP 1     // prepared kernel: the slices of a chunk of test sites on one XCD (4.168 -> 4.205 M windows/s, HBM reads / 8)
#endif
#ifndef BMX_SOLO_XCD_MAP
#define BMX_SOLO_XCD_MAP 1     // the slices of a chunk of test sites on one XCD: the chunk's streams come out of HBM once, not eight times
#endif
#ifndef BMX_PRIV01
#define BMX_PRIV01 1
#endif
#ifndef BMX_FAR_ORDER
#define BMX_FAR_ORDER 8
#endif
constexpr int FAR_ORDER = BMX_FAR_ORDER;      // ... to this power of alpha*R (log1p series): 4, 6 or 8
#ifndef BMX_MOM_COPIES
#define BMX_MOM_COPIES 8
#endif
constexpr int MOM_COPIES = BMX_MOM_COPIES;                 // copies of the most frequent row's moments (lane % 8): fewer LDS conflicts
#ifndef BMX_QSB
#define BMX_QSB 8
#endif
#ifndef BMX_FOLD_BATCH
#define BMX_FOLD_BATCH 2
#endif
constexpr int SITE_SCR = 72;                  // per-site kernel: two lists of at most 64 sites, each padded to a multiple of four
constexpr int SCR_CAP = 80;                   // grouped kernel: entries of a wave's scratch list: up to 64 pending + 16 neutral ones (padding of the
                                              // last block of 4 or 8 and what the block loops read one block ahead -- R from global memory must
                                              // never see a stale row reference)
constexpr int MID_CAP = 32;                   // grouped kernel: sites between the test sites of a group staged in LDS (12 B each)
constexpr int FAR_CAP = 8192;                 // ... at most this many sites per zone (exponent budget: 8192 * 0.05 * 1.49 bits < 1000)
constexpr double LN2 = 0.693147180559945309417232121458;
// Far field: log1p(x) = sum_k (-1)^(k+1) w_k x^k.  To 8th order the w_k are the Taylor coefficients 1/k economised on
// [-0.05, 0.05] (Taylor polynomial of degree 24 re-expanded in Chebyshev polynomials of x/0.05, cut after T_8, constant
// term -1.9e-17 dropped): max error 8.9e-16 on the whole interval, against 2.3e-13 for the plain Taylor cut -- Taylor's
// accuracy at |x| <= 0.0296 with the far field starting 0.5 units of A*d nearer to the test sites.  Lanes that drop
// high orders (x^k/k < 2e-15) see w_k - 1/k only as |w_1 - 1| |x| + |w_2 - 1/2| x^2 + ... < 1e-17.  Lower orders: Taylor.
#if BMX_FAR_ORDER >= 8
__device__ constexpr double FAR_W[8] = {0.9999999999998467, 0.4999999999996164, 0.3333333341509627, 0.25000000122701976,
                                        0.19999882322703955, 0.16666529319200146, 0.1434841013671569, 0.12562715480255862};
#else
__device__ constexpr double FAR_W[8] = {1.0, 0.5, 0.3333333333333333, 0.25, 0.2, 0.16666666666666666, 0.14285714285714285, 0.125};
#endif

// ----------------------------------------------------------------------------- K1
struct LutParams {
    int stat, min_count, n_sizes, rows, nx, nab, NP;
    const int32_t *sizes;
    const int32_t *row_off;
    const double *g;
    const double *prop;
    const double *x;
    const double *abeta;
    double *psel;  // [nx][nab][rows]
    double *R;     // [nx][nab][rows]
    double *Rt;    // [rows][NP]   kernel layout: pair index fastest, zero padded
    bmx::LogPatch patch;   // host-libm conformance exceptions for lgam's log (bmx_math.h)
};

// One thread per (grid pair, LUT row).  The thread needs the beta-binomial pmf at the site's own
// count (folded / B_1 forms, v1:375-396) and at the counts excluded from the support
// (v1:399-433), each for b(x) and b(1-x).  All of them go through ONE inlined pmf call site
// inside nested loops: the pmf body is large, and gfx950 device-function calls from a partially
// active wave proved unusable here (hang), so nothing in this kernel is an out-of-line call.
__global__ __launch_bounds__(128) void bb_lut_kernel(LutParams P) {   // 128 threads per workgroup: room for 512 registers, no scratch
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int npairs = P.nx * P.nab;
    if (gid >= (int64_t)npairs * P.rows) return;
    int p = (int)(gid / P.rows), r = (int)(gid % P.rows);
    int ix = p / P.nab, ia = p % P.nab;
    int j = 0;
    while (j + 1 < P.n_sizes && r >= P.row_off[j + 1]) j++;
    const int n = P.sizes[j], k = r - P.row_off[j];
    const double x = P.x[ix], a = P.abeta[ia];
    const double xm = 1. - x;
    const double b1 = a / x - a, b2 = a / xm - a;                 // v1:316
    const int m = P.min_count, stat = P.stat;
    const bool maf = (stat == BMX_STAT_B2MAF || stat == BMX_STAT_B0MAF);
    int nex = m;                                                  // excluded counts, v1:399-433
    if (stat == BMX_STAT_B2MAF) nex += (m - 1 > 0 ? m - 1 : 0);
    if (stat == BMX_STAT_B0) nex += 1;
    if (stat == BMX_STAT_B0MAF) nex += m;
    // numpy's pairwise summation of the excluded probabilities, streamed (np.sum, v1:402)
    const int nblk = nex < 8 ? 0 : nex - (nex % 8);
    double r8[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    double res = 0.;
    double raw = 0.;
    for (int it = 0; it <= nex; ++it) {
        const bool site = (it == nex);
        int c;
        if (site) c = (stat == BMX_STAT_B1) ? n : k;              // B_1 uses pmf(n) (v1:382)
        else if (it < m) c = it;
        else if (stat == BMX_STAT_B0) c = n;
        else c = n - m + 1 + (it - m);
        const int nfold = (site && maf) ? 2 : 1;                  // pmf(k) + pmf(n-k)  (v1:389)
        double v[2];
        for (int side = 0; side < 2; ++side) {
            const double b = side ? b2 : b1;
            double pr = 0.;
            for (int f = 0; f < nfold; ++f) {
                const double q = bmx::betabinom_pmf(f ? n - c : c, n, a, b, P.patch);
                pr = f ? pr + q : q;
            }
            if (site && maf && (n % 2 == 0) && c == n / 2) pr = pr / 2;      // v1:391-392
            if (site && stat == BMX_STAT_B1) pr = (k == 0) ? pr : (1. - pr - pr);
            v[side] = pr;
        }
        const double e = 0.5 * (v[0] + v[1]);
        if (site) {
            raw = e;
        } else if (nex < 8) {
            res += e;
        } else if (it < nblk) {
            if (it < 8) r8[it] = e; else r8[it & 7] += e;
            if (it == nblk - 1)
                res = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
        } else {
            res += e;
        }
    }
    const double base = 1. - res;
    const double psel = raw / base;
    const size_t o = ((size_t)ix * P.nab + ia) * P.rows + r;
    const double R = psel * P.prop[j] / P.g[r] - 1.0;
    P.psel[o] = psel;
    P.R[o] = R;
    P.Rt[(size_t)r * P.NP + p] = R;
}

// ----------------------------------------------------------------------------- locate
__global__ void locate_kernel(const double *genpos, int64_t N, const double *test_gen, int64_t M,
                              int64_t *center, int64_t *center_hi) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    double v = test_gen[t];
    int64_t a = 0, b = N;
    while (a < b) {
        int64_t m = (a + b) >> 1;
        if (genpos[m] < v) a = m + 1; else b = m;
    }
    center[t] = a;  // first index with genpos >= test position
    b = N;
    while (a < b) {
        int64_t m = (a + b) >> 1;
        if (genpos[m] <= v) a = m + 1; else b = m;
    }
    center_hi[t] = a;  // first index with genpos > test position (sites in between are ties)
}

// window bounds of the all-sites mode (set_tests without bounds): every window is [0, N - 1]
__global__ void all_sites_kernel(int64_t *lo, int64_t *hi, int64_t M, int64_t N) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    lo[k] = 0;
    hi[k] = N - 1;
}

// index gaps between neighbouring test sites at a strided sample of all of them (set_tests: test-site density)
__global__ void gap_kernel(const int64_t *center, int64_t stride, int64_t ns, int64_t *gap) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    gap[k] = center[k * stride + 1] - center[k * stride];
}

// LUT row of a site: 2 bytes per site normally (<= 65535 rows), 4 when the table is larger
// (hundreds of distinct sample sizes).
struct RowArray {
    const uint16_t *r16;
    const uint32_t *r32;
    __device__ __forceinline__ int operator[](int64_t i) const { return r32 ? (int)r32[i] : (int)r16[i]; }
};

// ----------------------------------------------------------------------------- K2
struct ScanParams {
    const double *genpos;
    RowArray row;
    int64_t N;
    const double *Rt;  // [rows][NP]
    int rows, NP, npairs, nslices;
    const double *A;
    int nA;
    const double *test_gen;
    const int64_t *win_lo;
    const int64_t *win_hi;
    const int64_t *center;     // first index with genpos >= test position
    const int64_t *center_hi;  // first index with genpos >  test position
    int64_t M;
    double zcut;       // alpha >= 1e-8  <=>  A*d <= zcut  (v1:455)
    int renorm_every;  // per-site kernel: sites between exponent extractions
    int span_hi;       // grouped kernel: bits by which one factor 1+alpha*R can exceed 1 (>= 1)
    double rmax;       // max(0, largest finite R of the table)
    double far_eps;    // grouped kernel, FARSUM: sites with E * rowmax[row] <= far_eps go through power sums
    // [nslices][rows]: max |R| of the row over the slice's 64 pairs, rounded up (NaN for absent rows), with the
    // row's far-field moment slot in the low mantissa byte: the rank of the row among the data's rows by
    // frequency (255: not ranked); row_of_slot is the inverse, kmom[iA] says how many slots pay at A
    const double *rowmax;
    const uint8_t *kmom;
    int row_of_slot[MOM_SLOTS];
    int row0;          // per-site kernel: the most frequent row of the data (-1: none)
    int mom_slots;     // slots per wave allocated in LDS (<= MOM_SLOTS)
    int wide_tab;      // the global R table has 2^32 bytes or more
    unsigned long long *prof;   // -DBMX_PROFILE builds: cycles per kernel section, summed over waves (else unused)
    float far_bits;    // far_eps * log2(e): exponent-budget bits per far site
    int sites_per_block;
    double *part_T;    // [nslices][M]
    int32_t *part_lin;
    int32_t *part_ns;
};

__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// exp(-z) for 0 <= z < ~700, without OCML's special-case handling: n = rint(-z*log2 e),
// r = -z - n*ln2 (two-term), degree-13 Taylor polynomial (|r| <= 0.347: truncation 4e-18), ldexp.
// ~19 instructions instead of ~34; about 1 ulp, which is all alpha needs (alpha only feeds
// log-likelihood terms; the window PREDICATE never goes through this function).
// p*r + c with the constant c in a scalar register pair.  gfx950 VALU instructions take no 64-bit literals, and left to
// itself the compiler keeps every polynomial coefficient of exp_neg (called from five places) in a VGPR pair for the whole
// kernel -- 20 VGPRs of a 256-VGPR budget -- and spends a v_mov_b64 per Horner step because it selects the two-address
// v_fmac, which overwrites its addend.  The "s" constraint makes the coefficient two s_mov_b32 (scalar issue slots,
// rematerialised wherever needed) and the step one three-address v_fma_f64.
__device__ __forceinline__ double fma_sc(double p, double r, double c) {
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
    return o;
}

__device__ __forceinline__ double exp_neg(double z) {
    const double t = -z;
    const double n = rint(t * 1.4426950408889634);
    double r = fma(-n, 0.6931471805599453, t);
    r = fma(-n, 2.3190468138462996e-17, r);
    double p = fma_sc(1.6059043836821613e-10, r, 2.08767569878681e-09);   // 1/13!, 1/12!
    p = fma_sc(p, r, 2.505210838544172e-08);         // 1/11!
    p = fma_sc(p, r, 2.755731922398589e-07);         // 1/10!
    p = fma_sc(p, r, 2.7557319223985893e-06);        // 1/9!
    p = fma_sc(p, r, 2.48015873015873e-05);          // 1/8!
    p = fma_sc(p, r, 0.0001984126984126984);         // 1/7!
    p = fma_sc(p, r, 0.001388888888888889);          // 1/6!
    p = fma_sc(p, r, 0.008333333333333333);          // 1/5!
    p = fma_sc(p, r, 0.041666666666666664);          // 1/4!
    p = fma_sc(p, r, 0.16666666666666666);           // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// Two independent exp_neg chains, written out interleaved: a chain of 19 dependent FP64 operations runs at the FP64
// latency (~10 cycles per step for a wave on its own), two interleaved chains at nearly the issue rate.  Used where exps
// come in batches (the far-field flush: one per test site).
__device__ __forceinline__ void exp_neg2(double z0, double z1, double &o0, double &o1) {
    const double t0 = -z0, t1 = -z1;
    const double n0 = rint(t0 * 1.4426950408889634), n1 = rint(t1 * 1.4426950408889634);
    double r0 = fma(-n0, 0.6931471805599453, t0), r1 = fma(-n1, 0.6931471805599453, t1);
    r0 = fma(-n0, 2.3190468138462996e-17, r0);
    r1 = fma(-n1, 2.3190468138462996e-17, r1);
    double p0 = fma_sc(1.6059043836821613e-10, r0, 2.08767569878681e-09), p1 = fma_sc(1.6059043836821613e-10, r1, 2.08767569878681e-09);
#define BMX_STEP2(c) p0 = fma_sc(p0, r0, c); p1 = fma_sc(p1, r1, c);
    BMX_STEP2(2.505210838544172e-08)
    BMX_STEP2(2.755731922398589e-07)
    BMX_STEP2(2.7557319223985893e-06)
    BMX_STEP2(2.48015873015873e-05)
    BMX_STEP2(0.0001984126984126984)
    BMX_STEP2(0.001388888888888889)
    BMX_STEP2(0.008333333333333333)
    BMX_STEP2(0.041666666666666664)
    BMX_STEP2(0.16666666666666666)
#undef BMX_STEP2
    p0 = fma(p0, r0, 0.5); p1 = fma(p1, r1, 0.5);
    p0 = fma(p0, r0, 1.0); p1 = fma(p1, r1, 1.0);
    p0 = fma(p0, r0, 1.0); p1 = fma(p1, r1, 1.0);
    o0 = ldexp(p0, (int)n0);
    o1 = ldexp(p1, (int)n1);
}

// Pull the binary exponent out of a non-negative product; a zero product is sticky (-> -inf).
__device__ __forceinline__ void renorm(double &acc, int &E) {
    unsigned hi = (unsigned)__double2hiint(acc);
    int e = (int)((hi >> 20) & 0x7ffu);
    E = (e == 0) ? -(1 << 28) : E + (e - 1023);
    hi = (hi & 0x800fffffu) | 0x3ff00000u;
    acc = __hiloint2double((int)hi, __double2loint(acc));
}

// The same with the hardware's frexp pair (v_frexp_exp_i32_f64 / v_frexp_mant_f64): three VALU instructions instead of six.  The
// mantissa lands in [1/2, 1) -- a convention of its own: whoever compares (exponent, mantissa) pairs must take both from this
// function -- and a zero product keeps mantissa 0 with its exponent unchanged, so the comparison has to ask for a positive mantissa.
#ifndef BMX_FREXP
#define BMX_FREXP 0     // prepared kernels: measured no gain (config-3 block 4.45 vs 4.43 M windows/s), so they keep the integer form
#endif
__device__ __forceinline__ void renorm_fx(double &acc, int &E) {
    E += __builtin_amdgcn_frexp_exp(acc);
    acc = __builtin_amdgcn_frexp_mant(acc);
}

// One (E or alpha, row * 64) entry of a wave's scratch list in LDS: written lanes-over-sites, read back with a uniform
// address (LDS broadcast) by all 64 grid-pair lanes.
struct alignas(16) ScratchEnt {
    double e;
    int ro;   // row * 64
    int pad;
};

// K2, one test site per wave: sparse test sets (the reference's -s with a large step), unsorted test positions, tables whose
// dynamic range is too wide for block-wise exponent extraction; also the independent cross-check of the grouped kernel in the
// tests.  Per A and side the wave walks the window 64 sites at a time: alpha = exp(-A d) lanes-over-sites, (alpha, row) through
// the wave's LDS scratch, then every lane (grid pair) multiplies 1 + alpha R four sites per step: 2.75 vector instructions
// per site (round 1: ~6 with v_readlane broadcasts and OCML's exp).  Best grid point tracked per lane as (exponent, mantissa),
// like the grouped kernel; finalize_kernel takes the one log and counts nSites.
template <bool USE_LDS>
__global__ __launch_bounds__(SITE_THREADS) void clr_scan_kernel(ScanParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds_R[];  // [rows][64] when USE_LDS, then 64 entries per wave
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x / WAVE;
    const int slice = blockIdx.x % P.nslices;   // blocks b, b+8 share an XCD: one R slice per L2
    const int64_t chunk = blockIdx.x / P.nslices;
    const int p = slice * WAVE + lane;
    const int N = (int)P.N;

    if (USE_LDS) {
        const int total = P.rows * WAVE;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x)
            lds_R[idx] = P.Rt[(size_t)(idx >> 6) * P.NP + slice * WAVE + (idx & 63)];
        __syncthreads();
    }
    // rowoff: index of the row's first value -- row * 64 in the LDS slice, row * NP in the global table (see the grouped kernel)
    const char *Rb = reinterpret_cast<const char *>(P.Rt + slice * WAVE);
    const unsigned lane8 = (unsigned)lane * 8u;
    const int rowmul = USE_LDS ? WAVE : P.wide_tab ? 1 : P.NP;      // tables of 4 GiB and more: rowoff = row, 64-bit arithmetic per load
    auto loadR = [&](int rowoff) -> double {
        if (USE_LDS) return lds_R[rowoff + lane];
        if (P.wide_tab) return *reinterpret_cast<const double *>(Rb + ((size_t)rowoff * P.NP * 8u + lane8));
        return *reinterpret_cast<const double *>(Rb + ((unsigned)rowoff * 8u + lane8));
    };
    ScratchEnt *scr = reinterpret_cast<ScratchEnt *>(lds_R + (USE_LDS ? P.rows * WAVE : 0)) + wave * SITE_SCR;
    // The most frequent row of the data (substitutions: ~70 % of the sites) stays in a register pair: its sites go through a list
    // of their own and need no R read at all -- this kernel is bound by LDS bandwidth (512 B of R per site and slice), not by
    // arithmetic, and by L2 bandwidth when the table does not fit the LDS.
    const int row0 = P.row0;
    const double R0 = row0 >= 0 ? loadR(row0 * rowmul) : 0.0;
    auto rank = [&](unsigned long long m) {
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    };

    const int64_t t_begin = chunk * P.sites_per_block;
    const int64_t t_end = min(t_begin + (int64_t)P.sites_per_block, P.M);
    for (int64_t t = t_begin + wave; t < t_end; t += nw) {
        const double tg = P.test_gen[t];
        const int lo = (int)max(P.win_lo[t], (int64_t)0);
        const int hi = (int)min(P.win_hi[t], (int64_t)N - 1);
        const int c = (int)P.center[t];
        double bestM = 1.0;
        int bestEc = 131072, bestA = -1;            // clamped exponent + 2^17 of the best product so far, and its A index (-1: none yet);
                                                    // not packed into one word as in the grouped kernel: this kernel takes any nA

        for (int iA = 0; iA < P.nA; ++iA) {
            const double Aval = P.A[iA];
            double acc = 1.0;
            int E = 0, since = 0;
            for (int dir = 0; dir < 2; ++dir) {
                // dir 0: indices c, c+1, ... up to hi;  dir 1: c-1, c-2, ... down to lo
                int base = dir == 0 ? max(c, lo) : min(c - 1, hi);
                int i = dir == 0 ? base + lane : base - lane;
                double g_nx = P.genpos[min(max(i, 0), N - 1)];          // the next pass's sites are requested one pass ahead
                int r_nx = (int)P.row[min(max(i, 0), N - 1)];
                while (true) {
                    const bool valid = (i >= lo) && (i <= hi);
                    const double g = g_nx;
                    const int rraw = r_nx;
                    const int inx = dir == 0 ? i + WAVE : i - WAVE;
                    g_nx = P.genpos[min(max(inx, 0), N - 1)];
                    r_nx = (int)P.row[min(max(inx, 0), N - 1)];
                    const double z = Aval * fabs(g - tg);
                    const bool in = valid && (z <= P.zcut) && (g != tg);
                    const bool beyond = valid && (z > P.zcut);
                    const unsigned long long m_in = __ballot(in);
                    if (m_in != 0ull) {
                        // two lists in the scratch: the sites of row0 at [0, n0), the others from n0r = n0 rounded up to 4 on,
                        // each padded to a multiple of four with neutral entries (alpha = 0; the row of a site of the window, so
                        // that 0*R is 0 and never 0*NaN: rows absent from the helper file hold NaN)
                        const bool is0 = in && rraw == row0;
                        const unsigned long long m0 = __ballot(is0), m1 = m_in & ~m0;
                        const int n0 = __popcll(m0), n1 = __popcll(m1), n0r = (n0 + 3) & ~3;
                        const int pad_ro = __builtin_amdgcn_readlane(rraw, __ffsll((long long)m_in) - 1) * rowmul;
                        if (in) scr[is0 ? rank(m0) : n0r + rank(m1)] = ScratchEnt{exp_neg(z), rraw * rowmul, 0};
                        if (lane < n0r - n0) scr[n0 + lane] = ScratchEnt{0.0, pad_ro, 0};
                        if (lane < 3) scr[n0r + n1 + lane] = ScratchEnt{0.0, pad_ro, 0};
                        __builtin_amdgcn_wave_barrier();
                        for (int l = 0; l < n0; l += 4) {
                            double f[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) f[u] = fma(scr[l + u].e, R0, 1.0);       // uniform address: LDS broadcast
                            acc *= (f[0] * f[1]) * (f[2] * f[3]);
                            since += 4;
                            if (since + 4 > P.renorm_every) {
                                renorm(acc, E);
                                since = 0;
                            }
                        }
                        // (no prefetch of the next step's entries, and the entry as an 8-byte + a 4-byte read rather than one 16-byte
                        // read: with 8 waves per SIMD the LDS round trips are covered by the other waves; both variants were
                        // measured slower, by 17 % and 4 %)
                        for (int l = n0r; l < n0r + n1; l += 4) {
                            double f[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const ScratchEnt en = scr[l + u];
                                f[u] = fma(en.e, loadR(en.ro), 1.0);
                            }
                            acc *= (f[0] * f[1]) * (f[2] * f[3]);
                            since += 4;
                            if (since + 4 > P.renorm_every) {
                                renorm(acc, E);
                                since = 0;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (__ballot(beyond) != 0ull || __ballot(valid) != ~0ull) break;
                    i = inx;
                }
            }
            renorm(acc, E);
            const int ec = min(max(E, -131071), 131071) + 131072;
            if (((ec > bestEc) || (ec == bestEc && acc > bestM)) && p < P.npairs) {       // strict '>' (v1:501); iA ascending
                bestM = acc;
                bestEc = ec;
                bestA = iA;
            }
        }
        // wave argmax on (exponent, mantissa), ties to the smaller linear index = the reference's first strict
        // maximum in (A, x, alpha_beta) loop order; the logarithm is finalize_kernel's
        int bE = bestEc;
        int bL = bestA < 0 ? 0x7fffffff : bestA * P.npairs + p;
        for (int off = 32; off > 0; off >>= 1) {
            const int oE = __shfl_xor(bE, off);
            const double oM = __shfl_xor(bestM, off);
            const int oL = __shfl_xor(bL, off);
            if (oE > bE || (oE == bE && (oM > bestM || (oM == bestM && oL < bL)))) { bE = oE; bestM = oM; bL = oL; }
        }
        if (lane == 0) {
            const size_t o = (size_t)slice * P.M + t;          // [slice][M]: coalesced for writer and reader
            P.part_T[o] = bestM;
            P.part_lin[o] = bL;
            P.part_ns[o] = bE;
        }
    }
}

// ----------------------------------------------------------------------------- K2, grouped
// J consecutive test sites (a "group") share one pass over the sites around them.  A lane
// still owns one (x, alpha_beta) pair but now carries J running products, one per test site.
//
//  * bulk zones: sites that lie to the right (left) of ALL J test sites and inside ALL J
//    windows.  There  exp(-A(g_i - t_j)) = exp(-A(g_i - t_last)) * exp(-A(t_last - t_j)) = E_i * F_j
//    with F_j constant for the whole zone (scalar registers), so one LDS read of R and one
//    broadcast of (E_i, row_i) feed J FMA+MUL pairs:  v = E_i*R;  acc_j *= fma(F_j, v, 1).
//    A site within the window of the farthest test site is within all of them because FP
//    subtraction and multiplication are monotone, so the reference's predicate is kept exactly.
//  * everything else (sites between the test sites, ties, the ragged window ends) goes through
//    generic passes of 64/J sites x J test sites with the reference's formula per (site, test
//    site): alpha = exp(-(A*|g_i - t_j|)) if i in window_j and A*d <= zcut and g_i != t_j.
//
// The argmax is tracked per lane on (binary exponent, mantissa) of the product -- an exact
// ordering that needs no logarithm; one log per test site is taken at the very end.
//
//  * dense test sets (every SNP a test site): the sites between the test sites are the J test sites themselves and
//    alpha_ij = exp(-A|t_i - t_j|) separates as well (G_i H_j), so that triangle runs in the pair form below.
//
// Inner-loop forms (template parameter MODE_; 3 is what ships, 0..2 stay for A/B runs):
//   0: (E_i, row_i) and alpha_ij reach the lanes by v_readlane, one site per step.
//   1: they are staged in a wave-private LDS scratch and read back with uniform-address
//      (broadcast) ds_read, and bulk sites are taken two at a time:
//          (1 + F v1)(1 + F v2) = 1 + F*(s + F*q),   s = v1 + v2,  q = v1*v2
//      i.e. 2 FMA + 1 MUL per test site per PAIR of sites (s, q shared by all J test sites).
//   2: as 1, and list entries with alpha <= 1/2 go FOUR sites per step (see the bulk loop).
//   3: as 2, and sites with alpha*max|R| <= far_eps are not multiplied at all: their alpha^k go to per-row moments.
// Measured on gfx950 (profiles/): every VALU instruction of this kernel -- FP64 or not -- costs
// ~5-6 SIMD cycles at 2 waves/SIMD and ~10 at one, so the design minimises instruction count:
// 0.81 VALU instructions per 64 evaluations in form 3 at J = 16 (1.98 in form 2), vs ~6 in the per-site kernel.
// -DBMX_PROFILE: s_memtime stamps between the sections of the grouped kernel (diagnostic builds only; the
// stamps serialise outstanding LDS/scalar loads, so the split is approximate).  Sections: 0 sites between
// the test sites, 1 zone set-up, 2 per-pass work, 3 products over the near list, 4 ragged-end masks,
// 5 fold of the moments, 6 flush (exp per test site), 7 generic walks past the zones, 8 best-tracking per A.
#ifdef BMX_PROFILE
#define PROF_MARK(k) do { const long long now_ = clock64(); prof_[k] += now_ - tprev_; tprev_ = now_; } while (0)
#else
#define PROF_MARK(k) do { } while (0)
#endif
// -DBMX_COUNT: event counters (wave-uniform adds; no stamps), summed over waves into P.prof[16..31]
#ifdef BMX_COUNT
#define CNT(k, n) do { cnt_[k] += (unsigned long long)(n); } while (0)
#else
#define CNT(k, n) do { } while (0)
#endif

template <int J, bool USE_LDS, int MODE_>
__global__ __launch_bounds__(SCAN_THREADS_MAX) void clr_scan_grouped_kernel(ScanParams P) {
#ifdef BMX_PROFILE
    long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = clock64();
#endif
#ifdef BMX_COUNT
    unsigned long long cnt_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) double lds_R[];  // [rows][64] when USE_LDS, then scratch
    constexpr int SP = WAVE / J;                                    // sites per generic pass
    constexpr bool QUAD = (MODE_ >= 2);                             // MODE_ 2 = MODE 1 + quads in far passes
    constexpr bool FARSUM = (MODE_ == 3);                           // MODE_ 3 = MODE 2 + far-field moments
    constexpr int MODE = MODE_ ? 1 : 0;
    constexpr bool MIDTRI = (MODE_ >= 2) && BMX_MIDTRI;           // dense test sets: the J x J triangle between the test sites in pair form
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably wave-uniform: group-level scalars live in SGPRs
    const int slice = blockIdx.x % P.nslices;
    const int64_t chunk = blockIdx.x / P.nslices;
    const int p = slice * WAVE + lane;
    const int jl = lane % J, sl = lane / J;
    const int N = (int)P.N;

    if (USE_LDS) {
        const int total = P.rows * WAVE;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x)
            lds_R[idx] = P.Rt[(size_t)(idx >> 6) * P.NP + slice * WAVE + (idx & 63)];
        for (int idx = threadIdx.x; idx < P.rows; idx += blockDim.x)
            lds_R[total + idx] = P.rowmax[(size_t)slice * P.rows + idx];
        __syncthreads();
    }
    // A row is referred to by the index of its first value: row * 64 in the LDS slice, row * NP in the global table (32 bits:
    // the host checks rows * NP * 8 < 2^31), so that either load is one address addition -- scalar base + 32-bit lane offset
    // for the global one, not a 64-bit multiply-add per load
    const char *Rb = reinterpret_cast<const char *>(P.Rt + slice * WAVE);
    const unsigned lane8 = (unsigned)lane * 8u;
    const int rowmul = USE_LDS ? WAVE : P.NP;                      // (tables of 4 GiB and more go to the per-site kernel)
    auto loadR = [&](int rowoff) -> double {
        return USE_LDS ? lds_R[rowoff + lane] : *reinterpret_cast<const double *>(Rb + ((unsigned)rowoff * 8u + lane8));
    };
    // per-row max |R| of this slice (behind the R slice), then the wave-private scratch: 64 x 16 B
    // then the wave-private scratch (64 x 16 B per wave) and the wave-private moments
    const int rows_pad = (P.rows + 1) & ~1;
    const double *rowmax = USE_LDS ? lds_R + P.rows * WAVE : P.rowmax + (size_t)slice * P.rows;
    double *lds_tail = lds_R + (USE_LDS ? P.rows * WAVE + rows_pad : 0);
    ScratchEnt *scr = reinterpret_cast<ScratchEnt *>(lds_tail) + wave * SCR_CAP;
    const int mom_len = (P.mom_slots + MOM_COPIES - 1 + 3) * FAR_ORDER;   // slot 0 in MOM_COPIES copies, slots 1.., 3 spare
    double *mom = lds_tail + (blockDim.x / WAVE) * SCR_CAP * 2 + wave * mom_len;
    // ... and the sites between the group's test sites (position, row * 64): read by the generic passes of all nA iterations
    double *mid_base = lds_tail + (blockDim.x / WAVE) * (SCR_CAP * 2 + mom_len);
    if (MODE_ == 3) {
        for (int idx = lane; idx < mom_len; idx += WAVE) mom[idx] = 0.0;
        __builtin_amdgcn_wave_barrier();
    }
    double *scr_d = reinterpret_cast<double *>(scr);

    const int64_t ngroups = (P.M + J - 1) / J;
    const int64_t gpb = P.sites_per_block / J;
    const int64_t g_end = min((chunk + 1) * gpb, ngroups);
    for (int64_t grp = chunk * gpb + wave; grp < g_end; grp += blockDim.x / WAVE) {
        const int64_t tb = grp * J;
        const int nvalid = (int)min((int64_t)J, P.M - tb);
        // the lane-derived addresses of the group prologue are recomputed per group: left loop-invariant, the compiler
        // keeps them in registers across the whole kernel and, at 256 VGPRs, in scratch
        int jl_g = jl, lane_g = lane, wave_g = wave;
        asm volatile("" : "+v"(jl_g), "+v"(lane_g), "+s"(wave_g));
        double *mid_g = mid_base + wave_g * (MID_CAP + MID_CAP / 2);
        int *mid_ro = reinterpret_cast<int *>(mid_g + MID_CAP);
        const int jj = min(jl_g, nvalid - 1);
        const double tj = P.test_gen[tb + jj];
        int lo_j = (int)max(P.win_lo[tb + jj], (int64_t)0);
        int hi_j = (int)min(P.win_hi[tb + jj], (int64_t)N - 1);
        if (jl >= nvalid) { lo_j = 1; hi_j = 0; }
        const double t0 = readlane_f64(tj, 0), tL = readlane_f64(tj, J - 1);
        const int c0 = (int)P.center[tb], cU = (int)P.center_hi[tb + nvalid - 1];
        int lo_max = 0, hi_min = N - 1;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int a = __builtin_amdgcn_readlane(lo_j, j), b = __builtin_amdgcn_readlane(hi_j, j);
            if (j < nvalid) { lo_max = max(lo_max, a); hi_min = min(hi_min, b); }
        }
        int L_int = min(c0, hi_min + 1), R_int = max(cU, lo_max);
        if (hi_min < lo_max) { L_int = c0; R_int = c0; hi_min = -1; lo_max = N; }   // no bulk zone
        // the sites between the test sites are visited once per A: keep them in LDS (global latency once per group)
        const bool staged = R_int - L_int <= MID_CAP;
        if (staged && lane_g < R_int - L_int) {
            mid_g[lane_g] = P.genpos[L_int + lane_g];
            mid_ro[lane_g] = (int)P.row[L_int + lane_g] * rowmul;
        }
        __builtin_amdgcn_wave_barrier();
        // Dense test sets: the sites between the test sites ARE the J test sites (site k at t_k, positions strictly
        // increasing, every site inside every window's index bounds).  Then alpha_ij = exp(-A|t_i - t_j|) separates into
        // G_i H_j (j < i) and H_i G_j (j > i), G_k = exp(-A(t_k - t_0)), H_k = 1/G_k, and the triangle of (site, test site)
        // pairs runs in the pair form of the bulk loops instead of generic passes (see the A loop).
        bool mid_tri = false;
        if (MIDTRI) {
            const bool same = staged && (R_int - L_int == J) && nvalid == J && lo_max <= L_int && hi_min >= R_int - 1;
            if (same) {
                const double gm = mid_g[jl_g];
                const double tprev = __shfl_up(tj, 1);
                mid_tri = __ballot(!(gm == tj && (jl_g == 0 || tj > tprev))) == 0ull;
            }
        }

        // running best per test site: key = (clamped exponent + 2^17) << 13 | iA  (iA = 8191: none yet)
        double acc[J], bestM[J];
        int E[J], bestK[J];
#pragma unroll
        for (int j = 0; j < J; ++j) { bestM[j] = 1.0; bestK[j] = (131072 << 13) | 8191; }

        for (int iA = 0; iA < P.nA; ++iA) {
            const double A = P.A[iA];
            const int kmom = MODE_ == 3 ? min((int)P.kmom[iA], P.mom_slots) : 0;
            // Exponent budget: every factor 1 + alpha*R lies in [1 - alpha, max(1, 1 + Rmax)], so a
            // block of 8 sites moves log2 of a product by at most 8*span bits, span being the
            // larger of span_hi and -log2(1 - alpha_max).  The products are pulled back to [1,2)
            // before the running total could pass 1000 bits (FP64 holds +-1022).
            int bits = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) { acc[j] = 1.0; E[j] = 0; }
            auto renorm_all = [&]() {
#pragma unroll
                for (int j = 0; j < J; ++j) renorm(acc[j], E[j]);
                bits = 0;
            };
            auto spend = [&](int nbits) {
                if (bits + nbits > 1000) renorm_all();
                bits += nbits;
            };
            const int span_generic = max(P.span_hi, 54);     // alpha < 1  =>  1 - alpha >= 2^-53

            // SP sites x J test sites with per-(site, test site) alpha (lane = site slot * J + test site): acc_j *= 1 + alpha R
            auto apply_pass = [&](double alpha, int rowoff, unsigned long long m_in, double *buf) {
                if (MODE == 1) {
                    buf[lane] = alpha;
                    __builtin_amdgcn_wave_barrier();
                }
                double Rpre[SP];
                if (!USE_LDS) {               // R from global memory: all rows of the pass requested at once (lanes without a site: row 0)
#pragma unroll
                    for (int s = 0; s < SP; ++s) Rpre[s] = loadR(__builtin_amdgcn_readlane(rowoff, s * J));
                }
#pragma unroll
                for (int s = 0; s < SP; ++s) {
                    if (((m_in >> (s * J)) & ((1ull << J) - 1ull)) == 0ull) continue;
                    const double R = USE_LDS ? loadR(__builtin_amdgcn_readlane(rowoff, s * J)) : Rpre[s];
                    if (MODE == 1) {
                        const double2 *a2 = reinterpret_cast<const double2 *>(buf + s * J);
#pragma unroll
                        for (int j = 0; j < J; j += 2) {
                            const double2 a = a2[j >> 1];              // uniform address: LDS broadcast
                            acc[j] *= fma(a.x, R, 1.0);
                            acc[j + 1] *= fma(a.y, R, 1.0);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < J; ++j) acc[j] *= fma(readlane_f64(alpha, s * J + j), R, 1.0);
                    }
                }
                if (MODE == 1) __builtin_amdgcn_wave_barrier();
            };

            // one generic pass over SP sites starting at b, stepping dir; returns "all finished"
            // (site i of the lane, its position and row; lanes with inr = false carry no site)
            auto generic_core = [&](int i, bool inr, double g, int rowoff, int dir) -> bool {
                const bool inwin = inr && i >= lo_j && i <= hi_j;
                const double z = A * fabs(g - tj);
                const bool in = inwin && (z <= P.zcut) && (g != tj);
                // a lane is finished once nothing further along the walk can be in its window; an empty
                // window (lo > hi: the padding test sites of a partial last group) is finished at once --
                // without this the left walk of a partial group runs to index 0
                const bool fin = lo_j > hi_j ||
                                 (dir > 0 ? (!inr || i > hi_j || (i >= lo_j && g > tj && z > P.zcut))
                                          : (!inr || i < lo_j || (i <= hi_j && g < tj && z > P.zcut)));
                const unsigned long long m_in = __ballot(in);
                CNT(7, 1);
                if (m_in != 0ull) {
                    CNT(6, 1);
                    const double alpha = in ? exp_neg(z) : 0.0;
                    // factors of this pass lie in [1 - a_max, 1 + a_max*Rmax]; away from the test sites
                    // (every alpha <= 1/2) that is [1/2, 2^span_hi], not the 54-bit worst case
                    spend(SP * (__ballot(in && z < 0.6931471805599453) == 0ull ? max(P.span_hi, 2) : span_generic));
                    apply_pass(alpha, rowoff, m_in, scr_d);
                }
                return __ballot(fin) == ~0ull;
            };
            auto generic_pass = [&](int b, int dir, int lim, bool from_lds) -> bool {
                const int i = b + dir * sl;
                const bool inr = dir > 0 ? (i < lim) : (i > lim);
                // both loads up front: the row is needed only when some site is in a window, but waiting for
                // the ballot would put two global round trips in series
                // (two branches, not a select between an LDS and a global address: that would be a flat load)
                double g = 0.0;
                int rowoff = 0;
                if (from_lds) {
                    if (inr) { g = mid_g[i - L_int]; rowoff = mid_ro[i - L_int]; }
                } else {
                    if (inr) { g = P.genpos[i]; rowoff = (int)P.row[i] * rowmul; }
                }
                return generic_core(i, inr, g, rowoff, dir);
            };

            // bulk zone: sites i = base, base+dir, ... ; ok(i) is a prefix property along the walk
            constexpr int ZONE_DONE = -0x7fffffff;
            auto bulk_zone = [&](int base, int dir, double tnear, double tfar) -> int {
                double F[J];
                {
                    const double fv = exp_neg(A * fabs(tnear - tj));
#pragma unroll
                    for (int j = 0; j < J; ++j) F[j] = readlane_f64(fv, j);
                }
                // the site data of the NEXT pass is requested before this pass's arithmetic starts,
                // so its L2 latency hides under ~4000 cycles of FP64 work
                int i = base + dir * lane;
                double g_nx = P.genpos[min(max(i, 0), N - 1)];
                int r_nx = (int)P.row[min(max(i, 0), N - 1)];
                int nfar_tot = 0;                                  // far-field sites of this zone (FARSUM)
                double m1p = 0.0, m2p = 0.0;                       // lane-private first and second moments of slot 0
                // MODE 1: the near list is carried from pass to pass -- entries pending in scr[0 .. fill), worked off in blocks
                // of four only when the next pass would not fit behind them or the zone has ended, so the padding of the last
                // block and the set-up of the block loops are paid once per ~60 near sites, not once per pass (~15)
                int fill = 0, pend_np = 0, pend_hi = 0, pend_lo = 0, pad_ro = 0;
                bool ended = false;
                auto rank = [&](unsigned long long m) {
                    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                };
                PROF_MARK(1);
                while (true) {
                    int cnt = 0, cnt_blk = 0, inx = i;
                    bool nearl = false;
                    double Ev = 0.0;
                    int rraw = 0;
                    if (!ended) {
                    const bool ok = dir > 0 ? (i <= hi_min) : (i >= lo_max);
                    const double g = g_nx;
                    rraw = r_nx;
                    inx = i + dir * WAVE;
                    g_nx = P.genpos[min(max(inx, 0), N - 1)];
                    r_nx = (int)P.row[min(max(inx, 0), N - 1)];
                    const bool bulk = ok && (A * fabs(g - tfar) <= P.zcut);
                    const unsigned long long mb = __ballot(bulk);
                    cnt = __popcll(mb);
                    CNT(13, 1);
                    if (cnt) {
                        CNT(0, 1);
                        CNT(14, cnt);
                        Ev = bulk ? exp_neg(A * fabs(g - tnear)) : 0.0;
                        cnt_blk = cnt;                                  // sites left to the block loops
                        if (MODE == 1) {
                            // FARSUM.  A site is FAR when alpha*|R| <= far_eps for every pair of the slice and every
                            // test site (alpha = E F <= E) and its row is one of the kmom most frequent rows of the
                            // data (substitutions, singletons, ...: nearly all sites).  For those,
                            //   sum_i log1p(F v_i) = sum_k (-1)^(k+1) F^k/k * sum_rows R[row]^k M_k[row],  M_k = sum_i E_i^k,
                            // so the pass only adds E, E^2, .. E^8 to the row's moments in LDS -- lane-parallel over
                            // the sites, nothing per pair -- and the zone's end folds the moments into the product.
                            bool moml = false;
                            if (FARSUM && kmom) {
                                const double ri = rowmax[rraw];                 // row's max |R|, its moment slot in the low byte
                                const int slot = __double2loint(ri) & 0xff;
                                const double xr = Ev * ri;
                                moml = bulk && slot < kmom && xr <= P.far_eps && nfar_tot < FAR_CAP;
                                const unsigned long long mm = __ballot(moml);
                                const int nfar = __popcll(mm);
                                PROF_MARK(10);
                                if (nfar) {
                                    if (moml) {
                                        // most sites carry the most frequent row (substitutions): its moments are kept in
                                        // MOM_COPIES copies so that one ds_add_f64 does not serialise ~45 lanes on one address
                                        double *mr = mom + (slot ? slot + MOM_COPIES - 1 : (lane & (MOM_COPIES - 1))) * FAR_ORDER;   // slot 0 = copies 0..C-1
                                        // ds_add_f64 costs ~0.65 FMA slots per active lane (scripts/ubench_lds_atomic), so a
                                        // lane only adds the powers it needs: x^k/k < 2e-15 is dropped (x = E * rowmax >= |F v|)
                                        const double E2 = Ev * Ev;
                                        if (BMX_PRIV01 && slot == 0) {
                                            // the most frequent row (70 % of the sites): first and second moments -- the two every far
                                            // site adds -- in two registers of the lane, joined to the LDS copies at the fold
                                            m1p += Ev;
                                            m2p += E2;
                                        } else {
                                            atomicAdd(mr, Ev);
                                            atomicAdd(mr + 1, E2);
                                        }
                                        if (xr > 1.8e-5) {
                                            const double E3 = E2 * Ev;
                                            atomicAdd(mr + 2, E3);
                                            if (xr > 3.0e-4) {
                                                const double E4 = E2 * E2;
                                                atomicAdd(mr + 3, E4);
                                                if (FAR_ORDER >= 6 && xr > 1.6e-3) {
                                                    atomicAdd(mr + 4, E4 * Ev);
                                                    if (xr > 4.8e-3) {
                                                        atomicAdd(mr + 5, E4 * E2);
                                                        if (FAR_ORDER >= 8 && xr > 0.0105) {
                                                            atomicAdd(mr + 6, E4 * E3);
                                                            if (xr > 0.0189) atomicAdd(mr + 7, E4 * E4);
                                                        }
                                                    }
                                                }
                                            }
                                        }
                                    }
                                    PROF_MARK(11);
                                    nfar_tot += nfar;
                                    cnt_blk = cnt - nfar;
                                    CNT(5, nfar);
                                }
                            }
                            nearl = bulk && !moml;
                            PROF_MARK(2);
                        } else {
                            int rowoff = rraw * rowmul;
                            rowoff = bulk ? rowoff : __builtin_amdgcn_readlane(rowoff, 0);
                            const double e0 = readlane_f64(Ev, 0);
                            const double om = 1.0 - e0, op = fma(e0, P.rmax, 1.0);
                            const int lowbits = 1024 - ((__double2hiint(om) >> 20) & 0x7ff);
                            const int hibits = ((__double2hiint(op) >> 20) & 0x7ff) - 1022;
                            const int span8 = 8 * min(max(hibits, lowbits), 125);
                            for (int l0 = 0; l0 < cnt; l0 += 8) {
                                spend(span8);
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int l = l0 + u;           // lanes >= cnt carry Ev = 0: factor 1
                                    const double v = readlane_f64(Ev, l) * loadR(__builtin_amdgcn_readlane(rowoff, l));
#pragma unroll
                                    for (int j = 0; j < J; ++j) acc[j] *= fma(F[j], v, 1.0);
                                }
                            }
                        }
                    }
                    }
                    if (MODE == 1) {
                        if (fill > 0 && (ended || fill + cnt_blk > WAVE - 4)) {
                            CNT(1, 1);
                            CNT(4, fill);
                            // neutral entries (E = 0, the row of a site of the zone: 0*R is 0, not 0*NaN from a row absent in the
                            // helper file) behind the list: the padding of the last block and what the loops read one block ahead
                            if (lane < 16) scr[fill + lane] = ScratchEnt{0.0, pad_ro, 0};
                            __builtin_amdgcn_wave_barrier();
                            // every factor of the pending sites lies in [1 - E0, 1 + E0*Rmax], E0 the largest alpha among them
                            const int span8 = 8 * min(max(pend_hi, pend_lo), 125);
                            constexpr int BS = J >= 16 ? 4 : 8;        // sites per unrolled block
                            // Sites with alpha <= 1/2 (factors >= 1/2: the expanded product is well conditioned) are
                            // taken FOUR per step:
                            //   prod_m (1 + F v_m) = 1 + F e1 + F^2 e2 + F^3 e3 + F^4 e4   (Horner in F),
                            // e_k = elementary symmetric polynomials of v_1..v_4 shared by all J test sites:
                            // 4 FMA + 1 MUL per test site per four sites.  The near list is ordered by distance, so the
                            // entries with E > 1/2 are a prefix: they go two per step (below), rounded up to whole blocks.
                            const int npair = !QUAD ? fill : min(fill, (pend_np + BS - 1) & ~(BS - 1));
                            const int span8q = 8 * min(max(pend_hi, 2), 125);
                            PROF_MARK(9);
                            if (QUAD && npair < fill) {
                                // the R rows of the NEXT four sites are requested before this block's arithmetic, so the
                                // two dependent LDS round trips (list entry -> row) of a block hide under the previous one
                                double Rn[4], en_e[4];
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const ScratchEnt en = scr[min(npair + u, SCR_CAP - 1)];
                                    en_e[u] = en.e;
                                    Rn[u] = loadR(en.ro);
                                }
                                for (int l0 = npair; l0 < fill; l0 += 4) {
                                    CNT(2, 1);
                                    spend(span8q / 2);
                                    double v[4];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) v[u] = en_e[u] * Rn[u];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        const ScratchEnt en = scr[min(l0 + 4 + u, SCR_CAP - 1)];
                                        en_e[u] = en.e;
                                        Rn[u] = loadR(en.ro);
                                    }
                                    const double s01 = v[0] + v[1], q01 = v[0] * v[1];
                                    const double s23 = v[2] + v[3], q23 = v[2] * v[3];
                                    const double e1 = s01 + s23;
                                    const double e2 = fma(s01, s23, q01 + q23);
                                    const double e3 = fma(q01, s23, q23 * s01);
                                    const double e4 = q01 * q23;
#pragma unroll
                                    for (int j = 0; j < J; ++j) {
                                        double t = fma(F[j], e4, e3);
                                        t = fma(F[j], t, e2);
                                        t = fma(F[j], t, e1);
                                        acc[j] *= fma(F[j], t, 1.0);
                                        if ((j & (BMX_QSB - 1)) == BMX_QSB - 1) __builtin_amdgcn_sched_barrier(0);   // at most BMX_QSB chains in flight
                                    }
                                }
                            }
                            // (R from global memory: the rows of the NEXT block are requested before this block's arithmetic, as in
                            // the quad loop -- an L2 round trip per block would otherwise be exposed)
                            double Rp[BS], ep[BS];
                            if (BMX_PAIR_PREFETCH && npair > 0) {
#pragma unroll
                                for (int u = 0; u < BS; ++u) {
                                    const ScratchEnt en = scr[u];
                                    ep[u] = en.e;
                                    Rp[u] = loadR(en.ro);
                                }
                            }
                            for (int l0 = 0; l0 < npair; l0 += BS) {
                                CNT(3, 1);
                                spend(span8 * BS / 8);
                                double v[BS];
                                if (BMX_PAIR_PREFETCH) {
#pragma unroll
                                    for (int u = 0; u < BS; ++u) v[u] = ep[u] * Rp[u];
#pragma unroll
                                    for (int u = 0; u < BS; ++u) {
                                        const ScratchEnt en = scr[min(l0 + BS + u, SCR_CAP - 1)];
                                        ep[u] = en.e;
                                        Rp[u] = loadR(en.ro);
                                    }
                                } else {
#pragma unroll
                                for (int u = 0; u < BS; ++u) {      // lanes >= cnt carry Ev = 0: factor 1
                                    const ScratchEnt en = scr[l0 + u];     // uniform address: LDS broadcast
                                    v[u] = en.e * loadR(en.ro);
                                }
                                }
#pragma unroll
                                for (int u = 0; u < BS; u += 2) {
                                    const double sv = v[u] + v[u + 1], qv = v[u] * v[u + 1];
#pragma unroll
                                    for (int j = 0; j < J; ++j) acc[j] *= fma(F[j], fma(F[j], qv, sv), 1.0);
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            fill = 0;
                            pend_np = 0;
                        }
                        PROF_MARK(3);
                        if (ended) break;
                        if (cnt_blk > 0) {
                            if (fill == 0) {
                                // lane 0 is the site of the pass nearest to the test sites: the largest alpha of everything pending
                                const double e0 = readlane_f64(Ev, 0);
                                const double om = 1.0 - e0, op = fma(e0, P.rmax, 1.0);
                                pend_lo = 1024 - ((__double2hiint(om) >> 20) & 0x7ff);
                                pend_hi = ((__double2hiint(op) >> 20) & 0x7ff) - 1022;
                                pad_ro = __builtin_amdgcn_readlane(rraw, 0) * rowmul;
                            }
                            const unsigned long long mn = __ballot(nearl);
                            if (nearl) scr[fill + rank(mn)] = ScratchEnt{Ev, rraw * rowmul, 0};
                            if (QUAD) pend_np += __popcll(__ballot(nearl && Ev > 0.5));
                            fill += cnt_blk;
                        }
                    }
                    base += dir * cnt;
                    if (cnt < WAVE) {
                        if (MODE != 1) break;
                        ended = true;                                  // one more turn: works off what is pending
                    }
                    i = inx;
                }
                PROF_MARK(2);
                // Ragged far end.  Beyond the zone common to all J windows, window j still holds n_j more sites.
                // When those are far-field sites (alpha |R| <= 3e-4: the log1p series to third order is exact
                // to 2e-15), the n_j grow along the order in which the windows end and all fit one pass, their
                // power sums are simply carried on from test site to test site in the flush below:
                // no generic passes at this end of the windows.
                int nrag_v = 0;                                    // lane: n_j of test site jl = lane % J
                int nrmax = 0;
                bool rag = false;
                if (FARSUM && kmom) {
                    const int ir = base + dir * lane;
                    const bool inr = ir >= 0 && ir < N;
                    const int ic = min(max(ir, 0), N - 1);
                    const double g = P.genpos[ic];
                    const int rr = (int)P.row[ic];
                    // alpha = E F needs the sites strictly beyond ALL test sites of the group (not so when the group has
                    // no common zone; a site AT a test position is not in that test site's window, v1:456), and the
                    // count below needs every window to be open already at `base`
                    bool okr = __ballot(inr && (dir > 0 ? g <= tnear : g >= tnear)) == 0ull &&
                               __ballot(dir > 0 ? lo_j > base : hi_j < base) == 0ull;
                    if (okr) {
                        // n_j = min(sites up to the window's index bound, sites with A*|g - t_j| <= zcut): the second
                        // count is a bisection over the pass's positions (the predicate falls monotonically along
                        // the walk: positions are sorted, FP subtraction and multiplication are monotone), done by
                        // every lane for its own test site -- the same predicate, bit for bit, as the scan's
                        scr_d[lane] = g;
                        __builtin_amdgcn_wave_barrier();
                        const int cnt1 = min(max(dir > 0 ? hi_j - base + 1 : base - lo_j + 1, 0), WAVE);
                        int lo_n = 0, hi_n = WAVE;                 // predicate true below lo_n, false from hi_n on
#pragma unroll
                        for (int it = 0; it < 7; ++it) {
                            const int mid = min((lo_n + hi_n) >> 1, WAVE - 1);
                            const bool pm = A * fabs(scr_d[mid] - tj) <= P.zcut;
                            const bool act = lo_n < hi_n;
                            lo_n = act && pm ? mid + 1 : lo_n;
                            hi_n = act && !pm ? mid : hi_n;
                        }
                        nrag_v = min(cnt1, lo_n);
                        __builtin_amdgcn_wave_barrier();
                        // the windows must end in walk order (t ascending to the right, descending to the left)
                        const int nb = dir > 0 ? __shfl_up(nrag_v, 1) : __shfl_down(nrag_v, 1);
                        const bool edge = dir > 0 ? jl == 0 : jl == J - 1;
                        okr = __ballot(!edge && nb > nrag_v) == 0ull;
                        nrmax = __builtin_amdgcn_readlane(nrag_v, dir > 0 ? J - 1 : 0);
                    }
                    if (okr && nrmax > 0 && nrmax < WAVE) {
                        const double Er = exp_neg(A * fabs(g - tnear));
                        const bool far3 = lane >= nrmax || Er * rowmax[rr] <= 3e-4;     // NaN (absent row): false
                        if (__ballot(far3) == ~0ull) {
                            rag = true;
                            scr[lane] = ScratchEnt{Er, rr * rowmul, 0};
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
                PROF_MARK(4);
                CNT(15, 1);
                if (FARSUM && (nfar_tot || rag)) {
                    CNT(8, 1);
                    CNT(11, rag ? 1 : 0);
                    // fold the moments: p_k = sum_rows M_k[row] R[row]^k, then acc_j *= exp(sum_k (-1)^(k+1) F_j^k p_k / k).
                    // |F v| <= far_eps = 0.05: 8th-order series with economised coefficients (see the flush), error < 9e-16 per
                    // site.  |sum| <= nfar_tot * far_eps * 1.03 < 422.
                    __builtin_amdgcn_wave_barrier();
                    double p[FAR_ORDER];
#pragma unroll
                    for (int k = 0; k < FAR_ORDER; ++k) p[k] = 0.0;
                    if (nfar_tot) {
                    auto fold = [&](const double (&m)[FAR_ORDER], const double R) {
                        const double R2 = R * R, R3 = R2 * R, R4 = R2 * R2;
                        p[0] = fma(m[0], R, p[0]);
                        p[1] = fma(m[1], R2, p[1]);
                        p[2] = fma(m[2], R3, p[2]);
                        p[3] = fma(m[3], R4, p[3]);
                        if constexpr (FAR_ORDER >= 6) {
                            p[4] = fma(m[4], R4 * R, p[4]);
                            p[5] = fma(m[5], R4 * R2, p[5]);
                        }
                        if constexpr (FAR_ORDER >= 8) {
                            p[6] = fma(m[6], R4 * R3, p[6]);
                            p[7] = fma(m[7], R4 * R4, p[7]);
                        }
                    };
                    {   // slot 0: its MOM_COPIES copies are read lane-parallel (copy = lane / order, moment = lane % order)
                        // and added up across lanes
                        static_assert(MOM_COPIES * FAR_ORDER <= WAVE, "slot-0 copies must fit one wave-wide read");
                        double x = 0.0;
                        if (lane < MOM_COPIES * FAR_ORDER) {
                            x = mom[lane];
                            mom[lane] = 0.0;                              // ready for the next zone
                        }
                        if (BMX_PRIV01) {
                            // lane = copy * 8 + order: the private sums are first added up over each group of eight lanes, then
                            // join the copies' first (order 0) and second (order 1) entries; the loop below sums over the copies
                            double y1 = m1p, y2 = m2p;
#pragma unroll
                            for (int off = 1; off < FAR_ORDER; off <<= 1) {
                                y1 += __shfl_xor(y1, off);
                                y2 += __shfl_xor(y2, off);
                            }
                            const int k8 = lane & (FAR_ORDER - 1);
                            x += k8 == 0 ? y1 : k8 == 1 ? y2 : 0.0;
                        }
#pragma unroll
                        for (int c = MOM_COPIES / 2; c >= 1; c >>= 1) x += __shfl_down(x, c * FAR_ORDER);
                        double m[FAR_ORDER];
#pragma unroll
                        for (int k = 0; k < FAR_ORDER; ++k) m[k] = readlane_f64(x, k);
                        if (m[0] != 0.0) fold(m, loadR(P.row_of_slot[0] * rowmul));
                    }
                    // the other slots, four at a time: all LDS reads of a batch are in flight together
                    CNT(10, kmom);
                    constexpr int FB = BMX_FOLD_BATCH;                       // slots per batch: their LDS reads are in flight together
                    for (int s0 = 1; s0 < kmom; s0 += FB) {
                        double2 m2[FB][FAR_ORDER / 2];
                        double Rs[FB];
                        if (BMX_FOLD_RPRE) {      // the batch's rows are requested together with the moments
#pragma unroll
                            for (int u = 0; u < FB; ++u) Rs[u] = loadR(P.row_of_slot[min(s0 + u, MOM_SLOTS - 1)] * rowmul);
                        }
#pragma unroll
                        for (int u = 0; u < FB; ++u) {
                            double2 *mp = reinterpret_cast<double2 *>(mom + (s0 + u + MOM_COPIES - 1) * FAR_ORDER);
#pragma unroll
                            for (int q = 0; q < FAR_ORDER / 2; ++q) m2[u][q] = mp[q];    // uniform address: LDS broadcast
                        }
#pragma unroll
                        for (int u = 0; u < FB; ++u) {
                            if (m2[u][0].x == 0.0) continue;              // no far site of this row in the zone
                            CNT(9, 1);
                            double m[FAR_ORDER];
#pragma unroll
                            for (int q = 0; q < FAR_ORDER / 2; ++q) {
                                m[2 * q] = m2[u][q].x;
                                m[2 * q + 1] = m2[u][q].y;
                            }
                            fold(m, BMX_FOLD_RPRE ? Rs[u] : loadR(P.row_of_slot[min(s0 + u, MOM_SLOTS - 1)] * rowmul));
                        }
                    }
                    // ready for the next zone: one lane-parallel sweep over the slots that may have been used
                    for (int idx = MOM_COPIES * FAR_ORDER + lane; idx < (kmom + MOM_COPIES - 1) * FAR_ORDER; idx += WAVE) mom[idx] = 0.0;
                    __builtin_amdgcn_wave_barrier();
                    }
                    PROF_MARK(5);
                    spend(2 + (int)((float)(nfar_tot + nrmax) * P.far_bits));
                    // t = p1 - f (p2/2 - f (p3/3 - ...)),  log product = f t
                    // t = w1 p1 - f (w2 p2 - f (w3 p3 - ...)),  log product = f t   (w_k: FAR_W above)
#pragma unroll
                    for (int k = 0; k < FAR_ORDER; ++k) p[k] *= FAR_W[k];
                    int l = 0;
                    double rag_e = 0.0, rag_R = 0.0;
                    if (rag) {
                        const ScratchEnt en = scr[0];
                        rag_e = en.e;
                        rag_R = loadR(en.ro);
                    }
#pragma unroll
                    for (int w = 0; w < J; w += 2) {
                        // test sites in pairs: the two exps of a pair are interleaved chains (exp_neg2), and the pair's
                        // products are final before the next pair starts (no tails of 16 exps kept in registers)
                        double arg[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int j = dir > 0 ? w + u : J - 1 - (w + u);
                            if (rag) {
                                const int nj = __builtin_amdgcn_readlane(nrag_v, j);
                                for (; l < nj; ++l) {                         // the ragged sites that window j adds
                                    const double v = rag_e * rag_R, v2 = v * v;
                                    const ScratchEnt en = scr[min(l + 1, WAVE - 1)];      // the next site's entry and row: one step ahead
                                    rag_e = en.e;
                                    rag_R = loadR(en.ro);
                                    p[0] += v;
                                    p[1] = fma(v2, 0.5, p[1]);
                                    p[2] = fma(v2 * v, 0.3333333333333333, p[2]);
                                }
                            }
                            const double f = F[j];
                            double t = p[FAR_ORDER - 1];
#pragma unroll
                            for (int k = FAR_ORDER - 2; k >= 0; --k) t = fma(-f, t, p[k]);
                            arg[u] = -f * t;
                        }
                        const int j0 = dir > 0 ? w : J - 1 - w, j1 = dir > 0 ? w + 1 : J - 2 - w;
                        double e0, e1;
                        exp_neg2(arg[0], arg[1], e0, e1);
                        acc[j0] *= e0;
                        acc[j1] *= e1;
                        asm volatile("" : "+v"(acc[j0]), "+v"(acc[j1]));
                    }
                    PROF_MARK(6);
                    if (rag) {
                        __builtin_amdgcn_wave_barrier();
                        return ZONE_DONE;      // every window of the group has ended: nothing left on this side
                    }
                }
                return base;
            };

            // sites between / at the test sites (and any part of the windows not covered by bulk)
            PROF_MARK(8);
            if (MIDTRI && mid_tri && A * (tL - t0) <= P.zcut) {
                // every pair is inside the cut-off (|t_i - t_j| <= tL - t0; FP subtraction and multiplication are monotone)
                const double xk = A * (tj - t0);
                double Gk, Hk;
                exp_neg2(xk, -xk, Gk, Hk);
                const int ro_k = mid_ro[jl];
                // each test site meets the other J - 1 sites once; every factor lies in [1 - alpha, 1 + alpha Rmax], alpha < 1
                spend((J - 1) * span_generic);
                double Fs[J];
                // sites i reach the test sites j < i:  alpha_ij = H_j G_i
#pragma unroll
                for (int j = 0; j < J; ++j) Fs[j] = readlane_f64(Hk, j);
                if (lane < J) scr[lane] = ScratchEnt{Gk, ro_k, 0};
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < J; i += 2) {
                    const ScratchEnt en0 = scr[i], en1 = scr[i + 1];
                    const double v1 = en1.e * loadR(en1.ro);
                    if (i > 0) {
                        const double v0 = en0.e * loadR(en0.ro);
                        const double sv = v0 + v1, qv = v0 * v1;
#pragma unroll
                        for (int j = 0; j < i; ++j) acc[j] *= fma(Fs[j], fma(Fs[j], qv, sv), 1.0);
                    }
                    acc[i] *= fma(Fs[i], v1, 1.0);                    // test site i: site i + 1 only
                }
                __builtin_amdgcn_wave_barrier();
                // sites i reach the test sites j > i:  alpha_ij = G_j H_i
#pragma unroll
                for (int j = 0; j < J; ++j) Fs[j] = readlane_f64(Gk, j);
                if (lane < J) scr[lane] = ScratchEnt{Hk, ro_k, 0};
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < J; i += 2) {
                    const ScratchEnt en0 = scr[i], en1 = scr[i + 1];
                    const double v0 = en0.e * loadR(en0.ro);
                    if (i + 2 < J) {
                        const double v1 = en1.e * loadR(en1.ro);
                        const double sv = v0 + v1, qv = v0 * v1;
#pragma unroll
                        for (int j = i + 2; j < J; ++j) acc[j] *= fma(Fs[j], fma(Fs[j], qv, sv), 1.0);
                    }
                    acc[i + 1] *= fma(Fs[i + 1], v0, 1.0);            // test site i + 1: site i only
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                for (int b = L_int; b < R_int; b += SP) generic_pass(b, +1, R_int, staged);
            }
            PROF_MARK(0);
            // right side
            int b = bulk_zone(R_int, +1, tL, t0);
            if (b != ZONE_DONE) { CNT(12, 1); while (!generic_pass(b, +1, N, false)) { b += SP; CNT(12, 1); } }
            PROF_MARK(7);
            // left side
            b = bulk_zone(L_int - 1, -1, t0, tL);
            if (b != ZONE_DONE) { CNT(12, 1); while (!generic_pass(b, -1, -1, false)) { b -= SP; CNT(12, 1); } }
            PROF_MARK(7);

            renorm_all();
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int ec = min(max(E[j], -131071), 131071) + 131072;
                const int eb = bestK[j] >> 13;
                const bool better = (ec > eb) || (ec == eb && acc[j] > bestM[j]);
                if (better && p < P.npairs) {            // strict '>' (v1:501); iA ascending
                    bestM[j] = acc[j];
                    bestK[j] = (ec << 13) | iA;
                }
            }
        }
#pragma unroll
        // slice winner per test site: (exponent, mantissa) compared exactly, ties to the smaller linear index; the one
        // logarithm per test site is taken by finalize_kernel (no log, and none of its constants, in this kernel)
        for (int j = 0; j < J; ++j) {
            const int biA = bestK[j] & 8191;
            int bE = bestK[j] >> 13;
            double bM = bestM[j];
            int bL = biA == 8191 ? 0x7fffffff : biA * P.npairs + p;
            for (int off = 32; off > 0; off >>= 1) {
                const int oE = __shfl_xor(bE, off);
                const double oM = __shfl_xor(bM, off);
                const int oL = __shfl_xor(bL, off);
                if (oE > bE || (oE == bE && (oM > bM || (oM == bM && oL < bL)))) { bE = oE; bM = oM; bL = oL; }
            }
            if (lane == 0 && j < nvalid) {
                const size_t o = (size_t)slice * P.M + (tb + j);   // [slice][M]
                P.part_T[o] = bM;
                P.part_lin[o] = bL;
                P.part_ns[o] = bE;
            }
        }
    }
#ifdef BMX_PROFILE
    if (lane == 0 && P.prof)
        for (int k = 0; k < 12; ++k) atomicAdd(P.prof + k, (unsigned long long)prof_[k]);
#endif
#ifdef BMX_COUNT
    if (lane == 0 && P.prof)
        for (int k = 0; k < 16; ++k) atomicAdd(P.prof + 16 + k, cnt_[k]);
#endif
}

// ----------------------------------------------------------------------------- K2, prepared (round 3)
// Everything the grouped kernel did per pass of 64 sites -- loading positions and rows, exp(-A d), the near / far
// classification, the ds_add_f64 moment sums, the ragged-end counts -- is the same for all eight 64-pair slices of a group of
// test sites, and the grouped kernel did it once per slice (a quarter of its cycles, round-2 profiles).  Here it is done ONCE
// per group, by a separate lanes-over-sites kernel with a small register footprint (prep_kernel), which writes what the
// pair-parallel kernel needs as one sequential stream per group ("blob") in HBM; clr_scan_prepared_kernel -- one wave per
// (group, slice) as before -- then only multiplies, folds and flushes, reading its blob through a ring in LDS.
//
//   blob(group)  = for iA in 0..nA-1: near(iA, right), near(iA, left), far(iA, right), far(iA, left);   16-byte units,
//                  padded to a multiple of 4
//   near(zone)   = header (3 units): {magic | rag, n_pair, n_quad, n_occ} {n_far, n_rag, end index | ZONE_DONE, n_ser | class counts << 8} {n_j bytes}
//                  near list: n_pair entries with alpha > 1/2 (padded to whole blocks), then n_quad entries (multiple of 4),
//                             each (E_i f64, row offset i32), in walk order; 8 neutral guard entries (what the block loops
//                             request one block ahead)
//                  ragged end (rag only): n_rag entries (E, row offset, flag) + 1 guard   (round 4: here, not behind the far field)
//   far(zone)    = moments:   n_occ entries of PREP_MOM units: (M_1, row offset) (M_2, M_3) (M_4, M_5) ...
//                  series entries: n_ser (a multiple of 4, <= 64) entries (E, row offset, order class) of far sites whose row has no
//                                  moment slot, highest class first
//   (both near lists first: the consumer multiplies, then takes ONE exp per test site for the two far fields together)
// Sizes come from a counting pass of the same code (prep_kernel<J, false>: identical predicates, no exp, no stores) run when
// the test sites are set, then an exclusive scan; the fill pass (prep_kernel<J, true>) and the consumer run per launch range.
// The far test is slice-independent now: alpha * max_grid |R[row]| <= far_eps, evaluated in the exponent domain
// (A d >= log(max|R| / far_eps), rounded up), so both passes classify from A d alone.
// Far field of the prepared kernels: log1p(x) = sum_k (-1)^(k+1) w_k x^k to order P_ORDER on |x| <= P_EPS, coefficients economised
// on that interval (scripts/far_series.py: Taylor polynomial of degree 40 re-expanded in Chebyshev polynomials of x / eps, cut after
// T_K, constant term < 3e-17 dropped).  Order 12 on [-0.15, 0.15]: max error 4.5e-16 (the round-2 kernels: order 8 on [-0.05, 0.05],
// 8.9e-16) -- the far field starts 1.1 units of A d nearer to the test sites, and the near lists, half of the scan kernel's
// instructions, lose a third of their entries.
#ifndef BMX_P_ORDER
#define BMX_P_ORDER 12
#endif
constexpr int P_ORDER = BMX_P_ORDER;
constexpr int P_COPIES = 4;                   // copies of the most frequent row's moments (lane % 4); P_COPIES * P_ORDER <= 64
static_assert(P_ORDER == 8 || P_ORDER == 12 || P_ORDER == 16, "far-field order of the prepared kernels: 8, 12 or 16");
static_assert(P_COPIES * P_ORDER <= WAVE, "slot-0 copies must fit one wave-wide read");
constexpr double P_EPS = P_ORDER == 16 ? 0.25 : P_ORDER == 12 ? 0.15 : 0.05;
constexpr int P_FAR_CAP = P_ORDER == 16 ? 2048 : P_ORDER == 12 ? 3584 : 8192;     // far sites per zone: P_FAR_CAP * P_EPS * 1.6 bits < 1000
#if BMX_P_ORDER == 16
__device__ constexpr double P_W[16] = {0.9999999999999954, 0.4999999999999791, 0.3333333333368442, 0.250000000008901, 0.19999999921757028,
                                       0.1666666652123534, 0.14285722101681053, 0.1250001188192722, 0.11110698172974029, 0.09999456222768308,
                                       0.09103227543438049, 0.08347880124963403, 0.07484961178075682, 0.06918273076087589, 0.08475562956871537,
                                       0.08074032343426385};
// a lane adds E^k (k >= 3) only while its term can exceed 2e-15: d = A d - threshold < P_D[k - 3]  (x = P_EPS exp(-d))
__device__ constexpr double P_D[14] = {9.539, 6.739, 5.071, 3.966, 3.181, 2.594, 2.14, 1.778, 1.483, 1.237, 1.03, 0.853, 0.7, 0.566};
#elif BMX_P_ORDER == 12
__device__ constexpr double P_W[12] = {0.9999999999999661, 0.49999999999988076, 0.3333333333754503, 0.25000000008463336, 0.19999998506452038,
                                       0.1666666441614055, 0.14285940976674, 0.1250028457483109, 0.11094429253027539, 0.09981580095968261,
                                       0.09676290425438011, 0.08920395582674094};
__device__ constexpr double P_D[10] = {9.029, 6.228, 4.56, 3.455, 2.67, 2.084, 1.629, 1.267, 0.972, 0.726};
#else
__device__ constexpr double P_W[8] = {0.9999999999998467, 0.4999999999996164, 0.3333333341509627, 0.25000000122701976,
                                      0.19999882322703955, 0.16666529319200146, 0.1434841013671569, 0.12562715480255862};
__device__ constexpr double P_D[6] = {7.94, 5.13, 3.462, 2.357, 1.571, 0.985};
#endif
constexpr double P_RAG_D = P_ORDER == 16 ? 6.74 : P_ORDER == 12 ? 6.23 : 5.13;     // log(P_EPS / 3e-4): the ragged end's third-order test
constexpr int PREP_HDR = 3, PREP_GUARD = 8, PREP_MOM = 1 + P_ORDER / 2, PREP_RAG_GUARD = 1;     // a moment entry: (M_1, row) + M_2 .. M_K in pairs
// Far sites of rows WITHOUT a moment slot (rare rows: fewer than ~1.3 sites per zone, so a slot's fold would cost more than it
// saves): not multiplied either -- they go into the stream as "series entries" (E, row, order class) and the scan kernel adds
// their (E R)^k to the same power sums p_k the moments feed, up to the order their alpha max|R| needs (2, 3, 5 or 8):
// 3 to 9 instructions per entry for all 16 test sites, against 24.5 in the product form.  A third of the old near lists were such sites.
constexpr double SER_DMIN = P_ORDER > 8 ? P_D[P_ORDER > 8 ? 6 : 0] : 0.0;      // order 9 and beyond below 2e-15
constexpr int SER_CAP = 64;                  // series entries per zone (buffered in prep_kernel's LDS until the zone's far part is written)
constexpr int PREP_MAGIC = 0x5a0e0000;
constexpr int RING_UNITS = 256, RING_MIRROR = 32, AUX_UNITS = 32;     // per wave: ring of 4 x 64 units + 32 mirrored + scratch
constexpr int PREP_ZONE_DONE = -0x7fffffff;
constexpr int PREP_THREADS = 256;
constexpr int PREP_THR_LDS_MAX = 4096;                                // rows whose far thresholds are staged in LDS
constexpr int64_t SOLO_GAP = 10;                                      // median gap between test sites beyond which groups stop paying (round 4, J = 8 at three waves per SIMD: 1.64 / 1.55 M, solo 1.575 / 1.574 M windows/s at stride 10 / 11)

struct PrepParams {
    const double *genpos;
    RowArray row;
    int64_t N;
    int rows, rowmul;             // consumer's row offset = row * rowmul (64: LDS slice, NP: global table)
    const double *A;
    int nA;
    const double *test_gen;       // the slot's test sites (absolute indexing: group g = test sites [g J, g J + J))
    const int64_t *win_lo, *win_hi, *center, *center_hi;
    int64_t M;
    double zcut;
    const double *rowthr;         // [rows]: A d from which a site of the row is far, the row's moment slot in the low byte
    int thr_in_lds;
    const uint8_t *kmom;
    const int *row_of_slot;       // [MOM_SLOTS]
    int mom_slots;
    int ser_cap;       // series entries per zone (SER_CAP; diagnostic builds: BMX_SER_CAP, 0 = none)
    int64_t g_begin, g_end;       // groups of this launch
    int32_t *blob_units;          // counting pass: [groups of the slot]
    const int64_t *blob_prefix;   // fill pass: exclusive prefix of blob_units
    int64_t prefix_base;          // ... of the launch range's first group: the arena holds the range's blobs from unit 0
    ScratchEnt *arena;
    int *status;                  // bit 0: a blob came out longer/shorter than counted, bit 1: bad header seen by the consumer
};

template <int J, bool FILL>
__global__ __launch_bounds__(PREP_THREADS) void prep_kernel(PrepParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds_p[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x / WAVE;
    const int N = (int)P.N;
    const int thr_len = P.thr_in_lds ? ((P.rows + 1) & ~1) : 0;
    if (P.thr_in_lds) {
        for (int idx = threadIdx.x; idx < P.rows; idx += blockDim.x) lds_p[idx] = P.rowthr[idx];
        __syncthreads();
    }
    // per wave: the moments of the right and of the left zone of one A (the stream holds both zones' near lists before either
    // zone's moments, so the right zone's sums wait while the left zone is walked), then 64 doubles of scratch.  P.mom_slots
    // is the largest number of slots any A uses (not the 64 / 254 the table would allow): LDS per wave decides how many waves
    // of this latency-bound kernel a CU holds.  The counting pass keeps one occupancy flag per slot (MS = 1).
    constexpr int MS = FILL ? P_ORDER : 1;
    const int mom_len = (P.mom_slots + P_COPIES - 1 + 3) * MS;
    double *mom_r = lds_p + thr_len + wave * (2 * mom_len + WAVE);
    double *mom_l = mom_r + mom_len;
    double *ragscr = mom_l + mom_len;
    // ... and (fill pass) the series entries of the two zones, kept until the zones' far parts are written
    ScratchEnt *ser_r = reinterpret_cast<ScratchEnt *>(lds_p + thr_len + nw * (2 * mom_len + WAVE)) + wave * (2 * SER_CAP);
    ScratchEnt *ser_l = ser_r + SER_CAP;
    for (int idx = lane; idx < 2 * mom_len; idx += WAVE) mom_r[idx] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int64_t grp = P.g_begin + (int64_t)blockIdx.x * nw + wave;
    if (grp >= P.g_end) return;                     // (no workgroup barrier below this line)
    auto thr_of = [&](int r) -> double {
        double v;
        if (P.thr_in_lds) v = lds_p[r]; else v = P.rowthr[r];
        return v;
    };
    auto rank = [&](unsigned long long m) {
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    };
    constexpr int BS = J >= 16 ? 4 : 8;             // sites per pair block of the consumer

    // group prologue: the consumer's, value for value
    const int64_t tb = grp * J;
    const int nvalid = (int)min((int64_t)J, P.M - tb);
    const int jl = lane % J;
    const int jj = min(jl, nvalid - 1);
    const double tj = P.test_gen[tb + jj];
    int lo_j = (int)max(P.win_lo[tb + jj], (int64_t)0);
    int hi_j = (int)min(P.win_hi[tb + jj], (int64_t)N - 1);
    if (jl >= nvalid) { lo_j = 1; hi_j = 0; }
    const double t0 = readlane_f64(tj, 0), tL = readlane_f64(tj, J - 1);
    const int c0 = (int)P.center[tb], cU = (int)P.center_hi[tb + nvalid - 1];
    int lo_max = 0, hi_min = N - 1;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int a = __builtin_amdgcn_readlane(lo_j, j), b = __builtin_amdgcn_readlane(hi_j, j);
        if (j < nvalid) { lo_max = max(lo_max, a); hi_min = min(hi_min, b); }
    }
    int L_int = min(c0, hi_min + 1), R_int = max(cU, lo_max);
    if (hi_min < lo_max) { L_int = c0; R_int = c0; hi_min = -1; lo_max = N; }   // no bulk zone

    int wpos = 0;                                   // units of the blob written (counted) so far
    ScratchEnt *out = nullptr;
    if (FILL) out = P.arena + (P.blob_prefix[grp] - P.prefix_base);

    for (int iA = 0; iA < P.nA; ++iA) {
        const double A = P.A[iA];
        const int kmom = min((int)P.kmom[iA], P.mom_slots);
        // One zone, first half: header slot, near list (pairs, then quads), guard; the far sites' moments go to `mom`; the
        // ragged end is sized.  What the second half needs comes back through the reference arguments.
        auto zone_near = [&](int base, int dir, double tnear, double tfar, double *mom, ScratchEnt *ser, int &zbase_o, int &npp_o, int &nqp_o,
                             int &nfar_o, int &base_o, double &m1p_o, double &m2p_o, int &nragv_o, int &nrmax_o, int &rag_o, double &zr_o,
                             int &rr_o, int &nser_o) {
            const int zbase = wpos, nbase = zbase + PREP_HDR;
            int n_pair = 0, n_pair_pad = 0, n_quad = 0, nfar_tot = 0, pad_ro = 0, n_ser = 0;
            bool pair_open = true, seen = false;
            double m1p = 0.0, m2p = 0.0;
            auto close_pairs = [&]() {
                n_pair_pad = (n_pair + BS - 1) & ~(BS - 1);
                if (FILL && lane < n_pair_pad - n_pair) out[nbase + n_pair + lane] = ScratchEnt{0.0, pad_ro, 0};
                pair_open = false;
            };
            int i = base + dir * lane;
            double g_nx = P.genpos[min(max(i, 0), N - 1)];
            int r_nx = (int)P.row[min(max(i, 0), N - 1)];
            while (true) {
                const bool ok = dir > 0 ? (i <= hi_min) : (i >= lo_max);
                const double g = g_nx;
                const int rraw = r_nx;
                const int inx = i + dir * WAVE;
                g_nx = P.genpos[min(max(inx, 0), N - 1)];
                r_nx = (int)P.row[min(max(inx, 0), N - 1)];
                const bool bulk = ok && (A * fabs(g - tfar) <= P.zcut);
                const int cnt = __popcll(__ballot(bulk));
                if (cnt) {
                    const double zn = A * fabs(g - tnear);
                    const double th = thr_of(rraw);
                    const int slot = __double2loint(th) & 0xff;
                    const bool farx = bulk && zn >= th && nfar_tot + n_ser < P_FAR_CAP;      // alpha max|R| <= P_EPS (NaN threshold: never)
                    const bool moml = farx && slot < kmom;
                    // a far site whose row has no moment slot: a series entry, while the zone's buffer has room for the whole pass
                    // (one that still needs more than 8 orders costs as much there as in the product: it stays near)
                    const bool serx = farx && slot >= kmom && !(zn - th < SER_DMIN);
                    const unsigned long long ms_ = __ballot(serx);
                    const bool ser_ok = n_ser + __popcll(ms_) <= P.ser_cap;
                    const bool serl = ser_ok && serx;
                    const bool nearl = bulk && !moml && !serl;
                    const bool pairl = nearl && zn < LN2;
                    const unsigned long long mm = __ballot(moml), mp = __ballot(pairl), mq = __ballot(nearl && !pairl);
                    const int nfar = __popcll(mm);
                    if (!seen) { pad_ro = __builtin_amdgcn_readlane(rraw, 0) * P.rowmul; seen = true; }   // lane 0: a site of the zone
                    double Ev = 0.0;
                    if (FILL) Ev = bulk ? exp_neg(zn) : 0.0;
                    if (nfar) {
                        if (moml) {
                            double *mr = mom + (slot ? slot + P_COPIES - 1 : (lane & (P_COPIES - 1))) * MS;
                            if (FILL) {
                                // a lane adds only the powers whose term can exceed 2e-15: x = alpha max|R| = far_eps exp(-(z - th))
                                const double d = zn - th;
                                const double E2 = Ev * Ev;
                                if (slot == 0) { m1p += Ev; m2p += E2; }
                                else { atomicAdd(mr, Ev); atomicAdd(mr + 1, E2); }
                                double Ek = E2;
#pragma unroll
                                for (int k = 3; k <= P_ORDER; ++k) {
                                    if (!(d < P_D[k - 3])) break;
                                    Ek *= Ev;
                                    atomicAdd(mr + (k - 1), Ek);
                                }
                            } else {
                                mom[(slot ? slot + P_COPIES - 1 : 0) * MS] = 1.0;      // occupancy only
                            }
                        }
                        nfar_tot += nfar;
                    }
                    if (ser_ok && ms_ != 0ull) {
                        if (FILL && serl) {
                            // order class by how far past the threshold the site lies: x = P_EPS exp(-(z - th)); order k + 1 matters while d < P_D[k - 2]
                            const double d = zn - th;
                            const int cls = !(d < P_D[0]) ? 2 : !(d < P_D[1]) ? 3 : !(d < P_D[3]) ? 5 : 8;
                            ser[n_ser + rank(ms_)] = ScratchEnt{Ev, rraw * P.rowmul, cls};
                        }
                        n_ser += __popcll(ms_);
                    }
                    if (pair_open) {
                        if (FILL && pairl) out[nbase + n_pair + rank(mp)] = ScratchEnt{Ev, rraw * P.rowmul, 0};
                        n_pair += __popcll(mp);
                        if (__ballot(bulk && !(zn < LN2)) != 0ull || cnt < WAVE) close_pairs();
                    }
                    if (!pair_open) {
                        if (FILL && nearl && !pairl) out[nbase + n_pair_pad + n_quad + rank(mq)] = ScratchEnt{Ev, rraw * P.rowmul, 0};
                        n_quad += __popcll(mq);
                    }
                }
                base += dir * cnt;
                if (cnt < WAVE) break;
                i = inx;
            }
            if (pair_open) close_pairs();
            const int n_quad_pad = (n_quad + 3) & ~3;
            if (FILL) {
                if (lane < n_quad_pad - n_quad) out[nbase + n_pair_pad + n_quad + lane] = ScratchEnt{0.0, pad_ro, 0};
                if (lane < PREP_GUARD) out[nbase + n_pair_pad + n_quad_pad + lane] = ScratchEnt{0.0, pad_ro, 0};
            }
            wpos = nbase + n_pair_pad + n_quad_pad + PREP_GUARD;

            // ragged far end (see the grouped kernel): per-window counts n_j by bisection with the scan's own predicate
            int nrag_v = 0, nrmax = 0;
            int rag = 0;                 // bit 0: the ragged end goes into the stream; bit 1: some of its sites need more than three orders
            double zr = 0.0;
            int rr = 0;
            {   // (also where no row has a moment slot: series entries and the ragged end do not need one)
                const int ir = base + dir * lane;
                const bool inr = ir >= 0 && ir < N;
                const int ic = min(max(ir, 0), N - 1);
                const double g = P.genpos[ic];
                rr = (int)P.row[ic];
                bool okr = __ballot(inr && (dir > 0 ? g <= tnear : g >= tnear)) == 0ull &&
                           __ballot(dir > 0 ? lo_j > base : hi_j < base) == 0ull;
                if (okr) {
                    ragscr[lane] = g;
                    __builtin_amdgcn_wave_barrier();
                    const int cnt1 = min(max(dir > 0 ? hi_j - base + 1 : base - lo_j + 1, 0), WAVE);
                    int lo_n = 0, hi_n = WAVE;
#pragma unroll
                    for (int it = 0; it < 7; ++it) {
                        const int mid = min((lo_n + hi_n) >> 1, WAVE - 1);
                        const bool pm = A * fabs(ragscr[mid] - tj) <= P.zcut;
                        const bool act = lo_n < hi_n;
                        lo_n = act && pm ? mid + 1 : lo_n;
                        hi_n = act && !pm ? mid : hi_n;
                    }
                    nrag_v = min(cnt1, lo_n);
                    __builtin_amdgcn_wave_barrier();
                    const int nb = dir > 0 ? __shfl_up(nrag_v, 1) : __shfl_down(nrag_v, 1);
                    const bool edge = dir > 0 ? jl == 0 : jl == J - 1;
                    okr = __ballot(!edge && nb > nrag_v) == 0ull;
                    nrmax = __builtin_amdgcn_readlane(nrag_v, dir > 0 ? J - 1 : 0);
                }
                if (okr && nrmax > 0 && nrmax < WAVE) {
                    zr = A * fabs(g - tnear);
                    // every site of it far enough for eight orders (alpha max|R| <= 0.03; NaN threshold, absent row: false); most of
                    // the time for three (<= 3e-4).  Anything nearer: the zone ends before it and the scan kernel walks it.
                    const bool far8 = lane >= nrmax || zr >= thr_of(rr) + SER_DMIN;
                    const bool far3 = lane >= nrmax || zr >= thr_of(rr) + P_RAG_D;
                    rag = __ballot(far8) == ~0ull ? (__ballot(far3) == ~0ull ? 1 : 3) : 0;
                }
            }
            // the ragged end's entries follow the near list (the consumer walks them there, before either far field)
            if (rag) {
                if (FILL) {
                    if (lane < nrmax) out[wpos + lane] = ScratchEnt{exp_neg(zr), rr * P.rowmul, zr >= thr_of(rr) + P_RAG_D ? 0 : 1};      // 1: orders 4..8 too
                    if (lane == nrmax) out[wpos + lane] = ScratchEnt{0.0, rr * P.rowmul, 0};
                }
                wpos += nrmax + PREP_RAG_GUARD;
            }
            zbase_o = zbase; npp_o = n_pair_pad; nqp_o = n_quad_pad; nfar_o = nfar_tot; base_o = base; m1p_o = m1p; m2p_o = m2p;
            nragv_o = nrag_v; nrmax_o = nrmax; rag_o = rag; zr_o = zr; rr_o = rr; nser_o = n_ser;
        };
        // ... second half: the moments of the occupied slots (slot order), the ragged end's entries, and the header
        auto zone_far = [&](double *mom, const ScratchEnt *ser, int n_ser, int zbase, int n_pair_pad, int n_quad_pad, int nfar_tot, int base,
                            double m1p, double m2p, int nrag_v, int nrmax, int rag, double zr, int rr) {
            int n_occ = 0;
            if (nfar_tot) {
                __builtin_amdgcn_wave_barrier();
                {   // slot 0: P_COPIES copies read lane-parallel (copy = lane / order, moment = lane % order) + the private sums
                    double x = 0.0;
                    if (lane < P_COPIES * MS) {
                        x = mom[lane];
                        mom[lane] = 0.0;
                    }
                    if (FILL) {
                        double y1 = m1p, y2 = m2p;
#pragma unroll
                        for (int off = 1; off < WAVE; off <<= 1) {
                            y1 += __shfl_xor(y1, off);
                            y2 += __shfl_xor(y2, off);
                        }
                        x += lane == 0 ? y1 : lane == 1 ? y2 : 0.0;       // copy 0, orders 1 and 2
                    }
#pragma unroll
                    for (int c = P_COPIES / 2; c >= 1; c >>= 1) x += __shfl_down(x, c * MS);
                    const double m0 = readlane_f64(x, 0);
                    if (m0 != 0.0) {
                        if (FILL) {
                            double m[P_ORDER];
#pragma unroll
                            for (int k = 0; k < P_ORDER; ++k) m[k] = readlane_f64(x, k);
                            if (lane == 0) {
                                ScratchEnt *o = out + wpos;
                                o[0] = ScratchEnt{m[0], P.row_of_slot[0] * P.rowmul, 0};
                                double2 *o2 = reinterpret_cast<double2 *>(o + 1);
#pragma unroll
                                for (int q = 0; q < P_ORDER / 2; ++q) o2[q] = double2{m[2 * q + 1], 2 * q + 2 < P_ORDER ? m[2 * q + 2] : 0.0};
                            }
                        }
                        n_occ = 1;
                    }
                }
                for (int s0 = 1; s0 < kmom; s0 += WAVE) {
                    const int s = s0 + lane;
                    double m[P_ORDER];
#pragma unroll
                    for (int k = 0; k < P_ORDER; ++k) m[k] = 0.0;
                    if (s < kmom) {
                        double *ms = mom + (s + P_COPIES - 1) * MS;
#pragma unroll
                        for (int k = 0; k < MS; ++k) { m[k] = ms[k]; ms[k] = 0.0; }
                    }
                    const bool occ = m[0] != 0.0;
                    const unsigned long long mo = __ballot(occ);
                    if (FILL && occ) {
                        ScratchEnt *o = out + wpos + PREP_MOM * (n_occ + rank(mo));
                        o[0] = ScratchEnt{m[0], P.row_of_slot[s] * P.rowmul, 0};
                        double2 *o2 = reinterpret_cast<double2 *>(o + 1);
#pragma unroll
                        for (int q = 0; q < P_ORDER / 2; ++q) o2[q] = double2{m[2 * q + 1], 2 * q + 2 < P_ORDER ? m[2 * q + 2] : 0.0};
                    }
                    n_occ += __popcll(mo);
                }
                __builtin_amdgcn_wave_barrier();
                wpos += PREP_MOM * n_occ;
            }
            // series entries, sorted by class (highest first) and padded to a multiple of four with neutral entries
            const int n_ser_pad = (n_ser + 3) & ~3;
            int ser_w = n_ser_pad;
            if (n_ser) {
                if (FILL) {
                    __builtin_amdgcn_wave_barrier();
                    const ScratchEnt en = lane < n_ser ? ser[lane] : ScratchEnt{0.0, P.row_of_slot[0] * P.rowmul, 2};
                    const int cls = lane < n_ser ? en.pad : 0;
                    const unsigned long long m8 = __ballot(cls == 8), m5 = __ballot(cls == 5), m3 = __ballot(cls == 3), m2 = __ballot(cls == 2);
                    const int c8 = __popcll(m8), c5 = c8 + __popcll(m5), c3 = c5 + __popcll(m3);
                    const int dst = cls == 8 ? rank(m8) : cls == 5 ? c8 + rank(m5) : cls == 3 ? c5 + rank(m3) : cls == 2 ? c3 + rank(m2) : lane;
                    if (lane < n_ser_pad) out[wpos + dst] = en;
                    ser_w |= c8 << 8 | c5 << 16 | c3 << 24;
                    __builtin_amdgcn_wave_barrier();
                }
                wpos += n_ser_pad;
            }
            if (FILL) {
                // n_j of the J test sites as bytes (n_j < 64)
                int w = nrag_v & 0xff;
                w |= (__shfl_down(nrag_v, 1) & 0xff) << 8;
                w |= (__shfl_down(nrag_v, 2) & 0xff) << 16;
                w |= (__shfl_down(nrag_v, 3) & 0xff) << 24;
                const int w0 = __builtin_amdgcn_readlane(w, 0), w1 = __builtin_amdgcn_readlane(w, 4 % J),
                          w2 = __builtin_amdgcn_readlane(w, 8 % J), w3 = __builtin_amdgcn_readlane(w, 12 % J);
                if (lane == 0) {
                    int4 *o = reinterpret_cast<int4 *>(out + zbase);
                    o[0] = int4{PREP_MAGIC | rag, n_pair_pad, n_quad_pad, n_occ};
                    o[1] = int4{nfar_tot + n_ser, rag ? nrmax : 0, rag ? PREP_ZONE_DONE : base, ser_w};      // n_far: every site of the far field
                    o[2] = int4{w0, w1, w2, w3};
                }
            }
        };
        // stream order: both zones' near lists, then both zones' far fields (the consumer multiplies first and takes ONE exp
        // per test site for the two far fields together)
        int zbR, nppR, nqpR, nfR, beR, nrvR, nrmR, rrR, nsR, zbL, nppL, nqpL, nfL, beL, nrvL, nrmL, rrL, nsL;
        int ragR, ragL;
        double m1R, m2R, zrR, m1L, m2L, zrL;
        zone_near(R_int, +1, tL, t0, mom_r, ser_r, zbR, nppR, nqpR, nfR, beR, m1R, m2R, nrvR, nrmR, ragR, zrR, rrR, nsR);
        zone_near(L_int - 1, -1, t0, tL, mom_l, ser_l, zbL, nppL, nqpL, nfL, beL, m1L, m2L, nrvL, nrmL, ragL, zrL, rrL, nsL);
        zone_far(mom_r, ser_r, nsR, zbR, nppR, nqpR, nfR, beR, m1R, m2R, nrvR, nrmR, ragR, zrR, rrR);
        zone_far(mom_l, ser_l, nsL, zbL, nppL, nqpL, nfL, beL, m1L, m2L, nrvL, nrmL, ragL, zrL, rrL);
    }
    const int units = (wpos + 3) & ~3;
    if (!FILL) {
        if (lane == 0) P.blob_units[grp] = units;
    } else {
        if (lane == 0 && (int64_t)units != P.blob_prefix[grp + 1] - P.blob_prefix[grp]) atomicOr(P.status, 1);
    }
}

// exclusive prefix sums of the blob sizes (one workgroup; the array has one entry per group of test sites)
__global__ __launch_bounds__(1024) void prefix_kernel(const int32_t *u, int64_t n, int64_t *pre) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = min((int64_t)t * per, n), e = min(b + per, n);
    int64_t s = 0;
    for (int64_t i = b; i < e; ++i) s += u[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = part[t] - s;
    for (int64_t i = b; i < e; ++i) { pre[i] = run; run += u[i]; }
    if (t == 1023) pre[n] = part[1023];
}

// The ragged end's walk from entry l up to (not including) nj, as ONE asm statement: entry l comes out of the registers that hold the
// zone's ragged entries lanes-over-entries (v_readlane), its R was requested ahead (three registers, rotated), and its first three
// Taylor terms go into the zone's power sums p0..p2.  The same instructions the compiler made of the C++ loop -- but as a statement
// without control flow it no longer puts sixteen copies of a loop into the unrolled loop over the test sites, which is what made the
// register allocator spill 49 registers around them (round 3).  An LDS read may still be in flight when the statement ends: the
// caller waits for it (rag_walk_settle) before anything else may touch R2.
__device__ __forceinline__ void rag_walk_lds(int &l, int nj, int nrmax, int e_lo, int e_hi, int ro, unsigned lds_lane_addr,
                                             double &R0, double &R1, double &R2, double &p0, double &p1, double &p2) {
    int tmp;
    unsigned addr;
    double v, v2;
    const double third = 0.3333333333333333;
    asm volatile(
        "s_cmp_ge_i32 %[l], %[nj]\n\t"
        "s_cbranch_scc1 2f\n"
        "1:\n\t"
        "v_readlane_b32 vcc_lo, %[elo], %[l]\n\t"
        "v_readlane_b32 vcc_hi, %[ehi], %[l]\n\t"
        "s_add_i32 %[tmp], %[l], 3\n\t"
        "s_min_i32 %[tmp], %[tmp], %[nrmax]\n\t"
        "v_readlane_b32 %[tmp], %[ro], %[tmp]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f64 %[v], %[R0], vcc\n\t"
        "v_mov_b64 %[R0], %[R1]\n\t"
        "v_mov_b64 %[R1], %[R2]\n\t"
        "v_lshl_add_u32 %[addr], %[tmp], 3, %[lane]\n\t"
        "ds_read_b64 %[R2], %[addr]\n\t"
        "v_mul_f64 %[v2], %[v], %[v]\n\t"
        "v_add_f64 %[p0], %[p0], %[v]\n\t"
        "v_fma_f64 %[p1], %[v2], 0.5, %[p1]\n\t"
        "v_mul_f64 %[v2], %[v2], %[v]\n\t"
        "v_fma_f64 %[p2], %[v2], %[third], %[p2]\n\t"
        "s_add_i32 %[l], %[l], 1\n\t"
        "s_cmp_lt_i32 %[l], %[nj]\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        : [l] "+s"(l), [R0] "+v"(R0), [R1] "+v"(R1), [R2] "+v"(R2), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2),
          [tmp] "=&s"(tmp), [v] "=&v"(v), [v2] "=&v"(v2), [addr] "=&v"(addr)
        : [nj] "s"(nj), [nrmax] "s"(nrmax), [elo] "v"(e_lo), [ehi] "v"(e_hi), [ro] "v"(ro), [lane] "v"(lds_lane_addr), [third] "s"(third)
        : "vcc", "scc", "memory");
}
__device__ __forceinline__ void rag_walk_settle(double &R0, double &R1, double &R2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R0), "+v"(R1), "+v"(R2) : : "memory");
}
// ... the same walk with the R table in global memory / L2 (tables too large for LDS): row offsets are indices into the table
// (row * NP), the load is global_load_dwordx2 with the slice's base in a scalar pair, and the wait is on vmcnt.  Here the three R
// registers are taken in turn, RA -> RB -> RC -> RA, with the turn `ph` carried from statement to statement: the R of entry l was
// requested three entries ago and the walk waits for exactly that one (vmcnt(2): loads return in order), no register is moved.
// Same box, 262k-window blocks: 11 / 41 sample sizes per file 3.459 -> 3.504 / 2.638 -> 2.710 M windows/s.  With the table in LDS
// the same form measured 0.4-0.8 % SLOWER than the rotation (three more scalar compares and branches per statement, for a latency
// that is short there), so rag_walk_lds keeps it.
__device__ __forceinline__ void rag_walk_global(int &l, int &ph, int nj, int nrmax, int e_lo, int e_hi, int ro, unsigned lane8, const char *Rb,
                                             double &RA, double &RB, double &RC, double &p0, double &p1, double &p2) {
    int tmp;
    unsigned addr;
    double v, v2;
    const double third = 0.3333333333333333;
    asm volatile(
        "s_cmp_ge_i32 %[l], %[nj]\n\t"
        "s_cbranch_scc1 9f\n\t"
        "s_cmp_eq_u32 %[ph], 1\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_cmp_eq_u32 %[ph], 2\n\t"
        "s_cbranch_scc1 3f\n"
        "1:\n\t"
        "v_readlane_b32 vcc_lo, %[elo], %[l]\n\t"
        "v_readlane_b32 vcc_hi, %[ehi], %[l]\n\t"
        "s_add_i32 %[tmp], %[l], 3\n\t"
        "s_min_i32 %[tmp], %[tmp], %[nrmax]\n\t"
        "v_readlane_b32 %[tmp], %[ro], %[tmp]\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_mul_f64 %[v], %[RA], vcc\n\t"
        "v_lshl_add_u32 %[addr], %[tmp], 3, %[lane]\n\t"
        "global_load_dwordx2 %[RA], %[addr], %[base]\n\t"
        "v_mul_f64 %[v2], %[v], %[v]\n\t"
        "v_add_f64 %[p0], %[p0], %[v]\n\t"
        "v_fma_f64 %[p1], %[v2], 0.5, %[p1]\n\t"
        "v_mul_f64 %[v2], %[v2], %[v]\n\t"
        "v_fma_f64 %[p2], %[v2], %[third], %[p2]\n\t"
        "s_add_i32 %[l], %[l], 1\n\t"
        "s_mov_b32 %[ph], 1\n\t"
        "s_cmp_ge_i32 %[l], %[nj]\n\t"
        "s_cbranch_scc1 9f\n"
        "2:\n\t"
        "v_readlane_b32 vcc_lo, %[elo], %[l]\n\t"
        "v_readlane_b32 vcc_hi, %[ehi], %[l]\n\t"
        "s_add_i32 %[tmp], %[l], 3\n\t"
        "s_min_i32 %[tmp], %[tmp], %[nrmax]\n\t"
        "v_readlane_b32 %[tmp], %[ro], %[tmp]\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_mul_f64 %[v], %[RB], vcc\n\t"
        "v_lshl_add_u32 %[addr], %[tmp], 3, %[lane]\n\t"
        "global_load_dwordx2 %[RB], %[addr], %[base]\n\t"
        "v_mul_f64 %[v2], %[v], %[v]\n\t"
        "v_add_f64 %[p0], %[p0], %[v]\n\t"
        "v_fma_f64 %[p1], %[v2], 0.5, %[p1]\n\t"
        "v_mul_f64 %[v2], %[v2], %[v]\n\t"
        "v_fma_f64 %[p2], %[v2], %[third], %[p2]\n\t"
        "s_add_i32 %[l], %[l], 1\n\t"
        "s_mov_b32 %[ph], 2\n\t"
        "s_cmp_ge_i32 %[l], %[nj]\n\t"
        "s_cbranch_scc1 9f\n"
        "3:\n\t"
        "v_readlane_b32 vcc_lo, %[elo], %[l]\n\t"
        "v_readlane_b32 vcc_hi, %[ehi], %[l]\n\t"
        "s_add_i32 %[tmp], %[l], 3\n\t"
        "s_min_i32 %[tmp], %[tmp], %[nrmax]\n\t"
        "v_readlane_b32 %[tmp], %[ro], %[tmp]\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_mul_f64 %[v], %[RC], vcc\n\t"
        "v_lshl_add_u32 %[addr], %[tmp], 3, %[lane]\n\t"
        "global_load_dwordx2 %[RC], %[addr], %[base]\n\t"
        "v_mul_f64 %[v2], %[v], %[v]\n\t"
        "v_add_f64 %[p0], %[p0], %[v]\n\t"
        "v_fma_f64 %[p1], %[v2], 0.5, %[p1]\n\t"
        "v_mul_f64 %[v2], %[v2], %[v]\n\t"
        "v_fma_f64 %[p2], %[v2], %[third], %[p2]\n\t"
        "s_add_i32 %[l], %[l], 1\n\t"
        "s_mov_b32 %[ph], 0\n\t"
        "s_cmp_lt_i32 %[l], %[nj]\n\t"
        "s_cbranch_scc1 1b\n"
        "9:\n\t"
        : [l] "+s"(l), [ph] "+s"(ph), [RA] "+v"(RA), [RB] "+v"(RB), [RC] "+v"(RC), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2),
          [tmp] "=&s"(tmp), [v] "=&v"(v), [v2] "=&v"(v2), [addr] "=&v"(addr)
        : [nj] "s"(nj), [nrmax] "s"(nrmax), [elo] "v"(e_lo), [ehi] "v"(e_hi), [ro] "v"(ro), [lane] "v"(lane8), [base] "s"(Rb), [third] "s"(third)
        : "vcc", "scc", "memory");
}
__device__ __forceinline__ void rag_walk_settle_global(double &RA, double &RB, double &RC) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(RA), "+v"(RB), "+v"(RC) : : "memory");
}

struct PrepView {
    const ScratchEnt *arena;
    const int64_t *blob_prefix;   // the slot's exclusive prefix, indexed by absolute group
    int64_t prefix_base;          // prefix of the launch range's first group
    int64_t grp_base;             // absolute index of the launch range's first group
    int *status;
};

// (groups of 8 with the table in LDS: 168 registers, so that one workgroup of twelve waves -- one R slice, twelve rings -- gives
// three waves per SIMD; the 16-test-site form needs 243 and runs two)
template <int J, bool USE_LDS>
__global__ __launch_bounds__((J == 8 && USE_LDS) ? SCAN_THREADS_J8 : SCAN_THREADS_MAX) void clr_scan_prepared_kernel(ScanParams P, PrepView V) {
#ifdef BMX_PROFILE
    // sections: 0 sites between the test sites, 1 zone header, 2 pair list, 3 quad list, 4 generic walks past the zones, 5 fold of the
    // moments, 6 series entries, 7 ragged end + Horner, 8 exp + apply, 9 renormalise + best-tracking, 10 group set-up, 11 winners out
    long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) double lds_R[];  // [rows][64] when USE_LDS, then per wave: ring + scratch, sites between the test sites
    // farg[j] as an LLVM vector: the ragged-end walk indexes it with a wave-uniform j, which the compiler turns into GPR indexing
    // (s_set_gpr_idx_on) -- a C array indexed dynamically would live in scratch memory
    typedef double FargVec __attribute__((ext_vector_type(J)));
    constexpr int SP = WAVE / J;
    constexpr int BS = J >= 16 ? 4 : 8;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#if BMX_XCD_MAP
    // The slices of one chunk of test sites read the SAME blobs: put them on one XCD (workgroups go to the XCDs round-robin,
    // b % 8), next to each other in its dispatch order, so that one of them brings a blob into the XCD's L2 and the others hit.
    // (The host pads the number of chunks to a multiple of 8; padding chunks have no groups.)
    const int xcd = blockIdx.x & 7;
    const int64_t q = blockIdx.x >> 3;
    const int slice = (int)(q % P.nslices);
    const int64_t chunk = (q / P.nslices) * 8 + xcd;
#else
    const int slice = blockIdx.x % P.nslices;
    const int64_t chunk = blockIdx.x / P.nslices;
#endif
    const int p = slice * WAVE + lane;
    const int jl = lane % J, sl = lane / J;
    const int N = (int)P.N;
    const int nwv = blockDim.x / WAVE;

    if (USE_LDS) {
        const int total = P.rows * WAVE;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x)
            lds_R[idx] = P.Rt[(size_t)(idx >> 6) * P.NP + slice * WAVE + (idx & 63)];
    }
    const char *Rb = reinterpret_cast<const char *>(P.Rt + slice * WAVE);
    const unsigned lane8 = (unsigned)lane * 8u;
    const int rowmul = USE_LDS ? WAVE : P.NP;
    auto loadR = [&](int rowoff) -> double {
        return USE_LDS ? lds_R[rowoff + lane] : *reinterpret_cast<const double *>(Rb + ((unsigned)rowoff * 8u + lane8));
    };
    // LDS byte address of this lane's column of the R slice (for the hand-written ragged-end walk)
    const unsigned lds_lane_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)(lds_R + lane);
    double *lds_tail = lds_R + (USE_LDS ? P.rows * WAVE : 0);
    constexpr int WAVE_UNITS = RING_UNITS + RING_MIRROR + AUX_UNITS;
    ScratchEnt *ring = reinterpret_cast<ScratchEnt *>(lds_tail) + wave * WAVE_UNITS;
    ScratchEnt *scr = ring + RING_UNITS + RING_MIRROR;              // 32 units of wave-private scratch
    double *scr_d = reinterpret_cast<double *>(scr);
    double *mid_base = lds_tail + nwv * WAVE_UNITS * 2;
    // the ring and the scratch start out as valid neutral entries (row offset 0): whatever is read ahead of the stream is a
    // legal row reference
    for (int idx = lane; idx < WAVE_UNITS; idx += WAVE) ring[idx] = ScratchEnt{0.0, 0, 0};
    if (USE_LDS) __syncthreads(); else __builtin_amdgcn_wave_barrier();

    const int64_t ngroups = (P.M + J - 1) / J;
    const int64_t gpb = P.sites_per_block / J;
    const int64_t g_end = min((chunk + 1) * gpb, ngroups);
    for (int64_t grp = chunk * gpb + wave; grp < g_end; grp += nwv) {
        const int64_t tb = grp * J;
        const int nvalid = (int)min((int64_t)J, P.M - tb);
        int jl_g = jl, lane_g = lane, wave_g = wave;
        asm volatile("" : "+v"(jl_g), "+v"(lane_g), "+s"(wave_g));
        double *mid_g = mid_base + wave_g * (MID_CAP + MID_CAP / 2);
        int *mid_ro = reinterpret_cast<int *>(mid_g + MID_CAP);
        const int jj = min(jl_g, nvalid - 1);
        const double tj = P.test_gen[tb + jj];
        int lo_j = (int)max(P.win_lo[tb + jj], (int64_t)0);
        int hi_j = (int)min(P.win_hi[tb + jj], (int64_t)N - 1);
        if (jl >= nvalid) { lo_j = 1; hi_j = 0; }
        const double t0 = readlane_f64(tj, 0), tL = readlane_f64(tj, J - 1);
        const int c0 = (int)P.center[tb], cU = (int)P.center_hi[tb + nvalid - 1];
        int lo_max = 0, hi_min = N - 1;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int a = __builtin_amdgcn_readlane(lo_j, j), b = __builtin_amdgcn_readlane(hi_j, j);
            if (j < nvalid) { lo_max = max(lo_max, a); hi_min = min(hi_min, b); }
        }
        int L_int = min(c0, hi_min + 1), R_int = max(cU, lo_max);
        if (hi_min < lo_max) { L_int = c0; R_int = c0; hi_min = -1; lo_max = N; }   // no bulk zone
        const bool staged = R_int - L_int <= MID_CAP;
        if (staged && lane_g < R_int - L_int) {
            mid_g[lane_g] = P.genpos[L_int + lane_g];
            mid_ro[lane_g] = (int)P.row[L_int + lane_g] * rowmul;
        }
        __builtin_amdgcn_wave_barrier();
        bool mid_tri = false;
        if (BMX_MIDTRI) {
            const bool same = staged && (R_int - L_int == J) && nvalid == J && lo_max <= L_int && hi_min >= R_int - 1;
            if (same) {
                const double gm = mid_g[jl_g];
                const double tprev = __shfl_up(tj, 1);
                mid_tri = __ballot(!(gm == tj && (jl_g == 0 || tj > tprev))) == 0ull;
            }
        }

        // the group's blob, streamed through the ring: `pos` units consumed, [.., staged_u) in the ring, the next 64 in flight
        // (the unit in flight is held as two doubles -- bit patterns, never arithmetic -- so that it stays in registers)
        const double2 *src = reinterpret_cast<const double2 *>(V.arena + (V.blob_prefix[V.grp_base + grp] - V.prefix_base));
        double2 *ring2 = reinterpret_cast<double2 *>(ring);
        int pos = 0, staged_u = 0;
        double nx_a, nx_b;
        {
            const double2 t = src[lane];
            nx_a = t.x; nx_b = t.y;
        }
        bool bad = false;
        auto stage = [&]() {
            const int slot = (staged_u >> 6) & 3;
            ring2[slot * WAVE + lane] = double2{nx_a, nx_b};
            if (slot == 0 && lane < RING_MIRROR) ring2[RING_UNITS + lane] = double2{nx_a, nx_b};   // units 0..15 again behind the ring's
            staged_u += WAVE;                                                       // end: any 16 consecutive units read without a wrap
            const double2 t = src[staged_u + lane];
            nx_a = t.x; nx_b = t.y;
            __builtin_amdgcn_wave_barrier();
        };
        // [pos, pos + 80) is in the ring.  `pos` moves by at most 64 units between two calls (a block, an entry, a guard, a
        // ragged end of < 64 sites), so one step restores the invariant
        auto need = [&]() {
            if (pos + 80 > staged_u) stage();
        };
        stage();
        stage();
        PROF_MARK(10);

        double acc[J], bestM[J];
        int E[J], bestK[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {        // the product 1 (T = 0: only a larger one wins, v1:451,501) in the convention of the renormalisation
            bestM[j] = BMX_FREXP ? 0.5 : 1.0;
            bestK[j] = ((131072 + (BMX_FREXP ? 1 : 0)) << 13) | 8191;
        }

        for (int iA = 0; iA < P.nA; ++iA) {
            const double A = P.A[iA];
            int bits = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) { acc[j] = 1.0; E[j] = 0; }
            auto renorm_all = [&]() {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    if (BMX_FREXP) renorm_fx(acc[j], E[j]); else renorm(acc[j], E[j]);
                }
                bits = 0;
            };
            auto spend = [&](int nbits) {
                if (bits + nbits > 1000) renorm_all();
                bits += nbits;
            };
            const int span_generic = max(P.span_hi, 54);

            auto apply_pass = [&](double alpha, int rowoff, unsigned long long m_in, double *buf) {
                buf[lane] = alpha;
                __builtin_amdgcn_wave_barrier();
                double Rpre[SP];
                if (!USE_LDS) {
#pragma unroll
                    for (int s = 0; s < SP; ++s) Rpre[s] = loadR(__builtin_amdgcn_readlane(rowoff, s * J));
                }
#pragma unroll
                for (int s = 0; s < SP; ++s) {
                    if (((m_in >> (s * J)) & ((1ull << J) - 1ull)) == 0ull) continue;
                    const double R = USE_LDS ? loadR(__builtin_amdgcn_readlane(rowoff, s * J)) : Rpre[s];
                    const double2 *a2 = reinterpret_cast<const double2 *>(buf + s * J);
#pragma unroll
                    for (int j = 0; j < J; j += 2) {
                        const double2 a = a2[j >> 1];
                        acc[j] *= fma(a.x, R, 1.0);
                        acc[j + 1] *= fma(a.y, R, 1.0);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            };
            auto generic_core = [&](int i, bool inr, double g, int rowoff, int dir) -> bool {
                const bool inwin = inr && i >= lo_j && i <= hi_j;
                const double z = A * fabs(g - tj);
                const bool in = inwin && (z <= P.zcut) && (g != tj);
                const bool fin = lo_j > hi_j ||
                                 (dir > 0 ? (!inr || i > hi_j || (i >= lo_j && g > tj && z > P.zcut))
                                          : (!inr || i < lo_j || (i <= hi_j && g < tj && z > P.zcut)));
                const unsigned long long m_in = __ballot(in);
                if (m_in != 0ull) {
                    const double alpha = in ? exp_neg(z) : 0.0;
                    spend(SP * (__ballot(in && z < 0.6931471805599453) == 0ull ? max(P.span_hi, 2) : span_generic));
                    apply_pass(alpha, rowoff, m_in, scr_d);
                }
                return __ballot(fin) == ~0ull;
            };
            auto generic_pass = [&](int b, int dir, int lim, bool from_lds) -> bool {
                const int i = b + dir * sl;
                const bool inr = dir > 0 ? (i < lim) : (i > lim);
                double g = 0.0;
                int rowoff = 0;
                if (from_lds) {
                    if (inr) { g = mid_g[i - L_int]; rowoff = mid_ro[i - L_int]; }
                } else {
                    if (inr) { g = P.genpos[i]; rowoff = (int)P.row[i] * rowmul; }
                }
                return generic_core(i, inr, g, rowoff, dir);
            };

            // One zone of the blob, first half: header and near-list products; returns where the generic walk goes on.  What
            // the far half needs later (after BOTH near lists) comes back through the references; fv: lane j holds F_j.
            auto zone_near = [&](auto dirk, double tnear, double &fv_o, int &nocc_o, int &nfar_o, int &nrmax_o, int &rag_o, int &nser_o,
                                 FargVec &farg) -> int {
                constexpr int dirc = decltype(dirk)::value;       // compile-time: farg[j] and F[j] of the ragged-end walk are registers
                int nragv_o = 0;
                nocc_o = 0; nfar_o = 0; nrmax_o = 0; rag_o = 0; nser_o = 0;
                fv_o = 0.0;
                if (bad) return PREP_ZONE_DONE;
                double F[J];
                {
                    const double fv = exp_neg(A * fabs(tnear - tj));
                    fv_o = fv;
#pragma unroll
                    for (int j = 0; j < J; ++j) F[j] = readlane_f64(fv, j);
                }
                need();
                const int4 *hp = reinterpret_cast<const int4 *>(ring + (pos & (RING_UNITS - 1)));
                const int4 h0 = hp[0], h1 = hp[1];
                const int magic = __builtin_amdgcn_readfirstlane(h0.x);
                const int n_pair = __builtin_amdgcn_readfirstlane(h0.y), n_quad = __builtin_amdgcn_readfirstlane(h0.z);
                const int n_occ = __builtin_amdgcn_readfirstlane(h0.w);
                const int nfar_tot = __builtin_amdgcn_readfirstlane(h1.x), nrmax = __builtin_amdgcn_readfirstlane(h1.y);
                const int base_end = __builtin_amdgcn_readfirstlane(h1.z), ser_w = __builtin_amdgcn_readfirstlane(h1.w);
                const int n_ser = ser_w & 0xff, c8 = (ser_w >> 8) & 0xff, c5 = (ser_w >> 16) & 0xff, c3 = (ser_w >> 24) & 0xff;
                const int rag = magic & 3;
                if ((magic & ~3) != PREP_MAGIC || rag == 2 || n_pair < 0 || n_quad < 0 || n_pair > N + 8 || n_quad > N + 8 || n_occ < 0 || n_occ > MOM_SLOTS ||
                    nrmax < 0 || nrmax >= WAVE || n_ser > SER_CAP || (n_ser & 3) || c8 > c5 || c5 > c3 || c3 > n_ser) {
                    bad = true;
                    return PREP_ZONE_DONE;
                }
                // n_j of this lane's test site (ragged end)
                nragv_o = (int)reinterpret_cast<const unsigned char *>(hp + 2)[jl];
                nocc_o = n_occ; nfar_o = nfar_tot; nrmax_o = nrmax; rag_o = rag; nser_o = ser_w;
                pos += PREP_HDR;
                PROF_MARK(1);

                if (n_pair > 0) {
                    need();
                    const double e0 = ring[pos & (RING_UNITS - 1)].e;       // the nearest site: the largest alpha of the list
                    const double om = 1.0 - e0, op = fma(e0, P.rmax, 1.0);
                    const int pend_lo = 1024 - ((__builtin_amdgcn_readfirstlane(__double2hiint(om)) >> 20) & 0x7ff);
                    const int pend_hi = ((__builtin_amdgcn_readfirstlane(__double2hiint(op)) >> 20) & 0x7ff) - 1022;
                    const int span8 = 8 * min(max(pend_hi, pend_lo), 125);
                    double Rp[BS], ep[BS];
                    if (BMX_PPAIR_PREFETCH) {
                        const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
#pragma unroll
                        for (int u = 0; u < BS; ++u) {
                            const ScratchEnt en = rp[u];
                            ep[u] = en.e;
                            Rp[u] = loadR(en.ro);
                        }
                    }
                    for (int l0 = 0; l0 < n_pair; l0 += BS) {
                        need();
                        const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
                        spend(span8 * BS / 8);
                        double v[BS];
                        if (BMX_PPAIR_PREFETCH) {
#pragma unroll
                            for (int u = 0; u < BS; ++u) v[u] = ep[u] * Rp[u];
#pragma unroll
                            for (int u = 0; u < BS; ++u) {
                                const ScratchEnt en = rp[BS + u];
                                ep[u] = en.e;
                                Rp[u] = loadR(en.ro);
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < BS; ++u) {
                                const ScratchEnt en = rp[u];
                                v[u] = en.e * loadR(en.ro);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < BS; u += 2) {
                            const double sv = v[u] + v[u + 1], qv = v[u] * v[u + 1];
#pragma unroll
                            for (int j = 0; j < J; ++j) acc[j] *= fma(F[j], fma(F[j], qv, sv), 1.0);
                        }
                        pos += BS;
                    }
                }
                PROF_MARK(2);
                if (n_quad > 0) {
                    need();
                    int span8q = 0;
                    double Rn[4], en_e[4];
                    {
                        const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const ScratchEnt en = rp[u];
                            en_e[u] = en.e;
                            Rn[u] = loadR(en.ro);
                        }
                    }
                    for (int l0 = 0; l0 < n_quad; l0 += 4) {
                        need();
                        const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
                        if ((l0 & 63) == 0) {
                            // every factor of the next 64 entries lies in [1/2, 1 + e0 Rmax], e0 the first (largest) alpha among them
                            const double op = fma(rp[0].e, P.rmax, 1.0);
                            const int hi = ((__builtin_amdgcn_readfirstlane(__double2hiint(op)) >> 20) & 0x7ff) - 1022;
                            span8q = 8 * min(max(hi, 2), 125);
                        }
                        spend(span8q / 2);
                        double v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = en_e[u] * Rn[u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const ScratchEnt en = rp[4 + u];
                            en_e[u] = en.e;
                            Rn[u] = loadR(en.ro);
                        }
                        const double s01 = v[0] + v[1], q01 = v[0] * v[1];
                        const double s23 = v[2] + v[3], q23 = v[2] * v[3];
                        const double e1 = s01 + s23;
                        const double e2 = fma(s01, s23, q01 + q23);
                        const double e3 = fma(q01, s23, q23 * s01);
                        const double e4 = q01 * q23;
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            double t = fma(F[j], e4, e3);
                            t = fma(F[j], t, e2);
                            t = fma(F[j], t, e1);
                            acc[j] *= fma(F[j], t, 1.0);
                            if ((j & (BMX_QSB - 1)) == BMX_QSB - 1) __builtin_amdgcn_sched_barrier(0);
                        }
                        pos += 4;
                    }
                }
                PROF_MARK(3);
                pos += PREP_GUARD;
                // The ragged end (< 64 entries and a guard) follows the near list in the stream: read at once, lanes over entries; the walk
                // takes an entry out of the registers and has its R loaded ahead.  Test site by test site, in the order in which the
                // windows end: its share of the entries into three sums of their own (third-order Taylor terms: these sites sit at
                // A d ~ 18.4), then the sums as they stand into farg[j].  Done HERE, where only the running products are live -- not
                // between the Horner chains of the far field, where round 3 had it and paid 49 spilled registers for it.
                if (rag) {
                    need();
                    const bool rag_hi = (rag & 2) != 0;
                    const ScratchEnt ragm = ring[(pos + lane) & (RING_UNITS - 1)];
                    double rag_R0 = loadR(__builtin_amdgcn_readlane(ragm.ro, 0));
                    double rag_R1 = loadR(__builtin_amdgcn_readlane(ragm.ro, min(1, nrmax)));
                    double rag_R2 = loadR(__builtin_amdgcn_readlane(ragm.ro, min(2, nrmax)));
                    const int nrag_v = nragv_o;
                    double r1 = 0.0, r2 = 0.0, r3 = 0.0;
                    int l = 0, ph = 0;            // the walk's position and whose turn it is among rag_R0 / rag_R1 / rag_R2
                    if (USE_LDS) {
                        // the walk itself is one asm statement per test site (rag_walk_lds): no control flow inside the unrolled loop
#pragma unroll
                        for (int w = 0; w < J; ++w) {
                            const int j = dirc > 0 ? w : J - 1 - w;
                            rag_walk_lds(l, __builtin_amdgcn_readlane(nrag_v, j), nrmax, __double2loint(ragm.e), __double2hiint(ragm.e), ragm.ro,
                                         lds_lane_addr, rag_R0, rag_R1, rag_R2, r1, r2, r3);
                            const double f = F[j];
                            farg[j] = fma(-f, fma(-f, fma(-f, r3, r2), r1), farg[j]);
                        }
                        rag_walk_settle(rag_R0, rag_R1, rag_R2);
                    } else {
#pragma unroll
                        for (int w = 0; w < J; ++w) {
                            const int j = dirc > 0 ? w : J - 1 - w;
                            rag_walk_global(l, ph, __builtin_amdgcn_readlane(nrag_v, j), nrmax, __double2loint(ragm.e), __double2hiint(ragm.e), ragm.ro,
                                            lane8, Rb, rag_R0, rag_R1, rag_R2, r1, r2, r3);
                            const double f = F[j];
                            farg[j] = fma(-f, fma(-f, fma(-f, r3, r2), r1), farg[j]);
                        }
                        rag_walk_settle_global(rag_R0, rag_R1, rag_R2);
                    }
                    // (rare) sites of the ragged end with alpha max|R| between 3e-4 and 0.03, flagged by the producer: orders 4 to 8 of each, for
                    // the test sites whose windows hold it -- apart from the walk above
                    if (rag_hi) {
                        for (int lh = 0; lh < nrmax; ++lh) {
                            if (__builtin_amdgcn_readlane(ragm.pad, lh) == 0) continue;
                            const double v = readlane_f64(ragm.e, lh) * loadR(__builtin_amdgcn_readlane(ragm.ro, lh));
#pragma unroll
                            for (int j = 0; j < J; ++j) {
                                const double u = (lh < __builtin_amdgcn_readlane(nrag_v, j)) ? F[j] * v : 0.0;
                                const double u2 = u * u;
                                double t = fma(-u, P_W[7], P_W[6]);
                                t = fma(-u, t, P_W[5]);
                                t = fma(-u, t, P_W[4]);
                                t = fma(-u, t, P_W[3]);
                                farg[j] = fma(u2 * u2, t, farg[j]);
                            }
                        }
                    }
                    pos += nrmax + PREP_RAG_GUARD;
                }
                PROF_MARK(7);
                return base_end;
            };

            // ... second half, after both zones' near lists: fold the zone's moments, walk its ragged end, and ADD the log of
            // the factor each test site's product has to pick up to farg (the exp is taken once for both zones)
            auto zone_far = [&](double fv, int n_occ, int ser_w, int nfar_tot, FargVec &farg) {
                const int n_ser = ser_w & 0xff, ser_cls = ser_w >> 8;        // entries; cumulative class counts (8, 5, 3), 8 bits each
                if (bad || !nfar_tot) return;
                double F[J];
#pragma unroll
                for (int j = 0; j < J; ++j) F[j] = readlane_f64(fv, j);
                double pk[P_ORDER];
#pragma unroll
                for (int k = 0; k < P_ORDER; ++k) pk[k] = 0.0;
                // p_k += M_k R^k for one moment entry (q: its units 1.. as (M_2, M_3), (M_4, M_5), ...); powers by halving: depth log2 k
                auto fold = [&](const double m1, const double2 *q, const double R) {
                    double pw[P_ORDER + 1];
                    pw[1] = R;
#pragma unroll
                    for (int k = 2; k <= P_ORDER; ++k) pw[k] = pw[k >> 1] * pw[k - (k >> 1)];
                    pk[0] = fma(m1, R, pk[0]);
#pragma unroll
                    for (int k = 2; k <= P_ORDER; ++k) {
                        const double2 v = q[(k - 2) >> 1];
                        pk[k - 1] = fma((k & 1) ? v.y : v.x, pw[k], pk[k - 1]);
                    }
                };
                // two slots per step.  With the table in L2 / HBM the heads (M_1, row) and the R of the NEXT two are requested before this
                // step's powers (+3.4 / +4.2 % at 11 / 41 sample sizes); with the table in LDS the extra live values cost more than the
                // short latency (-0.7 %), so that form asks for its R where it needs it
                if (BMX_FOLD_EARLY && n_occ > 0) {
                    need();
                    const ScratchEnt *rp0 = ring + (pos & (RING_UNITS - 1));
                    ScratchEnt ua = rp0[0], ub = rp0[n_occ > 1 ? PREP_MOM : 0];
                    double Ra = loadR(ua.ro), Rb2 = loadR(ub.ro);
                    for (int s = 0; s < n_occ; s += 2) {
                        need();
                        const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
                        const bool two = s + 1 < n_occ;
                        const double m1a = ua.e, m1b = ub.e, Rca = Ra, Rcb = Rb2;
                        const int na = s + 2 < n_occ ? 2 * PREP_MOM : 0, nb = s + 3 < n_occ ? 3 * PREP_MOM : na;     // (the last step: itself again)
                        ua = rp[na];
                        ub = rp[nb];
                        Ra = loadR(ua.ro);
                        Rb2 = loadR(ub.ro);
                        fold(m1a, reinterpret_cast<const double2 *>(rp + 1), Rca);
                        if (two) fold(m1b, reinterpret_cast<const double2 *>(rp + PREP_MOM + 1), Rcb);
                        pos += two ? 2 * PREP_MOM : PREP_MOM;
                    }
                }
                if (!BMX_FOLD_EARLY) {
                    for (int s = 0; s < n_occ; s += 2) {
                        need();
                        const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
                        const bool two = s + 1 < n_occ;
                        const ScratchEnt ua = rp[0], ub = rp[two ? PREP_MOM : 0];
                        const double Ra = loadR(ua.ro), Rb2 = loadR(ub.ro);
                        fold(ua.e, reinterpret_cast<const double2 *>(rp + 1), Ra);
                        if (two) fold(ub.e, reinterpret_cast<const double2 *>(rp + PREP_MOM + 1), Rb2);
                        pos += two ? 2 * PREP_MOM : PREP_MOM;
                    }
                }
                PROF_MARK(5);
                // (round 4, measured and dropped: E by an LDS broadcast read one batch ahead instead of two v_readlane per entry: -0.7 %)
                // series entries: p_k += (E R)^k up to the order the entry's class needs.  The wave reads them all at once, lanes over
                // entries; one entry at a time then comes out of the registers (readlane) -- its R is the only load in the loop, four
                // entries ahead.  The producer sorted them by class, highest first: one branch-free loop per class (a lower-class
                // entry that rides along in the last batch of a higher class only gets more terms than it needs).
                if (n_ser > 0) {
                    need();
                    const ScratchEnt mine = ring[(pos + lane) & (RING_UNITS - 1)];
                    const int e8 = (ser_cls & 0x7f) + 3 & ~3, e5 = max(e8, ((ser_cls >> 8) & 0x7f) + 3 & ~3), e3 = max(e5, ((ser_cls >> 16) & 0x7f) + 3 & ~3);
                    double Rn[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) Rn[u] = loadR(__builtin_amdgcn_readlane(mine.ro, u));
                    auto batch = [&](auto kc, int s0) {
                        constexpr int K = decltype(kc)::value;
                        double Rc[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) Rc[u] = Rn[u];
                        const int nx = s0 + 4 < n_ser ? s0 + 4 : s0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) Rn[u] = loadR(__builtin_amdgcn_readlane(mine.ro, nx + u));
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double v = readlane_f64(mine.e, s0 + u) * Rc[u];
                            pk[0] += v;
                            if (K == 2) {
                                pk[1] = fma(v, v, pk[1]);
                            } else {
                                const double v2 = v * v;
                                pk[1] += v2;
                                pk[2] = fma(v2, v, pk[2]);
                                if (K > 3) {
                                    const double v4 = v2 * v2;
                                    pk[3] += v4;
                                    pk[4] = fma(v4, v, pk[4]);
                                    if (K > 5) {
                                        pk[5] = fma(v4, v2, pk[5]);
                                        pk[6] = fma(v4 * v2, v, pk[6]);
                                        pk[7] = fma(v4, v4, pk[7]);
                                    }
                                }
                            }
                        }
                    };
                    int s0 = 0;
                    for (; s0 < e8; s0 += 4) batch(std::integral_constant<int, 8>{}, s0);
                    for (; s0 < e5; s0 += 4) batch(std::integral_constant<int, 5>{}, s0);
                    for (; s0 < e3; s0 += 4) batch(std::integral_constant<int, 3>{}, s0);
                    for (; s0 < n_ser; s0 += 4) batch(std::integral_constant<int, 2>{}, s0);
                    pos += n_ser;
                }
                PROF_MARK(6);
#pragma unroll
                for (int k = 0; k < P_ORDER; ++k) pk[k] *= P_W[k];
                // the Horner chains of the zone's far field, one per test site, in one straight line
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const double f = F[j];
                    double t = pk[P_ORDER - 1];
#pragma unroll
                    for (int k = P_ORDER - 2; k >= 0; --k) t = fma(-f, t, pk[k]);
                    farg[j] = fma(-f, t, farg[j]);              // exp_neg's argument: the factor is exp(f t)
                }
                PROF_MARK(7);
            };

            // sites between / at the test sites (and any part of the windows not covered by bulk)
            PROF_MARK(9);
            if (BMX_MIDTRI && mid_tri && A * (tL - t0) <= P.zcut) {
                const double xk = A * (tj - t0);
                double Gk, Hk;
                exp_neg2(xk, -xk, Gk, Hk);
                const int ro_k = mid_ro[jl];
                spend((J - 1) * span_generic);
                double Fs[J];
#pragma unroll
                for (int j = 0; j < J; ++j) Fs[j] = readlane_f64(Hk, j);
                if (lane < J) scr[lane] = ScratchEnt{Gk, ro_k, 0};
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < J; i += 2) {
                    const ScratchEnt en0 = scr[i], en1 = scr[i + 1];
                    const double v1 = en1.e * loadR(en1.ro);
                    if (i > 0) {
                        const double v0 = en0.e * loadR(en0.ro);
                        const double sv = v0 + v1, qv = v0 * v1;
#pragma unroll
                        for (int j = 0; j < i; ++j) acc[j] *= fma(Fs[j], fma(Fs[j], qv, sv), 1.0);
                    }
                    acc[i] *= fma(Fs[i], v1, 1.0);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < J; ++j) Fs[j] = readlane_f64(Gk, j);
                if (lane < J) scr[lane] = ScratchEnt{Hk, ro_k, 0};
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < J; i += 2) {
                    const ScratchEnt en0 = scr[i], en1 = scr[i + 1];
                    const double v0 = en0.e * loadR(en0.ro);
                    if (i + 2 < J) {
                        const double v1 = en1.e * loadR(en1.ro);
                        const double sv = v0 + v1, qv = v0 * v1;
#pragma unroll
                        for (int j = i + 2; j < J; ++j) acc[j] *= fma(Fs[j], fma(Fs[j], qv, sv), 1.0);
                    }
                    acc[i + 1] *= fma(Fs[i + 1], v0, 1.0);
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                for (int b = L_int; b < R_int; b += SP) generic_pass(b, +1, R_int, staged);
            }
            PROF_MARK(0);
            // right side, left side: near lists and whatever the zones do not cover
            double fvR, fvL;
            int noccR, nfarR, nrmR, nserR, noccL, nfarL, nrmL, nserL;
            int ragR, ragL;
            FargVec farg;               // per test site: minus the log of the factor its product picks up from the far fields (exp_neg's argument)
#pragma unroll
            for (int j = 0; j < J; ++j) farg[j] = 0.0;
            int b = zone_near(std::integral_constant<int, +1>{}, tL, fvR, noccR, nfarR, nrmR, ragR, nserR, farg);
            if (b != PREP_ZONE_DONE) { while (!generic_pass(b, +1, N, false)) b += SP; }
            PROF_MARK(4);
            b = zone_near(std::integral_constant<int, -1>{}, t0, fvL, noccL, nfarL, nrmL, ragL, nserL, farg);
            if (b != PREP_ZONE_DONE) { while (!generic_pass(b, -1, -1, false)) b -= SP; }
            PROF_MARK(4);
            // the far fields of both zones: ONE exp per test site -- test sites in pairs, two interleaved exp chains, each pair
            // final before the next one starts
            if (!bad && (nfarR || ragR || nfarL || ragL)) {
                auto apply_far = [&]() {
#pragma unroll
                    for (int w = 0; w < J; w += 2) {
                        double e0, e1;
                        exp_neg2(farg[w], farg[w + 1], e0, e1);
                        acc[w] *= e0;
                        acc[w + 1] *= e1;
                        asm volatile("" : "+v"(acc[w]), "+v"(acc[w + 1]));
                        farg[w] = 0.0;
                        farg[w + 1] = 0.0;
                    }
                    PROF_MARK(8);
                };
                // exponent budget: a zone's far field moves a product by at most (sites) * far_bits bits (< 860 by P_FAR_CAP); both
                // zones in one exp only while their sum fits, else one after the other with the exponents pulled out in between
                const int bitsR = 2 + (int)((float)(nfarR + nrmR) * P.far_bits), bitsL = 2 + (int)((float)(nfarL + nrmL) * P.far_bits);
                // (one copy of each zone's code: this kernel's per-A path is about as large as the instruction cache)
                const bool split = bitsR + bitsL > 900;
                zone_far(fvR, noccR, nserR, nfarR, farg);
                if (split) {
                    spend(bitsR);
                    apply_far();
                }
                zone_far(fvL, noccL, nserL, nfarL, farg);
                spend(split ? bitsL : bitsR + bitsL);
                apply_far();
            }

            renorm_all();
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int ec = min(max(E[j], -131071), 131071) + 131072;
                const int eb = bestK[j] >> 13;
                const bool better = ((ec > eb) || (ec == eb && acc[j] > bestM[j])) && (!BMX_FREXP || acc[j] > 0.0);
                if (better && p < P.npairs) {
                    bestM[j] = acc[j];
                    bestK[j] = (ec << 13) | iA;
                }
            }
            if (bad) break;
        }
        PROF_MARK(9);
        if (bad && lane == 0) atomicOr(V.status, 2);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int biA = bestK[j] & 8191;
            int bE = bestK[j] >> 13;
            double bM = bestM[j];
            int bL = (biA == 8191 || bad) ? 0x7fffffff : biA * P.npairs + p;
            for (int off = 32; off > 0; off >>= 1) {
                const int oE = __shfl_xor(bE, off);
                const double oM = __shfl_xor(bM, off);
                const int oL = __shfl_xor(bL, off);
                if (oE > bE || (oE == bE && (oM > bM || (oM == bM && oL < bL)))) { bE = oE; bM = oM; bL = oL; }
            }
            if (lane == 0 && j < nvalid) {
                const size_t o = (size_t)slice * P.M + (tb + j);
                P.part_T[o] = bM;
                P.part_lin[o] = bL;
                P.part_ns[o] = bE;
            }
        }
        PROF_MARK(11);
    }
#ifdef BMX_PROFILE
    if (lane == 0 && P.prof)
        for (int k = 0; k < 12; ++k) atomicAdd(P.prof + k, (unsigned long long)prof_[k]);
#endif
}

// ----------------------------------------------------------------------------- K2, prepared, one test site per wave
// Sparse test sets (the reference's -s with a large step), unsorted test positions: windows share too little for groups.
// The same split as above with J = 1: prep_solo_kernel walks each test site's window once per A -- both sides; with one
// test site the decay separates trivially (alpha_i = E_i, F = 1), so the two sides share the near lists and ONE set of
// moments -- and clr_scan_solo_kernel, one wave per (test site, slice), multiplies the near lists, folds the moments and takes
// one exp per A.  The round-2 per-site kernel did the walk (loads, exp, window test) in all eight slice waves and multiplied
// every site of the window.
//
// Round 3 read every entry back from the LDS ring with a broadcast ds_read_b128 next to the ds_read_b64 of its R: 9 LDS clocks per
// entry for each of 16 waves per CU -- the LDS pipe, not the vector unit, was the bound (8.7 cycles per VALU instruction).  Round 4:
// the sites of the most frequent row (substitutions: 70 % of the sites) need no R read at all -- that row's R sits in a register and
// their entries are the 8-byte E alone, two per broadcast read: 0.5 LDS instructions per entry instead of 2 -- and the slices of a
// chunk of test sites run on one XCD, so that a stream comes out of HBM once, not eight times.
// (Measured and dropped in round 4: the whole stream through the scalar cache -- s_load_dwordx16 from the constant address space,
// entries as scalar operands, no ring at all: 0.80 M windows/s against round 3's 1.35 M.  A wave gets one 64-byte line per ~2200
// cycles that way: the scalar cache is built for a few hot constants, not for 88 KB of stream per wave.)
//   blob(test site) = for iA:  one or more SEGMENTS, then the far field
//     segment   = header, 64 B: {magic | last, n0, n1, n_occ} {n_far, 0, 0, 0} ...
//                 n0 (a multiple of 8) entries of the most frequent row: E (8 B each)
//                 n1 (a multiple of 4) entries of the other rows: (E, row offset, 0) (16 B each); all 64-byte aligned;
//                 lists padded with neutral entries (E = 0); a segment is flushed from prep_solo_kernel's LDS staging lists
//                 whenever they fill, so any window size goes through fixed-size staging
//     far field = n_occ moment entries of 5 units: (c_1 .. c_8) (row offset), c_k = -+ w_k M_k, the series' coefficient already
//                 in (with F = 1 the far field of a row is one polynomial in R); padded to a multiple of 4 units
constexpr int SOLO_MAGIC = 0x50100000;
constexpr int SOLO_S0 = 256, SOLO_S1 = 192;        // staging lists per wave: E of the most frequent row / (E, row offset) of the others
// Far field of the solo kernels: order 8 on |x| <= 0.05 (the round-2 series).  The per-row thresholds on the device are
// those of the grouped kernels (P_EPS): a site is far here S_SHIFT = log(P_EPS / S_EPS) later.
#ifndef BMX_S_ORDER
#define BMX_S_ORDER 8
#endif
#if BMX_S_ORDER == 8
constexpr int S_ORDER = 8, S_COPIES = 8, S_MOM = 1 + S_ORDER / 2, S_FAR_CAP = 8192;
// (S_EPS = 0.05)
constexpr double S_SHIFT = P_ORDER == 16 ? 1.6095 : P_ORDER == 12 ? 1.0987 : 0.0;
__device__ constexpr double S_W[8] = {0.9999999999998467, 0.4999999999996164, 0.3333333341509627, 0.25000000122701976,
                                      0.19999882322703955, 0.16666529319200146, 0.1434841013671569, 0.12562715480255862};
__device__ constexpr double S_D[6] = {7.94, 5.13, 3.462, 2.357, 1.571, 0.985};
#else
// the prepared group kernels' series (order P_ORDER on |x| <= P_EPS): fewer near entries, longer Horner chains per occupied row
constexpr int S_ORDER = P_ORDER, S_COPIES = P_COPIES, S_MOM = 1 + S_ORDER / 2, S_FAR_CAP = P_FAR_CAP;
constexpr double S_SHIFT = 0.0;
#define S_W P_W
#define S_D P_D
#endif

template <bool FILL>
__global__ __launch_bounds__(PREP_THREADS) void prep_solo_kernel(PrepParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds_p[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x / WAVE;
    const int N = (int)P.N;
    const int thr_len = P.thr_in_lds ? ((P.rows + 1) & ~1) : 0;
    if (P.thr_in_lds) {
        for (int idx = threadIdx.x; idx < P.rows; idx += blockDim.x) lds_p[idx] = P.rowthr[idx];
        __syncthreads();
    }
    const int mom_len = (P.mom_slots + S_COPIES - 1 + 3) * S_ORDER;
    double *mom = lds_p + thr_len + wave * mom_len;
    // staging lists of the near entries (fill pass): the most frequent row's E, then the other rows' (E, row offset)
    double *st0 = lds_p + thr_len + nw * mom_len + wave * (SOLO_S0 + 2 * SOLO_S1);
    ScratchEnt *st1 = reinterpret_cast<ScratchEnt *>(st0 + SOLO_S0);
    for (int idx = lane; idx < mom_len; idx += WAVE) mom[idx] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int64_t t = P.g_begin + (int64_t)blockIdx.x * nw + wave;       // "group" = one test site
    if (t >= P.g_end) return;
    auto thr_of = [&](int r) -> double {
        double v;
        if (P.thr_in_lds) v = lds_p[r]; else v = P.rowthr[r];
        return v;
    };
    auto rank = [&](unsigned long long m) {
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    };
    const double tg = P.test_gen[t];
    const int lo = (int)max(P.win_lo[t], (int64_t)0);
    const int hi = (int)min(P.win_hi[t], (int64_t)N - 1);
    const int c = (int)P.center[t], ch = (int)P.center_hi[t];     // sites in [c, ch) sit AT the test position: never in its window (v1:455)
    const int row0 = P.row_of_slot[0];                            // the most frequent row of the data (its R stays in a register of the consumer)
    int wpos = 0;
    ScratchEnt *out = nullptr;
    if (FILL) out = P.arena + (P.blob_prefix[t] - P.prefix_base);

    for (int iA = 0; iA < P.nA; ++iA) {
        const double A = P.A[iA];
        const int kmom = min((int)P.kmom[iA], P.mom_slots);
        int n0 = 0, n1 = 0, nfar_tot = 0, pad_ro = 0;       // n0 / n1: entries staged for the next segment
        bool seen = false;
        double m1p = 0.0, m2p = 0.0;
        // one segment of the stream from the staging lists: header, the E of the most frequent row (padded to 8), the other rows'
        // entries (padded to 4)
        auto flush = [&](bool last, int n_occ) {
            const int n0p = (n0 + 7) & ~7, n1p = (n1 + 3) & ~3;
            if (FILL) {
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    int4 *h = reinterpret_cast<int4 *>(out + wpos);
                    h[0] = int4{SOLO_MAGIC | (last ? 1 : 0), n0p, n1p, n_occ};
                    h[1] = int4{nfar_tot, 0, 0, 0};
                    h[2] = int4{0, 0, 0, 0};
                    h[3] = int4{0, 0, 0, 0};
                }
                double *o0 = reinterpret_cast<double *>(out + wpos + 4);
                for (int i = lane; i < n0p; i += WAVE) o0[i] = i < n0 ? st0[i] : 0.0;
                ScratchEnt *o1 = out + wpos + 4 + n0p / 2;
                for (int i = lane; i < n1p; i += WAVE) o1[i] = i < n1 ? st1[i] : ScratchEnt{0.0, pad_ro, 0};
                __builtin_amdgcn_wave_barrier();
            }
            wpos += 4 + n0p / 2 + n1p;
            n0 = 0;
            n1 = 0;
        };
        for (int dir = 0; dir < 2; ++dir) {
            // dir 0: indices max(ch, lo), +1, ... up to hi;  dir 1: min(c - 1, hi), -1, ... down to lo
            const int base = dir == 0 ? max(ch, lo) : min(c - 1, hi);
            int i = dir == 0 ? base + lane : base - lane;
            double g_nx = P.genpos[min(max(i, 0), N - 1)];
            int r_nx = (int)P.row[min(max(i, 0), N - 1)];
            while (true) {
                const bool valid = (i >= lo) && (i <= hi);
                const double g = g_nx;
                const int rraw = r_nx;
                const int inx = dir == 0 ? i + WAVE : i - WAVE;
                g_nx = P.genpos[min(max(inx, 0), N - 1)];
                r_nx = (int)P.row[min(max(inx, 0), N - 1)];
                const double zn = A * fabs(g - tg);
                const bool in = valid && (zn <= P.zcut) && (g != tg);
                const unsigned long long m_in = __ballot(in);
                if (m_in != 0ull) {
                    const double th0 = thr_of(rraw);
                    const int slot = __double2loint(th0) & 0xff;
                    const double th = th0 + S_SHIFT;                    // (NaN stays NaN: rows absent from the helper file are never far)
                    const bool moml = in && slot < kmom && zn >= th && nfar_tot < S_FAR_CAP;
                    const bool nearl = in && !moml;
                    const bool near0 = nearl && rraw == row0;
                    const unsigned long long mm = __ballot(moml), mn0 = __ballot(near0), mn1 = __ballot(nearl && !near0);
                    const int nfar = __popcll(mm);
                    if (!seen) { pad_ro = __builtin_amdgcn_readlane(rraw, __ffsll((long long)m_in) - 1) * P.rowmul; seen = true; }
                    double Ev = 0.0;
                    if (FILL) Ev = in ? exp_neg(zn) : 0.0;
                    if (nfar) {
                        if (moml) {
                            double *mr = mom + (slot ? slot + S_COPIES - 1 : (lane & (S_COPIES - 1))) * S_ORDER;
                            if (FILL) {
                                const double d = zn - th;
                                const double E2 = Ev * Ev;
                                if (slot == 0) { m1p += Ev; m2p += E2; }
                                else { atomicAdd(mr, Ev); atomicAdd(mr + 1, E2); }
                                double Ek = E2;
#pragma unroll
                                for (int k = 3; k <= S_ORDER; ++k) {
                                    if (!(d < S_D[k - 3])) break;
                                    Ek *= Ev;
                                    atomicAdd(mr + (k - 1), Ek);
                                }
                            } else {
                                mom[(slot ? slot + S_COPIES - 1 : 0) * S_ORDER] = 1.0;
                            }
                        }
                        nfar_tot += nfar;
                    }
                    // a pass adds at most 64 entries to either list: a segment goes out before one could overflow
                    if (n0 + WAVE > SOLO_S0 || n1 + WAVE > SOLO_S1) flush(false, 0);
                    if (FILL) {
                        if (near0) st0[n0 + rank(mn0)] = Ev;
                        if (nearl && !near0) st1[n1 + rank(mn1)] = ScratchEnt{Ev, rraw * P.rowmul, 0};
                    }
                    n0 += __popcll(mn0);
                    n1 += __popcll(mn1);
                }
                // the walk ends where the window does: its index bound, or the first site past the cut-off (positions are sorted)
                if (__ballot(valid && zn > P.zcut) != 0ull || __ballot(valid) != ~0ull) break;
                i = inx;
            }
        }
        // the moments of the occupied slots, premultiplied by the series' coefficients; they follow the last segment
        int n_occ = 0;
        bool occ0 = false;
        double x0 = 0.0;
        if (nfar_tot) {
            __builtin_amdgcn_wave_barrier();
            if (lane < S_COPIES * S_ORDER) {
                x0 = mom[lane];
                mom[lane] = 0.0;
            }
            if (FILL) {
                double y1 = m1p, y2 = m2p;
#pragma unroll
                for (int off = 1; off < WAVE; off <<= 1) {
                    y1 += __shfl_xor(y1, off);
                    y2 += __shfl_xor(y2, off);
                }
                x0 += lane == 0 ? y1 : lane == 1 ? y2 : 0.0;
            }
#pragma unroll
            for (int cc = S_COPIES / 2; cc >= 1; cc >>= 1) x0 += __shfl_down(x0, cc * S_ORDER);
            occ0 = readlane_f64(x0, 0) != 0.0;
            n_occ = occ0 ? 1 : 0;
            // count the other occupied slots first: the last segment's header carries n_occ
            for (int s0 = 1; s0 < kmom; s0 += WAVE) {
                const int sl = s0 + lane;
                const bool occ = sl < kmom && mom[(sl + S_COPIES - 1) * S_ORDER] != 0.0;
                n_occ += __popcll(__ballot(occ));
            }
        }
        flush(true, n_occ);
        if (nfar_tot) {
            int at = 0;
            if (occ0) {
                if (FILL) {
                    double m[S_ORDER];
#pragma unroll
                    for (int k = 0; k < S_ORDER; ++k) m[k] = readlane_f64(x0, k) * ((k & 1) ? -S_W[k] : S_W[k]);     // c_k = -+ w_k M_k
                    if (lane == 0) {
                        double2 *o2 = reinterpret_cast<double2 *>(out + wpos);
#pragma unroll
                        for (int q = 0; q < S_ORDER / 2; ++q) o2[q] = double2{m[2 * q], m[2 * q + 1]};
                        *reinterpret_cast<int4 *>(out + wpos + S_ORDER / 2) = int4{row0 * P.rowmul, 0, 0, 0};
                    }
                }
                at = 1;
            }
            for (int s0 = 1; s0 < kmom; s0 += WAVE) {
                const int sl = s0 + lane;
                double m[S_ORDER];
#pragma unroll
                for (int k = 0; k < S_ORDER; ++k) m[k] = 0.0;
                if (sl < kmom) {
                    double *ms = mom + (sl + S_COPIES - 1) * S_ORDER;
#pragma unroll
                    for (int k = 0; k < S_ORDER; ++k) { m[k] = ms[k]; ms[k] = 0.0; }
                }
                const bool occ = m[0] != 0.0;
                const unsigned long long mo = __ballot(occ);
                if (FILL && occ) {
#pragma unroll
                    for (int k = 0; k < S_ORDER; ++k) m[k] *= (k & 1) ? -S_W[k] : S_W[k];
                    ScratchEnt *o = out + wpos + S_MOM * (at + rank(mo));
                    double2 *o2 = reinterpret_cast<double2 *>(o);
#pragma unroll
                    for (int q = 0; q < S_ORDER / 2; ++q) o2[q] = double2{m[2 * q], m[2 * q + 1]};
                    *reinterpret_cast<int4 *>(o + S_ORDER / 2) = int4{P.row_of_slot[sl] * P.rowmul, 0, 0, 0};
                }
                at += __popcll(mo);
            }
            __builtin_amdgcn_wave_barrier();
            // (at == n_occ: both loops apply the same test to the same LDS values)
            const int far_units = (S_MOM * n_occ + 3) & ~3;            // the next header on a 64-byte boundary
            if (FILL && lane < far_units - S_MOM * n_occ) out[wpos + S_MOM * n_occ + lane] = ScratchEnt{0.0, 0, 0};
            wpos += far_units;
        }
    }
    const int units = (wpos + 3) & ~3;
    if (!FILL) {
        if (lane == 0) P.blob_units[t] = units;
    } else {
        if (lane == 0 && (int64_t)units != P.blob_prefix[t + 1] - P.blob_prefix[t]) atomicOr(P.status, 1);
    }
}

template <bool USE_LDS>
__global__ __launch_bounds__(SITE_THREADS) void clr_scan_solo_kernel(ScanParams P, PrepView V) {
    extern __shared__ __attribute__((aligned(16))) double lds_R[];  // [rows][64] when USE_LDS, then one ring per wave
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x / WAVE;
#if BMX_SOLO_XCD_MAP
    // the slices of one chunk of test sites on one XCD, next to each other in its dispatch order (see the prepared kernel): every
    // slice reads all of the chunk's streams, and with this placement seven of the eight reads hit that XCD's L2
    const int xcd = blockIdx.x & 7;
    const int64_t q = blockIdx.x >> 3;
    const int slice = (int)(q % P.nslices);
    const int64_t chunk = (q / P.nslices) * 8 + xcd;
#else
    const int slice = blockIdx.x % P.nslices;
    const int64_t chunk = blockIdx.x / P.nslices;
#endif
    const int p = slice * WAVE + lane;
    if (USE_LDS) {
        const int total = P.rows * WAVE;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x)
            lds_R[idx] = P.Rt[(size_t)(idx >> 6) * P.NP + slice * WAVE + (idx & 63)];
    }
    const char *Rb = reinterpret_cast<const char *>(P.Rt + slice * WAVE);
    const unsigned lane8 = (unsigned)lane * 8u;
    auto loadR = [&](int rowoff) -> double {
        return USE_LDS ? lds_R[rowoff + lane] : *reinterpret_cast<const double *>(Rb + ((unsigned)rowoff * 8u + lane8));
    };
    constexpr int WAVE_UNITS = RING_UNITS + RING_MIRROR;
    ScratchEnt *ring = reinterpret_cast<ScratchEnt *>(lds_R + (USE_LDS ? P.rows * WAVE : 0)) + wave * WAVE_UNITS;
    double2 *ring2 = reinterpret_cast<double2 *>(ring);
    for (int idx = lane; idx < WAVE_UNITS; idx += WAVE) ring[idx] = ScratchEnt{0.0, 0, 0};
    if (USE_LDS) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    const int rowmul = USE_LDS ? WAVE : P.NP;
    const double R0 = P.row0 >= 0 ? loadR(P.row0 * rowmul) : 0.0;       // the most frequent row of the data: its entries carry no row offset
    // every factor 1 + E R lies within 2^-54 .. 2^span_hi (or is exactly 0): P.renorm_every of them between two exponent extractions
    const int lim = P.renorm_every;

    const int64_t t_begin = chunk * P.sites_per_block;
    const int64_t t_end = min(t_begin + (int64_t)P.sites_per_block, P.M);
    for (int64_t t = t_begin + wave; t < t_end; t += nw) {
        // the stream of this test site through the ring, as in the prepared kernel
        const double2 *src = reinterpret_cast<const double2 *>(V.arena + (V.blob_prefix[V.grp_base + t] - V.prefix_base));
        int pos = 0, staged_u = 0;
        double nx_a, nx_b;
        {
            const double2 q0 = src[lane];
            nx_a = q0.x; nx_b = q0.y;
        }
        auto stage = [&]() {
            const int slot = (staged_u >> 6) & 3;
            ring2[slot * WAVE + lane] = double2{nx_a, nx_b};
            if (slot == 0 && lane < RING_MIRROR) ring2[RING_UNITS + lane] = double2{nx_a, nx_b};
            staged_u += WAVE;
            const double2 q0 = src[staged_u + lane];
            nx_a = q0.x; nx_b = q0.y;
            __builtin_amdgcn_wave_barrier();
        };
        auto need = [&]() {                   // [pos, pos + 80) is in the ring
            if (pos + 80 > staged_u) stage();
        };
        stage();
        stage();
        bool bad = false;
        double bestM = 0.5;                         // the product 1 = 0.5 * 2^1 in renorm_fx's convention: only a larger one wins (v1:451,501)
        int bestEc = 131072 + 1, bestA = -1;
        for (int iA = 0; iA < P.nA && !bad; ++iA) {
            double acc = 1.0;
            int E = 0, since = 0;
            int n_occ = 0, nfar = 0;
            for (;;) {
                need();
                const int4 *hp = reinterpret_cast<const int4 *>(ring + (pos & (RING_UNITS - 1)));
                const int4 h0 = hp[0], h1 = hp[1];
                const int magic = __builtin_amdgcn_readfirstlane(h0.x), n0 = __builtin_amdgcn_readfirstlane(h0.y);
                const int n1 = __builtin_amdgcn_readfirstlane(h0.z), nocc = __builtin_amdgcn_readfirstlane(h0.w);
                if ((magic & ~1) != SOLO_MAGIC || n0 < 0 || (n0 & 7) || n1 < 0 || (n1 & 3) || n0 > (int)P.N + 8 || n1 > (int)P.N + 8 || nocc < 0 || nocc > MOM_SLOTS) {
                    bad = true;
                    break;
                }
                pos += 4;
                // the most frequent row: two E per broadcast read, R in a register, eight entries per step
                for (int b = 0; b < n0; b += 8) {
                    need();
                    const double2 *ep = reinterpret_cast<const double2 *>(ring + (pos & (RING_UNITS - 1)));
                    double f[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double2 e = ep[k];                       // uniform address: LDS broadcast
                        f[2 * k] = fma(e.x, R0, 1.0);
                        f[2 * k + 1] = fma(e.y, R0, 1.0);
                    }
                    if (since + 8 > lim) { renorm_fx(acc, E); since = 0; }
                    acc *= ((f[0] * f[1]) * (f[2] * f[3])) * ((f[4] * f[5]) * (f[6] * f[7]));
                    since += 8;
                    pos += 4;
                }
                // the other rows: (E, row offset) per broadcast read, R from the LDS slice (or L2), four entries per step
                for (int b = 0; b < n1; b += 4) {
                    need();
                    const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
                    double f[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const ScratchEnt en = rp[k];
                        f[k] = fma(en.e, loadR(en.ro), 1.0);
                    }
                    if (since + 4 > lim) { renorm_fx(acc, E); since = 0; }
                    acc *= (f[0] * f[1]) * (f[2] * f[3]);
                    since += 4;
                    pos += 4;
                }
                if (magic & 1) { n_occ = nocc; nfar = __builtin_amdgcn_readfirstlane(h1.x); break; }
            }
            if (bad) break;
            if (nfar) {
                // With F = 1 the series collapses per row: log factor = sum_rows sum_k c_k R^k, c_k = -+ w_k M_k premultiplied by the
                // preparation kernel -- one Horner chain in R per occupied row (both sides of the window at once)
                double tsum = 0.0;
                for (int s = 0; s < n_occ; ++s) {
                    need();
                    const ScratchEnt *rp = ring + (pos & (RING_UNITS - 1));
                    const double2 *qa = reinterpret_cast<const double2 *>(rp);
                    const int ro = reinterpret_cast<const int *>(rp + S_ORDER / 2)[0];
                    const double R = loadR(ro);
                    double hh = 0.0;
#pragma unroll
                    for (int k = S_ORDER - 1; k >= 0; --k) {
                        const double2 v = qa[k >> 1];
                        hh = fma(hh, R, (k & 1) ? v.y : v.x);
                    }
                    tsum = fma(hh, R, tsum);
                    pos += S_MOM;
                }
                pos = (pos + 3) & ~3;
                renorm_fx(acc, E);                   // |tsum| <= S_FAR_CAP * S_EPS * 1.1 < 600: 2^860 on top of [1/2, 1) is safe
                since = 0;
                acc *= exp_neg(-tsum);
            }
            renorm_fx(acc, E);
            const int ec = min(max(E, -131071), 131071) + 131072;
            if (((ec > bestEc) || (ec == bestEc && acc > bestM)) && acc > 0.0 && p < P.npairs) {       // strict '>' (v1:501); iA ascending
                bestM = acc;
                bestEc = ec;
                bestA = iA;
            }
        }
        if (bad && lane == 0) atomicOr(V.status, 2);
        int bE = bestEc;
        int bL = (bestA < 0 || bad) ? 0x7fffffff : bestA * P.npairs + p;
        for (int off = 32; off > 0; off >>= 1) {
            const int oE = __shfl_xor(bE, off);
            const double oM = __shfl_xor(bestM, off);
            const int oL = __shfl_xor(bL, off);
            if (oE > bE || (oE == bE && (oM > bestM || (oM == bestM && oL < bL)))) { bE = oE; bestM = oM; bL = oL; }
        }
        if (lane == 0) {
            const size_t o = (size_t)slice * P.M + t;
            P.part_T[o] = bestM;
            P.part_lin[o] = bL;
            P.part_ns[o] = bE;
        }
    }
}

// Combine the per-slice winners of a test site, take the one logarithm, and count nSites of the winning A
// (the scan kernels do not carry window sizes).  The count uses the scan's exact predicate:
// i in [lo,hi], A*|g_i - t| <= zcut, g_i != t; it is monotone on either side of the test site.
struct FinalParams {
    const double *part_T; const int32_t *part_lin; const int32_t *part_ns;
    int nslices, npairs;            // parts hold (mantissa, linear index, clamped exponent + 2^17) per slice
    int64_t M, N;
    const double *genpos, *A, *test_gen;
    const int64_t *win_lo, *win_hi, *center, *center_hi;
    double zcut;
    double *clr; int32_t *lin; int32_t *nsites;
    bmx_record *rec;   // the same three values as one 16-byte record per test site (what the multi-GPU gather moves)
};

__global__ void finalize_kernel(FinalParams F) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= F.M) return;
    double bT = 0.0;
    int bL = 0x7fffffff, bN = 0;
    double bM = 1.0;
    int bE = 131072;
    for (int s = 0; s < F.nslices; ++s) {
        const double m = F.part_T[(size_t)s * F.M + t];
        const int L = F.part_lin[(size_t)s * F.M + t];
        const int e = F.part_ns[(size_t)s * F.M + t];
        if (L != 0x7fffffff && (e > bE || (e == bE && (m > bM || (m == bM && L < bL))))) {
            bM = m;
            bE = e;
            bL = L;
        }
    }
    if (bL != 0x7fffffff) {
        // kernels that normalise with v_frexp hand over mantissas in [1/2, 1): back to [1, 2), or a product just above 1 would come
        // out as LN2 + log(0.5000...) -- a difference of two numbers of size 0.69
        if (bM < 1.0) { bM *= 2.0; bE -= 1; }
        bT = 2.0 * ((double)(bE - 131072) * LN2 + log(bM));
    }
    const bool none = (bL == 0x7fffffff);
    if (!none) {
        const double A = F.A[bL / F.npairs], tg = F.test_gen[t];
        const int64_t lo = max(F.win_lo[t], (int64_t)0), hi = min(F.win_hi[t], F.N - 1);
        // right of the test position: indices [a0, hi], predicate true on a prefix
        int64_t a0 = max(F.center_hi[t], lo), a = a0, b = hi + 1;
        while (a < b) {
            int64_t m = (a + b) >> 1;
            if (A * fabs(F.genpos[m] - tg) <= F.zcut) a = m + 1; else b = m;
        }
        int64_t cnt = max(a - a0, (int64_t)0);
        // left: indices [lo, b0], predicate true on a suffix
        int64_t b0 = min(F.center[t] - 1, hi);
        a = lo; b = b0 + 1;
        while (a < b) {
            int64_t m = (a + b) >> 1;
            if (A * fabs(F.genpos[m] - tg) <= F.zcut) b = m; else a = m + 1;
        }
        cnt += max(b0 + 1 - a, (int64_t)0);
        bN = (int)cnt;
    }
    F.clr[t] = none ? 0.0 : bT;
    F.lin[t] = none ? -1 : bL;
    F.nsites[t] = none ? 0 : bN;
    F.rec[t] = bmx_record{none ? 0.0 : bT, none ? -1 : bL, none ? 0 : bN};
}

// ----------------------------------------------------------------------------- surface
// Full likelihood surface T[iA][pair] of ONE test site (the reference keeps only the maximum and
// notes the surfaces as a wish, v1:449-450).  Deliberately a different arithmetic from the scan
// kernels -- a plain sum of log1p(alpha*R) per (A, pair) -- so that it doubles as an on-device
// cross-check of the product form.  One workgroup per (A, 64-pair slice); lanes = pairs.
struct SurfParams {
    const double *genpos; RowArray row; int64_t N;
    const double *Rt; int NP, npairs, nslices;
    const double *A; int nA;
    double tg; int64_t lo, hi;
    double zcut;
    double *T;        // [nA][npairs]
    int32_t *ns;      // [nA]
};

__global__ void surface_kernel(SurfParams S) {
    const int iA = blockIdx.x / S.nslices, slice = blockIdx.x % S.nslices;
    const int lane = threadIdx.x;
    const int p = slice * WAVE + lane;
    const double A = S.A[iA];
    double sum = 0.0;
    int ns = 0;
    for (int64_t base = S.lo; base <= S.hi; base += WAVE) {
        const int64_t i = base + lane;
        const bool valid = i <= S.hi;
        const double g = valid ? S.genpos[i] : S.tg;
        const double z = A * fabs(g - S.tg);
        const bool in = valid && z <= S.zcut && g != S.tg;
        const double alpha = in ? exp(-z) : 0.0;
        const int r = valid ? (int)S.row[i] : 0;
        const unsigned long long m = __ballot(in);
        ns += __popcll(m);
        for (unsigned long long mm = m; mm; mm &= mm - 1) {
            const int l = __ffsll((long long)mm) - 1;
            const double a_s = readlane_f64(alpha, l);
            const int r_s = __builtin_amdgcn_readlane(r, l);
            sum += log1p(a_s * S.Rt[(size_t)r_s * S.NP + p]);
        }
    }
    if (p < S.npairs) S.T[(size_t)iA * S.npairs + p] = ns ? 2.0 * sum : NAN;
    if (slice == 0 && lane == 0) S.ns[iA] = ns;
}

// Largest double z with exp(-z) >= 1e-8 under correct rounding of exp: bisection on the host.
double compute_zcut() {
    double lo = 18.0, hi = 19.0;
    for (;;) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        if (exp(-mid) >= 1e-8) lo = mid; else hi = mid;
    }
    return lo;
}

template <class T>
void dfree(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

}  // namespace

// -DBMX_DEVICE_PROBE=<kernel instantiation>: device code of that ONE kernel and nothing else (register / spill counts and the
// assembly of a kernel under work in seconds instead of a minute; `make probe K='clr_scan_prepared_kernel<16, true>'`)
#ifdef BMX_DEVICE_PROBE
namespace {
template __global__ void BMX_DEVICE_PROBE(ScanParams, PrepView);
}
#else

// bmx_io.cpp: rows [0, n) formatted into `out` (<= 512 bytes per row); grid indices either as (ix, ia, iA) or as the
// linear index lin = (iA*nx + ix)*nab + ia (< 0: the reference's all-zero row).  Returns the bytes written, 0 on a bad index.
struct bmx_row_tables_;
extern "C" bmx_row_tables_ *bmx_row_tables_new_(const char *xs, int nx, const char *abs_, int nab, const char *As, int nA);
extern "C" void bmx_row_tables_free_(bmx_row_tables_ *t);
extern "C" int bmx_write_chunk_(FILE *f, const bmx_row_tables_ *t, int64_t n, const int64_t *phys, const double *gen, const double *clr,
                                const int32_t *ix, const int32_t *ia, const int32_t *iA, const int32_t *lin, const int32_t *nsites);
// bmx_io.cpp: the threaded host passes of set_sites / set_tests
extern "C" int bmx_validate_sites_(int64_t N, const double *genpos, const int32_t *row, int32_t rows, const double *g,
                                   uint16_t *r16, uint32_t *r32, int64_t *cnt);
extern "C" int bmx_tests_sorted_(int64_t M, const double *test_gen);

// =================================================================================== ctx
namespace {

// Device buffer that only grows: set_sites / set_tests / surface of a long run (22 chromosomes, thousands of surfaces)
// reuse their allocations instead of paying hipMalloc/hipFree per call.
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(n, 1);
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// fn(t, begin, end) on T host threads over [0, n)
template <class F>
void parallel_ranges(int64_t n, int64_t grain, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 32), n / std::max<int64_t>(grain, 1) + 1);
    if (T <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(fn, t, n * t / T, n * (t + 1) / T);
    for (auto &x : th) x.join();
}

}  // namespace

struct ScanPlan;

// One launch range of the prepared pipeline: test sites [off, off + cnt) = groups [g0, g0 + ng), whose blobs take `units`
// 16-byte units of the arena starting at prefix `pbase`.
struct PrepRange {
    int64_t off, cnt, g0, ng, pbase, units;
};

// Everything a launch of the scan needs that does not depend on which test sites it covers.
struct ScanPlan {
    ScanParams P;
    const void *fn = nullptr;
    int J = 0, threads = SCAN_THREADS;
    size_t lds_bytes = 0;
    int spb = 0;            // test sites per workgroup
    int64_t range = 0;      // test sites per launch (multiple of spb): bounds the per-slice winner arrays
    bool use_lds = false;
    int mode = 3;           // inner-loop form of the grouped kernel (0..3), 4: prepared pipeline
    // prepared pipeline (mode 4): the per-group kernel's two forms and its launch shape
    const void *prep_count = nullptr, *prep_fill = nullptr;
    size_t prep_lds = 0, prep_lds_count = 0;
    int thr_in_lds = 0, prep_threads = PREP_THREADS;
};

// One chromosome of a context (a "slot"): its site arrays, its test sites, their results and the scan plan made for them.
// A context holds ONE model and any number of slots (bmx_ctx_select_slot); a whole-genome run keeps every chromosome
// resident and scans them back to back on the context's stream.
struct ChromSlot {
    // sites (grow-only buffers)
    bool has_sites = false;
    int64_t N = 0;
    DevBuf<double> genpos, rowmax, rowthr;
    DevBuf<uint16_t> row16;      // one of the two is used
    DevBuf<uint32_t> row32;
    bool wide_rows = false;
    DevBuf<uint8_t> kmom;        // far-field moment slots that pay at each A (set_sites)
    DevBuf<int> d_row_of_slot;
    int row_of_slot[MOM_SLOTS] = {0};
    int nslots = 0;              // rows ranked by frequency: row_of_slot[0 .. nslots)
    int kmom_max = 0;            // the most slots any A uses
    int kmom_max_ser = 0;        // ... in the prepared group kernels' table (second half of kmom)
    // tests
    bool has_tests = false;
    int64_t M = 0;
    DevBuf<double> test_gen;
    DevBuf<int64_t> win_lo, win_hi, center, center_hi;
    bool tests_sorted = false;
    int64_t test_gap = 1 << 30;  // median index gap between neighbouring test sites (strided sample of all of them)
    // results of all test sites
    bool timed = false;          // a scan has been launched since the test sites were set: results / events are valid
    DevBuf<double> clr;
    DevBuf<int32_t> lin, nsites;
    DevBuf<bmx_record> rec;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // plan of the scan, made once per (model, sites, tests, variant)
    bool plan_ok = false;
    int plan_variant = -1;
    ScanPlan plan;
    // prepared pipeline: blob sizes per group, their exclusive prefix, launch ranges
    bool prep_ok = false;
    DevBuf<int32_t> blob_units;
    DevBuf<int64_t> blob_prefix;
    std::vector<PrepRange> ranges;

    void release() {
        genpos.release(); rowmax.release(); rowthr.release(); row16.release(); row32.release(); kmom.release(); d_row_of_slot.release();
        test_gen.release(); win_lo.release(); win_hi.release(); center.release(); center_hi.release();
        clr.release(); lin.release(); nsites.release(); rec.release();
        blob_units.release(); blob_prefix.release();
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        ev0 = ev1 = nullptr;
    }
};

struct bmx_ctx {
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    hipEvent_t ev_done[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    int variant = 0;
    // model
    bool has_model = false;
    int stat = 0, min_count = 1, n_sizes = 0, rows = 0, nx = 0, nab = 0, npairs = 0, NP = 0, nslices = 0, nA = 0;
    int renorm_every = 16, span_hi = 1;
    double rmax = 0.0;
    int32_t *d_sizes = nullptr, *d_row_off = nullptr;
    double *d_g = nullptr, *d_prop = nullptr, *d_x = nullptr, *d_abeta = nullptr, *d_A = nullptr;
    double *d_psel = nullptr, *d_R = nullptr, *d_Rt = nullptr;
    std::vector<double> h_A, h_rowmax;   // h_rowmax: [nslices][rows] max |R| (+inf: absent row)
    uint64_t *d_patch_x = nullptr;
    double *d_patch_y = nullptr;
    int n_patch = 0;
    std::vector<double> h_g;
    // chromosomes
    std::vector<ChromSlot *> slots;      // slot 0 exists from creation; nullptr: never selected
    ChromSlot *cur = nullptr;
    int cur_index = 0;
    unsigned long long *d_prof = nullptr;   // -DBMX_PROFILE / -DBMX_COUNT builds
    // scratch shared by the slots (launches of one context are serialised on its stream)
    DevBuf<double> part_T;       // per-slice winners of one launch range
    DevBuf<int32_t> part_lin, part_ns;
    DevBuf<ScratchEnt> arena;    // prepared pipeline: the blobs of one launch range
    int *d_status = nullptr;     // ... and its error bits
    DevBuf<double> surf_T;
    DevBuf<int32_t> surf_ns;
    DevBuf<int64_t> gap_sample;
    // pinned host staging of the streaming writer: two slots of (clr, lin, nsites)
    void *h_stage[2] = {nullptr, nullptr};
    size_t h_stage_cap = 0;
    double zcut = 0;
};

namespace {

void free_model(bmx_ctx *c) {
    dfree(c->d_sizes); dfree(c->d_row_off); dfree(c->d_g); dfree(c->d_prop); dfree(c->d_x);
    dfree(c->d_abeta); dfree(c->d_A); dfree(c->d_psel); dfree(c->d_R); dfree(c->d_Rt);
    dfree(c->d_patch_x); dfree(c->d_patch_y);
    c->has_model = false;
}
// sites / tests: the buffers stay allocated for the next chromosome; only the state is dropped
void drop_tests(ChromSlot *s) {
    s->has_tests = false;
    s->timed = false;       // results belong to the test sites they were computed for
    s->plan_ok = false;
    s->prep_ok = false;
}
void drop_sites(ChromSlot *s) {
    s->has_sites = false;
    drop_tests(s);          // test sites were located in the old site array
}

template <class T>
int upload(T *&dst, const T *src, size_t n, hipStream_t s) {
    HIP_TRY(hipMalloc((void **)&dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n) HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    return BMX_OK;
}
template <class T>
int upload(DevBuf<T> &dst, const T *src, size_t n, hipStream_t s) {
    HIP_TRY(dst.ensure(n));
    if (n) HIP_TRY(hipMemcpyAsync(dst.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    return BMX_OK;
}

int validate_model(const bmx_model *m) {
    if (!m) return fail(BMX_E_INVALID, "model is NULL");
    if (m->stat < BMX_STAT_B2 || m->stat > BMX_STAT_B1) return fail(BMX_E_INVALID, "unknown statistic id");
    if (m->n_sizes < 1 || !m->sizes || !m->row_off || !m->g || !m->prop)
        return fail(BMX_E_INVALID, "model: sizes/row_off/g/prop must be given");
    if (m->nx < 1 || m->nab < 1 || !m->x || !m->abeta) return fail(BMX_E_INVALID, "model: empty x / alpha_beta grid");
    if (m->min_count < 0) return fail(BMX_E_INVALID, "model: negative min_count");
    if (m->row_off[0] != 0) return fail(BMX_E_INVALID, "model: row_off[0] must be 0");
    for (int j = 0; j < m->n_sizes; j++) {
        int want = m->stat == BMX_STAT_B1 ? 2 : m->sizes[j] + 1;
        if (m->sizes[j] < 1 || m->row_off[j + 1] - m->row_off[j] != want)
            return fail(BMX_E_INVALID, "model: row_off does not match sizes (n+1 rows per size, 2 for B1)");
    }
    if (m->row_off[m->n_sizes] > (1 << 24)) return fail(BMX_E_LIMIT, "model: more than 2^24 LUT rows");
    for (int i = 0; i < m->nx; i++)
        if (!(m->x[i] > 0.0 && m->x[i] < 1.0)) return fail(BMX_E_INVALID, "model: x grid must lie in (0,1)");
    for (int i = 0; i < m->nab; i++)
        if (!(m->abeta[i] > 0.0)) return fail(BMX_E_INVALID, "model: alpha_beta grid must be positive");
    return BMX_OK;
}

constexpr int MAX_SLOTS = 4096;

int ensure_plan(bmx_ctx *c, ChromSlot *s);

int select_slot(bmx_ctx *c, int slot) {
    if (slot < 0 || slot >= MAX_SLOTS) return fail(BMX_E_INVALID, "slot index out of range (0..4095)");
    if ((size_t)slot >= c->slots.size()) c->slots.resize((size_t)slot + 1, nullptr);
    if (!c->slots[(size_t)slot]) {
        ChromSlot *s = new ChromSlot();
        if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
            s->release();
            delete s;
            return fail(BMX_E_HIP, "event creation failed");
        }
        c->slots[(size_t)slot] = s;
    }
    c->cur = c->slots[(size_t)slot];
    c->cur_index = slot;
    return BMX_OK;
}

}  // namespace

extern "C" {

void bmx_version(int *major, int *minor) {
    if (major) *major = BMX_ABI_VERSION_MAJOR;
    if (minor) *minor = BMX_ABI_VERSION_MINOR;
}

/* A short hash of the kernel source this binary was built from (Makefile: -DBMX_SRC_HASH): the Python shim and the
 * tests compare it with the source in the tree, so that a stale libbmxscan.so cannot be tested or benchmarked. */
#ifndef BMX_SRC_HASH
#define BMX_SRC_HASH "unknown"
#endif
const char *bmx_build_id(void) { return BMX_SRC_HASH; }

const char *bmx_last_error(void) { return g_err.c_str(); }
void bmx_set_error_(const char *msg) { g_err = msg ? msg : ""; }   // for the library's other translation units

int bmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

double bmx_alpha_cut(void) { return compute_zcut(); }

int bmx_ctx_create(bmx_ctx **out, int device) {
    if (!out) return fail(BMX_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = bmx_device_count();
    if (n <= 0) return fail(BMX_E_NODEVICE, "no HIP device available (libbmxscan has no CPU fallback)");
    if (device < 0 || device >= n) return fail(BMX_E_NODEVICE, "device index out of range");
    TRACE("ctx_create: hipSetDevice(%d)", device);
    HIP_TRY(hipSetDevice(device));
    bmx_ctx *c = new bmx_ctx();
    c->device = device;
    c->zcut = compute_zcut();
    bool ok = hipStreamCreate(&c->stream) == hipSuccess && hipStreamCreate(&c->copy_stream) == hipSuccess;
    for (int k = 0; ok && k < 2; k++)
        ok = hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_copied[k], hipEventDisableTiming) == hipSuccess;
    if (ok) ok = hipMalloc((void **)&c->d_status, sizeof(int)) == hipSuccess && hipMemset(c->d_status, 0, sizeof(int)) == hipSuccess;
    if (ok) ok = select_slot(c, 0) == BMX_OK;
    if (!ok) {
        bmx_ctx_destroy(c);
        return fail(BMX_E_HIP, "stream/event creation failed");
    }
    TRACE("ctx_create: done");
    *out = c;
    return BMX_OK;
}

void bmx_ctx_destroy(bmx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    free_model(c);
    for (ChromSlot *s : c->slots)
        if (s) { s->release(); delete s; }
    c->slots.clear();
    c->part_T.release(); c->part_lin.release(); c->part_ns.release(); c->arena.release(); c->gap_sample.release();
    c->surf_T.release(); c->surf_ns.release();
    dfree(c->d_prof);
    dfree(c->d_status);
    for (int k = 0; k < 2; k++) {
        if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
        if (c->ev_copied[k]) (void)hipEventDestroy(c->ev_copied[k]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

int bmx_ctx_set_variant(bmx_ctx *c, int variant) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    c->variant = variant;
    return BMX_OK;
}

int bmx_ctx_select_slot(bmx_ctx *c, int32_t slot) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return select_slot(c, slot);
}

int bmx_ctx_slot_count(bmx_ctx *c) {
    if (!c) return 0;
    return (int)c->slots.size();
}

int bmx_ctx_set_model(bmx_ctx *c, const bmx_model *m, const double *A, int32_t nA) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    int rc = validate_model(m);
    if (rc) return rc;
    if (!A || nA < 1) return fail(BMX_E_INVALID, "empty A grid");
    for (int i = 0; i < nA; i++)
        if (!(A[i] > 0.0)) return fail(BMX_E_INVALID, "A grid must be positive");
    if ((int64_t)nA * m->nx * m->nab > 0x7ffffff0LL) return fail(BMX_E_LIMIT, "grid has more than 2^31 points");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_model(c);
    for (ChromSlot *s : c->slots)
        if (s) drop_sites(s);      // row indices and the moment slots belong to the model they were set under
    c->stat = m->stat; c->min_count = m->min_count; c->n_sizes = m->n_sizes;
    c->rows = m->row_off[m->n_sizes]; c->nx = m->nx; c->nab = m->nab; c->nA = nA;
    c->npairs = m->nx * m->nab;
    c->h_g.assign(m->g, m->g + c->rows);
    c->NP = (c->npairs + WAVE - 1) / WAVE * WAVE;
    c->nslices = c->NP / WAVE;
    if ((rc = upload(c->d_sizes, m->sizes, (size_t)m->n_sizes, c->stream))) return rc;
    if ((rc = upload(c->d_row_off, m->row_off, (size_t)m->n_sizes + 1, c->stream))) return rc;
    if ((rc = upload(c->d_g, m->g, (size_t)c->rows, c->stream))) return rc;
    if ((rc = upload(c->d_prop, m->prop, (size_t)m->n_sizes, c->stream))) return rc;
    if ((rc = upload(c->d_x, m->x, (size_t)m->nx, c->stream))) return rc;
    if ((rc = upload(c->d_abeta, m->abeta, (size_t)m->nab, c->stream))) return rc;
    if ((rc = upload(c->d_A, A, (size_t)nA, c->stream))) return rc;
    c->h_A.assign(A, A + nA);
    size_t tab = (size_t)c->npairs * c->rows;
    HIP_TRY(hipMalloc((void **)&c->d_psel, tab * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&c->d_R, tab * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&c->d_Rt, (size_t)c->rows * c->NP * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->d_Rt, 0, (size_t)c->rows * c->NP * sizeof(double), c->stream));
    // host-libm conformance patch (bmx_math.h): every argument that K1 takes a log of inside lgam's
    // Stirling branch, evaluated with this host's libm and with the device's correctly rounded log
    std::vector<uint64_t> px;
    std::vector<double> py;
    {
        std::unordered_set<uint64_t> seen;        // many (k, n) give the same argument bits: check each once
        auto consider = [&](double v) {
            if (!(v >= 13.0) || !(v < 1e300)) return;     // lgam's log(x) branch starts at 13
            uint64_t vb;
            memcpy(&vb, &v, sizeof(vb));
            if (!seen.insert(vb).second) return;
            const double hl = log(v);
            if (hl != bmx::crlog(v)) {
                uint64_t b;
                memcpy(&b, &v, sizeof(b));
                if (std::find(px.begin(), px.end(), b) == px.end() && px.size() < 4096) { px.push_back(b); py.push_back(hl); }
            }
        };
        int nmax = 0;
        for (int j = 0; j < m->n_sizes; j++) nmax = std::max(nmax, m->sizes[j]);
        for (int ix = 0; ix < m->nx; ix++)
            for (int ia = 0; ia < m->nab; ia++) {
                const double a = m->abeta[ia];
                for (int side = 0; side < 2; side++) {
                    const double xx = side ? 1. - m->x[ix] : m->x[ix];
                    const double b = a / xx - a;
                    consider(a); consider(b); consider(a + b);
                    for (int j = 0; j < m->n_sizes; j++) {
                        const int n = m->sizes[j];
                        for (int k = 0; k <= n; k++) {
                            const double p1 = k + a, q1 = n - k + b;
                            consider(p1); consider(q1); consider(p1 + q1);
                        }
                    }
                }
            }
    }
    c->n_patch = (int)px.size();
    TRACE("set_model: %d libm log exceptions", c->n_patch);
    if (c->n_patch) {
        if ((rc = upload(c->d_patch_x, (const uint64_t *)px.data(), px.size(), c->stream))) return rc;
        if ((rc = upload(c->d_patch_y, (const double *)py.data(), py.size(), c->stream))) return rc;
    }
    LutParams P;
    P.patch = bmx::LogPatch{c->d_patch_x, c->d_patch_y, c->n_patch};
    P.stat = c->stat; P.min_count = c->min_count; P.n_sizes = c->n_sizes; P.rows = c->rows;
    P.nx = c->nx; P.nab = c->nab; P.NP = c->NP;
    P.sizes = c->d_sizes; P.row_off = c->d_row_off; P.g = c->d_g; P.prop = c->d_prop;
    P.x = c->d_x; P.abeta = c->d_abeta; P.psel = c->d_psel; P.R = c->d_R; P.Rt = c->d_Rt;
    int threads = 128;
    int blocks = (int)((tab + threads - 1) / threads);
    TRACE("set_model: launching bb_lut_kernel, %d blocks x %d, rows=%d pairs=%d", blocks, threads, c->rows, c->npairs);
    hipLaunchKernelGGL(bb_lut_kernel, dim3(blocks), dim3(threads), 0, c->stream, P);
    HIP_TRY(hipGetLastError());
    // How many sites may be multiplied between exponent extractions: every factor
    // 1 + alpha*R lies in [min(1, 1+Rmin), max(1, 1+Rmax)]; keep the product inside 2^+-1000.
    std::vector<double> hR(tab);
    HIP_TRY(hipMemcpyAsync(hR.data(), c->d_R, tab * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    TRACE("set_model: table built");
    double fmax = 1.0, fmin = 1.0;
    for (size_t i = 0; i < tab; i++) {
        double v = hR[i];
        if (v != v) continue;  // rows absent from the helper file
        double f = 1.0 + v;
        fmax = std::max(fmax, f);
        if (f > 0.0) fmin = std::min(fmin, f);
    }
    if (!(fmax < 1e300)) return fail(BMX_E_INVALID, "selection table overflows (a neutral probability g is 0 or tiny)");
    c->span_hi = std::max(1, (int)std::ceil(std::log2(fmax)));
    if (c->span_hi > 240)     // four factors are multiplied between exponent extractions
        return fail(BMX_E_LIMIT, "selection table spans more than 2^240 (a neutral probability of ~1e-70?)");
    c->rmax = fmax - 1.0;
    {   // per slice and row: the largest |R| over the slice's pairs (grouped kernel's far-field test)
        std::vector<double> rm((size_t)c->nslices * c->rows, 0.0);
        for (int p = 0; p < c->npairs; p++)
            for (int r = 0; r < c->rows; r++) {
                const double v = std::fabs(hR[(size_t)p * c->rows + r]);
                double &m = rm[(size_t)(p / WAVE) * c->rows + r];
                m = (v != v) ? INFINITY : std::max(m, v);
            }
        c->h_rowmax.swap(rm);                 // uploaded by set_sites, with the rows' moment slots packed in
    }
    // per-site kernel: worst case per factor is max(span_hi, 54 bits for 1 - alpha) -- see the kernel
    c->renorm_every = std::max(1, std::min(16, 1000 / std::max(c->span_hi, 54)));
    c->has_model = true;
    return BMX_OK;
}

int bmx_ctx_set_sites(bmx_ctx *c, int64_t N, const double *genpos, const int32_t *row) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    if (!c->has_model) return fail(BMX_E_STATE, "set_model must precede set_sites");
    if (N < 1 || !genpos || !row) return fail(BMX_E_INVALID, "empty site arrays");
    if (N >= 0x7fffffffLL) return fail(BMX_E_LIMIT, "more than 2^31 sites in one array (scan chromosomes one at a time)");
    ChromSlot *s = c->cur;
    // One pass on the host's cores: every row index inside the table and on a (k, n) with a positive neutral probability
    // (the kernels index the LDS/L2 table with it unchecked), positions sorted and not NaN; 16-bit row indices and the
    // per-row site counts (moment slots) come out of the same pass.
    const bool wide = c->rows > 65535;
    std::vector<uint16_t> r16(wide ? 0 : (size_t)N);
    std::vector<uint32_t> r32(wide ? (size_t)N : 0);
    std::vector<int64_t> cnt((size_t)c->rows, 0);
    switch (bmx_validate_sites_(N, genpos, row, c->rows, c->h_g.data(), wide ? nullptr : r16.data(), wide ? r32.data() : nullptr, cnt.data())) {
        case 1: return fail(BMX_E_INVALID, "site row index outside the LUT");
        case 2: return fail(BMX_E_INVALID, "a site has a (count, sample size) whose neutral probability is missing or not positive");
        case 3: return fail(BMX_E_INVALID, "genetic positions must be non-decreasing");
        case 4: return fail(BMX_E_INVALID, "NaN genetic position");
        default: break;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_sites(s);
    int rc;
    if ((rc = upload(s->genpos, genpos, (size_t)N, c->stream))) return rc;
    if (wide) {
        if ((rc = upload(s->row32, (const uint32_t *)r32.data(), (size_t)N, c->stream))) return rc;
    } else {
        if ((rc = upload(s->row16, (const uint16_t *)r16.data(), (size_t)N, c->stream))) return rc;
    }
    s->wide_rows = wide;
    {
        // Moment slots for the far field: rank the rows by how many sites carry them.
        // Slot s pays at a given A when the ~23 instructions saved per far site of that row outweigh
        // the ~30 instructions its term costs at the end of each zone; the expected number of far sites
        // per zone follows from the mean site density (a performance heuristic only: any choice is exact).
        // The prepared group kernels have a cheaper place than the product for a far site without a slot
        // (series entries, ~11 instructions less than the product): a second table with that gain for them
        // (measured optimum 10-12, flat: profiles/r03_series_entries.txt).
        std::vector<int> order((size_t)c->rows);
        for (int r = 0; r < c->rows; r++) order[(size_t)r] = r;
        const size_t ns = std::min((size_t)MOM_SLOTS, order.size());
        std::partial_sort(order.begin(), order.begin() + ns, order.end(),
                          [&](int a, int b) { return cnt[(size_t)a] != cnt[(size_t)b] ? cnt[(size_t)a] > cnt[(size_t)b] : a < b; });
        std::vector<uint8_t> slot((size_t)c->rows, 255);
        int nslots = 0;
        for (size_t k = 0; k < ns; k++) {
            if (cnt[(size_t)order[k]] == 0) break;
            slot[(size_t)order[k]] = (uint8_t)k;
            s->row_of_slot[k] = order[k];
            nslots = (int)k + 1;
        }
        for (int k = nslots; k < MOM_SLOTS; k++) s->row_of_slot[k] = nslots ? s->row_of_slot[0] : 0;
        s->nslots = nslots;
        const int kcap = diag_env("BMX_MOM_SLOTS") ? std::min(std::max(atoi(diag_env("BMX_MOM_SLOTS")), 0), MOM_SLOTS) : MOM_SLOTS;
        const double range = genpos[N - 1] - genpos[0];
        std::vector<uint8_t> km(2 * (size_t)c->nA, 0);            // [0, nA): solo / round-2 kernels; [nA, 2 nA): prepared group kernels
        s->kmom_max = 0;
        s->kmom_max_ser = 0;
        for (int t = 0; t < 2; t++) {
            const double gain = diag_env("BMX_KMOM_GAIN") ? atof(diag_env("BMX_KMOM_GAIN")) : t ? 11.0 : 23.0;      // (threshold experiments)
            for (int a = 0; a < c->nA; a++) {
                const double nfar = range > 0 ? 0.8 * (double)(N - 1) / range * c->zcut / c->h_A[(size_t)a] : (double)N;
                int k = 0;
                while (k < nslots && k < kcap && (double)cnt[(size_t)s->row_of_slot[k]] / (double)N * nfar * gain > 30.0) k++;
                km[(size_t)t * c->nA + a] = (uint8_t)k;
                (t ? s->kmom_max_ser : s->kmom_max) = std::max(t ? s->kmom_max_ser : s->kmom_max, k);
            }
        }
        // the kernel reads max |R| and the slot of a row with one load: the slot sits in the low mantissa
        // byte of the (rounded up) maximum; +inf becomes NaN, which never compares as far
        std::vector<double> packed(c->h_rowmax.size());
        // BMX_ROWMAX_GLOBAL (diagnostic builds): one far threshold per row for all slices (its max |R| over the whole grid) -- what a
        // classification shared by the slices would have to use; measures how many more near sites that costs
        std::vector<double> rm_src(c->h_rowmax);
        if (diag_env("BMX_ROWMAX_GLOBAL")) {
            const size_t R_ = (size_t)c->rows, S_ = rm_src.size() / R_;
            for (size_t r = 0; r < R_; r++) {
                double m = 0.0;
                bool nan = false;
                for (size_t sl = 0; sl < S_; sl++) { const double v = rm_src[sl * R_ + r]; if (v != v) nan = true; else m = std::max(m, v); }
                for (size_t sl = 0; sl < S_; sl++) if (!nan) rm_src[sl * R_ + r] = m;
            }
        }
        for (size_t k = 0; k < packed.size(); k++) {
            uint64_t bits;
            memcpy(&bits, &rm_src[k], sizeof bits);
            bits = ((bits & ~0xffull) + 0x100ull) | slot[k % (size_t)c->rows];
            memcpy(&packed[k], &bits, sizeof bits);
        }
        if ((rc = upload(s->rowmax, (const double *)packed.data(), packed.size(), c->stream))) return rc;
        if ((rc = upload(s->kmom, (const uint8_t *)km.data(), km.size(), c->stream))) return rc;
        // prepared pipeline: ONE far threshold per row for all slices, in the exponent domain -- a site of the row at
        // z = A d >= thr has alpha max_grid|R| <= far_eps (log rounded up, the low mantissa byte replaced by the row's slot
        // after rounding up once more).  Rows with max|R| <= far_eps / e are far at any distance (thr = -1); rows absent from
        // the helper file (max|R| = inf) get NaN, which never compares as far.
        const double eps = P_EPS;
        std::vector<double> thr((size_t)c->rows);
        const size_t R_ = (size_t)c->rows, S_ = c->h_rowmax.size() / std::max<size_t>(R_, 1);
        for (size_t r = 0; r < R_; r++) {
            double m = 0.0;
            for (size_t sl = 0; sl < S_; sl++) {
                const double v = c->h_rowmax[sl * R_ + r];
                m = (v != v || m != m) ? NAN : std::max(m, v);
            }
            double t = (m != m) ? INFINITY : (m <= eps * 0.36) ? -1.0 : std::log(m / eps) * (1.0 + 1e-12) + 1e-9;
            if (t < 0.0 && t != -1.0) t = std::max(t, -1.0);
            uint64_t bits;
            memcpy(&bits, &t, sizeof bits);
            if (t > 0.0) bits = ((bits & ~0xffull) + 0x100ull) | slot[r];     // rounds the threshold up; +inf becomes NaN
            else bits = (bits & ~0xffull) | slot[r];                            // negative: far at any distance either way
            memcpy(&thr[r], &bits, sizeof bits);
        }
        if ((rc = upload(s->rowthr, (const double *)thr.data(), thr.size(), c->stream))) return rc;
        if ((rc = upload(s->d_row_of_slot, (const int *)s->row_of_slot, (size_t)MOM_SLOTS, c->stream))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));     // the staging vectors go out of scope here
    s->N = N;
    s->has_sites = true;
    return BMX_OK;
}

int bmx_ctx_set_tests(bmx_ctx *c, int64_t M, const double *test_gen, const int64_t *win_lo,
                      const int64_t *win_hi) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    ChromSlot *s = c->cur;
    if (!s->has_sites) return fail(BMX_E_STATE, "set_sites must precede set_tests");
    if (M < 1 || !test_gen) return fail(BMX_E_INVALID, "empty test-site arrays");
    if ((win_lo == nullptr) != (win_hi == nullptr)) return fail(BMX_E_INVALID, "window bounds: both arrays or neither (neither = all sites)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_tests(s);
    int rc;
    if ((rc = upload(s->test_gen, test_gen, (size_t)M, c->stream))) return rc;
    if (win_lo) {
        if ((rc = upload(s->win_lo, win_lo, (size_t)M, c->stream))) return rc;
        if ((rc = upload(s->win_hi, win_hi, (size_t)M, c->stream))) return rc;
    } else {        // the reference's default mode (v1:598-610): every window holds all sites -- nothing to copy
        HIP_TRY(s->win_lo.ensure((size_t)M));
        HIP_TRY(s->win_hi.ensure((size_t)M));
        hipLaunchKernelGGL(all_sites_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->stream, s->win_lo.p, s->win_hi.p, M, s->N);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(s->center.ensure((size_t)M));
    HIP_TRY(s->center_hi.ensure((size_t)M));
    HIP_TRY(s->clr.ensure((size_t)M));
    HIP_TRY(s->lin.ensure((size_t)M));
    HIP_TRY(s->nsites.ensure((size_t)M));
    HIP_TRY(s->rec.ensure((size_t)M));
    int threads = 256;
    hipLaunchKernelGGL(locate_kernel, dim3((unsigned)((M + threads - 1) / threads)), dim3(threads), 0, c->stream,
                       s->genpos.p, s->N, s->test_gen.p, M, s->center.p, s->center_hi.p);
    HIP_TRY(hipGetLastError());
    // test-site density: the median index gap between neighbouring test sites over a strided sample of ALL of them (decides
    // grouped vs per-site kernel and the group size; a chromosome whose head differs from its body is judged by its body)
    const int64_t ns = std::min<int64_t>(M - 1, 65536);
    if (ns > 0) {
        HIP_TRY(c->gap_sample.ensure((size_t)ns));
        hipLaunchKernelGGL(gap_kernel, dim3((unsigned)((ns + threads - 1) / threads)), dim3(threads), 0, c->stream,
                           s->center.p, (M - 1) / ns, ns, c->gap_sample.p);
        HIP_TRY(hipGetLastError());
    }
    // while the device locates the test sites: the grouped kernels need ascending test positions
    s->tests_sorted = bmx_tests_sorted_(M, test_gen) != 0;
    s->test_gap = 1 << 30;
    if (ns > 0) {
        std::vector<int64_t> gaps((size_t)ns);
        HIP_TRY(hipMemcpyAsync(gaps.data(), c->gap_sample.p, (size_t)ns * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::nth_element(gaps.begin(), gaps.begin() + gaps.size() / 2, gaps.end());
        s->test_gap = gaps[gaps.size() / 2];
    } else {
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    s->M = M;
    s->has_tests = true;
    // the plan of the scan, and for the prepared pipeline the sizes of its per-group blobs (a counting pass on the device),
    // are part of setting the test sites: bmx_ctx_scan itself only launches
    return ensure_plan(c, s);
}

}  // extern "C"

namespace {

// the dynamic-LDS limit of a kernel is set per FUNCTION and launches read it: attribute + launch pairs of different contexts
// (threads of bmx_scan_multi, a caller's own threads) must not interleave
std::mutex g_launch_mu;

int plan_scan(bmx_ctx *c, ChromSlot *s, ScanPlan &pl) {
    ScanParams &P = pl.P;
    P.genpos = s->genpos.p; P.row = RowArray{s->wide_rows ? nullptr : s->row16.p, s->wide_rows ? s->row32.p : nullptr}; P.N = s->N; P.Rt = c->d_Rt;
    P.rows = c->rows; P.NP = c->NP; P.npairs = c->npairs; P.nslices = c->nslices;
    P.A = c->d_A; P.nA = c->nA; P.zcut = c->zcut; P.renorm_every = c->renorm_every; P.span_hi = c->span_hi; P.rmax = c->rmax;
    {
        // 8th order: economised coefficients, valid on [-0.05, 0.05] (FAR_W); lower orders: Taylor, cut at ~6e-15
        const double eps_default = FAR_ORDER >= 8 ? 0.05 : FAR_ORDER >= 6 ? 0.0105 : 0.0016;
        double eps = diag_env("BMX_FAR_EPS") ? atof(diag_env("BMX_FAR_EPS")) : eps_default;   // accuracy experiments
        eps = std::min(std::max(eps, 0.0), FAR_ORDER >= 8 ? 0.05 : 0.035);
        P.far_eps = eps;
        P.rowmax = s->rowmax.p;
        P.kmom = s->kmom.p;
        P.prof = nullptr;
#if defined(BMX_PROFILE) || defined(BMX_COUNT)
        if (!c->d_prof) { HIP_TRY(hipMalloc((void **)&c->d_prof, 32 * sizeof(unsigned long long))); HIP_TRY(hipMemset(c->d_prof, 0, 32 * sizeof(unsigned long long))); }
        P.prof = c->d_prof;
#endif
        for (int k = 0; k < MOM_SLOTS; k++) P.row_of_slot[k] = s->row_of_slot[k];
        P.far_bits = (float)(eps * 1.4427 * 1.03);      // |log1p(x)| <= 1.027 |x| for |x| <= 0.05
    }
    size_t lds = (size_t)c->rows * WAVE * sizeof(double);
    if (const char *pad = diag_env("BMX_LDS_PAD")) lds += (size_t)std::max(atoi(pad), 0);   // occupancy experiments
    P.row0 = s->nslots > 0 ? s->row_of_slot[0] : -1;
    P.wide_tab = (size_t)c->rows * c->NP * sizeof(double) >= ((size_t)1 << 32) ? 1 : 0;
    // Grouping pays while neighbouring test sites share most of their windows.  Measured on config 3 in round 3 with the
    // prepared kernels (windows/s x1000 for J = 16 / 8 / 4 / one test site per wave; profiles/r03_stride_table.txt): stride 1:
    // 4175/-/-/-, 2: 3133/2860/1774/1339, 4: -/2433/1677/1343, 6: -/2058/1543/1342, 8: -/1741/1474/1337, 12: -/1354/1298/1333,
    // 16: -/1157/1158/1322, 24: -/916/957/1314, 32: -/772/836/1301, 48: -/585/687/1282, 64+: 1268 ... 1174 at 200
    // -> J = 16 up to a median gap of 3 sites between test sites, 8 up to 12 (round 4: up to 9, the solo pipeline having gained 15 %),
    //    beyond that one test site per wave.
    // (The round-2 kernels, variant 12, keep their own thresholds: 16 / 8 / 4 up to 3 / 28 / 56, then the per-site kernel.)
    const int64_t gap_max = diag_env("BMX_DENSE_GAP") ? atoll(diag_env("BMX_DENSE_GAP")) : 56;
    const bool can_group = s->tests_sorted && s->test_gap <= gap_max && c->span_hi <= 62 && s->N < 0x7fffffffLL && c->nA < 8191 && !P.wide_tab;
    int J = 0;
    // variants: 0 -> prepared pipeline (round 3: per-group work done once by prep_kernel), J by test-site gap;
    //           13 / 14 / 15 -> prepared, J = 16 / 8 / 4;
    //           12 -> the round-2 grouped kernel, J by test-site gap (pairs near / quads mid / power sums far);
    //           3 -> J=8, 4 -> J=4 (same form); 10/11 -> J=16/8 without the power sums (exact products);
    //           8/9 -> J=16/8 pairs only; 5/6/7 -> J=16/8/4 readlane single-site loop; 1, 2 -> per-site kernel
    const int v = c->variant;
    const bool prep_ok = !diag_env("BMX_FAR_EPS");
    // one test site per wave, prepared (mode 5): sparse or unsorted test sites (variant 0), or on request (variant 16)
    const int64_t solo_gap = diag_env("BMX_SOLO_GAP") ? atoll(diag_env("BMX_SOLO_GAP")) : SOLO_GAP;
    const bool solo = prep_ok && !P.wide_tab && s->N < 0x7fffffffLL && (v == 16 || (v == 0 && (!can_group || s->test_gap > solo_gap)));
    const bool prepared = !solo && can_group && (v == 0 || (v >= 13 && v <= 15)) && prep_ok;
    if (can_group && !solo) {
        J = (v == 0 || v == 12 || v == 5 || v == 8 || v == 10 || v == 13) ? 16 : (v == 3 || v == 6 || v == 9 || v == 11 || v == 14) ? 8
            : (v == 4 || v == 7 || v == 15) ? 4 : 0;
        if (v == 12) J = s->test_gap <= 3 ? 16 : s->test_gap <= 28 ? 8 : 4;
        if (v == 0) J = s->test_gap <= 3 ? 16 : 8;                  // (gaps beyond SOLO_GAP never get here: solo)
        if ((v == 0 || v == 12) && diag_env("BMX_FORCE_J")) {                    // threshold experiments: 16, 8 or 4
            const int fj = atoi(diag_env("BMX_FORCE_J"));
            if (fj != 16 && fj != 8 && fj != 4) return fail(BMX_E_INVALID, "BMX_FORCE_J must be 16, 8 or 4");
            J = fj;
        }
    }
    // LDS: the prepared kernel keeps per wave a ring of the blob stream and 512 B of scratch, the round-2 grouped kernel a
    // scratch list and the moments; both the sites between the test sites; the slice's per-row max |R| only the latter
    int mom_slots = MOM_SLOTS_LDS;
    auto wave_bytes = [&](int slots) {
        if (solo) return (size_t)(RING_UNITS + RING_MIRROR) * sizeof(ScratchEnt);
        if (prepared) return (size_t)(RING_UNITS + RING_MIRROR + AUX_UNITS) * sizeof(ScratchEnt) + (size_t)MID_CAP * 12;
        return SCR_CAP * sizeof(ScratchEnt) + (size_t)(slots + MOM_COPIES - 1 + 3) * FAR_ORDER * sizeof(double) + (size_t)MID_CAP * 12;
    };
    const size_t lds_rm = (prepared || solo) ? 0 : (size_t)((c->rows + 1) & ~1) * sizeof(double);
    auto lds_need = [&](int slots) { return lds + lds_rm + (size_t)((solo ? SITE_THREADS : SCAN_THREADS_MAX) / WAVE) * wave_bytes(slots); };
    // moment slots: as many (64, 32, 16, 8, 0) as leave the R slice in LDS; when the table is too large for LDS anyway
    // (many sample sizes: the sites spread over many rows), all MOM_SLOTS.  (The prepared pipeline's moments live in
    // prep_kernel's LDS: 64 slots with the table in LDS, all of them otherwise -- the same rule, so that both forms
    // classify alike.)
    while (!prepared && !solo && mom_slots >= 8 && lds_need(mom_slots) > (size_t)LDS_LIMIT_BYTES) mom_slots /= 2;
    if (mom_slots < 8) mom_slots = 0;
    const bool fits = lds_need(mom_slots) <= (size_t)LDS_LIMIT_BYTES && !diag_env("BMX_NO_LDS");   // BMX_NO_LDS: R from L2 (A/B runs)
    if (!fits || c->variant == 1) mom_slots = MOM_SLOTS;
    P.mom_slots = mom_slots;
    // test sites per workgroup: every workgroup loads its R slice into LDS first (52 KB at n = 100, 103 KB at n = 200), so large scans
    // give each wave two groups of 16 (128 test sites per 4 waves: +0.6 % at config 3, +2 % at config 5 over 64; 256 gains nothing more)
    int spb = J ? (s->M >= 65536 ? 128 : 4 * J) : (s->M >= 65536 ? 32 : SITE_THREADS / WAVE);   // per-site kernel: >= one test site per wave
    if (J == 16 && spb < 64) spb = 64;
    if (J && diag_env("BMX_SPB")) spb = std::max(4 * J, atoi(diag_env("BMX_SPB")) / (4 * J) * (4 * J));   // experiments
    const bool use_lds = fits && c->variant != 1;
    const void *fn = nullptr;
#define PICK(K) (use_lds ? (const void *)K<true> : (const void *)K<false>)
    // inner-loop form: 0 readlane / one site per step; 1 LDS broadcast + pairs; 2 = 1 + four sites per
    // step where alpha <= 1/2; 3 = 2 + power sums where alpha*max|R| <= far_eps; 4 = 3 with the per-group work prepared
    const int mode = solo ? 5 : prepared ? 4 : (v >= 5 && v <= 7) ? 0 : (v >= 8 && v <= 9) ? 1 : (v >= 10 && v <= 11) ? 2 : 3;
#define GP2(JJ, MM) (use_lds ? (const void *)clr_scan_grouped_kernel<JJ, true, MM> : (const void *)clr_scan_grouped_kernel<JJ, false, MM>)
#define GP4(JJ) (use_lds ? (const void *)clr_scan_prepared_kernel<JJ, true> : (const void *)clr_scan_prepared_kernel<JJ, false>)
#define GPICK(JJ) (mode == 4 ? GP4(JJ) : mode == 3 ? GP2(JJ, 3) : mode == 2 ? GP2(JJ, 2) : mode == 1 ? GP2(JJ, 1) : GP2(JJ, 0))
    if (J == 8) fn = GPICK(8);
    else if (J == 16) fn = GPICK(16);
    else if (J == 4) fn = GPICK(4);
    else if (solo) fn = PICK(clr_scan_solo_kernel);
    else fn = PICK(clr_scan_kernel);
#undef GP2
#undef GP4
#undef GPICK
#undef PICK
    // One wave per SIMD issues FP64 at half rate (measured), so a workgroup whose LDS footprint
    // allows only one resident workgroup per CU gets 8 waves instead of 4.
    int threads = SCAN_THREADS;
    size_t lds_bytes = (use_lds ? lds + (J ? lds_rm : 0) : 0) + (J ? (size_t)(threads / WAVE) * wave_bytes(mom_slots) : 0);
    if (J && 2 * lds_bytes > (size_t)LDS_LIMIT_BYTES) {
        threads = SCAN_THREADS_MAX;
        lds_bytes = (use_lds ? lds + lds_rm : 0) + (size_t)(threads / WAVE) * wave_bytes(mom_slots);
        spb *= 2;
    }
    if (prepared && J == 8 && use_lds) {
        // three waves per SIMD: ONE workgroup of twelve waves per CU (its registers allow it, see the kernel), if slice + twelve rings fit
        const size_t lds12 = lds + (size_t)(SCAN_THREADS_J8 / WAVE) * wave_bytes(mom_slots);
        if (lds12 <= (size_t)LDS_LIMIT_BYTES) {
            if (threads == SCAN_THREADS_MAX) spb /= 2;
            threads = SCAN_THREADS_J8;
            lds_bytes = lds12;
            spb = spb * 3;                 // the same number of groups per wave as with four waves
        }
    }
    if (!J) {       // per-site kernels: 16 waves, the R slice (if it fits) + 1 KB of scratch list (solo: 4.3 KB of ring) per wave
        threads = SITE_THREADS;
        lds_bytes = (use_lds ? lds : 0) + (size_t)(threads / WAVE) * (solo ? wave_bytes(0) : SITE_SCR * sizeof(ScratchEnt));
    }
    if (lds_bytes > (size_t)LDS_LIMIT_BYTES) return fail(BMX_E_LIMIT, "LDS budget exceeded");
    if (lds_bytes) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    P.sites_per_block = spb;
    pl.fn = fn; pl.J = solo ? 1 : J; pl.threads = threads; pl.lds_bytes = lds_bytes; pl.spb = spb; pl.use_lds = use_lds;
    pl.mode = solo ? 5 : J ? mode : -1;
    // test sites per launch: keeps the per-slice winners (16 B x slices per test site) within ~512 MB and the grid
    // within 2^31 workgroups; a multiple of the workgroup's share, so ranges cut the test sites where workgroups do
    int64_t range = std::max<int64_t>((int64_t)(512u << 20) / (16 * (int64_t)c->nslices), spb);
    range = std::min<int64_t>(range, (int64_t)0x7fffff00LL / c->nslices * spb);
    range = std::max<int64_t>(range / spb, 1) * spb;
    pl.range = range;
    if (prepared || solo) {
        pl.thr_in_lds = c->rows <= PREP_THR_LDS_MAX ? 1 : 0;
        // the moment slots of prep_kernel: 64 with the table in LDS, all of them otherwise (see above) -- but no more than any A
        // uses: the moment arrays are most of that kernel's LDS, and LDS per wave decides how many of its waves a CU holds
        const int pm = std::max(1, std::min(use_lds ? MOM_SLOTS_LDS : MOM_SLOTS, (prepared && J == 16) ? s->kmom_max_ser : s->kmom_max));
        P.mom_slots = pm;
        const size_t mom_fill = solo ? (size_t)(pm + S_COPIES - 1 + 3) * S_ORDER : 2 * (size_t)(pm + P_COPIES - 1 + 3) * P_ORDER + WAVE;
        const size_t mom_count = solo ? mom_fill : 2 * (size_t)(pm + P_COPIES - 1 + 3) + WAVE;      // grouped counting pass: one flag per slot
        pl.prep_threads = mom_fill * sizeof(double) > 20480 ? PREP_THREADS / 4 : PREP_THREADS;          // (all 254 slots: 25 KB per moment array)
        const size_t thr_b = pl.thr_in_lds ? (size_t)((c->rows + 1) & ~1) * sizeof(double) : 0;
        const size_t stage_b = solo ? (size_t)(SOLO_S0 + 2 * SOLO_S1) * sizeof(double) : 2 * SER_CAP * sizeof(ScratchEnt);   // near-entry staging / series entries
        pl.prep_lds = thr_b + (size_t)(pl.prep_threads / WAVE) * (mom_fill * sizeof(double) + stage_b);
        pl.prep_lds_count = thr_b + (size_t)(pl.prep_threads / WAVE) * (mom_count * sizeof(double) + (solo ? stage_b : 0));
        P.far_bits = (float)(P_EPS * 1.4427 * 1.1);          // |log1p(x)| <= 1.09 |x| for |x| <= 0.15 (1.16 at 0.25: order 16 uses 1.2)
        if (P_ORDER == 16) P.far_bits = (float)(P_EPS * 1.4427 * 1.2);
#define PP(JJ, FF) (const void *)prep_kernel<JJ, FF>
        if (solo) {
            pl.prep_count = (const void *)prep_solo_kernel<false>;
            pl.prep_fill = (const void *)prep_solo_kernel<true>;
        } else {
            pl.prep_count = J == 16 ? PP(16, false) : J == 8 ? PP(8, false) : PP(4, false);
            pl.prep_fill = J == 16 ? PP(16, true) : J == 8 ? PP(8, true) : PP(4, true);
        }
#undef PP
        if (pl.prep_lds > (size_t)LDS_LIMIT_BYTES) return fail(BMX_E_LIMIT, "LDS budget of the preparation kernel exceeded");
        HIP_TRY(hipFuncSetAttribute(pl.prep_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.prep_lds_count));
        HIP_TRY(hipFuncSetAttribute(pl.prep_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.prep_lds));
    }
    TRACE("scan plan: lds=%zu use_lds=%d span_hi=%d spb=%d J=%d threads=%d range=%lld mode=%d", lds_bytes, (int)use_lds, c->span_hi, spb, J, threads,
          (long long)range, pl.mode);
    return BMX_OK;
}

PrepParams prep_params(bmx_ctx *c, ChromSlot *s, const ScanPlan &pl) {
    PrepParams Q;
    Q.genpos = s->genpos.p; Q.row = pl.P.row; Q.N = s->N;
    Q.rows = c->rows; Q.rowmul = pl.use_lds ? WAVE : c->NP;
    Q.A = c->d_A; Q.nA = c->nA;
    Q.test_gen = s->test_gen.p; Q.win_lo = s->win_lo.p; Q.win_hi = s->win_hi.p; Q.center = s->center.p; Q.center_hi = s->center_hi.p;
    Q.M = s->M; Q.zcut = c->zcut;
    Q.rowthr = s->rowthr.p; Q.thr_in_lds = pl.thr_in_lds;
    // series entries only where they are cheaper than the product: groups of 16 (a product entry costs 1.5 instructions per test site)
    const bool series = pl.mode == 4 && pl.J == 16;
    Q.kmom = s->kmom.p + (series ? c->nA : 0); Q.row_of_slot = s->d_row_of_slot.p; Q.mom_slots = pl.P.mom_slots;
    Q.ser_cap = !series ? 0 : diag_env("BMX_SER_CAP") ? std::min(std::max(atoi(diag_env("BMX_SER_CAP")), 0), SER_CAP) : SER_CAP;
    Q.g_begin = 0; Q.g_end = 0;
    Q.blob_units = s->blob_units.p; Q.blob_prefix = s->blob_prefix.p; Q.prefix_base = 0;
    Q.arena = c->arena.p; Q.status = c->d_status;
    return Q;
}

// Prepared pipeline, once per (model, sites, tests, variant): the counting pass over all groups, the prefix of the blob
// sizes, and the launch ranges -- cut where the per-slice winner arrays (pl.range) or the arena would overflow.
int ensure_prep(bmx_ctx *c, ChromSlot *s) {
    ScanPlan &pl = s->plan;
    if (pl.mode < 4 || s->prep_ok) return BMX_OK;
    const int J = pl.J;
    const int64_t ngroups = (s->M + J - 1) / J;
    HIP_TRY(s->blob_units.ensure((size_t)ngroups));
    HIP_TRY(s->blob_prefix.ensure((size_t)ngroups + 1));
    PrepParams Q = prep_params(c, s, pl);
    Q.g_begin = 0; Q.g_end = ngroups;
    const int gpw = pl.prep_threads / WAVE;
    void *kargs[] = {&Q};
    {
        std::lock_guard<std::mutex> launch_lock(g_launch_mu);
        if (pl.prep_lds_count) HIP_TRY(hipFuncSetAttribute(pl.prep_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.prep_lds_count));
        HIP_TRY(hipLaunchKernel(pl.prep_count, dim3((unsigned)((ngroups + gpw - 1) / gpw)), dim3(pl.prep_threads), kargs, pl.prep_lds_count, c->stream));
    }
    hipLaunchKernelGGL(prefix_kernel, dim3(1), dim3(1024), 0, c->stream, (const int32_t *)s->blob_units.p, ngroups, s->blob_prefix.p);
    HIP_TRY(hipGetLastError());
    std::vector<int64_t> pre((size_t)ngroups + 1);
    HIP_TRY(hipMemcpyAsync(pre.data(), s->blob_prefix.p, pre.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // arena budget: a quarter of what is free now (plus what the arena already holds), at most 16 GiB
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const int64_t cap_units = (int64_t)(std::min<size_t>((free_b + c->arena.cap * sizeof(ScratchEnt)) / 4, (size_t)16 << 30) / sizeof(ScratchEnt));
    const int64_t gstep = std::max<int64_t>(pl.spb / J, 1);       // groups per workgroup of the consumer
    s->ranges.clear();
    int64_t need = 0;
    for (int64_t g0 = 0; g0 < ngroups;) {
        int64_t g1 = g0;
        while (g1 < ngroups) {
            const int64_t g2 = std::min(g1 + gstep, ngroups);
            if (g1 > g0 && ((g2 - g0) * J > pl.range || pre[(size_t)g2] - pre[(size_t)g0] > cap_units)) break;
            g1 = g2;
        }
        const int64_t units = pre[(size_t)g1] - pre[(size_t)g0];
        if (units > cap_units) return fail(BMX_E_LIMIT, "prepared scan: one workgroup's share of the stream does not fit the device memory left");
        s->ranges.push_back(PrepRange{g0 * J, std::min(g1 * J, s->M) - g0 * J, g0, g1 - g0, pre[(size_t)g0], units});
        need = std::max(need, units);
        g0 = g1;
    }
    HIP_TRY(c->arena.ensure((size_t)need + 8 * WAVE));     // the consumers' read-ahead runs up to five chunks past a blob's end
    TRACE("prepared: %lld groups, %.1f MB of blobs, %zu launch range(s), arena %.1f MB", (long long)ngroups, (double)pre[(size_t)ngroups] * 16e-6,
          s->ranges.size(), (double)c->arena.cap * 16e-6);
    s->prep_ok = true;
    return BMX_OK;
}

int ensure_plan(bmx_ctx *c, ChromSlot *s) {
    if (!s->plan_ok || s->plan_variant != c->variant) {
        s->plan_ok = false;
        s->prep_ok = false;
        int rc = plan_scan(c, s, s->plan);
        if (rc) return rc;
        s->plan_ok = true;
        s->plan_variant = c->variant;
    }
    return ensure_prep(c, s);
}

// scan + finalize of test sites [off, off + cnt) on the context's stream (asynchronous)
int launch_range(bmx_ctx *c, ChromSlot *s, ScanPlan &pl, int64_t off, int64_t cnt, const PrepRange *pr) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    ScanParams P = pl.P;
    const size_t np = (size_t)cnt * c->nslices;
    HIP_TRY(c->part_T.ensure(np));
    HIP_TRY(c->part_lin.ensure(np));
    HIP_TRY(c->part_ns.ensure(np));
    P.test_gen = s->test_gen.p + off; P.win_lo = s->win_lo.p + off; P.win_hi = s->win_hi.p + off;
    P.center = s->center.p + off; P.center_hi = s->center_hi.p + off; P.M = cnt;
    P.part_T = c->part_T.p; P.part_lin = c->part_lin.p; P.part_ns = c->part_ns.p;
    if (pl.lds_bytes) HIP_TRY(hipFuncSetAttribute(pl.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
    int64_t blocks = (cnt + pl.spb - 1) / pl.spb * c->nslices;
    if (pr && (pl.mode == 5 ? BMX_SOLO_XCD_MAP : BMX_XCD_MAP)) blocks = ((cnt + pl.spb - 1) / pl.spb + 7) / 8 * 8 * c->nslices;     // chunks padded to whole XCD rounds
    if (pr) {
        // the range's blobs: filled by the per-group kernel, then consumed by one wave per (group, slice)
        if ((size_t)pr->units + 8 * WAVE > c->arena.cap) HIP_TRY(c->arena.ensure((size_t)pr->units + 8 * WAVE));
        PrepParams Q = prep_params(c, s, pl);
        Q.g_begin = pr->g0; Q.g_end = pr->g0 + pr->ng; Q.prefix_base = pr->pbase;
        const int gpw = pl.prep_threads / WAVE;
        void *qargs[] = {&Q};
        // (the dynamic-LDS limit is an attribute of the FUNCTION, shared by every slot and context; another slot's plan may have
        // lowered it since this one was made)
        if (pl.prep_lds) HIP_TRY(hipFuncSetAttribute(pl.prep_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.prep_lds));
        HIP_TRY(hipLaunchKernel(pl.prep_fill, dim3((unsigned)((pr->ng + gpw - 1) / gpw)), dim3(pl.prep_threads), qargs, pl.prep_lds, c->stream));
        PrepView V;
        V.arena = c->arena.p; V.blob_prefix = s->blob_prefix.p; V.prefix_base = pr->pbase; V.grp_base = pr->g0; V.status = c->d_status;
        void *kargs[] = {&P, &V};
        HIP_TRY(hipLaunchKernel(pl.fn, dim3((unsigned)blocks), dim3(pl.threads), kargs, pl.lds_bytes, c->stream));
    } else {
        void *kargs[] = {&P};
        HIP_TRY(hipLaunchKernel(pl.fn, dim3((unsigned)blocks), dim3(pl.threads), kargs, pl.lds_bytes, c->stream));
    }
    FinalParams F;
    F.part_T = c->part_T.p; F.part_lin = c->part_lin.p; F.part_ns = c->part_ns.p;
    F.nslices = c->nslices; F.npairs = c->npairs; F.M = cnt; F.N = s->N;
    F.genpos = s->genpos.p; F.A = c->d_A; F.test_gen = P.test_gen; F.win_lo = P.win_lo; F.win_hi = P.win_hi;
    F.center = P.center; F.center_hi = P.center_hi; F.zcut = c->zcut;
    F.clr = s->clr.p + off; F.lin = s->lin.p + off; F.nsites = s->nsites.p + off; F.rec = s->rec.p + off;
    const int fthreads = 256;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((cnt + fthreads - 1) / fthreads)), dim3(fthreads), 0, c->stream, F);
    HIP_TRY(hipGetLastError());
    return BMX_OK;
}

// the launch ranges of a slot's scan: the prepared pipeline's own, or plain cuts of pl.range test sites
int scan_ranges(bmx_ctx *c, ChromSlot *s, std::vector<PrepRange> &out) {
    int rc = ensure_plan(c, s);
    if (rc) return rc;
    out.clear();
    if (s->plan.mode >= 4) { out = s->ranges; return BMX_OK; }
    for (int64_t off = 0; off < s->M; off += s->plan.range)
        out.push_back(PrepRange{off, std::min(s->plan.range, s->M - off), 0, 0, 0, 0});
    return BMX_OK;
}

// the device-side error bits of the prepared pipeline, checked wherever results leave the library
int check_status(bmx_ctx *c) {
    int st = 0;
    HIP_TRY(hipMemcpy(&st, c->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        HIP_TRY(hipMemset(c->d_status, 0, sizeof(int)));
        return fail(BMX_E_HIP, st & 1 ? "prepared scan: a group's stream differs in size from the counting pass" : "prepared scan: bad zone header in a group's stream");
    }
    return BMX_OK;
}

}  // namespace

extern "C" {

int bmx_ctx_scan(bmx_ctx *c) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    ChromSlot *s = c->cur;
    if (!c->has_model || !s->has_sites || !s->has_tests) return fail(BMX_E_STATE, "model, sites and tests must be set before scan");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<PrepRange> rs;
    int rc = scan_ranges(c, s, rs);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev0, c->stream));
    for (const PrepRange &r : rs)
        if ((rc = launch_range(c, s, s->plan, r.off, r.cnt, s->plan.mode >= 4 ? &r : nullptr))) return rc;
    HIP_TRY(hipEventRecord(s->ev1, c->stream));
    s->timed = true;
    return BMX_OK;
}

/* What the scan of the selected slot will launch (valid once the test sites are set): group size J (0: one test site per
 * wave), 1 if the R slice is read from LDS, the inner-loop mode (4: prepared pipeline, 0..3: round-2 grouped forms, -1:
 * per-site kernel), and the bytes of the prepared stream of all test sites (0 otherwise).  Any pointer may be NULL. */
int bmx_ctx_plan(bmx_ctx *c, int32_t *J, int32_t *use_lds, int32_t *mode, int64_t *stream_bytes) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    ChromSlot *s = c->cur;
    if (!c->has_model || !s->has_sites || !s->has_tests) return fail(BMX_E_STATE, "model, sites and tests must be set before the plan exists");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_plan(c, s);
    if (rc) return rc;
    if (J) *J = s->plan.J;
    if (use_lds) *use_lds = s->plan.use_lds ? 1 : 0;
    if (mode) *mode = s->plan.mode;
    if (stream_bytes) {
        int64_t u = 0;
        if (s->plan.mode >= 4) for (const PrepRange &r : s->ranges) u += r.units;
        *stream_bytes = u * (int64_t)sizeof(ScratchEnt);
    }
    return BMX_OK;
}

int bmx_ctx_launch_ranges(bmx_ctx *c, int64_t *offs, int32_t cap, int32_t *n_out) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    ChromSlot *s = c->cur;
    if (!c->has_model || !s->has_sites || !s->has_tests) return fail(BMX_E_STATE, "model, sites and tests must be set before the plan exists");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<PrepRange> rs;
    int rc = scan_ranges(c, s, rs);
    if (rc) return rc;
    if (n_out) *n_out = (int32_t)rs.size();
    for (size_t i = 0; i < rs.size() && (int64_t)i < cap && offs; i++) offs[i] = rs[i].off;
    return BMX_OK;
}

int bmx_ctx_sync(bmx_ctx *c) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
#if defined(BMX_PROFILE) || defined(BMX_COUNT)
    if (c->d_prof) {
        unsigned long long h[32];
        HIP_TRY(hipMemcpy(h, c->d_prof, sizeof h, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(c->d_prof, 0, sizeof h));
#ifdef BMX_COUNT
        static const char *cn[16] = {"bulk passes with sites", "passes with a near list", "quad blocks", "pair blocks", "near-list sites",
                                     "far (moment) sites", "generic passes with sites", "generic passes", "zones folded/flushed",
                                     "fold slots non-empty", "fold slots scanned (kmom)", "ragged zones", "generic walk passes",
                                     "bulk pass iterations", "bulk sites", "zones"};
        for (int k = 0; k < 16; k++) fprintf(stderr, "[bmx count] %-28s %llu\n", cn[k], h[16 + k]);
#endif
        double tot = 0;
        for (int k = 0; k < 12; k++) tot += (double)h[k];
        static const char *nm[12] = {"between test sites", "zone set-up", "per-pass: rank, list position, tail", "near-list block loops", "ragged-end masks",
                                    "fold of moments", "flush", "generic walks past zones", "best-tracking", "near-list set-up",
                                    "per-pass: loads, exp, classify", "per-pass: moment adds"};
        static const char *np[12] = {"between test sites", "zone header", "pair lists", "quad lists", "generic walks past zones", "fold of moments",
                                    "series entries", "ragged end + Horner", "exp + apply", "renormalise + best-tracking", "group set-up", "winners out"};
        const bool prep_names = diag_env("BMX_PROF_PREPARED") != nullptr;
        if (tot > 0) for (int k = 0; k < 12; k++) fprintf(stderr, "[bmx prof] %-36s %5.1f %%\n", (prep_names ? np : nm)[k], 100.0 * (double)h[k] / tot);
    }
#endif
    return check_status(c);
}

/* The records of every slot that holds scan results, in slot order, back to back: the whole genome's results with one
 * call (and, on a device buffer, what ONE gather then moves to the writing rank).  dst: room for `cap` records, in host
 * memory (dst_on_device = 0) or device memory of this context's GPU (1); n_out: records written.  Blocks until done. */
int bmx_ctx_pack_records(bmx_ctx *c, void *dst, int64_t cap, int32_t dst_on_device, int64_t *n_out) {
    if (!c || (!dst && cap > 0)) return fail(BMX_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    int64_t n = 0;
    for (ChromSlot *s : c->slots)
        if (s && s->has_tests && s->timed) n += s->M;
    if (n_out) *n_out = n;
    if (n > cap) return fail(BMX_E_INVALID, "pack_records: destination too small");
    int64_t at = 0;
    for (ChromSlot *s : c->slots) {
        if (!s || !s->has_tests || !s->timed) continue;
        HIP_TRY(hipMemcpyAsync((bmx_record *)dst + at, s->rec.p, (size_t)s->M * sizeof(bmx_record),
                               dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        at += s->M;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int bmx_ctx_copy_records(bmx_ctx *c, void *dst_device, int64_t cap) {
    if (!c || !dst_device) return fail(BMX_E_INVALID, "NULL argument");
    ChromSlot *s = c->cur;
    if (!s->has_tests || !s->timed) return fail(BMX_E_STATE, "no scan results in the selected slot");
    if (cap < s->M) return fail(BMX_E_INVALID, "copy_records: destination too small");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst_device, s->rec.p, (size_t)s->M * sizeof(bmx_record), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int bmx_ctx_last_scan_ms(bmx_ctx *c, double *ms) {
    if (!c || !ms) return fail(BMX_E_INVALID, "NULL argument");
    ChromSlot *s = c->cur;
    if (!s->timed) return fail(BMX_E_STATE, "no scan has been launched");
    HIP_TRY(hipEventSynchronize(s->ev1));
    float f = 0;
    HIP_TRY(hipEventElapsedTime(&f, s->ev0, s->ev1));
    *ms = f;
    return BMX_OK;
}

int bmx_ctx_result_ptrs(bmx_ctx *c, void **d_clr, void **d_lin, void **d_nsites) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    ChromSlot *s = c->cur;
    if (!s->has_tests || !s->timed) return fail(BMX_E_STATE, "no scan results: call bmx_ctx_scan after bmx_ctx_set_tests");
    if (d_clr) *d_clr = s->clr.p;
    if (d_lin) *d_lin = s->lin.p;
    if (d_nsites) *d_nsites = s->nsites.p;
    return BMX_OK;
}

int bmx_ctx_records(bmx_ctx *c, void **d_rec) {
    if (!c || !d_rec) return fail(BMX_E_INVALID, "NULL argument");
    ChromSlot *s = c->cur;
    if (!s->has_tests || !s->timed) return fail(BMX_E_STATE, "no scan results: call bmx_ctx_scan after bmx_ctx_set_tests");
    *d_rec = s->rec.p;
    return BMX_OK;
}

int bmx_ctx_fetch_records(bmx_ctx *c, bmx_record *rec) {
    if (!c || !rec) return fail(BMX_E_INVALID, "NULL argument");
    ChromSlot *s = c->cur;
    if (!s->has_tests || !s->timed) return fail(BMX_E_STATE, "no scan results to fetch");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int st = check_status(c)) return st;
    HIP_TRY(hipMemcpy(rec, s->rec.p, (size_t)s->M * sizeof(bmx_record), hipMemcpyDeviceToHost));
    return BMX_OK;
}

int bmx_ctx_fetch(bmx_ctx *c, double *clr, int32_t *ix, int32_t *ia, int32_t *iA, int32_t *nsites) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    ChromSlot *s = c->cur;
    if (!s->has_tests || !s->timed) return fail(BMX_E_STATE, "no scan results to fetch");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<int32_t> lin((size_t)s->M);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int st = check_status(c)) return st;
    if (clr) HIP_TRY(hipMemcpy(clr, s->clr.p, (size_t)s->M * sizeof(double), hipMemcpyDeviceToHost));
    if (nsites) HIP_TRY(hipMemcpy(nsites, s->nsites.p, (size_t)s->M * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lin.data(), s->lin.p, (size_t)s->M * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int32_t npairs = c->npairs, nab = c->nab;
    parallel_ranges(s->M, 1 << 18, [&](int, int64_t b, int64_t e) {
        for (int64_t t = b; t < e; t++) {
            const int32_t L = lin[(size_t)t];
            int32_t a = -1, bb = -1, d = -1;
            if (L >= 0) {
                d = L / npairs;
                const int32_t p = L % npairs;
                a = p / nab;
                bb = p % nab;
            }
            if (ix) ix[t] = a;
            if (ia) ia[t] = bb;
            if (iA) iA[t] = d;
        }
    });
    return BMX_OK;
}

int bmx_ctx_fetch_lut(bmx_ctx *c, double *psel_out, double *R_out) {
    if (!c) return fail(BMX_E_INVALID, "ctx is NULL");
    if (!c->has_model) return fail(BMX_E_STATE, "no model set");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    size_t tab = (size_t)c->npairs * c->rows * sizeof(double);
    if (psel_out) HIP_TRY(hipMemcpy(psel_out, c->d_psel, tab, hipMemcpyDeviceToHost));
    if (R_out) HIP_TRY(hipMemcpy(R_out, c->d_R, tab, hipMemcpyDeviceToHost));
    return BMX_OK;
}

int bmx_ctx_surface(bmx_ctx *c, double test_gen, int64_t win_lo, int64_t win_hi, double *T_out, int32_t *nsites_out) {
    if (!c || !T_out) return fail(BMX_E_INVALID, "NULL argument");
    ChromSlot *s = c->cur;
    if (!c->has_model || !s->has_sites) return fail(BMX_E_STATE, "model and sites must be set before surface");
    HIP_TRY(hipSetDevice(c->device));
    SurfParams S;
    S.genpos = s->genpos.p; S.row = RowArray{s->wide_rows ? nullptr : s->row16.p, s->wide_rows ? s->row32.p : nullptr}; S.N = s->N; S.Rt = c->d_Rt; S.NP = c->NP; S.npairs = c->npairs;
    S.nslices = c->nslices; S.A = c->d_A; S.nA = c->nA; S.tg = test_gen;
    S.lo = std::max<int64_t>(win_lo, 0); S.hi = std::min<int64_t>(win_hi, s->N - 1); S.zcut = c->zcut;
    HIP_TRY(c->surf_T.ensure((size_t)c->nA * c->npairs));
    HIP_TRY(c->surf_ns.ensure((size_t)c->nA));
    S.T = c->surf_T.p; S.ns = c->surf_ns.p;
    hipLaunchKernelGGL(surface_kernel, dim3((unsigned)(c->nA * c->nslices)), dim3(WAVE), 0, c->stream, S);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(T_out, c->surf_T.p, (size_t)c->nA * c->npairs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (nsites_out) HIP_TRY(hipMemcpyAsync(nsites_out, c->surf_ns.p, (size_t)c->nA * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BMX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------- RCCL gather
// librccl.so through dlopen: only these entry points, resolved once
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

struct bmx_comm {
    bmx_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf<bmx_record> send, recv;
};

namespace {

RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("librccl.so cannot be opened: ") + dlerror(); return; }
        auto sym = [&](const char *n) -> void * {
            void *f = dlsym(api.handle, n);
            if (!f && api.error.empty()) api.error = std::string("librccl.so lacks ") + n;
            return f;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &api;
}

#define RCCL_TRY(api, expr)                                                                                        \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fail(BMX_E_HIP, std::string(#expr) + ": " + (api)->GetErrorString(r_));      \
    } while (0)

}  // namespace

extern "C" {

int bmx_comm_unique_id(char *id) {
    if (!id) return fail(BMX_E_INVALID, "id is NULL");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(BMX_E_HIP, api->error);
    static_assert(sizeof(ncclUniqueId) == BMX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    RCCL_TRY(api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return BMX_OK;
}

int bmx_comm_create(bmx_comm **out, bmx_ctx *c, const char *id, int32_t rank, int32_t world) {
    if (!out || !c || !id) return fail(BMX_E_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(BMX_E_INVALID, "rank / world out of range");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(BMX_E_HIP, api->error);
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    RCCL_TRY(api, api->CommInitRank(&comm, world, u, rank));
    bmx_comm *cm = new bmx_comm();
    cm->ctx = c; cm->comm = comm; cm->rank = rank; cm->world = world;
    *out = cm;
    return BMX_OK;
}

void bmx_comm_destroy(bmx_comm *cm) {
    if (!cm) return;
    (void)hipSetDevice(cm->ctx->device);
    RcclApi *api = rccl_api();
    if (cm->comm && api->CommDestroy) (void)api->CommDestroy(cm->comm);
    cm->send.release();
    cm->recv.release();
    delete cm;
}

int bmx_comm_gather_records(bmx_comm *cm, const int64_t *counts, int32_t root, bmx_record *dst_host, void **d_out) {
    if (!cm || !counts) return fail(BMX_E_INVALID, "NULL argument");
    if (root < 0 || root >= cm->world) return fail(BMX_E_INVALID, "root out of range");
    RcclApi *api = rccl_api();
    bmx_ctx *c = cm->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int64_t total = 0, my_off = 0;
    for (int r = 0; r < cm->world; r++) {
        if (counts[r] < 0) return fail(BMX_E_INVALID, "negative record count");
        if (r < cm->rank) my_off += counts[r];
        total += counts[r];
    }
    const int64_t mine = counts[cm->rank];
    const bool is_root = cm->rank == root;
    // this rank's records, packed: straight into the root's receive buffer at the rank's own offset, or into the send buffer
    bmx_record *pack_to = nullptr;
    if (is_root) {
        HIP_TRY(cm->recv.ensure((size_t)total));
        pack_to = cm->recv.p + my_off;
    } else {
        HIP_TRY(cm->send.ensure((size_t)mine));
        pack_to = cm->send.p;
    }
    int64_t n = 0;
    int rc = bmx_ctx_pack_records(c, pack_to, mine, 1, &n);
    if (rc) return rc;
    if (n != mine) return fail(BMX_E_INVALID, "gather_records: counts[rank] differs from the records this context holds");
    // one group: the root posts a receive per peer (each at that rank's offset), every peer one send
    RCCL_TRY(api, api->GroupStart());
    if (is_root) {
        int64_t off = 0;
        for (int r = 0; r < cm->world; r++) {
            if (r != root && counts[r] > 0)
                RCCL_TRY(api, api->Recv(cm->recv.p + off, (size_t)counts[r] * sizeof(bmx_record), ncclChar, r, cm->comm, c->stream));
            off += counts[r];
        }
    } else if (mine > 0) {
        RCCL_TRY(api, api->Send(cm->send.p, (size_t)mine * sizeof(bmx_record), ncclChar, root, cm->comm, c->stream));
    }
    RCCL_TRY(api, api->GroupEnd());
    if (is_root && dst_host && total > 0)
        HIP_TRY(hipMemcpyAsync(dst_host, cm->recv.p, (size_t)total * sizeof(bmx_record), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (d_out) *d_out = is_root ? (void *)cm->recv.p : nullptr;
    return BMX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------- streaming output
// The reference writes each row as soon as calcBaller returns it (BalLeRMix+_v1.py:599-608).  Here the test sites are
// scanned in chunks on the context's stream; chunk i's results travel to pinned host memory on a second stream and are
// formatted and appended to the output file by a writer thread while chunk i+1 is being scanned.
#include <condition_variable>
#include <deque>
#include <mutex>

extern "C" int bmx_ctx_scan_write(bmx_ctx *c, const char *path, const int64_t *phys, const double *gen,
                                  const char *xs, int nx, const char *abs_, int nab, const char *As, int nA, int64_t chunk) {
    if (!c || !path || !phys || !gen || !xs || !abs_ || !As) return fail(BMX_E_INVALID, "NULL argument");
    ChromSlot *s = c->cur;
    if (!c->has_model || !s->has_sites || !s->has_tests) return fail(BMX_E_STATE, "model, sites and tests must be set before scan");
    if (nx != c->nx || nab != c->nab || nA != c->nA) return fail(BMX_E_INVALID, "printed grids do not match the model's grids");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<PrepRange> rs;
    int rc = scan_ranges(c, s, rs);
    if (rc) return rc;
    ScanPlan &pl = s->plan;
    // chunks: the prepared pipeline's launch ranges as they are (their blobs were sized per range); otherwise `chunk` test
    // sites at a time (0: 65536; whole workgroups, which keeps every result bit-identical to bmx_ctx_scan)
    std::vector<PrepRange> chunks;
    if (pl.mode >= 4) {
        // a prepared range is cut further into chunks of whole workgroups: each chunk's groups are a sub-range of the
        // range's blobs (same arena, same prefix base), so nothing has to be re-planned
        if (chunk <= 0) chunk = 65536;
        chunk = std::max<int64_t>(chunk / pl.spb, 1) * pl.spb;
        for (const PrepRange &r : rs)
            for (int64_t o = 0; o < r.cnt; o += chunk) {
                PrepRange q = r;
                q.off = r.off + o;
                q.cnt = std::min(chunk, r.cnt - o);
                chunks.push_back(q);
            }
    } else {
        if (chunk <= 0) chunk = 65536;
        chunk = std::min<int64_t>(std::max<int64_t>(chunk / pl.spb, 1) * pl.spb, pl.range);
        for (int64_t off = 0; off < s->M; off += chunk) chunks.push_back(PrepRange{off, std::min(chunk, s->M - off), 0, 0, 0, 0});
    }
    int64_t chunk_max = 1;
    for (const PrepRange &q : chunks) chunk_max = std::max(chunk_max, q.cnt);
    const size_t slot_bytes = (size_t)chunk_max * 16;
    if (c->h_stage_cap < slot_bytes) {
        for (int k = 0; k < 2; k++) {
            if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
            c->h_stage[k] = nullptr;
        }
        c->h_stage_cap = 0;
        for (int k = 0; k < 2; k++) HIP_TRY(hipHostMalloc(&c->h_stage[k], slot_bytes, hipHostMallocDefault));
        c->h_stage_cap = slot_bytes;
    }
    bmx_row_tables_ *tabs = bmx_row_tables_new_(xs, nx, abs_, nab, As, nA);
    FILE *f = fopen(path, "a");
    if (!f) {
        bmx_row_tables_free_(tabs);
        return fail(BMX_E_INVALID, std::string("cannot open ") + path + ": " + strerror(errno));
    }
    struct Job { int slot; int64_t off, cnt; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    bool slot_free[2] = {true, true}, done = false;
    int werr = 0;                  // 1: event wait failed, 2: formatting/writing failed
    std::string wmsg;              // the writer thread's own message (bmx_last_error is thread-local)
    std::thread writer([&]() {
        (void)hipSetDevice(c->device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !jobs.empty() || done; });
                if (jobs.empty()) return;
                j = jobs.front();
                jobs.pop_front();
            }
            int e = 0;
            std::string msg;
            if (hipEventSynchronize(c->ev_copied[j.slot]) != hipSuccess) e = 1;
            if (!e && !werr) {
                const char *base = (const char *)c->h_stage[j.slot];
                const double *hclr = (const double *)base;
                const int32_t *hlin = (const int32_t *)(base + (size_t)chunk_max * 8);
                const int32_t *hns = (const int32_t *)(base + (size_t)chunk_max * 12);
                if (bmx_write_chunk_(f, tabs, j.cnt, phys + j.off, gen + j.off, hclr, nullptr, nullptr, nullptr, hlin, hns)) {
                    e = 2;
                    msg = bmx_last_error();       // set on THIS thread by bmx_write_chunk_
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (e && !werr) { werr = e; wmsg = msg; }
                slot_free[j.slot] = true;
            }
            cv.notify_all();
        }
    });
    bool all_launched = false;
    auto finish = [&](int code, const std::string &msg) {
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
        writer.join();
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->copy_stream);
        const bool wfail = fclose(f) != 0;
        bmx_row_tables_free_(tabs);
        s->timed = all_launched && !werr && !code;     // results are fetchable only when every chunk was scanned
        if (code) return fail(code, msg);
        if (werr == 1) return fail(BMX_E_HIP, "streaming writer: waiting for a result copy failed");
        if (werr == 2) return fail(BMX_E_INVALID, std::string("streaming writer: ") + wmsg);
        if (wfail) return fail(BMX_E_INVALID, "write failed");
        return check_status(c);
    };
#define STREAM_TRY(expr)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return finish(BMX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    STREAM_TRY(hipEventRecord(s->ev0, c->stream));
    size_t k = 0;
    bool stopped = false;
    for (; k < chunks.size(); ++k) {
        const PrepRange &q = chunks[k];
        const int slot = (int)(k & 1);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return slot_free[slot]; });
            if (werr) { stopped = true; break; }
            slot_free[slot] = false;
        }
        if (pl.mode >= 4) {
            // the chunk's groups within its range: fill + consume exactly those
            PrepRange sub = q;
            sub.g0 = q.off / pl.J;
            sub.ng = (q.cnt + pl.J - 1) / pl.J;
            if ((rc = launch_range(c, s, pl, q.off, q.cnt, &sub))) return finish(rc, g_err);
        } else if ((rc = launch_range(c, s, pl, q.off, q.cnt, nullptr))) return finish(rc, g_err);
        STREAM_TRY(hipEventRecord(c->ev_done[slot], c->stream));
        STREAM_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_done[slot], 0));
        char *base = (char *)c->h_stage[slot];
        STREAM_TRY(hipMemcpyAsync(base, s->clr.p + q.off, (size_t)q.cnt * 8, hipMemcpyDeviceToHost, c->copy_stream));
        STREAM_TRY(hipMemcpyAsync(base + (size_t)chunk_max * 8, s->lin.p + q.off, (size_t)q.cnt * 4, hipMemcpyDeviceToHost, c->copy_stream));
        STREAM_TRY(hipMemcpyAsync(base + (size_t)chunk_max * 12, s->nsites.p + q.off, (size_t)q.cnt * 4, hipMemcpyDeviceToHost, c->copy_stream));
        STREAM_TRY(hipEventRecord(c->ev_copied[slot], c->copy_stream));
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs.push_back(Job{slot, q.off, q.cnt});
        }
        cv.notify_all();
    }
    STREAM_TRY(hipEventRecord(s->ev1, c->stream));
#undef STREAM_TRY
    all_launched = !stopped;
    return finish(BMX_OK, "");
}

extern "C" {

int bmx_lut_build(const bmx_model *m, double *psel_out, double *R_out, int device) {
    bmx_ctx *c = nullptr;
    int rc = bmx_ctx_create(&c, device);
    if (rc) return rc;
    const double A1 = 1.0;
    rc = bmx_ctx_set_model(c, m, &A1, 1);
    if (!rc) rc = bmx_ctx_fetch_lut(c, psel_out, R_out);
    std::string keep = g_err;
    bmx_ctx_destroy(c);
    g_err = keep;
    return rc;
}

int bmx_scan(const bmx_model *m, const double *A, int32_t nA, int64_t N, const double *genpos,
             const int32_t *row, int64_t M, const double *test_gen, const int64_t *win_lo,
             const int64_t *win_hi, double *clr, int32_t *ix, int32_t *ia, int32_t *iA,
             int32_t *nsites, int device) {
    bmx_ctx *c = nullptr;
    int rc = bmx_ctx_create(&c, device);
    if (rc) return rc;
    rc = bmx_ctx_set_model(c, m, A, nA);
    if (!rc) rc = bmx_ctx_set_sites(c, N, genpos, row);
    if (!rc) rc = bmx_ctx_set_tests(c, M, test_gen, win_lo, win_hi);
    if (!rc) rc = bmx_ctx_scan(c);
    if (!rc) rc = bmx_ctx_fetch(c, clr, ix, ia, iA, nsites);
    std::string keep = g_err;
    bmx_ctx_destroy(c);
    g_err = keep;
    return rc;
}


/* bmx_scan on several GPUs of this node, inside the library (no torch, no process group): one host thread and one context
 * per entry of `devices` (NULL: 0 .. n_devices-1; an index may repeat -- two contexts on one GPU), test sites dealt to the
 * workers in blocks of 4096 consecutive test sites round-robin -- the sharding of ballermixplus_amd/distributed.py, so every
 * row is bitwise what one GPU computes -- and each worker's results copied into the caller's buffers. */
int bmx_scan_multi(const bmx_model *m, const double *A, int32_t nA, int64_t N, const double *genpos,
                   const int32_t *row, int64_t M, const double *test_gen, const int64_t *win_lo,
                   const int64_t *win_hi, double *clr, int32_t *ix, int32_t *ia, int32_t *iA,
                   int32_t *nsites, int32_t n_devices, const int32_t *devices) {
    if (n_devices < 1 || n_devices > 64) return fail(BMX_E_INVALID, "n_devices must be 1..64");
    if (M < 1 || !test_gen || !win_lo || !win_hi) return fail(BMX_E_INVALID, "empty test-site arrays");
    constexpr int64_t BLOCK = 4096;
    const int64_t nblk = (M + BLOCK - 1) / BLOCK;
    std::vector<int> rcs((size_t)n_devices, BMX_OK);
    std::vector<std::string> msgs((size_t)n_devices);
    auto work = [&](int w) {
        // this worker's test sites: blocks w, w + n, w + 2n, ...
        std::vector<int64_t> idx;
        for (int64_t b = w; b < nblk; b += n_devices)
            for (int64_t t = b * BLOCK; t < std::min((b + 1) * BLOCK, M); ++t) idx.push_back(t);
        if (idx.empty()) return;
        const size_t n = idx.size();
        std::vector<double> tg(n), c_(n);
        std::vector<int64_t> lo(n), hi(n);
        std::vector<int32_t> x_(n), a_(n), A_(n), ns_(n);
        for (size_t i = 0; i < n; ++i) { tg[i] = test_gen[idx[i]]; lo[i] = win_lo[idx[i]]; hi[i] = win_hi[idx[i]]; }
        bmx_ctx *c = nullptr;
        int rc = bmx_ctx_create(&c, devices ? devices[w] : w);
        if (!rc) rc = bmx_ctx_set_model(c, m, A, nA);
        if (!rc) rc = bmx_ctx_set_sites(c, N, genpos, row);
        if (!rc) rc = bmx_ctx_set_tests(c, (int64_t)n, tg.data(), lo.data(), hi.data());
        if (!rc) rc = bmx_ctx_scan(c);
        if (!rc) rc = bmx_ctx_fetch(c, c_.data(), x_.data(), a_.data(), A_.data(), ns_.data());
        if (rc) msgs[(size_t)w] = bmx_last_error();          // this thread's message
        bmx_ctx_destroy(c);
        rcs[(size_t)w] = rc;
        if (rc) return;
        for (size_t i = 0; i < n; ++i) {
            const int64_t t = idx[i];
            if (clr) clr[t] = c_[i];
            if (ix) ix[t] = x_[i];
            if (ia) ia[t] = a_[i];
            if (iA) iA[t] = A_[i];
            if (nsites) nsites[t] = ns_[i];
        }
    };
    std::vector<std::thread> th;
    for (int w = 0; w < n_devices; ++w) th.emplace_back(work, w);
    for (auto &t : th) t.join();
    for (int w = 0; w < n_devices; ++w)
        if (rcs[(size_t)w]) return fail(rcs[(size_t)w], "worker " + std::to_string(w) + ": " + msgs[(size_t)w]);
    return BMX_OK;
}

}  // extern "C"

#endif  // BMX_DEVICE_PROBE
