/* bmxscan.h -- C ABI of libbmxscan.so, the MI355X (gfx950) composite-likelihood scan.
 *
 * The reference (bioXiaoheng/BallerMixPlus, BalLeRMix+_v1.py) is a single Python
 * script with no FFI of its own; the seam this library drops into is the pair of
 * Python calls on its hot path:
 *
 *   NormalizedBetaBinom(InputData, Grids, nofreq, MAF, nosub)      BalLeRMix+_v1.py:793 (class at :310-433)
 *   calcBaller(window_indice, testSite, InputData, NeutralSFS,
 *              NormalizedBetaBinom, Grids) -> [T, x, abeta, A, nSites]
 *                                                                   BalLeRMix+_v1.py:539,573,590,606 (def at :436-507)
 *
 * Every entry point is extern "C", takes plain pointers and sizes, never throws,
 * never exits the process and retains no caller memory after it returns.
 * Return value: 0 on success, a negative BMX_E_* code otherwise; the message is
 * available from bmx_last_error() (thread-local).  All host buffers are
 * caller-allocated, C-contiguous, native endian.  There is NO CPU fallback: when
 * no HIP device is usable every compute entry point fails with BMX_E_NODEVICE.
 */
#ifndef BMXSCAN_H
#define BMXSCAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMX_ABI_VERSION_MAJOR 1
#define BMX_ABI_VERSION_MINOR 5

enum {
    BMX_OK = 0,
    BMX_E_INVALID = -1,   /* bad argument (null pointer, size, unsorted positions, ...) */
    BMX_E_NODEVICE = -2,  /* no usable HIP device / device index out of range */
    BMX_E_HIP = -3,       /* a HIP runtime call failed; see bmx_last_error() */
    BMX_E_LIMIT = -4,     /* a documented size limit was exceeded (2^24 LUT rows, 2^31 sites, 2^31 grid points) */
    BMX_E_STATE = -5      /* call order violated (e.g. scan before model/sites were set) */
};

/* Which B statistic the selection table is built for.
 * Replaces the (nofreq, MAF, nosub) flag triple of NormalizedBetaBinom.__init__
 * (BalLeRMix+_v1.py:319, dispatch at :336-352). */
enum {
    BMX_STAT_B2 = 0,     /* default                      v1:351-352, normBase :399-404 */
    BMX_STAT_B2MAF = 1,  /* --MAF                        v1:344-345, :389-392, :406-413 */
    BMX_STAT_B0 = 2,     /* --noSub                      v1:348-349, :419-425 */
    BMX_STAT_B0MAF = 3,  /* --noSub --MAF                v1:341-342, :427-433 */
    BMX_STAT_B1 = 4      /* --noFreq                     v1:337-338, :378-383, :415-417 */
};

/* The model the grid search runs over.  It carries what calcBaller reads from
 * InputData / NeutralSFS / Grids, re-indexed by LUT row instead of by site:
 * a site with derived count k and sample size sizes[j] maps to row
 * row_off[j] + k  (B1: k in {0,1}).  rows = row_off[n_sizes]. */
typedef struct bmx_model {
    int32_t stat;            /* BMX_STAT_* */
    int32_t min_count;       /* InputData.minCount (v1:59,74) */
    int32_t n_sizes;         /* number of distinct sample sizes (InputData.sampSizes, v1:76) */
    const int32_t *sizes;    /* [n_sizes] sample sizes n */
    const int32_t *row_off;  /* [n_sizes+1] first LUT row of each size */
    const double *g;         /* [rows] neutral probability NeutralSFS.spect[(k,n)] (v1:208,292);
                                NaN for (k,n) absent from the helper file (never referenced by a site) */
    const double *prop;      /* [n_sizes] NeutralSFS.sampProps[n] (v1:212-214,304) */
    int32_t nx;              /* grids in the reference's ITERATION order: list(set(Grids.x)) etc. */
    const double *x;         /* [nx]   (v1:473) */
    int32_t nab;
    const double *abeta;     /* [nab]  (v1:474) */
} bmx_model;

/* One result row as the multi-GPU gather moves it (16 bytes): CLR, linear grid index
 * (iA*nx + ix)*nab + ia (-1: no grid point had T > 0) and nSites -- Tmax[0], Tmax[1:4], Tmax[4] of
 * calcBaller's return value (BalLeRMix+_v1.py:451,502,507). */
typedef struct bmx_record {
    double clr;
    int32_t lin;
    int32_t nsites;
} bmx_record;

/* ---- library-level queries -------------------------------------------------------- */
void bmx_version(int *major, int *minor);
/* Hash of the kernel/host sources this binary was built from (first 16 hex digits of their SHA-256, set by the
 * Makefile): the Python shim refuses a library whose id differs from the sources next to it. */
const char *bmx_build_id(void);
const char *bmx_last_error(void);
/* Number of visible HIP devices (0 when none / no driver). */
int bmx_device_count(void);
/* The window cut-off in the exponent domain: largest double z with exp(-z) >= 1e-8, so that the
 * reference's predicate `np.exp(-A*dist) >= 1e-8` (BalLeRMix+_v1.py:454-455) is `A*dist <= z`. */
double bmx_alpha_cut(void);

/* ---- one-shot entry points (host buffers in, host buffers out) ---------------------- */

/* Replaces NormalizedBetaBinom.__init__ (BalLeRMix+_v1.py:319-359) + get() (:362-363).
 * Runs the device lgamma / beta-binomial kernel and returns, for every grid pair
 * and LUT row,
 *     psel_out[ix][ia][row] = normProbs[(x,a)] value of a site on that row      (optional, may be NULL)
 *     R_out   [ix][ia][row] = psel * prop(n) / g(k,n) - 1                       (optional, may be NULL)
 * so that calcBaller's mixture log-ratio of one site is log1p(alpha * R) (v1:494-499). */
int bmx_lut_build(const bmx_model *m, double *psel_out, double *R_out, int device);

/* Replaces the calcBaller call of every Scan mode (BalLeRMix+_v1.py:539,573,590,606).
 *   A[nA]        linkage grid in iteration order list(set(Grids.A))                    (v1:453)
 *   genpos[N]    InputData.genPos, non-decreasing;  row[N] LUT row of each site
 *   test_gen[M]  genetic position of each test site (testSite argument, v1:436)
 *   win_lo/hi[M] inclusive index bounds of window_indice (0, N-1 for the default mode)
 * Outputs, length M:  clr = Tmax[0];  ix/ia/iA = indices of x_hat / alpha_hat / A_hat in the
 * grids passed in;  nsites = Tmax[4].  iA == -1 (with clr = 0, ix = ia = -1, nsites = 0)
 * means no grid point had T > 0: the reference prints its all-zero initial row (v1:451). */
int bmx_scan(const bmx_model *m, const double *A, int32_t nA, int64_t N, const double *genpos,
             const int32_t *row, int64_t M, const double *test_gen, const int64_t *win_lo,
             const int64_t *win_hi, double *clr, int32_t *ix, int32_t *ia, int32_t *iA,
             int32_t *nsites, int device);

/* bmx_scan on several GPUs of this node, inside the library (no Python, no torch, no process group): one host thread and
 * one context per entry of devices[n_devices] (NULL: GPUs 0 .. n_devices-1; an index may repeat), test sites dealt to the
 * workers in blocks of 4096 consecutive test sites round-robin, each worker's results copied into the caller's buffers.
 * Every output row is bitwise what bmx_scan returns on one GPU (a window's arithmetic never depends on the sharding). */
int bmx_scan_multi(const bmx_model *m, const double *A, int32_t nA, int64_t N, const double *genpos,
                   const int32_t *row, int64_t M, const double *test_gen, const int64_t *win_lo,
                   const int64_t *win_hi, double *clr, int32_t *ix, int32_t *ia, int32_t *iA,
                   int32_t *nsites, int32_t n_devices, const int32_t *devices);

/* ---- resident-context entry points ------------------------------------------------- */
/* Same computation split so that inputs stay resident in HBM across scans (one context per
 * process per GPU; the multi-GPU driver shards test sites across processes).
 * A context holds ONE model (selection table + A grid) and any number of chromosome SLOTS, each with its own site
 * arrays, test sites and results: the reference runs one input file per process (BalLeRMix+_v1.py:777-799); a
 * whole-genome run here selects slot k, sets chromosome k's sites and test sites, and scans the slots back to back on
 * the context's stream with the table built once.  Slot 0 exists from creation and is selected; set_sites, set_tests,
 * scan, fetch*, records, result_ptrs, last_scan_ms, scan_write, surface and plan act on the selected slot. */
typedef struct bmx_ctx bmx_ctx;

int bmx_ctx_create(bmx_ctx **out, int device);
void bmx_ctx_destroy(bmx_ctx *c);
/* Build the selection table on the device (K1) and keep it resident.  Call order: set_model, then
 * set_sites, then set_tests; a new model discards the site and test arrays of EVERY slot (their
 * row indices belong to the old one), and new sites discard the slot's test sites (located in the old array).
 * The table's bit pattern follows the HOST's libm: scipy evaluates lgam's log with glibc's log, which is not correctly
 * rounded on ~0.015 % of arguments, so set_model evaluates this host's log on every argument the table build will use
 * and ships the exceptions to the device (bmx_math.h LogPatch).  The device table is therefore a function of the
 * model AND of the host library -- by design: it reproduces what scipy computes on this box. */
int bmx_ctx_set_model(bmx_ctx *c, const bmx_model *m, const double *A, int32_t nA);
/* Copy the site arrays of one chromosome to the device (and rank its rows by frequency for the scan
 * kernel's far-field moments). */
int bmx_ctx_set_sites(bmx_ctx *c, int64_t N, const double *genpos, const int32_t *row);
/* Copy test sites + window bounds to the device (and locate each test site).  win_lo == win_hi == NULL: every window
 * holds all sites of the chromosome, [0, N - 1] -- the reference's default mode (Scan._alpha, BalLeRMix+_v1.py:598-610);
 * the bounds are then written on the device, nothing is copied. */
int bmx_ctx_set_tests(bmx_ctx *c, int64_t M, const double *test_gen, const int64_t *win_lo,
                      const int64_t *win_hi);
/* Launch the scan (K2 + argmax finalisation) on the context's stream; asynchronous. */
int bmx_ctx_scan(bmx_ctx *c);
/* Block until the stream is idle. */
int bmx_ctx_sync(bmx_ctx *c);
/* Milliseconds the last bmx_ctx_scan spent in the scan kernel(s), from HIP events
 * recorded on the context's stream (valid after bmx_ctx_sync). */
int bmx_ctx_last_scan_ms(bmx_ctx *c, double *ms);
/* Copy results of the last scan to host buffers of length M. */
int bmx_ctx_fetch(bmx_ctx *c, double *clr, int32_t *ix, int32_t *ia, int32_t *iA, int32_t *nsites);
/* Device addresses of the last scan's results: clr f64[M], lin i32[M] (linear grid index
 * (iA*nx + ix)*nab + ia, or -1), nsites i32[M].  Valid until the next set_tests/destroy;
 * used for the RCCL gather without a host round trip. */
int bmx_ctx_result_ptrs(bmx_ctx *c, void **d_clr, void **d_lin, void **d_nsites);
/* The same results as one array of M bmx_record: device address (for a single RCCL gather to the
 * writing rank) and host copy.  Valid until the next set_tests/destroy. */
int bmx_ctx_records(bmx_ctx *c, void **d_rec);
int bmx_ctx_fetch_records(bmx_ctx *c, bmx_record *rec);
/* Scan and stream: the rows of `scores.write(...)` (BalLeRMix+_v1.py:599-608) are appended to `path`
 * while the scan is still running.  Test sites go to the device `chunk` at a time (0: 65536; rounded to
 * whole workgroups, which keeps every result bit-identical to bmx_ctx_scan); each chunk's results are copied
 * to pinned host memory on a second stream and formatted/written by a host thread while the next chunk is
 * scanned.  phys[M], gen[M]: the first two columns of each row; xs/abs_/As: the grids' printed forms as for
 * bmx_write_rows (they must have the model's nx/nab/nA entries).  Results stay fetchable afterwards. */
int bmx_ctx_scan_write(bmx_ctx *c, const char *path, const int64_t *phys, const double *gen,
                       const char *xs, int nx, const char *abs_, int nab, const char *As, int nA, int64_t chunk);
/* Copy the resident tables back: psel/R as in bmx_lut_build (either may be NULL). */
int bmx_ctx_fetch_lut(bmx_ctx *c, double *psel_out, double *R_out);
/* Full likelihood surface of ONE test site: T_out[nA][nx][nab] = T(A, x, alpha_beta) in the grids'
 * iteration order (NaN where the window of that A is empty -- the reference `continue`s there,
 * BalLeRMix+_v1.py:458-459), nsites_out[nA] = window size per A (may be NULL).  The reference only
 * keeps the maximum and lists the surfaces as future work (v1:449-450).  Computed as a plain sum of
 * log1p(alpha*R), independently of the scan kernels' product form. */
int bmx_ctx_surface(bmx_ctx *c, double test_gen, int64_t win_lo, int64_t win_hi, double *T_out,
                    int32_t *nsites_out);
/* Choose the scan kernel variant (0 = default). For A/B measurements only. */
int bmx_ctx_set_variant(bmx_ctx *c, int variant);
/* Select (creating it on first use) chromosome slot `slot`, 0 <= slot < 4096. */
int bmx_ctx_select_slot(bmx_ctx *c, int32_t slot);
/* Number of slot indices in use (highest selected slot + 1). */
int bmx_ctx_slot_count(bmx_ctx *c);
/* The records of every slot that holds scan results, in slot order, back to back: the whole genome's rows with one
 * call -- to host memory (dst_on_device = 0), or to device memory of this context's GPU (1), where ONE gather then
 * moves them to the writing rank.  dst: room for `cap` records; *n_out (may be NULL): records written.  Blocks. */
int bmx_ctx_pack_records(bmx_ctx *c, void *dst, int64_t cap, int32_t dst_on_device, int64_t *n_out);
/* The SELECTED slot's M records to device memory of this context's GPU (room for `cap` >= M records), device to device on the
 * context's stream; blocks until done.  What a sharded run gathers per input file when the context also holds other slots. */
int bmx_ctx_copy_records(bmx_ctx *c, void *dst_device, int64_t cap);
/* What the scan of the selected slot launches (valid once its test sites are set): *J = test sites per wave-group (0: one
 * test site per wave), *use_lds = 1 if the R slice is read from LDS, *mode = 4 prepared pipeline (prep_kernel +
 * clr_scan_prepared_kernel), 5 prepared pipeline with one test site per wave (prep_solo_kernel + clr_scan_solo_kernel, *J = 1:
 * sparse or unsorted test sites), 0..3 the round-2 grouped forms, -1 the round-2 per-site kernel; *stream_bytes = bytes of the prepared
 * per-group streams of all test sites (0 otherwise).  Any pointer may be NULL. */
int bmx_ctx_plan(bmx_ctx *c, int32_t *J, int32_t *use_lds, int32_t *mode, int64_t *stream_bytes);
/* Where the selected slot's scan is cut into launches: offs[i] = index of the first test site of launch range i (offs[0] = 0;
 * range i ends where range i + 1 begins, the last one at M).  At most `cap` entries are written; *n_out = number of ranges.
 * A window's result must not depend on the cut -- the parity tests recompute the windows on either side of every cut
 * independently.  Valid once the test sites are set. */
int bmx_ctx_launch_ranges(bmx_ctx *c, int64_t *offs, int32_t cap, int32_t *n_out);

/* ---- the final gather over RCCL, inside the library (north_star: "only a final RCCL gather over xGMI") --
 * One process per GPU, each with its own context.  Rank 0 makes an id (bmx_comm_unique_id: 128 bytes) and hands it to the other
 * ranks by whatever channel the caller has (MPI, a file, a socket, torch's store); every rank then calls bmx_comm_create with
 * its context -- collectively, like ncclCommInitRank.  librccl.so is opened when the first of these functions is called
 * (dlopen): a program that never gathers never loads it.  No torch, no Python. */
typedef struct bmx_comm bmx_comm;
#define BMX_COMM_ID_BYTES 128
int bmx_comm_unique_id(char *id /* BMX_COMM_ID_BYTES */);
int bmx_comm_create(bmx_comm **out, bmx_ctx *c, const char *id, int32_t rank, int32_t world);
void bmx_comm_destroy(bmx_comm *cm);
/* ONE gather of the records of every slot of the communicator's context that holds results (slot order, as
 * bmx_ctx_pack_records) to rank `root`: the packed bmx_record buffers move device to device in one group of ncclSend /
 * ncclRecv on the context's stream -- every peer has its own xGMI link to the root, nothing goes to the other ranks.
 * counts[world]: the number of records of every rank (each rank knows them all: test sites are dealt deterministically).
 * On the root, dst_host (may be NULL) receives all records, rank after rank, and *d_out (may be NULL) the device address of
 * the same array (valid until the next gather or bmx_comm_destroy); other ranks ignore both.  Collective; blocks until done. */
int bmx_comm_gather_records(bmx_comm *cm, const int64_t *counts, int32_t root, bmx_record *dst_host, void **d_out);

/* ---- input ingest (host only) ------------------------------ */
/* Native reader of the 4-column input that InputData.readCounts / readPolyCalls parse line by
 * line in Python (BalLeRMix+_v1.py:80-131): header skipped, tab-separated physPos, genPos, k, n.
 * bmx_input_count returns the number of data lines; bmx_input_parse fills caller-allocated arrays
 * of that length: phys = int(float(col0)), coord = float(col[pos_col]) (pos_col 0: physical,
 * 1: genetic), k = int(col2), n = int(col3).  Bit-identical to the Python parse (strtod). */
int bmx_input_count(const char *path, int64_t *n_out);
int bmx_input_parse(const char *path, int64_t N, int pos_col, int64_t *phys, double *coord,
                    int64_t *k, int64_t *n);

/* ---- output (host only) ------------------------------------- */
/* Appends M result rows to `path` in the reference's format (BalLeRMix+_v1.py:607), floats printed
 * exactly as Python's repr().  xs/abs/As: the grids' printed forms, '\0'-separated, in grid order.
 * iA[t] < 0 writes the all-zero row (v1:451).  Integer physPos only (the --noCenter mode prints a
 * float there and stays in Python). */
int bmx_write_rows(const char *path, int64_t M, const int64_t *phys, const double *gen, const double *clr,
                   const int32_t *ix, const int32_t *ia, const int32_t *iA, const int32_t *nsites,
                   const char *xs, int nx, const char *abs_, int nab, const char *As, int nA);
/* The same rows from 16-byte records as a sharded run gathers them on the writing rank: per_rank[r] = the records of rank
 * r's test sites in its own order, test sites dealt to the `world` ranks in blocks of `block` consecutive test sites
 * round-robin.  world = 1 (block: any value >= 1): M records in output order.  Appends to `path`. */
int bmx_write_records(const char *path, int64_t M, const int64_t *phys, const double *gen,
                      const bmx_record *const *per_rank, int32_t world, int64_t block,
                      const char *xs, int nx, const char *abs_, int nab, const char *As, int nA);
/* repr(v) as Python prints it, into buf (>= 32 bytes); returns the length. */
int bmx_py_repr(double v, char *buf);

#ifdef __cplusplus
}
#endif
#endif /* BMXSCAN_H */
