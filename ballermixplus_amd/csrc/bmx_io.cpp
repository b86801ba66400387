// bmx_io.cpp -- native reader for the 4-column input format (host code, no GPU involved).
//
// Replaces the text loop of InputData.readCounts / readPolyCalls (reference BalLeRMix+_v1.py:80-131),
// which costs ~1.5-2.6 s per million lines in Python: header line skipped, fields separated by
// tabs, physPos = int(float(col0)), coordinate = float(col[pos_col]), k = int(col2), n = int(col3).
// strtod is correctly rounded, as Python's float() is, so the arrays are bit-identical.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <string>

#include "../../include/bmxscan.h"

extern "C" void bmx_set_error_(const char *msg);   // defined next to bmx_last_error()

namespace {
struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p && n) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};
int map_file(const char *path, Mapped &m) {
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) { bmx_set_error_((std::string("cannot open ") + path + ": " + strerror(errno)).c_str()); return BMX_E_INVALID; }
    struct stat st;
    if (fstat(m.fd, &st) != 0) { bmx_set_error_("fstat failed"); return BMX_E_INVALID; }
    m.n = (size_t)st.st_size;
    if (m.n == 0) return BMX_OK;
    void *q = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (q == MAP_FAILED) { m.n = 0; bmx_set_error_("mmap failed"); return BMX_E_INVALID; }
    m.p = (const char *)q;
    return BMX_OK;
}
inline const char *next_line(const char *s, const char *end) {
    const char *nl = (const char *)memchr(s, '\n', (size_t)(end - s));
    return nl ? nl + 1 : end;
}
inline bool blank(const char *s, const char *e) {
    for (; s < e; ++s)
        if (*s != ' ' && *s != '\t' && *s != '\r' && *s != '\n') return false;
    return true;
}
}  // namespace

extern "C" {

// Number of data lines (everything after the header line; a trailing empty line is not counted).
int bmx_input_count(const char *path, int64_t *n_out) {
    if (!path || !n_out) { bmx_set_error_("NULL argument"); return BMX_E_INVALID; }
    Mapped m;
    int rc = map_file(path, m);
    if (rc) return rc;
    const char *s = m.p, *end = m.p + m.n;
    int64_t n = 0;
    if (s < end) s = next_line(s, end);   // header
    while (s < end) {
        const char *e = next_line(s, end);
        if (!(e == end && blank(s, e))) n++;
        s = e;
    }
    *n_out = n;
    return BMX_OK;
}

// Parse N data lines into caller-allocated arrays.  coord = column pos_col (0 physical, 1 genetic).
// The data bytes are cut into one range per thread at line boundaries: a counting pass gives every range its
// first row index, a second pass parses.  Anything this strict reader does not recognise (a field that does not
// start with a digit, sign or '.', hex floats, inf/nan, trailing junk) is reported as malformed, and the caller's
// Python reader then reproduces the reference's own behaviour (and error) for that file.
}  // extern "C"

#include <thread>
#include <vector>

namespace {
inline bool num_start(char c) { return (c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.'; }
inline bool float_chars(const char *a, const char *b) {
    for (; a < b; ++a) {
        const char c = *a;
        if (!((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E')) return false;
    }
    return true;
}

// one '\n'-terminated line; false = malformed
inline bool parse_line(const char *line, int pos_col, int64_t &phys, double &coord, int64_t &k, int64_t &n) {
    char *q;
    const char *f0 = line;
    while (*f0 == ' ') ++f0;                       // the reference strips the line
    if (!num_start(*f0)) return false;
    const double c0 = strtod(f0, &q);
    if (q == f0 || *q != '\t' || !float_chars(f0, q)) return false;
    const char *f1 = q + 1;
    if (!num_start(*f1)) return false;
    const double c1 = strtod(f1, &q);
    if (q == f1 || *q != '\t' || !float_chars(f1, q)) return false;
    const char *f2 = q + 1;
    if (!num_start(*f2) || *f2 == '.') return false;
    const long long kk = strtoll(f2, &q, 10);
    if (q == f2 || *q != '\t') return false;
    const char *f3 = q + 1;
    if (!num_start(*f3) || *f3 == '.') return false;      // an empty last column must not swallow the next line
    const long long nn = strtoll(f3, &q, 10);
    if (q == f3) return false;
    while (*q == ' ' || *q == '\r') ++q;
    if (*q != '\n' && *q != '\t') return false;
    phys = (int64_t)c0;                            // int(float(col0)): truncation toward zero
    coord = pos_col == 0 ? c0 : c1;
    k = kk;
    n = nn;
    return true;
}
}  // namespace

extern "C" {

int bmx_input_parse(const char *path, int64_t N, int pos_col, int64_t *phys, double *coord, int64_t *k, int64_t *n) {
    if (!path || !phys || !coord || !k || !n || (pos_col != 0 && pos_col != 1)) { bmx_set_error_("bad argument"); return BMX_E_INVALID; }
    Mapped m;
    int rc = map_file(path, m);
    if (rc) return rc;
    const char *s = m.p, *end = m.p + m.n;
    if (s < end) s = next_line(s, end);            // header
    // strtod/strtoll need a terminator: an unterminated last line is parsed from a copy
    const char *body_end = end;
    std::string tail;
    if (s < end && end[-1] != '\n') {
        const char *last = end;
        while (last > s && last[-1] != '\n') --last;
        body_end = last;
        if (!blank(last, end)) { tail.assign(last, end); tail.push_back('\n'); }
    }
    if (tail.empty() && body_end > s) {            // like bmx_input_count: one blank last line is not data
        const char *last = body_end - 1;
        while (last > s && last[-1] != '\n') --last;
        if (blank(last, body_end)) body_end = last;
    }
    const size_t bytes = (size_t)(body_end - s);
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 32), bytes / (1 << 20) + 1);
    std::vector<const char *> cut((size_t)T + 1);
    cut[0] = s;
    for (int t = 1; t < T; ++t) {
        const char *c = s + bytes * (size_t)t / (size_t)T;
        cut[(size_t)t] = c <= cut[(size_t)t - 1] ? cut[(size_t)t - 1] : next_line(c - 1, body_end);   // start of the line after the cut byte
    }
    cut[(size_t)T] = body_end;
    std::vector<int64_t> first((size_t)T + 1, 0), bad((size_t)T, -1);
    auto for_ranges = [&](auto fn) {
        if (T == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    for_ranges([&](int t) {
        int64_t c = 0;
        for (const char *a = cut[(size_t)t]; a < cut[(size_t)t + 1]; a = next_line(a, cut[(size_t)t + 1])) c++;
        first[(size_t)t + 1] = c;
    });
    for (int t = 0; t < T; ++t) first[(size_t)t + 1] += first[(size_t)t];
    const int64_t total = first[(size_t)T] + (tail.empty() ? 0 : 1);
    if (total != N) { bmx_set_error_(total < N ? "fewer data lines than announced" : "more data lines than announced"); return BMX_E_INVALID; }
    for_ranges([&](int t) {
        int64_t i = first[(size_t)t];
        for (const char *a = cut[(size_t)t]; a < cut[(size_t)t + 1]; a = next_line(a, cut[(size_t)t + 1]), ++i)
            if (!parse_line(a, pos_col, phys[i], coord[i], k[i], n[i])) { bad[(size_t)t] = i; return; }
    });
    int64_t first_bad = -1;
    for (int t = 0; t < T; ++t)
        if (bad[(size_t)t] >= 0 && (first_bad < 0 || bad[(size_t)t] < first_bad)) first_bad = bad[(size_t)t];
    if (first_bad < 0 && !tail.empty() && !parse_line(tail.c_str(), pos_col, phys[N - 1], coord[N - 1], k[N - 1], n[N - 1])) first_bad = N - 1;
    if (first_bad >= 0) {
        bmx_set_error_(("malformed input at data line " + std::to_string((long long)first_bad + 1)).c_str());
        return BMX_E_INVALID;
    }
    return BMX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Host-side validation passes of bmx_ctx_set_sites / bmx_ctx_set_tests, on several host threads (here, not next to the
// kernels, so that a ThreadSanitizer build of this file can run them without a GPU: tests/test_host_tsan.py).
#include <atomic>
#include <thread>
#include <vector>

namespace {
// fn(t, begin, end) on T host threads over [0, n)
template <class F>
void io_parallel_ranges(int64_t n, int64_t grain, int max_threads, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, (unsigned)max_threads), n / std::max<int64_t>(grain, 1) + 1);
    if (T <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(fn, t, n * t / T, n * (t + 1) / T);
    for (auto &x : th) x.join();
}
}  // namespace

// One pass over the site arrays: every row index inside the table and on a (k, n) with a positive neutral probability (the
// kernels index the LDS/L2 table with it unchecked), positions sorted and not NaN; the 16- or 32-bit row indices the device
// uses and the per-row site counts come out of the same pass.  Returns 0, or 1 row outside the table, 2 neutral probability
// missing / not positive, 3 positions not sorted, 4 NaN position.  r16 / r32: exactly one is non-NULL; cnt[rows] is zeroed here.
extern "C" int bmx_validate_sites_(int64_t N, const double *genpos, const int32_t *row, int32_t rows, const double *g,
                                   uint16_t *r16, uint32_t *r32, int64_t *cnt) {
    constexpr int MAXT = 32;
    std::atomic<int> bad{0};             // the kind of fault seen by any thread (which one wins does not matter)
    std::vector<int64_t> cnt_t[MAXT];
    io_parallel_ranges(N, 1 << 18, MAXT, [&](int t, int64_t b, int64_t e) {
        std::vector<int64_t> &c = cnt_t[t];
        c.assign((size_t)rows, 0);
        for (int64_t i = b; i < e; i++) {
            const int32_t r = row[i];
            int f = 0;
            if (r < 0 || r >= rows) f = 1;
            else if (!(g[(size_t)r] > 0.0)) f = 2;
            else if (i && genpos[i] < genpos[i - 1]) f = 3;
            else if (!(genpos[i] == genpos[i])) f = 4;
            if (f) { bad.store(f, std::memory_order_relaxed); return; }
            if (r32) r32[(size_t)i] = (uint32_t)r; else r16[(size_t)i] = (uint16_t)r;
            c[(size_t)r]++;
        }
    });
    for (int32_t r = 0; r < rows; r++) cnt[r] = 0;
    for (int t = 0; t < MAXT; t++)
        for (size_t r = 0; r < cnt_t[t].size(); r++) cnt[r] += cnt_t[t][r];
    return bad.load();
}

// 1 if test_gen[0..M) is non-decreasing (NaN counts as unsorted), else 0
extern "C" int bmx_tests_sorted_(int64_t M, const double *test_gen) {
    std::atomic<int> unsorted{0};
    io_parallel_ranges(M, 1 << 18, 32, [&](int, int64_t b, int64_t e) {
        for (int64_t t = std::max<int64_t>(b, 1); t < e; t++)
            if (!(test_gen[t] >= test_gen[t - 1])) { unsorted.store(1, std::memory_order_relaxed); return; }
    });
    return unsorted.load() ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// Output writer: the 7-column rows of `scores.write(f'{phys}\t{gen}\t{T}\t{x}\t{a}\t{A}\t{n}\n')`
// (reference BalLeRMix+_v1.py:607) for integer physPos.  Floats are printed exactly as Python's
// repr() does: shortest digits that round-trip (std::to_chars), fixed notation when the decimal
// exponent is in [-4, 16), otherwise d.ddde+XX with at least two exponent digits.
#include <charconv>
#include <stdio.h>
#include <vector>

namespace {
// appends repr(v) to out; returns new end
char *py_repr(char *out, double v) {
    if (v != v) { memcpy(out, "nan", 3); return out + 3; }
    if (v == 0.0) {
        if (std::signbit(v)) *out++ = '-';
        memcpy(out, "0.0", 3);
        return out + 3;
    }
    if (v < 0) { *out++ = '-'; v = -v; }
    if (v > 1.7976931348623157e308) { memcpy(out, "inf", 3); return out + 3; }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    *r.ptr = 0;
    char *e = strchr(buf, 'e');
    int exp10 = atoi(e + 1);
    char digits[32];
    int nd = 0;
    for (char *p = buf; p < e; ++p)
        if (*p != '.') digits[nd++] = *p;
    if (exp10 >= -4 && exp10 < 16) {          // fixed
        if (exp10 >= 0) {
            int i = 0;
            for (; i <= exp10; ++i) *out++ = i < nd ? digits[i] : '0';
            *out++ = '.';
            if (i >= nd) *out++ = '0';
            for (; i < nd; ++i) *out++ = digits[i];
        } else {
            *out++ = '0';
            *out++ = '.';
            for (int z = 0; z < -exp10 - 1; ++z) *out++ = '0';
            for (int i = 0; i < nd; ++i) *out++ = digits[i];
        }
    } else {                                   // exponent form
        *out++ = digits[0];
        if (nd > 1) {
            *out++ = '.';
            for (int i = 1; i < nd; ++i) *out++ = digits[i];
        }
        *out++ = 'e';
        *out++ = exp10 < 0 ? '-' : '+';
        int a = exp10 < 0 ? -exp10 : exp10;
        if (a < 10) *out++ = '0';
        out += sprintf(out, "%d", a);
    }
    return out;
}
}  // namespace

// The grids' printed forms (Python's str of the grid objects, made by the caller), split once.
struct bmx_row_tables_ {
    std::vector<std::pair<const char *, int>> tx, ta, tA;
    std::string store;
};

extern "C" bmx_row_tables_ *bmx_row_tables_new_(const char *xs, int nx, const char *abs_, int nab, const char *As, int nA) {
    bmx_row_tables_ *t = new bmx_row_tables_();
    size_t total = 0;
    auto measure = [&](const char *s, int n) { for (int i = 0; i < n; ++i) { size_t l = strlen(s) + 1; total += l; s += l; } };
    measure(xs, nx); measure(abs_, nab); measure(As, nA);
    t->store.reserve(total);
    auto take = [&](const char *s, int n, std::vector<std::pair<const char *, int>> &v) {
        for (int i = 0; i < n; ++i) {
            const int len = (int)strlen(s);
            const size_t at = t->store.size();
            t->store.append(s, (size_t)len + 1);
            v.push_back({(const char *)at, len});          // offsets first: the string may not move after reserve, but stay safe
            s += len + 1;
        }
    };
    take(xs, nx, t->tx); take(abs_, nab, t->ta); take(As, nA, t->tA);
    for (auto *v : {&t->tx, &t->ta, &t->tA})
        for (auto &e : *v) e.first = t->store.data() + (size_t)e.first;
    return t;
}
extern "C" void bmx_row_tables_free_(bmx_row_tables_ *t) { delete t; }

namespace {
// rows [b, e) into out; false on a grid index outside the tables
bool format_rows(std::vector<char> &out, const bmx_row_tables_ *t, int64_t b, int64_t e, const int64_t *phys, const double *gen,
                 const double *clr, const int32_t *ix, const int32_t *ia, const int32_t *iA, const int32_t *lin, const int32_t *nsites) {
    const int nx = (int)t->tx.size(), nab = (int)t->ta.size(), nA = (int)t->tA.size();
    out.resize((size_t)(e - b) * 96 + 512);
    size_t used = 0;
    for (int64_t r = b; r < e; ++r) {
        if (used + 512 > out.size()) out.resize(out.size() * 3 / 2 + 512);
        char *o = out.data() + used;
        o += sprintf(o, "%lld\t", (long long)phys[r]);
        o = py_repr(o, gen[r]);
        int a, bb, d;
        if (lin) {
            const int32_t L = lin[r];
            d = L < 0 ? -1 : L / (nx * nab);
            const int32_t p = L < 0 ? 0 : L % (nx * nab);
            a = p / nab;
            bb = p % nab;
        } else {
            a = ix[r]; bb = ia[r]; d = iA[r];
        }
        if (d < 0) {
            memcpy(o, "\t0.0\t0.0\t0.0\t0.0\t0.0\n", 21);
            o += 21;
        } else {
            if (a < 0 || a >= nx || bb < 0 || bb >= nab || d >= nA) return false;
            *o++ = '\t';
            o = py_repr(o, clr[r]);
            *o++ = '\t';
            memcpy(o, t->tx[(size_t)a].first, (size_t)t->tx[(size_t)a].second); o += t->tx[(size_t)a].second;
            *o++ = '\t';
            memcpy(o, t->ta[(size_t)bb].first, (size_t)t->ta[(size_t)bb].second); o += t->ta[(size_t)bb].second;
            *o++ = '\t';
            memcpy(o, t->tA[(size_t)d].first, (size_t)t->tA[(size_t)d].second); o += t->tA[(size_t)d].second;
            o += sprintf(o, "\t%d\n", nsites[r]);
        }
        used = (size_t)(o - out.data());
    }
    out.resize(used);
    return true;
}
}  // namespace

// n rows to f: formatted on several threads (Python-repr-exact floats are the cost of a million-row file), written in order
extern "C" int bmx_write_chunk_(FILE *f, const bmx_row_tables_ *t, int64_t n, const int64_t *phys, const double *gen, const double *clr,
                                const int32_t *ix, const int32_t *ia, const int32_t *iA, const int32_t *lin, const int32_t *nsites) {
    if (n <= 0) return BMX_OK;
    unsigned hw = std::thread::hardware_concurrency();
    const int T = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 16), n / 8192 + 1);
    std::vector<std::vector<char>> bufs((size_t)T);
    std::vector<char> okv((size_t)T, 1);
    auto work = [&](int k) {
        okv[(size_t)k] = format_rows(bufs[(size_t)k], t, n * k / T, n * (k + 1) / T, phys, gen, clr, ix, ia, iA, lin, nsites) ? 1 : 0;
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int k = 0; k < T; ++k) th.emplace_back(work, k);
        for (auto &x : th) x.join();
    }
    for (int k = 0; k < T; ++k)
        if (!okv[(size_t)k]) { bmx_set_error_("grid index out of range"); return BMX_E_INVALID; }
    for (int k = 0; k < T; ++k)
        if (!bufs[(size_t)k].empty() && fwrite(bufs[(size_t)k].data(), 1, bufs[(size_t)k].size(), f) != bufs[(size_t)k].size()) {
            bmx_set_error_("write failed");
            return BMX_E_INVALID;
        }
    return BMX_OK;
}

extern "C" {

// Rows from 16-byte records as a sharded run gathers them: per_rank[r] = the records of rank r's test sites in its own
// order, test sites dealt to the ranks in blocks of `block` round-robin (distributed.assign).  world = 1: the records in order.
int bmx_write_records(const char *path, int64_t M, const int64_t *phys, const double *gen, const bmx_record *const *per_rank,
                      int32_t world, int64_t block, const char *xs, int nx, const char *abs_, int nab, const char *As, int nA) {
    if (!path || (M > 0 && (!phys || !gen || !per_rank)) || world < 1 || block < 1 || !xs || !abs_ || !As) {
        bmx_set_error_("bad argument");
        return BMX_E_INVALID;
    }
    for (int r = 0; r < world; ++r) {
        const int64_t nblk = (M + block - 1) / block;
        if (!per_rank[r] && nblk > r) { bmx_set_error_("bad argument: a rank's records are missing"); return BMX_E_INVALID; }
    }
    FILE *f = fopen(path, "a");
    if (!f) { bmx_set_error_((std::string("cannot open ") + path + ": " + strerror(errno)).c_str()); return BMX_E_INVALID; }
    bmx_row_tables_ *t = bmx_row_tables_new_(xs, nx, abs_, nab, As, nA);
    int rc = BMX_OK;
    constexpr int64_t CH = 1 << 18;                         // bounded staging and formatting buffers
    std::vector<double> clr((size_t)std::min(CH, std::max<int64_t>(M, 1)));
    std::vector<int32_t> lin(clr.size()), ns(clr.size());
    for (int64_t off = 0; off < M && !rc; off += CH) {
        const int64_t cnt = std::min<int64_t>(CH, M - off);
        for (int64_t i = 0; i < cnt; ++i) {
            const int64_t tt = off + i, b = tt / block;
            const bmx_record &q = per_rank[b % world][(b / world) * block + tt % block];
            clr[(size_t)i] = q.clr; lin[(size_t)i] = q.lin; ns[(size_t)i] = q.nsites;
        }
        rc = bmx_write_chunk_(f, t, cnt, phys + off, gen + off, clr.data(), nullptr, nullptr, nullptr, lin.data(), ns.data());
    }
    bmx_row_tables_free_(t);
    if (fclose(f) != 0 && !rc) { bmx_set_error_("write failed"); rc = BMX_E_INVALID; }
    return rc;
}

// repr(v) into buf (>= 32 bytes); returns its length.  Exposed for the tests.
int bmx_py_repr(double v, char *buf) {
    char *e = py_repr(buf, v);
    *e = 0;
    return (int)(e - buf);
}

// Appends M rows to `path` (opened "a"): tables are '\0'-separated, concatenated strings of the
// grids' printed forms, in grid order.  iA[t] < 0 writes the reference's all-zero row (v1:451).
int bmx_write_rows(const char *path, int64_t M, const int64_t *phys, const double *gen, const double *clr,
                   const int32_t *ix, const int32_t *ia, const int32_t *iA, const int32_t *nsites,
                   const char *xs, int nx, const char *abs_, int nab, const char *As, int nA) {
    if (!path || (M > 0 && (!phys || !gen || !clr || !ix || !ia || !iA || !nsites)) || !xs || !abs_ || !As) {
        bmx_set_error_("bad argument");
        return BMX_E_INVALID;
    }
    FILE *f = fopen(path, "a");
    if (!f) { bmx_set_error_((std::string("cannot open ") + path + ": " + strerror(errno)).c_str()); return BMX_E_INVALID; }
    bmx_row_tables_ *t = bmx_row_tables_new_(xs, nx, abs_, nab, As, nA);
    int rc = BMX_OK;
    for (int64_t off = 0; off < M && !rc; off += 1 << 18) {      // bounded formatting buffers
        const int64_t cnt = std::min<int64_t>(1 << 18, M - off);
        rc = bmx_write_chunk_(f, t, cnt, phys + off, gen + off, clr + off, ix + off, ia + off, iA + off, nullptr, nsites + off);
    }
    bmx_row_tables_free_(t);
    if (fclose(f) != 0 && !rc) { bmx_set_error_("write failed"); rc = BMX_E_INVALID; }
    return rc;
}

}  // extern "C"
