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
int bmx_input_parse(const char *path, int64_t N, int pos_col, int64_t *phys, double *coord, int64_t *k, int64_t *n) {
    if (!path || !phys || !coord || !k || !n || (pos_col != 0 && pos_col != 1)) { bmx_set_error_("bad argument"); return BMX_E_INVALID; }
    Mapped m;
    int rc = map_file(path, m);
    if (rc) return rc;
    // strtod/strtoll need a terminator: copy the last line if the file does not end in '\n'
    std::string tail;
    const char *s = m.p, *end = m.p + m.n;
    if (s < end) s = next_line(s, end);
    int64_t i = 0;
    while (s < end && i < N) {
        const char *e = next_line(s, end);
        const char *line = s;
        if (e == end && (e == s || e[-1] != '\n')) {   // unterminated last line
            if (blank(s, e)) break;
            tail.assign(s, e);
            tail.push_back('\n');
            line = tail.c_str();
        }
        char *q;
        const char *f0 = line;
        while (*f0 == ' ') ++f0;
        double c0 = strtod(f0, &q);
        if (q == f0 || *q != '\t') goto bad;
        {
            const char *f1 = q + 1;
            double c1 = strtod(f1, &q);
            if (q == f1 || *q != '\t') goto bad;
            const char *f2 = q + 1;
            long long kk = strtoll(f2, &q, 10);
            if (q == f2 || *q != '\t') goto bad;
            const char *f3 = q + 1;
            long long nn = strtoll(f3, &q, 10);
            if (q == f3) goto bad;
            while (*q == ' ' || *q == '\r') ++q;
            if (*q != '\n' && *q != '\t') goto bad;
            phys[i] = (int64_t)c0;            // int(float(col0)): truncation toward zero
            coord[i] = pos_col == 0 ? c0 : c1;
            k[i] = kk;
            n[i] = nn;
        }
        ++i;
        s = e;
        continue;
    bad:
        bmx_set_error_(("malformed input at data line " + std::to_string((long long)i + 1)).c_str());
        return BMX_E_INVALID;
    }
    if (i != N) { bmx_set_error_("fewer data lines than announced"); return BMX_E_INVALID; }
    return BMX_OK;
}

}  // extern "C"

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

extern "C" {

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
    auto split = [](const char *s, int n, std::vector<std::pair<const char *, int>> &v) {
        for (int i = 0; i < n; ++i) {
            int len = (int)strlen(s);
            v.push_back({s, len});
            s += len + 1;
        }
    };
    std::vector<std::pair<const char *, int>> tx, ta, tA;
    split(xs, nx, tx); split(abs_, nab, ta); split(As, nA, tA);
    FILE *f = fopen(path, "a");
    if (!f) { bmx_set_error_((std::string("cannot open ") + path + ": " + strerror(errno)).c_str()); return BMX_E_INVALID; }
    std::vector<char> buf(1 << 20);
    size_t used = 0;
    for (int64_t t = 0; t < M; ++t) {
        if (used + 512 > buf.size()) { fwrite(buf.data(), 1, used, f); used = 0; }
        char *o = buf.data() + used;
        o += sprintf(o, "%lld\t", (long long)phys[t]);
        o = py_repr(o, gen[t]);
        if (iA[t] < 0) {
            memcpy(o, "\t0.0\t0.0\t0.0\t0.0\t0.0\n", 21);
            o += 21;
        } else {
            if (ix[t] < 0 || ix[t] >= nx || ia[t] < 0 || ia[t] >= nab || iA[t] >= nA) {
                fclose(f);
                bmx_set_error_("grid index out of range");
                return BMX_E_INVALID;
            }
            *o++ = '\t';
            o = py_repr(o, clr[t]);
            *o++ = '\t';
            memcpy(o, tx[ix[t]].first, tx[ix[t]].second); o += tx[ix[t]].second;
            *o++ = '\t';
            memcpy(o, ta[ia[t]].first, ta[ia[t]].second); o += ta[ia[t]].second;
            *o++ = '\t';
            memcpy(o, tA[iA[t]].first, tA[iA[t]].second); o += tA[iA[t]].second;
            o += sprintf(o, "\t%d\n", nsites[t]);
        }
        used = (size_t)(o - buf.data());
    }
    if (used) fwrite(buf.data(), 1, used, f);
    if (fclose(f) != 0) { bmx_set_error_("write failed"); return BMX_E_INVALID; }
    return BMX_OK;
}

}  // extern "C"
