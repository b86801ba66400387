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
