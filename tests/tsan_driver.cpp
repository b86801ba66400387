// ThreadSanitizer driver of libbmxscan's host-side threaded code (tests/test_host_tsan.py builds and runs it with
// g++ -fsanitize=thread together with ballermixplus_amd/csrc/bmx_io.cpp; no GPU, no HIP): the mmap/strtod input reader,
// the validation passes of bmx_ctx_set_sites / bmx_ctx_set_tests, and the multi-threaded row formatter behind
// bmx_write_rows / bmx_write_records.  Exit code 0 = every call returned what it should; data races are reported by TSan
// on stderr (the test fails on any "ThreadSanitizer" line).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/bmxscan.h"

static std::string g_msg;
extern "C" void bmx_set_error_(const char *msg) { g_msg = msg ? msg : ""; }
extern "C" int bmx_validate_sites_(int64_t N, const double *genpos, const int32_t *row, int32_t rows, const double *g,
                                   uint16_t *r16, uint32_t *r32, int64_t *cnt);
extern "C" int bmx_tests_sorted_(int64_t M, const double *test_gen);

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            fprintf(stderr, "FAILED line %d: %s (%s)\n", __LINE__, #cond, g_msg.c_str()); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    const int64_t N = 1500000;                    // several threads in every pass (grain 2^18)
    // ---- input reader
    std::string in = std::string(dir) + "/tsan_in.txt";
    {
        FILE *f = fopen(in.c_str(), "w");
        CHECK(f);
        fprintf(f, "physPos\tgenPos\tx\tn\n");
        for (int64_t i = 0; i < N; i++) fprintf(f, "%lld\t%.6f\t%d\t%d\n", (long long)(100 + 7 * i), (100 + 7 * i) * 1e-6, (int)(i % 50) + 1, 50);
        fclose(f);
    }
    int64_t n = 0;
    CHECK(bmx_input_count(in.c_str(), &n) == 0 && n == N);
    std::vector<int64_t> phys((size_t)N), k((size_t)N), nn((size_t)N);
    std::vector<double> gen((size_t)N);
    CHECK(bmx_input_parse(in.c_str(), N, 1, phys.data(), gen.data(), k.data(), nn.data()) == 0);
    CHECK(phys[0] == 100 && phys[(size_t)N - 1] == 100 + 7 * (N - 1) && k[49] == 50 && nn[7] == 50);
    // ---- set_sites validation: fine input, then each kind of fault from the middle of some thread's range
    const int rows = 51;
    std::vector<double> g((size_t)rows, 0.02);
    g[0] = 0.0;                                    // k = 0 never occurs above
    std::vector<int32_t> row((size_t)N);
    for (int64_t i = 0; i < N; i++) row[(size_t)i] = (int32_t)k[(size_t)i];
    std::vector<uint16_t> r16((size_t)N);
    std::vector<uint32_t> r32((size_t)N);
    std::vector<int64_t> cnt((size_t)rows);
    CHECK(bmx_validate_sites_(N, gen.data(), row.data(), rows, g.data(), r16.data(), nullptr, cnt.data()) == 0);
    int64_t tot = 0;
    for (int r = 0; r < rows; r++) tot += cnt[(size_t)r];
    CHECK(tot == N && cnt[1] == N / 50 && r16[123] == (uint16_t)row[123]);
    CHECK(bmx_validate_sites_(N, gen.data(), row.data(), rows, g.data(), nullptr, r32.data(), cnt.data()) == 0 && r32[777] == (uint32_t)row[777]);
    row[900001] = 51;
    CHECK(bmx_validate_sites_(N, gen.data(), row.data(), rows, g.data(), r16.data(), nullptr, cnt.data()) == 1);
    row[900001] = 0;
    CHECK(bmx_validate_sites_(N, gen.data(), row.data(), rows, g.data(), r16.data(), nullptr, cnt.data()) == 2);
    row[900001] = 5;
    const double keep = gen[400000];
    gen[400000] = gen[399999] - 1e-9;
    CHECK(bmx_validate_sites_(N, gen.data(), row.data(), rows, g.data(), r16.data(), nullptr, cnt.data()) == 3);
    gen[400000] = keep;
    // two faults in two threads' ranges at once: either code is a correct answer, the write itself must not race
    row[100] = -1;
    row[1400000] = 99;
    CHECK(bmx_validate_sites_(N, gen.data(), row.data(), rows, g.data(), r16.data(), nullptr, cnt.data()) == 1);
    row[100] = 3;
    row[1400000] = 3;
    // ---- set_tests order check
    CHECK(bmx_tests_sorted_(N, gen.data()) == 1);
    std::vector<double> tg(gen);
    tg[300000] = tg[299999] - 1.0;
    tg[1200000] = tg[1199999] - 1.0;
    CHECK(bmx_tests_sorted_(N, tg.data()) == 0);
    // ---- writers: rows formatted on several threads
    const int nx = 2, nab = 3, nA = 2;
    const char xs[] = "0.05\0000.1", abs_[] = "1\0005\0001000000000.0", As[] = "100\000200";
    std::vector<double> clr((size_t)N);
    std::vector<int32_t> ix((size_t)N), ia((size_t)N), iA((size_t)N), ns((size_t)N);
    std::vector<bmx_record> rec((size_t)N);
    for (int64_t i = 0; i < N; i++) {
        clr[(size_t)i] = 1.0 / (double)(i + 3);
        ix[(size_t)i] = (int32_t)(i % nx); ia[(size_t)i] = (int32_t)(i % nab); iA[(size_t)i] = i % 11 == 0 ? -1 : (int32_t)(i % nA);
        ns[(size_t)i] = (int32_t)(i % 5000);
        rec[(size_t)i].clr = clr[(size_t)i];
        rec[(size_t)i].lin = iA[(size_t)i] < 0 ? -1 : (iA[(size_t)i] * nx + ix[(size_t)i]) * nab + ia[(size_t)i];
        rec[(size_t)i].nsites = ns[(size_t)i];
    }
    std::string o1 = std::string(dir) + "/tsan_rows.txt", o2 = std::string(dir) + "/tsan_rec1.txt", o3 = std::string(dir) + "/tsan_rec3.txt";
    remove(o1.c_str()); remove(o2.c_str()); remove(o3.c_str());
    CHECK(bmx_write_rows(o1.c_str(), N, phys.data(), gen.data(), clr.data(), ix.data(), ia.data(), iA.data(), ns.data(), xs, nx, abs_, nab, As, nA) == 0);
    const bmx_record *one[1] = {rec.data()};
    CHECK(bmx_write_records(o2.c_str(), N, phys.data(), gen.data(), one, 1, 4096, xs, nx, abs_, nab, As, nA) == 0);
    // three ranks, blocks of 4096 dealt round-robin
    std::vector<bmx_record> part[3];
    for (int64_t t = 0; t < N; t++) part[(t / 4096) % 3].push_back(rec[(size_t)t]);
    const bmx_record *three[3] = {part[0].data(), part[1].data(), part[2].data()};
    CHECK(bmx_write_records(o3.c_str(), N, phys.data(), gen.data(), three, 3, 4096, xs, nx, abs_, nab, As, nA) == 0);
    auto slurp = [](const std::string &p) {
        std::string s;
        FILE *f = fopen(p.c_str(), "rb");
        if (!f) return s;
        char buf[1 << 16];
        size_t r;
        while ((r = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, r);
        fclose(f);
        return s;
    };
    const std::string a = slurp(o1), b = slurp(o2), c = slurp(o3);
    CHECK(a.size() > (size_t)N * 20 && a == b && a == c);
    ix[5] = 7;                                       // a grid index outside the tables is refused, not formatted
    CHECK(bmx_write_rows(o1.c_str(), 100, phys.data(), gen.data(), clr.data(), ix.data(), ia.data(), iA.data(), ns.data(), xs, nx, abs_, nab, As, nA) != 0);
    remove(in.c_str()); remove(o1.c_str()); remove(o2.c_str()); remove(o3.c_str());
    printf("tsan driver ok\n");
    return 0;
}
