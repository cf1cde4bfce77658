// tools/sanitize/hostonly_driver.cpp -- drives the context-free exports of include/csmp.h (host/hostonly.hpp: dictionary files, the
// sharded gather's wire layout) under gcc's AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only: no HIP, no GPU.
// Built and run by tools/sanitize_cpu.sh; argv[1] = tests/golden (the committed dictionary files), argv[2] = a scratch directory.
#include "../../compressedsensing.jl_amd/csrc/host/hostonly.hpp"
#include <cstdlib>
#include <string>
#include <vector>

static int fails = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++fails;                                                        \
        }                                                                   \
    } while (0)

static std::vector<char> slurp(const std::string& p) {
    std::vector<char> v;
    if (FILE* f = std::fopen(p.c_str(), "rb")) {
        char buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
        std::fclose(f);
    }
    return v;
}
static void spit(const std::string& p, const std::vector<char>& v, size_t n) {
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) { ++fails; return; }
    if (n && std::fwrite(v.data(), 1, n, f) != n) ++fails;
    std::fclose(f);
}

static void dictionary_files(const std::string& golden, const std::string& tmp) {
    struct Case { const char* name; int64_t M, N; int dtype; } cases[] = {{"dict_3x4_f64.csmp", 3, 4, CSMP_F64}, {"dict_5x3_f32.csmp", 5, 3, CSMP_F32}};
    for (const Case& c : cases) {
        const std::string path = golden + "/" + c.name;
        int64_t M = -1, N = -1;
        int dt = -1;
        EXPECT(csmp_dictionary_file_info(path.c_str(), &M, &N, &dt) == CSMP_OK);
        EXPECT(M == c.M && N == c.N && dt == c.dtype);
        EXPECT(csmp_dictionary_file_info(path.c_str(), nullptr, nullptr, nullptr) == CSMP_OK);  // every out pointer is optional
        // the committed file, column by column, through the writer again: the same bytes (leading dimension = the file's, then a
        // caller's array with a LARGER leading dimension)
        const std::vector<char> bytes = slurp(path);
        const size_t es = c.dtype == CSMP_F32 ? 4 : 8;
        const int64_t vec = 16 / (int64_t)es, ld = ((c.M + vec - 1) / vec) * vec;
        EXPECT(bytes.size() == 64 + (size_t)ld * (size_t)c.N * es);
        if (bytes.size() != 64 + (size_t)ld * (size_t)c.N * es) continue;
        const std::string out = tmp + "/rewrite_" + c.name;
        EXPECT(csmp_dictionary_file_write(out.c_str(), bytes.data() + 64, c.M, c.N, ld, c.dtype) == CSMP_OK);
        EXPECT(slurp(out) == bytes);
        const int64_t ld2 = ld + 5;
        std::vector<char> wide((size_t)ld2 * (size_t)c.N * es, (char)0x5a);  // (poison between the columns: it must not reach the file)
        for (int64_t j = 0; j < c.N; ++j) std::memcpy(wide.data() + (size_t)j * ld2 * es, bytes.data() + 64 + (size_t)j * ld * es, (size_t)c.M * es);
        EXPECT(csmp_dictionary_file_write(out.c_str(), wide.data(), c.M, c.N, ld2, c.dtype) == CSMP_OK);
        EXPECT(slurp(out) == bytes);
        // an exactly-sized caller array (ldA == M): the writer must not read past its end (ASan sees it if it does)
        std::vector<char> tight((size_t)c.M * (size_t)c.N * es);
        for (int64_t j = 0; j < c.N; ++j) std::memcpy(tight.data() + (size_t)j * c.M * es, bytes.data() + 64 + (size_t)j * ld * es, (size_t)c.M * es);
        EXPECT(csmp_dictionary_file_write(out.c_str(), tight.data(), c.M, c.N, c.M, c.dtype) == CSMP_OK);
        EXPECT(slurp(out) == bytes);
        // damaged files: every truncation of the header, a wrong magic, a wrong version, a leading dimension that is not M rounded up
        for (size_t cut : {(size_t)0, (size_t)7, (size_t)8, (size_t)40, (size_t)63}) {
            const std::string bad = tmp + "/cut.csmp";
            spit(bad, bytes, cut);
            EXPECT(csmp_dictionary_file_info(bad.c_str(), &M, &N, &dt) == CSMP_EIO);
        }
        for (size_t off : {(size_t)0, (size_t)8, (size_t)12, (size_t)16, (size_t)24, (size_t)32}) {  // magic, version, dtype, M, N, ld
            std::vector<char> bad = bytes;
            bad[off] = (char)(bad[off] ^ 0x7f);
            const std::string bp = tmp + "/bad.csmp";
            spit(bp, bad, bad.size());
            const int rc = csmp_dictionary_file_info(bp.c_str(), &M, &N, &dt);
            EXPECT(rc == CSMP_EIO || (off == 24 && rc == CSMP_OK));  // (N is not cross-checked against the file's length by the header reader)
        }
    }
    int64_t M = 0, N = 0;
    int dt = 0;
    EXPECT(csmp_dictionary_file_info(nullptr, &M, &N, &dt) == CSMP_EINVAL);
    EXPECT(csmp_dictionary_file_info((tmp + "/does_not_exist.csmp").c_str(), &M, &N, &dt) == CSMP_EIO);
    const double a[6] = {1, 2, 3, 4, 5, 6};
    const std::string out = tmp + "/args.csmp";
    EXPECT(csmp_dictionary_file_write(nullptr, a, 3, 2, 3, CSMP_F64) == CSMP_EINVAL);
    EXPECT(csmp_dictionary_file_write(out.c_str(), nullptr, 3, 2, 3, CSMP_F64) == CSMP_EINVAL);
    EXPECT(csmp_dictionary_file_write(out.c_str(), a, 0, 2, 3, CSMP_F64) == CSMP_EINVAL);
    EXPECT(csmp_dictionary_file_write(out.c_str(), a, 3, 0, 3, CSMP_F64) == CSMP_EINVAL);
    EXPECT(csmp_dictionary_file_write(out.c_str(), a, 3, 2, 2, CSMP_F64) == CSMP_EINVAL);   // ldA < M
    EXPECT(csmp_dictionary_file_write(out.c_str(), a, 3, 2, 3, 7) == CSMP_EINVAL);          // no such element type
    EXPECT(csmp_dictionary_file_write((tmp + "/no/such/dir/x.csmp").c_str(), a, 3, 2, 3, CSMP_F64) == CSMP_EIO);
    EXPECT(csmp_dictionary_file_write(out.c_str(), a, 3, 2, 3, CSMP_F64) == CSMP_OK);
    EXPECT(csmp_dictionary_file_info(out.c_str(), &M, &N, &dt) == CSMP_OK && M == 3 && N == 2 && dt == CSMP_F64);
}

static void shard_ranges() {
    for (int64_t nsig = 0; nsig <= 67; ++nsig)
        for (int world = 1; world <= 9; ++world) {
            int64_t expect_lo = 0, smallest = nsig, largest = 0;
            for (int r = 0; r < world; ++r) {
                int64_t lo = -1, hi = -1;
                EXPECT(csmp_shard_range(nsig, r, world, &lo, &hi) == CSMP_OK);
                EXPECT(lo == expect_lo && hi >= lo);
                expect_lo = hi;
                smallest = std::min(smallest, hi - lo);
                largest = std::max(largest, hi - lo);
            }
            EXPECT(expect_lo == nsig);          // contiguous blocks that cover 0 .. nsig
            EXPECT(largest - smallest <= 1);    // sizes differ by at most one
        }
    int64_t lo, hi;
    EXPECT(csmp_shard_range(-1, 0, 1, &lo, &hi) == CSMP_EINVAL);
    EXPECT(csmp_shard_range(4, 0, 0, &lo, &hi) == CSMP_EINVAL);
    EXPECT(csmp_shard_range(4, -1, 2, &lo, &hi) == CSMP_EINVAL);
    EXPECT(csmp_shard_range(4, 2, 2, &lo, &hi) == CSMP_EINVAL);
    EXPECT(csmp_shard_range(4, 0, 2, nullptr, &hi) == CSMP_EINVAL);
    EXPECT(csmp_shard_range(4, 0, 2, &lo, nullptr) == CSMP_EINVAL);
}

static void wire_layout() {
    for (int64_t k : {(int64_t)0, (int64_t)1, (int64_t)5, (int64_t)128})
        for (int64_t nsig : {(int64_t)0, (int64_t)1, (int64_t)7}) {
            // exactly-sized arrays (std::vector: heap, so that ASan sees one element too far)
            std::vector<int64_t> idx((size_t)(k * nsig)), nnz((size_t)nsig), idx2((size_t)(k * nsig), -7), nnz2((size_t)nsig, -7);
            std::vector<double> val((size_t)(k * nsig)), val2((size_t)(k * nsig), -7.0), packed((size_t)((2 * k + 1) * nsig), -1.0);
            uint64_t sd = 0x243F6A8885A308D3ull + (uint64_t)k * 977 + (uint64_t)nsig;
            auto rnd = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return sd >> 11; };
            for (int64_t s = 0; s < nsig; ++s) {
                nnz[(size_t)s] = k ? (int64_t)(rnd() % (uint64_t)(k + 1)) : 0;
                for (int64_t t = 0; t < k; ++t) {
                    idx[(size_t)(s * k + t)] = t < nnz[(size_t)s] ? (int64_t)(rnd() % ((uint64_t)1 << 52)) : -1;  // (exact in Float64 below 2^53)
                    val[(size_t)(s * k + t)] = (double)(int64_t)(rnd() % 2000001) / 1e3 - 1000.0;
                }
            }
            // (data() of an empty vector may be null: the functions take that as a missing argument, so hand them a valid address)
            int64_t di = 0;
            double dd = 0.0;
            auto P = [&](std::vector<int64_t>& v) { return v.empty() ? &di : v.data(); };
            auto Pd = [&](std::vector<double>& v) { return v.empty() ? &dd : v.data(); };
            EXPECT(csmp_pack_results(P(idx), Pd(val), P(nnz), k, nsig, Pd(packed)) == CSMP_OK);
            EXPECT(csmp_unpack_results(Pd(packed), k, nsig, P(idx2), Pd(val2), P(nnz2)) == CSMP_OK);
            EXPECT(idx == idx2 && val == val2 && nnz == nnz2);
            for (int64_t s = 0; s < nsig; ++s) EXPECT(packed[(size_t)(s * (2 * k + 1) + 2 * k)] == (double)nnz[(size_t)s]);
        }
    int64_t i1 = 0, n1 = 0;
    double v1 = 0, p3[3] = {0, 0, 0};
    EXPECT(csmp_pack_results(nullptr, &v1, &n1, 1, 1, p3) == CSMP_EINVAL);
    EXPECT(csmp_pack_results(&i1, nullptr, &n1, 1, 1, p3) == CSMP_EINVAL);
    EXPECT(csmp_pack_results(&i1, &v1, nullptr, 1, 1, p3) == CSMP_EINVAL);
    EXPECT(csmp_pack_results(&i1, &v1, &n1, 1, 1, nullptr) == CSMP_EINVAL);
    EXPECT(csmp_pack_results(&i1, &v1, &n1, -1, 1, p3) == CSMP_EINVAL);
    EXPECT(csmp_pack_results(&i1, &v1, &n1, 1, -1, p3) == CSMP_EINVAL);
    EXPECT(csmp_unpack_results(nullptr, 1, 1, &i1, &v1, &n1) == CSMP_EINVAL);
    EXPECT(csmp_unpack_results(p3, 1, 1, nullptr, &v1, &n1) == CSMP_EINVAL);
    EXPECT(csmp_unpack_results(p3, -1, 1, &i1, &v1, &n1) == CSMP_EINVAL);
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <tests/golden> <scratch directory>\n", argv[0]);
        return 2;
    }
    dictionary_files(argv[1], argv[2]);
    shard_ranges();
    wire_layout();
    std::printf("hostonly_driver: %s (%d failed expectation(s))\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
