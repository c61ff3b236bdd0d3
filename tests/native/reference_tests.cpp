// reference_tests.cpp — the reference crate's own tests, written in C++ against the
// C ABI (include/csvsimd.h) and the C++ host mirror (csv-simd_amd/host/csv_simd.hpp): what a C++
// host of libcsvsimd_hip.so looks like, and a check that it behaves as the Rust does.
//
//   reference_tests cpu <golden_dir>   tests that need no GPU (doc-test boundaries, blsr identity,
//                                       Header::new on the fixtures)
//   reference_tests gpu <golden_dir>   + reader::tests::mk_index and csv_simd::create end to end
//
// Test infrastructure; built by tests/native/Makefile, driven by tests/test_native_cpp.py.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "csv_simd.hpp"
#include "csvsimd.h"

namespace {

int g_failed = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("  FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);  \
            ++g_failed;                                                      \
        }                                                                    \
    } while (0)

// the reference maps the file (memmap::Mmap::map, src/lib.rs:64-65)
struct Mmap {
    const std::uint8_t* ptr = nullptr;
    std::uint64_t len = 0;
    bool open(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        len = (std::uint64_t)st.st_size;
        void* p = len ? mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        ::close(fd);
        if (len && p == MAP_FAILED) return false;
        ptr = (const std::uint8_t*)p;
        return true;
    }
    ~Mmap() { if (ptr) munmap((void*)ptr, len); }
};

// drop-in for `reader::read(&Mmap) -> StructureIndex` (src/reader.rs:150): the protocol of
// INTEGRATION.md §2 — guess a capacity, retry once with the exact size
bool reader_read(csvsimd_ctx* ctx, const Mmap& m, std::vector<std::uint64_t>& index) {
    index.assign(m.len / 8 + 64, 0);
    std::uint64_t n = 0;
    int rc = csvsimd_stage1_index(ctx, m.ptr, m.len, index.data(), index.size(), &n, nullptr);
    if (rc == CSVSIMD_ERR_TAPE_CAPACITY) {
        index.assign(n, 0);
        rc = csvsimd_stage1_index(ctx, m.ptr, m.len, index.data(), index.size(), &n, nullptr);
    }
    if (rc != CSVSIMD_OK) {
        std::printf("  csvsimd_stage1_index: %s (%s)\n", csvsimd_strerror(rc), csvsimd_last_error());
        return false;
    }
    index.resize(n);
    return true;
}

// `csv_simd::create(filename) -> Result<Tape, StructureError>` (src/lib.rs:61-74)
csv_simd::StructureError create(csvsimd_ctx* ctx, const std::string& filename, Mmap& memmap,
                                std::vector<std::uint64_t>& index, csv_simd::Tape& tape) {
    using csv_simd::StructureError;
    if (!memmap.open(filename)) return StructureError::Io;
    csv_simd::Header header;
    const StructureError e = csv_simd::Header::create(memmap.ptr, memmap.len, header);
    if (e != StructureError::Ok) return e;
    if (!reader_read(ctx, memmap, index)) return StructureError::InvalidState;
    return csv_simd::Tape::from_core(memmap.ptr, memmap.len, csv_simd::StructureIndex{index.data(), index.size()},
                                     std::move(header), tape);
}

std::string text(const csv_simd::Tape& t, const std::optional<std::pair<std::uint64_t, std::uint64_t>>& s) {
    if (!s) return "<none>";
    return std::string((const char*)t.data_bytes() + s->first, s->second - s->first);
}

// ---- tests that need no GPU -------------------------------------------------------------------
void doc_test_boundaries() {  // src/tape.rs:362-384
    using csv_simd::Boundary;
    using csv_simd::boundaries;
    auto b = boundaries(8, 3);
    CHECK(b && b->size() == 3 && (*b)[0] == (Boundary{0, 3}) && (*b)[1] == (Boundary{3, 3}) && (*b)[2] == (Boundary{6, 2}));
    b = boundaries(1000, 12);
    CHECK(b && b->size() == 12 && (*b)[0] == (Boundary{0, 84}) && (*b)[1] == (Boundary{84, 84}) &&
          b->back() == (Boundary{917, 83}));
    std::uint64_t sum = 0;
    if (b) for (const auto& x : *b) sum += x.len;
    CHECK(sum == 1000);
    b = boundaries(8, 12);
    CHECK(b && b->size() == 1 && (*b)[0] == (Boundary{0, 8}));
    CHECK(!boundaries(0, 3));
}

void binary_manipulations() {  // src/lib.rs:120-153: the blsr identity crush_set_bits relies on
    const std::uint64_t x = 0b01011100;
    CHECK((x & (x - 1)) == 0b01011000);
    std::uint64_t y = x;
    int bits = 0;
    while (y) { y &= y - 1; ++bits; }
    CHECK(bits == 4);
}

void header_new(const std::string& dir) {  // Header::new, src/tape.rs:226-273, on the reference's fixtures
    Mmap m;
    CHECK(m.open(dir + "/sample.csv"));
    csv_simd::Header h;
    CHECK(csv_simd::Header::create(m.ptr, m.len, h) == csv_simd::StructureError::Ok);
    CHECK(h.field_cnt == 3 && h.new_line == csv_simd::NewLine::LF && h.record_offset == 18);
    CHECK(h.header.size() == 3 && h.header[0] == "Name" && h.header[1] == "Number" && h.header[2] == "Done");
    Mmap rx;
    CHECK(rx.open(dir + "/sample_rx.csv"));
    csv_simd::Header hx;
    CHECK(csv_simd::Header::create(rx.ptr, rx.len, hx) == csv_simd::StructureError::Ok);
    CHECK(hx.field_cnt == 8 && hx.new_line == csv_simd::NewLine::CRLF);  // BOM skipped, CRLF sniffed
}

// ---- tests that run stage 1 on the GPU --------------------------------------------------------
void mk_index(csvsimd_ctx* ctx, const std::string& dir) {  // reader::tests::mk_index, src/reader.rs:318-327
    Mmap memmap;
    CHECK(memmap.open(dir + "/reader_test01.csv"));
    std::vector<std::uint64_t> index;
    CHECK(reader_read(ctx, memmap, index));
    CHECK(index.size() == 17);
    CHECK(index[1] == 4);
    CHECK(index[index.size() - 1] == 95);
}

void create_sample(csvsimd_ctx* ctx, const std::string& dir) {  // BASELINE config 1: res/sample.csv -> tape
    using RS = csv_simd::RecordSource<csv_simd::Tape>;
    Mmap memmap;
    std::vector<std::uint64_t> index;
    csv_simd::Tape tape;
    CHECK(create(ctx, dir + "/sample.csv", memmap, index, tape) == csv_simd::StructureError::Ok);
    CHECK(index.size() == 46 && tape.field_cnt() == 3 && *tape.record_cnt() == 15 && *tape.record_jump_size() == 3);
    std::optional<RS::Span> s;
    CHECK(RS::seek_record(tape, 0, s) == csv_simd::StructureError::Ok && text(tape, s) == "Edm nd,3, \"o\"");
    CHECK(RS::seek_field(tape, 0, 0, s) == csv_simd::StructureError::Ok && text(tape, s) == "Edm nd");
    CHECK(RS::seek_field(tape, 0, 3, s) == csv_simd::StructureError::Ok && !s);   // Ok(None)
    CHECK(RS::seek_record(tape, 14, s) == csv_simd::StructureError::Ok && !s);    // past the last data row
    std::vector<csv_simd::Chunk> chunks;
    CHECK(tape.chunks(4, chunks) == csv_simd::StructureError::Ok && chunks.size() == 4);
    std::uint32_t rows = 0;
    for (const auto& c : chunks) rows += c.record_cnt;
    CHECK(rows == 14 && chunks[0].start == 3);  // the header row is skipped

    // ragged file: TapeCore::init refuses it (src/tape.rs:342-344)
    Mmap m2;
    std::vector<std::uint64_t> i2;
    csv_simd::Tape t2;
    CHECK(create(ctx, dir + "/reader_test01.csv", m2, i2, t2) == csv_simd::StructureError::InvalidCsvFormat);
    Mmap m3;
    CHECK(create(ctx, dir + "/no_such_file.csv", m3, i2, t2) == csv_simd::StructureError::Io);
}

void create_sample_rx(csvsimd_ctx* ctx, const std::string& dir) {  // BOM + CRLF + quoted commas
    using RS = csv_simd::RecordSource<csv_simd::Tape>;
    Mmap memmap;
    std::vector<std::uint64_t> index;
    csv_simd::Tape tape;
    CHECK(create(ctx, dir + "/sample_rx.csv", memmap, index, tape) == csv_simd::StructureError::Ok);
    CHECK(index.size() == 73 && tape.field_cnt() == 8 && *tape.record_cnt() == 8 && *tape.record_jump_size() == 9);
    std::optional<RS::Span> s;
    // the quoted field keeps its comma: one field, not two
    bool saw_quoted_comma = false;
    for (std::uint32_t r = 0; r + 1 < *tape.record_cnt(); ++r)
        for (std::uint32_t f = 0; f < 8; ++f) {
            CHECK(RS::seek_field(tape, r, f, s) == csv_simd::StructureError::Ok && s);
            const std::string v = text(tape, s);
            if (v.size() >= 2 && v.front() == '"' && v.find(',') != std::string::npos) saw_quoted_comma = true;
        }
    CHECK(saw_quoted_comma);
}

// A host that reads file after file keeps ONE context (rust/reader_hip.rs holds it per thread): what a read of the
// reference's own 300-byte fixture costs once the context exists.  (The figure is reported by bench.py's latency leg;
// the bound here only catches a context rebuilt per call: hipMalloc + pinned allocations cost milliseconds.)
void repeated_reads_one_context(csvsimd_ctx* ctx, const std::string& dir) {
    Mmap memmap;
    CHECK(memmap.open(dir + "/sample.csv"));
    std::vector<std::uint64_t> first, again;
    CHECK(reader_read(ctx, memmap, first) && first.size() == 46);
    double best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(reader_read(ctx, memmap, again));
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        CHECK(again == first);
    }
    std::printf("  second read() of sample.csv (300 B), same context: %.1f us (best of 20)\n", best);
    CHECK(best < 1000.0);
}

// A directory of small files in ONE call (round 5: csvsimd_stage1_index_batch): the reference reads file by file
// (csv_simd::create, src/lib.rs:61-74); every item must come back exactly as csvsimd_stage1_index returns it alone —
// here the crate's three fixtures, each three times, in one batch.
void read_many_fixtures(csvsimd_ctx* ctx, const std::string& dir) {
    const char* names[3] = {"/reader_test01.csv", "/sample.csv", "/sample_rx.csv"};
    Mmap maps[3];
    std::vector<std::uint64_t> alone[3];
    for (int i = 0; i < 3; ++i) {
        CHECK(maps[i].open(dir + names[i]));
        CHECK(reader_read(ctx, maps[i], alone[i]));
    }
    std::vector<std::vector<std::uint64_t>> tapes(9);
    std::vector<csvsimd_host_batch_item> items(9);
    for (int k = 0; k < 9; ++k) {
        const Mmap& m = maps[k % 3];
        tapes[k].assign(m.len + 1, ~0ull);
        items[k] = csvsimd_host_batch_item{m.ptr, m.len, tapes[k].data(), tapes[k].size(), 0, 0, 0};
    }
    const int rc = csvsimd_stage1_index_batch(ctx, items.data(), (std::uint32_t)items.size());
    CHECK(rc == CSVSIMD_OK);
    for (int k = 0; k < 9; ++k) {
        CHECK(items[k].status == CSVSIMD_OK && items[k].tape_len == alone[k % 3].size());
        tapes[k].resize(items[k].tape_len);
        CHECK(tapes[k] == alone[k % 3]);
    }
    CHECK(tapes[0][1] == 4 && tapes[0].back() == 95);  // reader::tests::mk_index's two values, through the batch
    // csvsimd_create (csv_simd::create in one call of the library) agrees with the C++ re-assembly above
    csvsimd_tape* t = nullptr;
    CHECK(csvsimd_create(ctx, (dir + "/sample.csv").c_str(), &t) == CSVSIMD_OK && t != nullptr);
    if (t) {
        std::uint64_t n = 0;
        const std::uint64_t* idx = csvsimd_tape_index(t, &n);
        CHECK(n == alone[1].size() && std::equal(idx, idx + n, alone[1].begin()));
        CHECK(csvsimd_tape_field_cnt(t) == 3 && csvsimd_tape_record_cnt(t) == 15);
        csvsimd_tape_destroy(t);
    }
}

}  // namespace

// in-process entry (tests/test_native_cpp.py loads libreference_tests.so with ctypes for the GPU part:
// a test runner that has already initialised the GPU must not exec another program)
extern "C" int run_reference_tests(int gpu, const char* golden_dir) {
    g_failed = 0;
    const std::string dir = golden_dir;
    std::printf("test doc_test_boundaries\n");
    doc_test_boundaries();
    std::printf("test binary_manipulations\n");
    binary_manipulations();
    std::printf("test header_new\n");
    header_new(dir);
    if (gpu) {
        csvsimd_ctx* ctx = nullptr;
        const int rc = csvsimd_ctx_create(0, &ctx);
        if (rc != CSVSIMD_OK) {
            std::printf("csvsimd_ctx_create: %s\n", csvsimd_strerror(rc));
            return 1;
        }
        std::printf("test reader::tests::mk_index\n");
        mk_index(ctx, dir);
        std::printf("test create(sample.csv)\n");
        create_sample(ctx, dir);
        std::printf("test create(sample_rx.csv)\n");
        create_sample_rx(ctx, dir);
        std::printf("test repeated reads, one context\n");
        repeated_reads_one_context(ctx, dir);
        std::printf("test read_many(fixtures) + csvsimd_create\n");
        read_many_fixtures(ctx, dir);
        csvsimd_ctx_destroy(ctx);
    } else {
        // no GPU: the library must refuse loudly, never fall back
        csvsimd_ctx* ctx = nullptr;
        if (csvsimd_device_count() <= 0) CHECK(csvsimd_ctx_create(0, &ctx) == CSVSIMD_ERR_NO_DEVICE && ctx == nullptr);
        else if (csvsimd_ctx_create(0, &ctx) == CSVSIMD_OK) csvsimd_ctx_destroy(ctx);
    }
    std::printf(g_failed ? "FAILED: %d check(s)\n" : "ok: all checks passed\n", g_failed);
    std::fflush(stdout);
    return g_failed ? 1 : 0;
}

#ifndef REFERENCE_TESTS_AS_LIBRARY
int main(int argc, char** argv) {
    if (argc < 3) {
        std::printf("usage: %s cpu|gpu <golden_dir>\n", argv[0]);
        return 2;
    }
    return run_reference_tests(std::strcmp(argv[1], "gpu") == 0, argv[2]);
}
#endif
