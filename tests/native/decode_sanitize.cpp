// Sanitizer harness for the host decoders (csrc/ftk_decode.cpp), CPU build only:
//   g++ -fsanitize=address,undefined ... ; run on the committed fixtures and on
// malformed inputs.  GPU sanitizers are not available on the pool; the decoder is
// the only component that parses untrusted bytes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ftk.h"
#include "ftk_bamparse.h"
#include "ftk_inflate.h"
#include "ftk_textparse.h"

// The device row parser's kernels are not part of this host-only build; the harness opens host-mode
// streams only, so the launcher is never reached.
namespace ftk {
size_t textparse_scratch_bytes(size_t n) { return n / 512 + 64; }
void textparse_launch(hipStream_t, const uint8_t*, size_t, bool, void*, size_t, int32_t*, int32_t*, uint8_t*, uint8_t*,
                      TextSummary*) {
    fprintf(stderr, "textparse_launch called in the sanitizer harness\n");
    abort();
}
void textparse_launch_inflated(hipStream_t, uint8_t*, uint32_t, uint32_t, const uint8_t*, const TextSummary*, uint32_t, bool, bool,
                               void*, size_t, int32_t*, int32_t*, uint8_t*, uint8_t*, TextSummary*) {
    fprintf(stderr, "textparse_launch_inflated called in the sanitizer harness\n");
    abort();
}
void append_rows_launch(hipStream_t, int32_t*, int32_t*, uint8_t*, uint8_t*, int32_t*, int32_t*, const int32_t*, const int32_t*,
                        const uint8_t*, const uint8_t*, const int32_t*, const int32_t*, size_t) {
    fprintf(stderr, "append_rows_launch called in the sanitizer harness\n");
    abort();
}
size_t inflate_release_scratch() { return 0; }
void inflate_launch(hipStream_t, const uint8_t*, const InflateBlock*, int, uint8_t*, InflateStatus*, uint32_t*, bool) {
    fprintf(stderr, "inflate_launch called in the sanitizer harness\n");
    abort();
}
void bamparse_launch(hipStream_t, uint8_t*, uint32_t, uint32_t, const uint8_t*, const BamSummary*, uint32_t, const uint8_t*, int,
                     uint32_t, uint32_t*, size_t, size_t, int32_t*, int32_t*, uint8_t*, uint8_t*, int32_t*, int32_t*, int32_t*,
                     BamSummary*) {
    fprintf(stderr, "bamparse_launch called in the sanitizer harness\n");
    abort();
}
size_t bam_sort_tmp_bytes(size_t) { return 0; }
int bam_sort_contig(hipStream_t, size_t, const int32_t*, const int32_t*, const uint8_t*, const uint8_t*, const int32_t*,
                    const int32_t*, int32_t*, int32_t*, uint8_t*, uint8_t*, int32_t*, int32_t*, int32_t*, void*, size_t) {
    fprintf(stderr, "bam_sort_contig called in the sanitizer harness\n");
    abort();
}
}  // namespace ftk

static int decode(const char* path, bool bam, int threads, long* rows_out) {
    ftk_fragtable* t = nullptr;
    int rc = bam ? ftk_bam_decode(path, nullptr, threads, &t) : ftk_fragfile_decode(path, nullptr, threads, &t);
    if (rc != FTK_OK) return rc;
    long rows = 0;
    for (int i = 0; i < ftk_fragtable_n_contigs(t); ++i) {
        const int32_t *s, *e, *r1s, *r1e;
        const uint8_t *q, *st;
        ftk_fragtable_columns(t, i, &s, &e, &q, &st, &r1s, &r1e);
        long n = (long)ftk_fragtable_contig_rows(t, i);
        long acc = 0;
        for (long k = 0; k < n; ++k) acc += s[k] + e[k] + q[k] + st[k] + (r1s ? r1s[k] + r1e[k] : 0);
        rows += n;
        if (acc == 42) printf(" ");
    }
    ftk_fragtable_free(t);
    *rows_out = rows;
    return FTK_OK;
}

// the streaming decoder on the same file: rows must match (or fail cleanly); small pieces via FTK_STREAM_PIECE
static int stream(const char* path, bool bam, int threads, long* rows_out) {
    ftk_fragstream* s = nullptr;
    int rc = ftk_fragstream_open(path, nullptr, bam ? 1 : 0, threads, 1, &s);
    if (rc != FTK_OK) return rc;
    long rows = 0;
    if (bam) (void)ftk_fragstream_n_refs(s);
    for (;;) {
        ftk_fragtable* t = nullptr;
        rc = ftk_fragstream_next(s, &t);
        if (rc != FTK_OK || !t) break;
        const int32_t *st, *en, *r1s, *r1e;
        const uint8_t *q, *sd;
        ftk_fragtable_columns(t, 0, &st, &en, &q, &sd, &r1s, &r1e);
        long n = (long)ftk_fragtable_contig_rows(t, 0), acc = 0;
        for (long k = 0; k < n; ++k) acc += st[k] + en[k] + q[k] + sd[k] + (r1s ? r1s[k] + r1e[k] : 0);
        if (acc == 42) printf(" ");
        rows += n;
        ftk_fragtable_free(t);
    }
    ftk_fragstream_close(s);
    *rows_out = rows;
    return rc;
}

// extra mode (the ThreadSanitizer run): `prog --files a.frag.gz b.bam ...` decodes each file whole and
// streamed (many pieces when FTK_STREAM_PIECE is small) on several threads; the row counts must agree
static int files_mode(int argc, char** argv) {
    for (int i = 2; i < argc; ++i) {
        const std::string f = argv[i];
        const bool bam = f.size() > 4 && f.compare(f.size() - 4, 4, ".bam") == 0;
        long whole = -1, streamed = -2;
        if (decode(f.c_str(), bam, 4, &whole) != FTK_OK || stream(f.c_str(), bam, 4, &streamed) != FTK_OK || whole != streamed ||
            whole <= 0) {
            fprintf(stderr, "FAIL %s whole=%ld streamed=%ld (%s)\n", f.c_str(), whole, streamed, ftk_fragtable_error());
            return 1;
        }
    }
    printf("decode_sanitize ok\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (std::string(argv[1]) == "--files") return files_mode(argc, argv);
    std::string data = argv[1];
    long rows = 0;
    struct { const char* f; bool bam; long want; } cases[] = {
        {"/12.3444.b37.frag.gz", false, 17}, {"/12.3444.b37.frag.bed.gz", false, 17}, {"/12.3444.b37.bam", true, 17}};
    for (auto& c : cases)
        for (int th : {1, 3}) {
            if (decode((data + c.f).c_str(), c.bam, th, &rows) != FTK_OK || rows != c.want) {
                fprintf(stderr, "FAIL %s rows=%ld (%s)\n", c.f, rows, ftk_fragtable_error());
                return 1;
            }
            if (stream((data + c.f).c_str(), c.bam, th, &rows) != FTK_OK || rows != c.want) {
                fprintf(stderr, "FAIL stream %s rows=%ld (%s)\n", c.f, rows, ftk_fragtable_error());
                return 1;
            }
        }
    // truncated / corrupted copies must fail cleanly or decode a prefix, never crash
    for (auto& c : cases) {
        FILE* fp = fopen((data + c.f).c_str(), "rb");
        std::vector<unsigned char> buf(1 << 16);
        size_t n = fread(buf.data(), 1, buf.size(), fp);
        fclose(fp);
        for (size_t cut : {n / 2, n - 5, (size_t)20, (size_t)0}) {
            std::string tmp = std::string(argv[2]) + "/cut.bin";
            FILE* out = fopen(tmp.c_str(), "wb");
            fwrite(buf.data(), 1, cut, out);
            fclose(out);
            decode(tmp.c_str(), c.bam, 2, &rows);
            stream(tmp.c_str(), c.bam, 2, &rows);
        }
        for (size_t pos = 0; pos < n; pos += 7) {
            std::vector<unsigned char> b2(buf.begin(), buf.begin() + n);
            b2[pos] ^= 0x5a;
            std::string tmp = std::string(argv[2]) + "/flip.bin";
            FILE* out = fopen(tmp.c_str(), "wb");
            fwrite(b2.data(), 1, n, out);
            fclose(out);
            decode(tmp.c_str(), c.bam, 2, &rows);
            if (pos % 35 == 0) stream(tmp.c_str(), c.bam, 2, &rows);
        }
    }
    // single-contig requests read the tabix / BAI index: corrupt and truncated indexes must be survived
    struct { const char* f; const char* ix; bool bam; } icases[] = {
        {"/12.3444.b37.frag.gz", ".tbi", false}, {"/12.3444.b37.bam", ".bai", true}};
    for (auto& c : icases) {
        auto slurp = [](const std::string& path) {
            std::vector<unsigned char> b;
            FILE* fp = fopen(path.c_str(), "rb");
            if (!fp) return b;
            b.resize(1 << 20);
            b.resize(fread(b.data(), 1, b.size(), fp));
            fclose(fp);
            return b;
        };
        const std::vector<unsigned char> dat = slurp(data + c.f), idx = slurp(data + c.f + c.ix);
        const std::string base = std::string(argv[2]) + (c.bam ? "/ix.bam" : "/ix.frag.gz");
        FILE* out = fopen(base.c_str(), "wb");
        fwrite(dat.data(), 1, dat.size(), out);
        fclose(out);
        auto run = [&](const std::vector<unsigned char>& index_bytes, long want) {
            FILE* o2 = fopen((base + c.ix).c_str(), "wb");
            fwrite(index_bytes.data(), 1, index_bytes.size(), o2);
            fclose(o2);
            ftk_fragstream* s = nullptr;
            long got = 0;
            if (ftk_fragstream_open(base.c_str(), "12", c.bam ? 1 : 0, 2, 1, &s) == FTK_OK) {
                for (;;) {
                    ftk_fragtable* t = nullptr;
                    if (ftk_fragstream_next(s, &t) != FTK_OK || !t) break;
                    got += (long)ftk_fragtable_contig_rows(t, 0);
                    ftk_fragtable_free(t);
                }
                ftk_fragstream_close(s);
            }
            if (!c.bam) {
                char names[256];
                int64_t need = 0;
                int bed6 = 0;
                (void)ftk_fragfile_index_contigs(base.c_str(), names, sizeof(names), &need, &bed6);
            }
            if (want >= 0 && got != want) { fprintf(stderr, "FAIL index %s rows=%ld\n", c.f, got); exit(1); }
        };
        run(idx, 17);
        for (size_t cut : {idx.size() / 2, (size_t)40, (size_t)9, (size_t)0})
            run(std::vector<unsigned char>(idx.begin(), idx.begin() + cut), -1);
        for (size_t pos = 0; pos < idx.size(); pos += (c.bam ? 13 : 97)) {
            std::vector<unsigned char> b2 = idx;
            b2[pos] ^= 0xa5;
            run(b2, -1);
        }
    }
    printf("decode_sanitize ok\n");
    return 0;
}
