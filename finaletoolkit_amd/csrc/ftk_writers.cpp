// Host-side output writers of the per-base features: the text the reference's writers produce --
// fixedStep WIG bodies (frag/_wps.py:208-229), per-base bedGraph rows (frag/_multi_wps.py:328-341,
// frag/_cleavage_profile.py multi writer) -- formatted by all host threads instead of one Python f-string
// per base, gzip output as independent members compressed in parallel, and the zlib-compressed fixedStep
// sections of a bigWig (what pyBigWig's addEntries(values=..., span=1, step=1) emits,
// frag/_multi_wps.py:300-325).  Pure host code: no GPU needed.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include "ftk.h"
#include "ftk_host.h"

namespace {

int wfail(int code, const char* fmt, const char* a = "", const char* b = "") {
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, a, b);
    ftk_host::set_decode_error(buf);
    return code;
}

// ---- integers and floats as Python prints them ---------------------------------------------------------
inline char* put_i64(char* p, int64_t v) {
    // str(int): optional '-', decimal digits
    char tmp[24];
    uint64_t u = v < 0 ? 0 - (uint64_t)v : (uint64_t)v;
    int n = 0;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

// repr(float) (what an f-string prints): the shortest digit string that round-trips, in fixed notation when
// the decimal point lands within -4 < decpt <= 16 and with an exponent of at least two digits otherwise
// (CPython format_float_short, mode 'r', Py_DTSF_ADD_DOT_0); nan / inf / -inf spelled as Python does.
inline char* put_f64(char* p, double x) {
    if (std::isnan(x)) { memcpy(p, "nan", 3); return p + 3; }
    if (std::isinf(x)) {
        if (x < 0) *p++ = '-';
        memcpy(p, "inf", 3);
        return p + 3;
    }
    if (std::signbit(x)) { *p++ = '-'; x = -x; }
    if (x == 0.0) { memcpy(p, "0.0", 3); return p + 3; }
    char sci[40];  // d[.ddd]e[+-]XX
    *std::to_chars(sci, sci + sizeof(sci) - 1, x, std::chars_format::scientific).ptr = '\0';
    char* e = sci;
    while (*e != 'e') ++e;
    char digits[24];
    int nd = 0;
    for (char* c = sci; c < e; ++c)
        if (*c != '.') digits[nd++] = *c;
    const int exp10 = atoi(e + 1);
    const int decpt = exp10 + 1;  // value = 0.d1d2... * 10^decpt
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            *p++ = '0';
            *p++ = '.';
            for (int k = 0; k < -decpt; ++k) *p++ = '0';
            memcpy(p, digits, nd);
            p += nd;
        } else if (decpt >= nd) {
            memcpy(p, digits, nd);
            p += nd;
            for (int k = nd; k < decpt; ++k) *p++ = '0';
            *p++ = '.';
            *p++ = '0';
        } else {
            memcpy(p, digits, decpt);
            p += decpt;
            *p++ = '.';
            memcpy(p, digits + decpt, nd - decpt);
            p += nd - decpt;
        }
        return p;
    }
    *p++ = digits[0];
    if (nd > 1) {
        *p++ = '.';
        memcpy(p, digits + 1, nd - 1);
        p += nd - 1;
    }
    *p++ = 'e';
    int ex = decpt - 1;
    *p++ = ex < 0 ? '-' : '+';
    if (ex < 0) ex = -ex;
    if (ex < 10) *p++ = '0';
    return put_i64(p, ex);
}

struct Piece {
    char* p = nullptr;
    size_t len = 0;
};

// Format rows [0, n) on n_threads threads: every thread formats its contiguous share [a, b) with
// `body(a, b, out)` (returns the end of what it wrote) into its own buffer (an upper bound of max_row bytes per
// row), the shares are then copied side by side.
template <class Body>
int format_rows(int64_t n, size_t max_row, int n_threads, char** out, int64_t* out_len, Body body) {
    *out = nullptr;
    *out_len = 0;
    if (n <= 0) {
        *out = (char*)malloc(1);
        return *out ? FTK_OK : wfail(FTK_ERR_OOM, "out of host memory");
    }
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : ftk_host::default_threads(), (n + 65535) / 65536));
    std::vector<Piece> pieces(nt);
    std::atomic<bool> oom{false};
    ftk_host::parallel_run(nt, [&](int t) {
        const int64_t a = n * t / nt, b = n * (t + 1) / nt;
        char* buf = (char*)malloc((size_t)(b - a) * max_row + 1);
        if (!buf) { oom = true; return; }
        pieces[t].p = buf;
        pieces[t].len = (size_t)(body(a, b, buf) - buf);
    });
    size_t total = 0;
    for (auto& pc : pieces) total += pc.len;
    char* all = oom ? nullptr : (char*)malloc(total + 1);
    if (!all) {
        for (auto& pc : pieces) free(pc.p);
        return wfail(FTK_ERR_OOM, "out of host memory");
    }
    std::vector<size_t> off(nt + 1, 0);
    for (int t = 0; t < nt; ++t) off[t + 1] = off[t] + pieces[t].len;
    ftk_host::parallel_run(nt, [&](int t) {
        memcpy(all + off[t], pieces[t].p, pieces[t].len);
        free(pieces[t].p);
    });
    *out = all;
    *out_len = (int64_t)total;
    return FTK_OK;
}

// rows of n_iv runs laid end to end: run k holds values [offsets[k], offsets[k+1]) and starts at base iv_start[k]
template <class V, class Put>
int format_bedgraph(const char* contig, const int64_t* iv_start, const int64_t* offsets, int64_t n_iv, const V* values,
                    size_t max_value, int n_threads, char** out, int64_t* out_len, Put put) {
    if (!contig || !out || !out_len || n_iv < 0 || (n_iv > 0 && (!iv_start || !offsets)))
        return wfail(FTK_ERR_INVALID, "bad arguments");
    const int64_t n = n_iv ? offsets[n_iv] - offsets[0] : 0;
    if (n > 0 && !values) return wfail(FTK_ERR_INVALID, "values is NULL");
    for (int64_t k = 0; k < n_iv; ++k)
        if (offsets[k + 1] < offsets[k]) return wfail(FTK_ERR_INVALID, "offsets must not decrease");
    const size_t cl = strlen(contig);
    const int64_t o0 = n_iv ? offsets[0] : 0;
    return format_rows(n, cl + 2 * 21 + max_value + 4, n_threads, out, out_len, [=](int64_t a, int64_t b, char* p) {
        int64_t k = std::upper_bound(offsets, offsets + n_iv + 1, a + o0) - offsets - 1;  // run holding row a
        for (int64_t i = a + o0; i < b + o0; ++i) {
            while (i >= offsets[k + 1]) ++k;
            const int64_t pos = iv_start[k] + (i - offsets[k]);
            memcpy(p, contig, cl);
            p += cl;
            *p++ = '\t';
            p = put_i64(p, pos);
            *p++ = '\t';
            p = put_i64(p, pos + 1);
            *p++ = '\t';
            p = put(p, values[i]);
            *p++ = '\n';
        }
        return p;
    });
}

// ---- deflate: libdeflate when the runtime library is there (about twice zlib's speed), else zlib --------
struct Deflater {
    void* (*alloc)(int) = nullptr;
    size_t (*gzip)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*gzip_bound)(void*, size_t) = nullptr;
    size_t (*zlib_c)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*zlib_bound)(void*, size_t) = nullptr;
    void (*release)(void*) = nullptr;
    bool ok = false;
};

const Deflater& deflater() {
    static Deflater d = [] {
        Deflater x;
        if (getenv("FTK_NO_LIBDEFLATE")) return x;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return x;
        x.alloc = (decltype(x.alloc))dlsym(h, "libdeflate_alloc_compressor");
        x.gzip = (decltype(x.gzip))dlsym(h, "libdeflate_gzip_compress");
        x.gzip_bound = (decltype(x.gzip_bound))dlsym(h, "libdeflate_gzip_compress_bound");
        x.zlib_c = (decltype(x.zlib_c))dlsym(h, "libdeflate_zlib_compress");
        x.zlib_bound = (decltype(x.zlib_bound))dlsym(h, "libdeflate_zlib_compress_bound");
        x.release = (decltype(x.release))dlsym(h, "libdeflate_free_compressor");
        x.ok = x.alloc && x.gzip && x.gzip_bound && x.zlib_c && x.zlib_bound && x.release;
        return x;
    }();
    return d;
}

// one gzip member (gz) or one zlib stream holding `in`; returns the compressed size, 0 on failure
size_t deflate_block(void* ld, bool gz, int level, const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    const Deflater& D = deflater();
    if (ld) return gz ? D.gzip(ld, in, n, out, cap) : D.zlib_c(ld, in, n, out, cap);
    z_stream zs{};
    if (deflateInit2(&zs, std::min(level, 9), Z_DEFLATED, gz ? 15 + 16 : 15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
    zs.next_in = const_cast<Bytef*>(in);
    zs.avail_in = (uInt)n;
    zs.next_out = out;
    zs.avail_out = (uInt)cap;
    const int rc = deflate(&zs, Z_FINISH);
    const size_t got = rc == Z_STREAM_END ? (size_t)zs.total_out : 0;
    deflateEnd(&zs);
    return got;
}

size_t deflate_bound(bool gz, size_t n) {
    const Deflater& D = deflater();
    size_t b = compressBound((uLong)n) + 32;
    if (D.ok) {
        void* c = nullptr;  // bounds do not depend on the compressor object
        b = std::max(b, gz ? D.gzip_bound(c, n) : D.zlib_bound(c, n));
    }
    return b;
}

int write_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t w = write(fd, p, std::min<size_t>(n, size_t(1) << 30));
        if (w < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        p += w;
        n -= (size_t)w;
    }
    return 0;
}

}  // namespace

// Independent gzip members of 1 MB of text each, compressed in parallel, handed to `sink` in order (the body of
// ftk_file_write's .gz mode and of ftk_gzip_members: the same bytes either way).
template <class Sink>
static int gzip_members_to(const char* data, int64_t n, int gzip_level, int n_threads, Sink&& sink) {
    constexpr size_t kBlock = size_t(1) << 20;
    const int64_t n_blocks = (n + (int64_t)kBlock - 1) / (int64_t)kBlock;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : ftk_host::default_threads(), n_blocks));
    const size_t bound = deflate_bound(true, kBlock);
    const int64_t batch = (int64_t)nt * 4;  // blocks per round
    std::vector<uint8_t> outbuf((size_t)std::min<int64_t>(batch, n_blocks) * bound);
    std::vector<size_t> sizes((size_t)std::min<int64_t>(batch, n_blocks));
    const Deflater& D = deflater();
    std::vector<void*> comps(nt, nullptr);
    if (D.ok)
        for (auto& c : comps) c = D.alloc(std::min(gzip_level, 12));
    int rc = FTK_OK;
    for (int64_t b0 = 0; b0 < n_blocks && rc == FTK_OK; b0 += batch) {
        const int64_t nb = std::min(batch, n_blocks - b0);
        std::atomic<int64_t> next{0};
        std::atomic<bool> bad{false};
        ftk_host::parallel_run(nt, [&](int t) {
            for (;;) {
                const int64_t k = next.fetch_add(1);
                if (k >= nb) break;
                const size_t off = (size_t)(b0 + k) * kBlock;
                const size_t len = std::min(kBlock, (size_t)n - off);
                sizes[k] = deflate_block(comps[t], true, gzip_level, (const uint8_t*)data + off, len,
                                         outbuf.data() + (size_t)k * bound, bound);
                if (!sizes[k]) bad = true;
            }
        });
        if (bad) { rc = wfail(FTK_ERR_OOM, "deflate failed"); break; }
        for (int64_t k = 0; k < nb && rc == FTK_OK; ++k) rc = sink((const char*)outbuf.data() + (size_t)k * bound, sizes[k]);
    }
    if (D.ok)
        for (auto c : comps)
            if (c) D.release(c);
    return rc;
}

extern "C" {

int ftk_format_wig_i64(const int64_t* values, int64_t n, int n_threads, char** out, int64_t* out_len) {
    if (!out || !out_len || (n > 0 && !values)) return wfail(FTK_ERR_INVALID, "bad arguments");
    return format_rows(n, 22, n_threads, out, out_len, [values](int64_t a, int64_t b, char* p) {
        for (int64_t i = a; i < b; ++i) {
            p = put_i64(p, values[i]);
            *p++ = '\n';
        }
        return p;
    });
}

int ftk_format_bedgraph_i64(const char* contig, const int64_t* iv_start, const int64_t* offsets, int64_t n_iv,
                            const int64_t* values, int n_threads, char** out, int64_t* out_len) {
    return format_bedgraph(contig, iv_start, offsets, n_iv, values, 21, n_threads, out, out_len,
                           [](char* p, int64_t v) { return put_i64(p, v); });
}

int ftk_format_bedgraph_f64(const char* contig, const int64_t* iv_start, const int64_t* offsets, int64_t n_iv,
                            const double* values, int n_threads, char** out, int64_t* out_len) {
    return format_bedgraph(contig, iv_start, offsets, n_iv, values, 32, n_threads, out, out_len,
                           [](char* p, double v) { return put_f64(p, v); });
}

void ftk_buffer_free(void* p) { free(p); }

int ftk_file_write(const char* path, const char* data, int64_t n, int gzip_level, int n_threads, int append) {
    if (!path || n < 0 || (n > 0 && !data)) return wfail(FTK_ERR_INVALID, "bad arguments");
    const int fd = open(path, O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0644);
    if (fd < 0) return wfail(FTK_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    int rc = FTK_OK;
    if (gzip_level <= 0) {
        if (write_all(fd, data, (size_t)n)) rc = wfail(FTK_ERR_IO, "write to %s failed: %s", path, strerror(errno));
    } else if (n == 0) {
        if (!append) {  // an empty gzip member: a valid, empty .gz
            uint8_t buf[64];
            const size_t got = deflate_block(nullptr, true, gzip_level, (const uint8_t*)"", 0, buf, sizeof(buf));
            if (!got || write_all(fd, (const char*)buf, got)) rc = wfail(FTK_ERR_IO, "write to %s failed", path);
        }
    } else {
        rc = gzip_members_to(data, n, gzip_level, n_threads, [&](const char* p, size_t len) {
            return write_all(fd, p, len) ? wfail(FTK_ERR_IO, "write to %s failed: %s", path, strerror(errno)) : FTK_OK;
        });
    }
    if (close(fd) && rc == FTK_OK) rc = wfail(FTK_ERR_IO, "close of %s failed: %s", path, strerror(errno));
    return rc;
}

int ftk_gzip_members(const char* data, int64_t n, int gzip_level, int n_threads, char** out, int64_t* out_len) {
    if (!out || !out_len || n < 0 || (n > 0 && !data) || gzip_level <= 0) return wfail(FTK_ERR_INVALID, "bad arguments");
    *out = nullptr;
    *out_len = 0;
    std::vector<char> acc;
    if (n > 0) {
        const int rc = gzip_members_to(data, n, gzip_level, n_threads, [&](const char* p, size_t len) {
            acc.insert(acc.end(), p, p + len);
            return (int)FTK_OK;
        });
        if (rc != FTK_OK) return rc;
    }
    char* buf = (char*)malloc(acc.size() + 1);
    if (!buf) return wfail(FTK_ERR_OOM, "out of host memory");
    if (!acc.empty()) memcpy(buf, acc.data(), acc.size());
    *out = buf;
    *out_len = (int64_t)acc.size();
    return FTK_OK;
}

int ftk_fill_wps_records(void* dst, int64_t n, const uint32_t contig_ucs4[16], int64_t start, const int64_t* values,
                         int n_threads) {
    if (n < 0 || (n > 0 && (!dst || !values || !contig_ucs4))) return wfail(FTK_ERR_INVALID, "bad arguments");
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : ftk_host::default_threads(), (n + 65535) / 65536));
    ftk_host::parallel_run(nt, [&](int t) {
        const int64_t a = n * t / nt, b = n * (t + 1) / nt;
        uint8_t* p = (uint8_t*)dst + (size_t)a * 80;
        for (int64_t i = a; i < b; ++i, p += 80) {  // ('contig', 'U16'), ('start', 'i8'), ('wps', 'i8'): 64 + 8 + 8 bytes
            memcpy(p, contig_ucs4, 64);
            const int64_t pos = start + i;
            memcpy(p + 64, &pos, 8);
            memcpy(p + 72, values + i, 8);
        }
    });
    return FTK_OK;
}

int ftk_format_frag_rows(const char* contig, const int32_t* start, const int32_t* end, const uint8_t* mapq,
                         const uint8_t* strand, int64_t n, int bed6, int n_threads, char** out, int64_t* out_len) {
    if (!contig || !out || !out_len || (n > 0 && (!start || !end || !mapq || !strand)))
        return wfail(FTK_ERR_INVALID, "bad arguments");
    const size_t cl = strlen(contig);
    return format_rows(n, cl + 2 * 12 + 4 + 8, n_threads, out, out_len, [=](int64_t a, int64_t b, char* p) {
        for (int64_t i = a; i < b; ++i) {
            memcpy(p, contig, cl);
            p += cl;
            *p++ = '\t';
            p = put_i64(p, start[i]);
            *p++ = '\t';
            p = put_i64(p, end[i]);
            *p++ = '\t';
            if (bed6) { *p++ = '.'; *p++ = '\t'; }
            p = put_i64(p, mapq[i]);
            *p++ = '\t';
            *p++ = strand[i] ? '+' : '-';
            *p++ = '\n';
        }
        return p;
    });
}

int ftk_bgzf_write(const char* path, const char* data, int64_t n, int level, int n_threads, int append, int write_eof,
                   int64_t* block_offsets /* [ceil(n / 0xFF00) + 1] or NULL */) {
    if (!path || n < 0 || (n > 0 && !data)) return wfail(FTK_ERR_INVALID, "bad arguments");
    const int fd = open(path, O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0644);
    if (fd < 0) return wfail(FTK_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    int64_t file_pos = append ? (int64_t)lseek(fd, 0, SEEK_END) : 0;
    constexpr size_t kBlock = 0xFF00;  // uncompressed bytes per BGZF block (htslib's choice)
    const int64_t n_blocks = (n + (int64_t)kBlock - 1) / (int64_t)kBlock;
    const size_t bound = compressBound(kBlock) + 64;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : ftk_host::default_threads(), (n_blocks + 15) / 16));
    const int64_t batch = (int64_t)nt * 64;
    std::vector<uint8_t> outbuf((size_t)std::max<int64_t>(1, std::min(batch, n_blocks)) * bound);
    std::vector<size_t> sizes((size_t)std::max<int64_t>(1, std::min(batch, n_blocks)));
    const Deflater& D = deflater();
    // libdeflate's raw deflate entry point is not bound here: a BGZF block is a gzip member whose header carries
    // the BC extra field, so the member is built around a raw deflate stream from zlib (level as given) or, with
    // libdeflate, by re-wrapping its gzip member (10-byte header replaced by the 18-byte BGZF one).
    std::vector<void*> comps(nt, nullptr);
    if (D.ok)
        for (auto& c : comps) c = D.alloc(std::min(std::max(level, 1), 12));
    int rc = FTK_OK;
    for (int64_t b0 = 0; b0 < n_blocks && rc == FTK_OK; b0 += batch) {
        const int64_t nb = std::min(batch, n_blocks - b0);
        std::atomic<int64_t> next{0};
        std::atomic<bool> bad{false};
        ftk_host::parallel_run(nt, [&](int t) {
            std::vector<uint8_t> tmp(bound);
            for (;;) {
                const int64_t k = next.fetch_add(1);
                if (k >= nb) break;
                const size_t off = (size_t)(b0 + k) * kBlock;
                const size_t len = std::min(kBlock, (size_t)n - off);
                uint8_t* dst = outbuf.data() + (size_t)k * bound;
                // gzip member: 10-byte header, raw deflate, crc32, isize
                const size_t got = deflate_block(comps[t], true, level, (const uint8_t*)data + off, len, tmp.data(), bound);
                if (got < 18 || got - 10 + 18 > 65536) { bad = true; sizes[k] = 0; continue; }
                const size_t payload = got - 10 - 8;  // raw deflate bytes
                const size_t total = 18 + payload + 8;
                const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0,
                                          (uint8_t)((total - 1) & 255), (uint8_t)((total - 1) >> 8)};
                memcpy(dst, head, 18);
                memcpy(dst + 18, tmp.data() + 10, payload + 8);
                sizes[k] = total;
            }
        });
        if (bad) { rc = wfail(FTK_ERR_INVALID, "BGZF block did not fit (incompressible data)"); break; }
        for (int64_t k = 0; k < nb; ++k) {
            if (block_offsets) block_offsets[b0 + k] = file_pos;
            file_pos += (int64_t)sizes[k];
        }
        // the blocks of a batch are written with one call per block run: gather them first
        size_t w = 0;
        for (int64_t k = 0; k < nb; ++k) {
            if (w != (size_t)k * bound) memmove(outbuf.data() + w, outbuf.data() + (size_t)k * bound, sizes[k]);
            w += sizes[k];
        }
        if (write_all(fd, (const char*)outbuf.data(), w)) rc = wfail(FTK_ERR_IO, "write to %s failed: %s", path, strerror(errno));
    }
    if (D.ok)
        for (auto c : comps)
            if (c) D.release(c);
    if (rc == FTK_OK) {
        if (block_offsets) block_offsets[n_blocks] = file_pos;
        static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0,
                                        0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (write_eof && write_all(fd, (const char*)eof, sizeof(eof))) rc = wfail(FTK_ERR_IO, "write to %s failed", path);
    }
    if (close(fd) && rc == FTK_OK) rc = wfail(FTK_ERR_IO, "close of %s failed: %s", path, strerror(errno));
    return rc;
}

// ---- a synthetic paired-end BAM, one contig per call (test / bench tooling like ftk_format_frag_rows) ---------------
// BASELINE config 5 reads a whole-genome 60x BAM: 1.2 G records, ~145 GB of record bytes.  Assembled as numpy structured
// arrays the records of a chr1-sized contig alone took 53 s (round 4); here the host threads build, sort and deflate
// them window by window, so that writing the file is bound by libdeflate and the disk, not by Python.
namespace {

inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one BGZF block around `len` (<= 0xFF00) data bytes; returns its size, 0 on failure (dst holds >= bound bytes)
size_t bgzf_block(void* comp, int level, const uint8_t* in, size_t len, uint8_t* tmp, uint8_t* dst, size_t bound) {
    const size_t got = deflate_block(comp, true, level, in, len, tmp, bound);
    if (got < 18 || got - 10 + 18 > 65536) return 0;
    const size_t payload = got - 10 - 8;  // raw deflate bytes
    const size_t total = 18 + payload + 8;
    const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0,
                              (uint8_t)((total - 1) & 255), (uint8_t)((total - 1) >> 8)};
    memcpy(dst, head, 18);
    memcpy(dst + 18, tmp + 10, payload + 8);
    return total;
}

struct BamWindowJob {
    std::vector<uint8_t> out;           // the window's BGZF blocks
    std::vector<uint32_t> block_end;    // cumulative compressed size behind block k
    std::vector<std::pair<int64_t, int64_t>> first;  // (16 kb window, byte offset of its first overlapping record in the window's stream)
    int64_t n_records = 0;
    bool bad = false;
};

}  // namespace

int ftk_synth_bam_contig(const char* path, int32_t ref_id, int64_t contig_len, const int32_t* start, const int32_t* end,
                         const uint8_t* mapq, const uint8_t* strand, int64_t n, int32_t read_len, int32_t name_len,
                         uint64_t seed, int level, int n_threads, uint64_t* linear, int64_t n_linear,
                         int64_t* first_off, int64_t* end_off, int64_t* n_records_out) {
    if (!path || contig_len <= 0 || n < 0 || (n > 0 && (!start || !end || !mapq || !strand)) || read_len < 2 ||
        read_len > 1000 || name_len < 2 || name_len > 32 || (linear && n_linear < ((contig_len + read_len) >> 14) + 1))
        return wfail(FTK_ERR_INVALID, "bad arguments");
    const int fd = open(path, O_WRONLY | O_APPEND);
    if (fd < 0) return wfail(FTK_ERR_IO, "cannot open %s for appending: %s", path, strerror(errno));
    int64_t file_pos = (int64_t)lseek(fd, 0, SEEK_END);
    if (first_off) *first_off = -1;
    if (end_off) *end_off = -1;
    int64_t max_len = 0;
    for (int64_t i = 0; i < n; ++i) max_len = std::max<int64_t>(max_len, (int64_t)end[i] - start[i]);
    const size_t rec_bytes = 36 + (size_t)name_len + 4 + (size_t)(read_len + 1) / 2 + (size_t)read_len;
    constexpr int64_t kStep = 1 << 19;  // bases per window: ~24 MB of records at 60x
    constexpr size_t kBlock = 0xFF00;
    const int64_t n_win = (contig_len + kStep - 1) / kStep;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : ftk_host::default_threads(), n_win));
    const size_t bound = compressBound(kBlock) + 64;
    const Deflater& D = deflater();
    std::vector<void*> comps(nt, nullptr);
    if (D.ok)
        for (auto& c : comps) c = D.alloc(std::min(std::max(level, 1), 12));
    uint8_t qual_lut[256];  // 3 / 7 / 20 / 70 % of the qualities: 2, 11, 25, 37 (synth.write_paired_bam)
    for (int k = 0; k < 256; ++k) qual_lut[k] = k < 8 ? 2 : k < 26 ? 11 : k < 77 ? 25 : 37;
    const int64_t batch = (int64_t)nt * 2;
    int rc = FTK_OK;
    int64_t n_records = 0;
    // two sets of window jobs: while one batch's blocks go to the file (a writer thread: the disk is the slowest stage
    // wherever the file does not fit the page cache), the host threads build the next batch
    std::vector<BamWindowJob> jobs_ab[2];
    jobs_ab[0].resize((size_t)std::min(batch, n_win));
    jobs_ab[1].resize((size_t)std::min(batch, n_win));
    std::thread writer;
    int write_rc = FTK_OK;
    int side = 0;
    for (int64_t w0 = 0; w0 < n_win && rc == FTK_OK; w0 += batch, side ^= 1) {
        std::vector<BamWindowJob>& jobs = jobs_ab[side];
        const int64_t nw = std::min(batch, n_win - w0);
        std::atomic<int64_t> next{0};
        ftk_host::parallel_run(nt, [&](int t) {
            std::vector<uint64_t> keys;
            std::vector<uint8_t> raw, tmp(bound);
            for (;;) {
                const int64_t k = next.fetch_add(1);
                if (k >= nw) break;
                BamWindowJob& J = jobs[(size_t)k];
                J.out.clear(); J.block_end.clear(); J.first.clear(); J.n_records = 0; J.bad = false;
                const int64_t a = (w0 + k) * kStep, b = std::min(a + kStep, contig_len);
                const int64_t lo = std::lower_bound(start, start + n, (int32_t)std::max<int64_t>(a - max_len, INT32_MIN)) - start;
                const int64_t hi = std::lower_bound(start, start + n, (int32_t)std::min<int64_t>(b, INT32_MAX)) - start;
                // records of the window: key = (pos - a) << 40 | mate << 39 | fragment index (stable order of the numpy
                // writer: by position, read1 records before read2 records, then by fragment)
                keys.clear();
                for (int64_t i = lo; i < hi; ++i) {
                    const bool fwd = strand[i] != 0;
                    const int64_t p1 = fwd ? start[i] : (int64_t)end[i] - read_len, p2 = fwd ? (int64_t)end[i] - read_len : start[i];
                    if (p1 >= a && p1 < b) keys.push_back((uint64_t)(p1 - a) << 40 | (uint64_t)i);
                    if (p2 >= a && p2 < b) keys.push_back((uint64_t)(p2 - a) << 40 | 1ull << 39 | (uint64_t)i);
                }
                std::sort(keys.begin(), keys.end());
                J.n_records = (int64_t)keys.size();
                raw.resize(keys.size() * rec_bytes);
                uint8_t* p = raw.data();
                int64_t seen_w = -1;  // highest 16 kb window a record of this job has touched (records come by position)
                for (size_t r = 0; r < keys.size(); ++r, p += rec_bytes) {
                    const uint64_t key = keys[r];
                    const int64_t i = (int64_t)(key & ((1ull << 39) - 1));
                    const bool mate2 = (key >> 39) & 1;
                    const int64_t pos = a + (int64_t)(key >> 40);
                    const bool fwd = strand[i] != 0;
                    const int64_t len = (int64_t)end[i] - start[i];
                    const int64_t p1 = fwd ? start[i] : (int64_t)end[i] - read_len, p2 = fwd ? (int64_t)end[i] - read_len : start[i];
                    int32_t h[9];
                    h[0] = (int32_t)rec_bytes - 4;
                    h[1] = ref_id;
                    h[2] = (int32_t)pos;
                    h[3] = (int32_t)((uint32_t)name_len | (uint32_t)mapq[i] << 8);  // l_read_name, mapq, bin = 0
                    const uint32_t flag = mate2 ? (fwd ? 147u : 163u) : (fwd ? 99u : 83u);
                    h[4] = (int32_t)(1u | flag << 16);                               // n_cigar_op = 1, flag
                    h[5] = read_len;
                    h[6] = ref_id;
                    h[7] = (int32_t)(mate2 ? p1 : p2);
                    h[8] = (int32_t)((mate2 != fwd) ? len : -len);                   // read1 fwd: +len; read1 rev: -len; mates mirrored
                    memcpy(p, h, 36);
                    uint8_t* q = p + 36;
                    int64_t v = i;
                    for (int d = name_len - 2; d >= 0; --d) { q[d] = (uint8_t)('0' + v % 10); v /= 10; }
                    q[name_len - 1] = 0;
                    q += name_len;
                    const uint32_t cig = (uint32_t)read_len << 4;
                    memcpy(q, &cig, 4);
                    q += 4;
                    uint64_t st = seed ^ ((uint64_t)ref_id << 48) ^ ((uint64_t)i << 1) ^ (uint64_t)mate2;
                    const size_t n_seq = (size_t)(read_len + 1) / 2;
                    for (size_t j = 0; j < n_seq; j += 8) {
                        const uint64_t x = splitmix64(st);
                        memcpy(q + j, &x, std::min<size_t>(8, n_seq - j));
                    }
                    q += n_seq;
                    for (size_t j = 0; j < (size_t)read_len; j += 8) {
                        uint64_t x = splitmix64(st);
                        for (size_t u = j; u < std::min<size_t>(j + 8, read_len); ++u, x >>= 8) q[u] = qual_lut[x & 255];
                    }
                    // linear index: the first record overlapping each 16 kb window (a read covers at most two)
                    const int64_t wb = (pos + read_len - 1) >> 14;
                    for (int64_t wq = std::max(pos >> 14, seen_w + 1); wq <= wb; ++wq) J.first.emplace_back(wq, (int64_t)(r * rec_bytes));
                    seen_w = std::max(seen_w, wb);
                }
                const size_t n_blocks = (raw.size() + kBlock - 1) / kBlock;
                J.out.resize(n_blocks * bound);
                size_t w = 0;
                for (size_t bk = 0; bk < n_blocks; ++bk) {
                    const size_t off = bk * kBlock, len = std::min(kBlock, raw.size() - off);
                    const size_t got = bgzf_block(comps[t], level, raw.data() + off, len, tmp.data(), J.out.data() + w, bound);
                    if (!got) { J.bad = true; break; }
                    w += got;
                    J.block_end.push_back((uint32_t)w);
                }
                J.out.resize(w);
            }
        });
        if (writer.joinable()) writer.join();  // the previous batch is on the disk
        if (write_rc != FTK_OK) { rc = write_rc; break; }
        for (int64_t k = 0; k < nw && rc == FTK_OK; ++k) {
            BamWindowJob& J = jobs[(size_t)k];
            if (J.bad) { rc = wfail(FTK_ERR_INVALID, "BGZF block did not fit (incompressible data)"); break; }
            if (J.out.empty()) continue;
            if (first_off && *first_off < 0) *first_off = file_pos;
            if (linear)
                for (const auto& f : J.first) {
                    const size_t bk = (size_t)f.second / kBlock;
                    const uint64_t voff = (uint64_t)(file_pos + (bk ? J.block_end[bk - 1] : 0)) << 16 | (uint64_t)((size_t)f.second % kBlock);
                    if (f.first >= 0 && f.first < n_linear) linear[f.first] = std::min(linear[f.first], voff);
                }
            file_pos += (int64_t)J.out.size();
            n_records += J.n_records;
        }
        if (rc != FTK_OK) break;
        writer = std::thread([&jobs, nw, fd, &write_rc, path] {
            for (int64_t k = 0; k < nw; ++k) {
                const BamWindowJob& J = jobs[(size_t)k];
                if (!J.out.empty() && write_all(fd, (const char*)J.out.data(), J.out.size())) {
                    write_rc = wfail(FTK_ERR_IO, "write to %s failed: %s", path, strerror(errno));
                    return;
                }
            }
        });
    }
    if (writer.joinable()) writer.join();
    if (rc == FTK_OK) rc = write_rc;
    if (D.ok)
        for (auto c : comps)
            if (c) D.release(c);
    if (end_off) *end_off = file_pos;
    if (n_records_out) *n_records_out = n_records;
    if (close(fd) && rc == FTK_OK) rc = wfail(FTK_ERR_IO, "close of %s failed: %s", path, strerror(errno));
    return rc;
}

int ftk_bigwig_fixedstep_sections(uint32_t chrom_id, const int64_t* iv_start, const int64_t* offsets, int64_t n_iv,
                                  const void* values, int value_kind, int32_t items_per_section, int level,
                                  int n_threads, char** out, int64_t* out_len, int64_t* n_sections_out,
                                  int64_t** section_table_out, double** section_stats_out) {
    if (!out || !out_len || !n_sections_out || !section_table_out || !section_stats_out || n_iv < 0 ||
        (n_iv > 0 && (!iv_start || !offsets)) || items_per_section <= 0 || items_per_section > 65535 ||
        (value_kind != 0 && value_kind != 1))
        return wfail(FTK_ERR_INVALID, "bad arguments");
    *out = nullptr;
    *out_len = 0;
    *n_sections_out = 0;
    *section_table_out = nullptr;
    *section_stats_out = nullptr;
    const int64_t ips = items_per_section;
    // sections never span runs (each run is its own addEntries call): first section of every run
    std::vector<int64_t> first(n_iv + 1, 0);
    for (int64_t k = 0; k < n_iv; ++k) {
        const int64_t len = offsets[k + 1] - offsets[k];
        if (len < 0) return wfail(FTK_ERR_INVALID, "offsets must not decrease");
        if (len > 0 && (iv_start[k] < 0 || iv_start[k] + len > (int64_t)UINT32_MAX))
            return wfail(FTK_ERR_INVALID, "interval outside the bigWig coordinate range");
        first[k + 1] = first[k] + (len + ips - 1) / ips;
    }
    const int64_t n_sec = first[n_iv];
    if (n_sec > 0 && !values) return wfail(FTK_ERR_INVALID, "values is NULL");
    // per section: start, end, compressed bytes
    int64_t* table = (int64_t*)malloc((size_t)std::max<int64_t>(n_sec, 1) * 3 * sizeof(int64_t));
    double* stats = (double*)malloc((size_t)std::max<int64_t>(n_sec, 1) * 4 * sizeof(double));
    const size_t raw_max = 24 + 4 * (size_t)ips;
    const size_t bound = deflate_bound(false, raw_max);
    uint8_t* scratch = n_sec ? (uint8_t*)malloc((size_t)n_sec * bound) : nullptr;
    if (!table || !stats || (n_sec && !scratch)) {
        free(table); free(stats); free(scratch);
        return wfail(FTK_ERR_OOM, "out of host memory");
    }
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : ftk_host::default_threads(), n_sec));
    const Deflater& D = deflater();
    std::atomic<int64_t> next{0};
    std::atomic<bool> bad{false};
    if (n_sec) ftk_host::parallel_run(nt, [&](int) {
        void* comp = D.ok ? D.alloc(std::min(level, 12)) : nullptr;
        std::vector<uint8_t> raw(raw_max);
        constexpr int64_t kGrab = 8;
        for (;;) {
            const int64_t s0 = next.fetch_add(kGrab);
            if (s0 >= n_sec) break;
            int64_t k = std::upper_bound(first.begin(), first.end(), s0) - first.begin() - 1;  // run of section s0
            for (int64_t sidx = s0; sidx < std::min(s0 + kGrab, n_sec); ++sidx) {
                while (sidx >= first[k + 1]) ++k;
                const int64_t a = offsets[k] + (sidx - first[k]) * ips;
                const int64_t cnt = std::min(ips, offsets[k + 1] - a);
                const int64_t pos = iv_start[k] + (a - offsets[k]);
                // bigWig data section: chromId, chromStart, chromEnd, itemStep, itemSpan, type 3 (fixedStep),
                // reserved, itemCount, then float32 values
                const uint32_t head[5] = {chrom_id, (uint32_t)pos, (uint32_t)(pos + cnt), 1u, 1u};
                memcpy(raw.data(), head, 20);
                raw[20] = 3;
                raw[21] = 0;
                const uint16_t cnt16 = (uint16_t)cnt;
                memcpy(raw.data() + 22, &cnt16, 2);
                float* fv = reinterpret_cast<float*>(raw.data() + 24);
                double mn = std::numeric_limits<double>::infinity(), mx = -mn, sum = 0.0, sq = 0.0;
                for (int64_t i = 0; i < cnt; ++i) {
                    const double d = value_kind == 0 ? (double)((const int64_t*)values)[a + i] : ((const double*)values)[a + i];
                    const float f = (float)d;  // values.astype(float64).astype(float32)
                    fv[i] = f;
                    const double v = (double)f;
                    mn = std::min(mn, v);
                    mx = std::max(mx, v);
                    sum += v;
                    sq += v * v;
                }
                stats[4 * sidx + 0] = mn;
                stats[4 * sidx + 1] = mx;
                stats[4 * sidx + 2] = sum;
                stats[4 * sidx + 3] = sq;
                const size_t got = deflate_block(comp, false, level, raw.data(), 24 + 4 * (size_t)cnt,
                                                 scratch + (size_t)sidx * bound, bound);
                if (!got) bad = true;
                table[3 * sidx + 0] = pos;
                table[3 * sidx + 1] = pos + cnt;
                table[3 * sidx + 2] = (int64_t)got;
            }
        }
        if (comp) D.release(comp);
    });
    size_t total = 0;
    for (int64_t k = 0; k < n_sec; ++k) total += (size_t)table[3 * k + 2];
    char* all = bad ? nullptr : (char*)malloc(total + 1);
    if (!all) {
        free(table); free(stats); free(scratch);
        return wfail(bad ? FTK_ERR_INVALID : FTK_ERR_OOM, bad ? "deflate failed" : "out of host memory");
    }
    size_t off = 0;
    for (int64_t k = 0; k < n_sec; ++k) {
        memcpy(all + off, scratch + (size_t)k * bound, (size_t)table[3 * k + 2]);
        off += (size_t)table[3 * k + 2];
    }
    free(scratch);
    *out = all;
    *out_len = (int64_t)total;
    *n_sections_out = n_sec;
    *section_table_out = table;
    *section_stats_out = stats;
    return FTK_OK;
}

}  // extern "C"
