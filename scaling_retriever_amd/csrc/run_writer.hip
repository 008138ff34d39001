// run.json straight from the result arrays (host code; no kernel in this file).
//
// The reference's retrieval drivers turn the top-k arrays into a nested dict with a per-hit Python loop and json.dump it:
//   eval_dense.py:225-241          qid_to_rankdata[str(qid)][str(docid)] = float(score); ujson.dump(...)
//   scaling_retriever/indexer.py:405-474,530-540   res[str(qid)][str(doc_ids[id_])] = float(sc); json.dump(res, handler)
// At MS MARCO Dev (6 980 queries x 1 000 hits) that loop is 7 M dict insertions + a 217 MB dump: ~15 s of host time behind a
// 0.25 s search.  sr_write_run_json writes the same file from the arrays the search returns - byte for byte what Python's
// json.dump writes for that dict (", " / ": " separators, keys in row order, float repr of the fp32 score widened to double:
// shortest round-trip digits, exponent form outside 1e-4 <= |x| < 1e16, "NaN" / "Infinity" / "-Infinity") - formatting the
// queries in parallel into per-thread buffers and writing them in order.
#include "common.h"

#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace {

// float.__repr__ (CPython format_float_short, mode 'r', Py_DTSF_ADD_DOT_0) of v
inline char* py_float_repr(char* out, double v) {
    if (std::isnan(v)) { memcpy(out, "NaN", 3); return out + 3; }
    if (std::isinf(v)) {
        if (v < 0) { memcpy(out, "-Infinity", 9); return out + 9; }
        memcpy(out, "Infinity", 8);
        return out + 8;
    }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest round trip
    const char* p = sci;
    if (*p == '-') { *out++ = '-'; ++p; }
    char digits[24];
    int nd = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int ex = 0;
    {
        const char* q = p + 1;
        const bool neg = *q == '-';
        ++q;
        for (; q < r.ptr; ++q) ex = ex * 10 + (*q - '0');
        if (neg) ex = -ex;
    }
    while (nd > 1 && digits[nd - 1] == '0') --nd;        // to_chars does not pad, but stay safe
    const int decpt = ex + 1;                              // value = 0.d1 d2 ... x 10^decpt
    if (decpt <= -4 || decpt > 16) {                       // exponent form: d[.ddd]e[+-]XX (at least two exponent digits)
        *out++ = digits[0];
        if (nd > 1) {
            *out++ = '.';
            memcpy(out, digits + 1, (size_t)nd - 1);
            out += nd - 1;
        }
        *out++ = 'e';
        int e = decpt - 1;
        *out++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + e % 10); e /= 10; } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *out++ = eb[--ne];
        return out;
    }
    if (decpt <= 0) {
        *out++ = '0';
        *out++ = '.';
        for (int i = 0; i < -decpt; ++i) *out++ = '0';
        memcpy(out, digits, (size_t)nd);
        return out + nd;
    }
    if (decpt >= nd) {
        memcpy(out, digits, (size_t)nd);
        out += nd;
        for (int i = 0; i < decpt - nd; ++i) *out++ = '0';
        *out++ = '.';
        *out++ = '0';
        return out;
    }
    memcpy(out, digits, (size_t)decpt);
    out += decpt;
    *out++ = '.';
    memcpy(out, digits + decpt, (size_t)(nd - decpt));
    return out + (nd - decpt);
}

inline char* put_i64(char* out, int64_t v) {
    const auto r = std::to_chars(out, out + 24, v);
    return r.ptr;
}

struct Keys {
    const int64_t* i64;      // decimal keys, or
    const char* bytes;       // body of a JSON string per key (already escaped): offsets off[i] .. off[i + 1], or - off == NULL -
    const int64_t* off;      // fixed-width entries of `width` bytes, NUL-padded (a numpy 'S' array of ASCII ids)
    int64_t width;
    inline size_t max_len(int64_t i) const { return i64 ? 21 : (off ? (size_t)(off[i + 1] - off[i]) : (size_t)width); }
    inline char* put(char* out, int64_t i) const {
        *out++ = '"';
        if (i64) out = put_i64(out, i64[i]);
        else if (!off) {
            const char* p = bytes + i * width;
            for (int64_t c = 0; c < width && p[c]; ++c) *out++ = p[c];
        } else {
            const size_t n = (size_t)(off[i + 1] - off[i]);
            memcpy(out, bytes + off[i], n);
            out += n;
        }
        *out++ = '"';
        return out;
    }
};

}  // namespace

static std::atomic<int64_t> g_mapped_rounds{0};
// rounds of sr_write_run_json that went through the shared file mapping since the library was loaded (test hook)
extern "C" int64_t sr_run_writer_mapped_rounds(void) { return g_mapped_rounds.load(); }

static int write_run_json_impl(const char* path, int64_t nq, int64_t k, const float* h_scores, const int64_t* h_idx,
                               const int32_t* h_counts, const int64_t* h_qid_i64, const char* h_qid_bytes, const int64_t* h_qid_off,
                               int64_t qid_width, const int64_t* h_doc_i64, const char* h_doc_bytes, const int64_t* h_doc_off,
                               int64_t doc_width, int64_t n_docs, int32_t n_threads, int64_t* bytes_written, int part);

extern "C" int sr_write_run_json(const char* path, int64_t nq, int64_t k, const float* h_scores, const int64_t* h_idx,
                                 const int32_t* h_counts, const int64_t* h_qid_i64, const char* h_qid_bytes, const int64_t* h_qid_off,
                                 int64_t qid_width, const int64_t* h_doc_i64, const char* h_doc_bytes, const int64_t* h_doc_off,
                                 int64_t doc_width, int64_t n_docs, int32_t n_threads, int64_t* bytes_written) {
    return write_run_json_impl(path, nq, k, h_scores, h_idx, h_counts, h_qid_i64, h_qid_bytes, h_qid_off, qid_width, h_doc_i64, h_doc_bytes,
                               h_doc_off, doc_width, n_docs, n_threads, bytes_written, 0);
}

// The same file in pieces (the GPU searches the next piece of the query set while the host writes this one): part 1 = first piece
// (creates the file, no closing brace), 2 = a middle piece (appends), 3 = the last piece (appends and closes).  The bytes of the finished
// file are those of ONE sr_write_run_json call over the concatenated pieces.  *bytes_written = file size so far.
extern "C" int sr_write_run_json_part(const char* path, int32_t part, int64_t nq, int64_t k, const float* h_scores, const int64_t* h_idx,
                                      const int32_t* h_counts, const int64_t* h_qid_i64, const char* h_qid_bytes, const int64_t* h_qid_off,
                                      int64_t qid_width, const int64_t* h_doc_i64, const char* h_doc_bytes, const int64_t* h_doc_off,
                                      int64_t doc_width, int64_t n_docs, int32_t n_threads, int64_t* bytes_written) {
    SR_REQUIRE(part >= 1 && part <= 3, "sr_write_run_json_part: part must be 1 (first), 2 (middle) or 3 (last)");
    return write_run_json_impl(path, nq, k, h_scores, h_idx, h_counts, h_qid_i64, h_qid_bytes, h_qid_off, qid_width, h_doc_i64, h_doc_bytes,
                               h_doc_off, doc_width, n_docs, n_threads, bytes_written, part);
}

static int write_run_json_impl(const char* path, int64_t nq, int64_t k, const float* h_scores, const int64_t* h_idx,
                               const int32_t* h_counts, const int64_t* h_qid_i64, const char* h_qid_bytes, const int64_t* h_qid_off,
                               int64_t qid_width, const int64_t* h_doc_i64, const char* h_doc_bytes, const int64_t* h_doc_off,
                               int64_t doc_width, int64_t n_docs, int32_t n_threads, int64_t* bytes_written, int part) {
    SR_REQUIRE(path && nq >= 0 && k >= 0, "sr_write_run_json: bad argument");
    SR_REQUIRE(nq == 0 || k == 0 || (h_scores && h_idx), "sr_write_run_json: null result arrays");
    SR_REQUIRE(nq == 0 || h_qid_i64 || (h_qid_bytes && (h_qid_off || qid_width > 0)), "sr_write_run_json: no query ids");
    SR_REQUIRE(n_docs >= 0 && (n_docs == 0 || h_doc_i64 || (h_doc_bytes && (h_doc_off || doc_width > 0))),
               "sr_write_run_json: no document ids");
    const Keys qk{h_qid_i64, h_qid_bytes, h_qid_off, qid_width}, dk{h_doc_i64, h_doc_bytes, h_doc_off, doc_width};
    const bool opens = part == 0 || part == 1, closes = part == 0 || part == 3;
    // O_RDWR: a shared writable mapping needs read access (EACCES with O_WRONLY)
    const int fd = open(path, opens ? (O_RDWR | O_CREAT | O_TRUNC) : O_RDWR, 0644);
    if (fd < 0) {
        sr_set_error("sr_write_run_json: cannot open %s: %s", path, strerror(errno));
        return SR_ERR_INVALID;
    }
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((int64_t)nt > nq) nt = nq > 0 ? (int)nq : 1;
    // Thread t formats a contiguous run of queries (a slab) into its own buffer; once a round of `nt` slabs is formatted the
    // buffers' file offsets are known and every thread writes its own buffer with pwrite (the copy into the page cache is most of
    // the time of a 200 MB file and parallelises like the formatting).  Slabs are sized so that a Dev-sized result is one round.
    int64_t slab = (nq + nt - 1) / (nt > 0 ? nt : 1);
    if (slab < 32) slab = 32;
    if (slab > 1024) slab = 1024;
    std::vector<std::string> bufs((size_t)nt);
    std::vector<int> bad((size_t)nt, 0), io_bad((size_t)nt, 0);
    std::vector<int64_t> off((size_t)nt, 0);
    std::vector<size_t> skip((size_t)nt, 0);
    int64_t total = 1;                                        // the opening brace
    bool ok = true, any_entry = false;
    if (opens) ok = pwrite(fd, "{", 1, 0) == 1;
    else {                                                    // a continuation: behind what the earlier pieces wrote
        const off_t end = lseek(fd, 0, SEEK_END);
        ok = end >= 1;
        total = (int64_t)end;
        any_entry = end > 1;
    }
    const bool timing = getenv("SR_RUN_WRITER_TIMING") != nullptr;      // per-phase wall times on stderr
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_fmt = 0, t_copy = 0, t_alloc = 0;
    const double t_begin = now_ms();
    for (int64_t q0 = 0; q0 < nq && ok; q0 += slab * nt) {
        const double t_r0 = now_ms();
        auto work = [&](int t) {
            const int64_t a = q0 + (int64_t)t * slab, b = a + slab < nq ? a + slab : nq;
            std::string& s = bufs[(size_t)t];
            s.clear();
            if (a >= b) return;
            size_t used = 0;
            for (int64_t q = a; q < b; ++q) {
                const int64_t n = h_counts ? (h_counts[q] < k ? (h_counts[q] < 0 ? 0 : h_counts[q]) : k) : k;
                // worst case of this query's text
                size_t need = qk.max_len(q) + 8;
                for (int64_t j = 0; j < n; ++j) {
                    const int64_t d = h_idx[q * k + j];
                    if (d >= 0 && d < n_docs) need += dk.max_len(d) + 2 + 2 + 26 + 2;
                }
                if (s.size() < used + need) s.resize((used + need) * 2);
                char* const o_begin = &s[used];
                char* o = o_begin;
                *o++ = ','; *o++ = ' ';                        // the writer drops the separator of the file's first entry
                o = qk.put(o, q);
                *o++ = ':'; *o++ = ' '; *o++ = '{';
                bool first = true;
                for (int64_t j = 0; j < n; ++j) {
                    const int64_t d = h_idx[q * k + j];
                    if (d < 0) continue;                       // padding (fewer than k hits)
                    if (d >= n_docs) { bad[(size_t)t] = 1; continue; }
                    if (!first) { *o++ = ','; *o++ = ' '; }
                    first = false;
                    o = dk.put(o, d);
                    *o++ = ':'; *o++ = ' ';
                    o = py_float_repr(o, (double)h_scores[q * k + j]);
                }
                *o++ = '}';
                // a query without a hit has no entry: the reference creates qid's dict with its first hit (eval_dense.py:228-234,
                // indexer.py:431-432)
                if (!first) used = (size_t)(o - s.data());
            }
            s.resize(used);
        };
        auto put = [&](int t) {
            const std::string& s = bufs[(size_t)t];
            size_t done = skip[(size_t)t];
            while (done < s.size()) {
                const ssize_t w = pwrite(fd, s.data() + done, s.size() - done, (off_t)(off[(size_t)t] + (int64_t)(done - skip[(size_t)t])));
                if (w <= 0) { io_bad[(size_t)t] = 1; return; }
                done += (size_t)w;
            }
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
            work(0);
            for (auto& x : th) x.join();
        }
        const double t_r1 = now_ms();
        t_fmt += t_r1 - t_r0;
        for (int t = 0; t < nt; ++t) {
            const std::string& s = bufs[(size_t)t];
            skip[(size_t)t] = (!s.empty() && !any_entry) ? 2 : 0;          // the file's first entry has no separator in front
            if (!s.empty()) any_entry = true;
            off[(size_t)t] = total;
            total += (int64_t)(s.size() - skip[(size_t)t]);
        }
        // Large rounds go through a shared mapping of the file: buffered writes to one file serialise on its inode (the copy of a
        // 200 MB run into the page cache was most of the call), page faults of a mapping do not.  The blocks are reserved first
        // (posix_fallocate: a full disk is an error code here, not a SIGBUS in a thread); any failure falls back to pwrite.
        const int64_t round_begin = off[0], round_bytes = total - off[0];
        bool mapped = false;
        const bool reserved = round_bytes >= (8 << 20) && nt > 1 && posix_fallocate(fd, 0, (off_t)total) == 0;
        const double t_r2 = now_ms();
        t_alloc += t_r2 - t_r1;
        if (reserved) {
            const int64_t page = (int64_t)sysconf(_SC_PAGESIZE);
            const int64_t map_begin = round_begin / page * page;
            void* m = mmap(nullptr, (size_t)(total - map_begin), PROT_WRITE, MAP_SHARED, fd, (off_t)map_begin);
            if (m != MAP_FAILED) {
                char* base = static_cast<char*>(m) - map_begin;          // base + file offset
                auto copy = [&](int t) {
                    const std::string& s = bufs[(size_t)t];
                    if (s.size() > skip[(size_t)t]) memcpy(base + off[(size_t)t], s.data() + skip[(size_t)t], s.size() - skip[(size_t)t]);
                };
                // The copy is page faults of a fresh mapping, and those contend in the kernel: 64 copying threads take 60-75 ms for a 70 MB round
                // where 16 take 18-22 (tools/micro/run_writer_threads.py; formatting, which is pure CPU, wants all 64).  So at most 16 copiers,
                // each walking a contiguous run of the round's buffers.
                const int nc = nt < 16 ? nt : 16;
                auto copy_run = [&](int c) {
                    for (int t = (int)((int64_t)c * nt / nc); t < (int)((int64_t)(c + 1) * nt / nc); ++t) copy(t);
                };
                std::vector<std::thread> th;
                for (int c = 1; c < nc; ++c) th.emplace_back(copy_run, c);
                copy_run(0);
                for (auto& x : th) x.join();
                mapped = munmap(m, (size_t)(total - map_begin)) == 0;
                if (mapped) g_mapped_rounds.fetch_add(1);
            }
        }
        if (!mapped) {
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t)
                if (!bufs[(size_t)t].empty()) th.emplace_back(put, t);
            put(0);
            for (auto& x : th) x.join();
        }
        for (int t = 0; t < nt; ++t) ok = ok && !io_bad[(size_t)t];
        t_copy += now_ms() - t_r2;
    }
    if (timing)
        fprintf(stderr, "[run writer] %lld queries, %d threads, slabs of %lld: format %.1f ms, reserve %.1f ms, copy into the file %.1f ms, total %.1f ms\n",
                (long long)nq, nt, (long long)slab, t_fmt, t_alloc, t_copy, now_ms() - t_begin);
    if (closes) {
        ok = ok && pwrite(fd, "}", 1, (off_t)total) == 1;
        total += 1;
    }
    ok = (close(fd) == 0) && ok;
    for (int t = 0; t < nt; ++t)
        if (bad[(size_t)t]) {
            sr_set_error("sr_write_run_json: a result index lies outside the document id table of %lld entries", (long long)n_docs);
            return SR_ERR_INVALID;
        }
    if (!ok) {
        sr_set_error("sr_write_run_json: write to %s failed: %s", path, strerror(errno));
        return SR_ERR_INVALID;
    }
    if (bytes_written) *bytes_written = total;
    return SR_OK;
}
