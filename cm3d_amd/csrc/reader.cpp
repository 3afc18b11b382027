// libcm3d_reader.so: host-side loader of the lifting path (include/cm3d_reader.h).
//   reference: src/nuscenes/2d_to_3d.py:422-428 (pickle.load of <f>_masks.pkl + pycocotools decode) and :437-441 with
//   utils/pcd.py:246-257 (np.fromfile of every sweep) -- per frame, single-threaded Python there; per batch, on a thread
//   pool, straight into the caller's page-locked staging buffers here.
// Plain C++17 + pthreads; no HIP, no Python.
#include "../../include/cm3d_reader.h"

#include <atomic>
#include <limits>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------- thread pool
// parallel_for over [0, n): items are handed out one by one through an atomic counter (files differ in size)
struct Pool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> fn;
    std::atomic<int> next{0};
    int n = 0, active = 0;
    uint64_t generation = 0;
    bool stop = false;

    explicit Pool(int nt)
    {
        for (int t = 0; t < nt; ++t) threads.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n) break;
                fn(i);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }
    void run(int count, std::function<void(int)> f)
    {
        if (count <= 0) return;
        if (threads.empty() || count == 1) {
            for (int i = 0; i < count; ++i) f(i);
            return;
        }
        std::unique_lock<std::mutex> lk(mu);
        fn = std::move(f);
        n = count;
        next = 0;
        active = (int)threads.size();
        ++generation;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
    }
};

bool read_fully(int fd, void *dst, size_t bytes)
{
    char *p = (char *)dst;
    size_t off = 0;
    while (off < bytes) {
        const ssize_t got = pread(fd, p + off, bytes - off, (off_t)off);
        if (got <= 0) return false;
        off += (size_t)got;
    }
    return true;
}

bool read_file(const char *path, std::vector<uint8_t> &buf)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return false; }
    buf.resize((size_t)st.st_size);
    const bool ok = read_fully(fd, buf.data(), buf.size());
    close(fd);
    return ok;
}

// ---------------------------------------------------------------------------------------------- RLE strings
// pycocotools rleFrString: 5-bit groups (char - 48), bit 0x20 = continuation, bit 0x10 of the last group = sign
// extension; values from the 4th on are deltas against the value two positions back.
int64_t rle_from_string(const uint8_t *s, int64_t len, uint32_t *out, int64_t cap)
{
    int64_t m = 0, p = 0;
    long long prev1 = 0, prev2 = 0;           // values at m-1 and m-2
    while (p < len) {
        long long x = 0;
        int k = 0;
        bool more = true;
        while (more) {
            if (p >= len || k > 12) return CM3D_RD_ERR_FORMAT;
            const long long c = (long long)s[p] - 48;
            x |= (c & 0x1f) << (5 * k);
            more = (c & 0x20) != 0;
            ++p;
            ++k;
            if (!more && (c & 0x10) && 5 * k < 64) x |= (long long)(~0ull << (5 * k));     // (13 groups fill all 64 bits: nothing to extend)
        }
        if (m > 2) x += prev2;
        if (x < 0 || x > 0xFFFFFFFFll) return CM3D_RD_ERR_FORMAT;
        if (out) {
            if (m >= cap) return CM3D_RD_ERR_CAPACITY;
            out[m] = (uint32_t)x;
        }
        prev2 = prev1;
        prev1 = x;
        ++m;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------- pickle subset
// Values a mask file can hold: ints, bytes / str (views into the file), lists / tuples, dicts, None / bools / floats.
struct Val {
    enum Kind : uint8_t { NONE, INT, BYTES, LIST, DICT, MARK, OTHER, ENCODE_FN } kind = NONE;     // ENCODE_FN = the global _codecs.encode
    long long i = 0;                // INT
    const uint8_t *p = nullptr;     // BYTES (also str)
    int64_t len = 0;
    int node = -1;                  // LIST / DICT: index into the container tables
};

struct Unpickler {
    const uint8_t *b, *e, *b0;
    std::vector<Val> stack, memo;
    std::vector<std::vector<Val>> lists;                       // LIST nodes (tuples too)
    std::vector<std::vector<std::pair<Val, Val>>> dicts;       // DICT nodes
    std::vector<std::string> arena;                            // bytes rebuilt from protocol-2 str pickles
    bool ok = true;

    Unpickler(const uint8_t *data, size_t n) : b(data), e(data + n), b0(data) {}
    bool need(size_t n) { if ((size_t)(e - b) < n) { ok = false; return false; } return true; }
    uint64_t le(int n) { uint64_t v = 0; for (int k = 0; k < n; ++k) v |= (uint64_t)b[k] << (8 * k); b += n; return v; }
    Val bytes_of(size_t n) { Val v; v.kind = Val::BYTES; v.p = b; v.len = (int64_t)n; b += n; return v; }
    // memo slots are numbered by the pickler in the order it meets objects, so a valid index never exceeds the number of
    // bytes read so far; anything larger is a corrupt (or hostile) file, not a reason to allocate gigabytes
    void put(size_t idx, const Val &v)
    {
        if (idx > (size_t)(b - b0)) { ok = false; return; }
        if (memo.size() <= idx) memo.resize(idx + 1);
        memo[idx] = v;
    }
    int mark_pos()
    {
        for (int k = (int)stack.size() - 1; k >= 0; --k)
            if (stack[k].kind == Val::MARK) return k;
        ok = false;
        return -1;
    }
    Val new_list() { Val v; v.kind = Val::LIST; v.node = (int)lists.size(); lists.emplace_back(); return v; }
    Val new_dict() { Val v; v.kind = Val::DICT; v.node = (int)dicts.size(); dicts.emplace_back(); return v; }

    // runs until STOP; the result is the top of the stack
    bool run(Val &result)
    {
        while (ok && b < e) {
            const uint8_t op = *b++;
            switch (op) {
            case 0x80: if (need(1)) b += 1; break;                                            // PROTO
            case 0x95: if (need(8)) b += 8; break;                                            // FRAME
            case ']': stack.push_back(new_list()); break;                                     // EMPTY_LIST
            case ')': stack.push_back(new_list()); break;                                     // EMPTY_TUPLE
            case '}': stack.push_back(new_dict()); break;                                     // EMPTY_DICT
            case '(': { Val m; m.kind = Val::MARK; stack.push_back(m); break; }               // MARK
            case 'N': stack.emplace_back(); break;                                            // NONE
            case 0x88: case 0x89: { Val v; v.kind = Val::INT; v.i = op == 0x88; stack.push_back(v); break; }
            case 'K': if (need(1)) { Val v; v.kind = Val::INT; v.i = (long long)le(1); stack.push_back(v); } break;
            case 'M': if (need(2)) { Val v; v.kind = Val::INT; v.i = (long long)le(2); stack.push_back(v); } break;
            case 'J': if (need(4)) { Val v; v.kind = Val::INT; v.i = (int32_t)le(4); stack.push_back(v); } break;
            case 0x8a: {                                                                      // LONG1
                if (!need(1)) break;
                const int n = *b++;
                if (n > 8 || !need((size_t)n)) { ok = false; break; }
                Val v; v.kind = Val::INT;
                uint64_t u = 0;
                for (int k = 0; k < n; ++k) u |= (uint64_t)b[k] << (8 * k);
                if (n && n < 8 && (b[n - 1] & 0x80)) u |= ~0ull << (8 * n);
                b += n;
                v.i = (long long)u;
                stack.push_back(v);
                break;
            }
            case 'G': if (need(8)) { b += 8; Val v; v.kind = Val::OTHER; stack.push_back(v); } break;      // BINFLOAT
            case 'C': case 'U': case 0x8c:                                                    // SHORT_BINBYTES / SHORT_BINSTRING / SHORT_BINUNICODE
                if (need(1)) { const size_t n = *b++; if (need(n)) stack.push_back(bytes_of(n)); }
                break;
            case 'B': case 'T': case 'X':                                                     // BINBYTES / BINSTRING / BINUNICODE
                if (need(4)) { const size_t n = (size_t)le(4); if (need(n)) stack.push_back(bytes_of(n)); }
                break;
            case 0x8e: case 0x8d:                                                             // BINBYTES8 / BINUNICODE8
                if (need(8)) { const size_t n = (size_t)le(8); if (need(n)) stack.push_back(bytes_of(n)); }
                break;
            case 0x94: if (stack.empty()) ok = false; else put(memo.size(), stack.back()); break;          // MEMOIZE
            case 'q': if (need(1) && !stack.empty()) put(*b++, stack.back()); else ok = false; break;      // BINPUT
            case 'r': if (need(4) && !stack.empty()) put((size_t)le(4), stack.back()); else ok = false; break;
            case 'h': if (need(1)) { const size_t k = *b++; if (k < memo.size()) stack.push_back(memo[k]); else ok = false; } break;
            case 'j': if (need(4)) { const size_t k = (size_t)le(4); if (k < memo.size()) stack.push_back(memo[k]); else ok = false; } break;
            case 'a': {                                                                       // APPEND
                if (stack.size() < 2 || stack[stack.size() - 2].kind != Val::LIST) { ok = false; break; }
                const Val v = stack.back();
                stack.pop_back();
                lists[stack.back().node].push_back(v);
                break;
            }
            case 'e': {                                                                       // APPENDS
                const int m = mark_pos();
                if (m < 1 || stack[m - 1].kind != Val::LIST) { ok = false; break; }
                auto &dst = lists[stack[m - 1].node];
                dst.insert(dst.end(), stack.begin() + m + 1, stack.end());
                stack.resize(m);
                break;
            }
            case 'l': case 't': {                                                             // LIST / TUPLE from mark
                const int m = mark_pos();
                if (m < 0) break;
                Val v = new_list();
                lists[v.node].assign(stack.begin() + m + 1, stack.end());
                stack.resize(m);
                stack.push_back(v);
                break;
            }
            case 0x85: case 0x86: case 0x87: {                                                // TUPLE1..3
                const size_t n = op - 0x84;
                if (stack.size() < n) { ok = false; break; }
                Val v = new_list();
                lists[v.node].assign(stack.end() - n, stack.end());
                stack.resize(stack.size() - n);
                stack.push_back(v);
                break;
            }
            case 's': {                                                                       // SETITEM
                if (stack.size() < 3 || stack[stack.size() - 3].kind != Val::DICT) { ok = false; break; }
                const Val val = stack.back(), key = stack[stack.size() - 2];
                stack.resize(stack.size() - 2);
                dicts[stack.back().node].emplace_back(key, val);
                break;
            }
            case 'u': {                                                                       // SETITEMS
                const int m = mark_pos();
                if (m < 1 || stack[m - 1].kind != Val::DICT || ((stack.size() - m - 1) & 1)) { ok = false; break; }
                auto &dst = dicts[stack[m - 1].node];
                for (size_t k = m + 1; k + 1 < stack.size(); k += 2) dst.emplace_back(stack[k], stack[k + 1]);
                stack.resize(m);
                break;
            }
            case 'c': {                                                                       // GLOBAL: "module\nname\n"
                const uint8_t *m0 = b;
                while (b < e && *b != '\n') ++b;
                const uint8_t *n0 = b < e ? b + 1 : b;
                if (b < e) ++b;
                while (b < e && *b != '\n') ++b;
                if (b >= e) { ok = false; break; }
                const std::string mod((const char *)m0, (size_t)(n0 - 1 - m0)), name((const char *)n0, (size_t)(b - n0));
                ++b;
                Val v;
                v.kind = (mod == "_codecs" && name == "encode") ? Val::ENCODE_FN : Val::OTHER;
                stack.push_back(v);
                break;
            }
            case 'R': {                                                                       // REDUCE: only _codecs.encode(str, 'latin1')
                // -- how protocols 0-2 spell a bytes object
                if (stack.size() < 2 || stack[stack.size() - 2].kind != Val::ENCODE_FN || stack.back().kind != Val::LIST) { ok = false; break; }
                const auto &args = lists[stack.back().node];
                if (args.size() != 2 || args[0].kind != Val::BYTES || args[1].kind != Val::BYTES || args[1].len != 6 ||
                    memcmp(args[1].p, "latin1", 6) != 0) { ok = false; break; }
                arena.emplace_back();
                std::string &out = arena.back();
                for (int64_t k = 0; k < args[0].len; ++k) {                     // UTF-8 -> latin-1
                    const uint8_t c = args[0].p[k];
                    if (c < 0x80) out.push_back((char)c);
                    else if ((c & 0xFC) == 0xC0 && k + 1 < args[0].len) out.push_back((char)(((c & 3) << 6) | (args[0].p[++k] & 0x3F)));
                    else { ok = false; break; }
                }
                stack.resize(stack.size() - 2);
                Val v; v.kind = Val::BYTES; v.p = (const uint8_t *)out.data(); v.len = (int64_t)out.size();
                stack.push_back(v);
                break;
            }
            case '.':                                                                         // STOP
                if (stack.empty()) return false;
                result = stack.back();
                return ok;
            default: ok = false; break;                        // an opcode a plain list of RLE dicts never contains
            }
        }
        return false;
    }
};

bool key_is(const Val &k, const char *name)
{
    const size_t n = strlen(name);
    return k.kind == Val::BYTES && (size_t)k.len == n && memcmp(k.p, name, n) == 0;
}

// one mask file -> run lengths of its masks (appended to cnts), one offset per mask, one (W, H) per mask
int parse_mask_file(const std::vector<uint8_t> &buf, std::vector<uint32_t> &cnts, std::vector<int64_t> &off, std::vector<int32_t> &wh)
{
    Unpickler up(buf.data(), buf.size());
    Val top;
    if (!up.run(top) || top.kind != Val::LIST) return CM3D_RD_ERR_FORMAT;
    for (const Val &item : up.lists[top.node]) {
        if (item.kind != Val::DICT) return CM3D_RD_ERR_FORMAT;
        const Val *size = nullptr, *counts = nullptr;
        for (const auto &kv : up.dicts[item.node]) {
            if (key_is(kv.first, "size")) size = &kv.second;
            else if (key_is(kv.first, "counts")) counts = &kv.second;
        }
        if (!size || !counts || size->kind != Val::LIST || counts->kind != Val::BYTES) return CM3D_RD_ERR_FORMAT;
        const auto &sz = up.lists[size->node];
        if (sz.size() != 2 || sz[0].kind != Val::INT || sz[1].kind != Val::INT || sz[0].i <= 0 || sz[1].i <= 0) return CM3D_RD_ERR_FORMAT;
        const int64_t n = rle_from_string(counts->p, counts->len, nullptr, 0);
        if (n < 0) return (int)n;
        const size_t at = cnts.size();
        cnts.resize(at + (size_t)n);
        if (rle_from_string(counts->p, counts->len, cnts.data() + at, n) != n) return CM3D_RD_ERR_FORMAT;
        unsigned long long total = 0;
        for (int64_t k = 0; k < n; ++k) total += cnts[at + (size_t)k];
        if (total != (unsigned long long)sz[0].i * (unsigned long long)sz[1].i) return CM3D_RD_ERR_FORMAT;      // must cover the mask
        off.push_back((int64_t)cnts.size());
        wh.push_back((int32_t)sz[0].i);
        wh.push_back((int32_t)sz[1].i);
    }
    return CM3D_RD_OK;
}

}   // namespace

struct cm3d_reader {
    Pool pool;
    int n_threads;
    explicit cm3d_reader(int nt) : pool(nt > 1 ? nt : 0), n_threads(nt > 1 ? nt : 1) {}
};

extern "C" cm3d_reader *cm3d_reader_open(int32_t n_threads)
{
    if (n_threads <= 0) {
        long n = sysconf(_SC_NPROCESSORS_ONLN);
        n_threads = (int32_t)(n < 1 ? 1 : (n > 64 ? 64 : n));
    }
    if (n_threads > 256) n_threads = 256;
    try {
        return new cm3d_reader(n_threads);
    } catch (...) {
        return nullptr;
    }
}

extern "C" void cm3d_reader_close(cm3d_reader *r) { delete r; }
extern "C" int32_t cm3d_reader_threads(const cm3d_reader *r) { return r ? r->n_threads : 0; }

extern "C" int64_t cm3d_rle_string_to_counts(const uint8_t *s, int64_t len, uint32_t *counts_out, int64_t cap)
{
    if ((!s && len) || len < 0) return CM3D_RD_ERR_ARG;
    return rle_from_string(s, len, counts_out, cap);
}

extern "C" int cm3d_reader_load_sweeps(cm3d_reader *r, const char *const *paths, int32_t n_files, int32_t stride, float *raw_out,
                                       int64_t cap_rows, int32_t *sweep_row_off, int32_t *bad_index)
{
    if (!r || !paths || n_files < 0 || stride <= 0 || !sweep_row_off || (!raw_out && cap_rows > 0)) return CM3D_RD_ERR_ARG;
    if (bad_index) *bad_index = -1;
    // sizes first (stat, serial: a few microseconds per file), then every file opened, read straight to its place and closed
    // inside its own task: a batch may name thousands of sweeps, far more than a process may hold open at once
    try {
        int64_t rows = 0;
        sweep_row_off[0] = 0;
        const int64_t row_bytes = (int64_t)stride * 4;
        for (int i = 0; i < n_files; ++i) {
            struct stat st;
            if (!paths[i] || stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode)) { if (bad_index) *bad_index = i; return CM3D_RD_ERR_IO; }
            if (st.st_size % row_bytes) { if (bad_index) *bad_index = i; return CM3D_RD_ERR_FORMAT; }
            rows += st.st_size / row_bytes;
            if (rows > 0x7FFFFFFF) { if (bad_index) *bad_index = i; return CM3D_RD_ERR_CAPACITY; }
            sweep_row_off[i + 1] = (int32_t)rows;
        }
        if (rows > cap_rows) return CM3D_RD_ERR_CAPACITY;
        std::atomic<int> bad{-1};
        r->pool.run(n_files, [&](int i) {
            const int64_t a = sweep_row_off[i], n = sweep_row_off[i + 1] - a;
            if (n == 0) return;
            const int fd = open(paths[i], O_RDONLY);
            struct stat st;
            // (a file that changed size between the two looks is an I/O error, not a buffer overrun)
            if (fd < 0 || fstat(fd, &st) != 0 || st.st_size != n * row_bytes || !read_fully(fd, raw_out + a * stride, (size_t)(n * row_bytes))) bad = i;
            if (fd >= 0) close(fd);
        });
        if (bad >= 0) { if (bad_index) *bad_index = bad; return CM3D_RD_ERR_IO; }
        return CM3D_RD_OK;
    } catch (...) {
        return CM3D_RD_ERR_IO;
    }
}

// The same files into the QUAD layout of include/cm3d_hip.h (cm3d_sweep_prep): x, y, z of batch rows 4q..4q+3 side by side (12 floats per
// quad), every frame padded to a multiple of 4 rows with NaN rows that belong to its last sweep.  Of a row's 20 bytes in the file 12 reach
// the batch (16 with the intensity plane); the ring index never leaves the page cache.  A task reads its file in pieces of PIECE rows into a
// buffer of its own (L2-sized) and scatters them: 3 (4) stores per row -- a sweep may start anywhere inside a quad, so the rows of one file
// and the next can share a quad; they never share a float.
extern "C" int cm3d_reader_load_sweeps_quads(cm3d_reader *r, const char *const *paths, int32_t n_files, int32_t file_stride,
                                             const int32_t *frame_sweep_off, int32_t n_frames, float *quads_out, float *intensity_out,
                                             int64_t cap_rows, int32_t *sweep_row_off, int32_t *frame_rows, int32_t *bad_index)
{
    if (!r || !paths || n_files < 0 || file_stride < 4 || !frame_sweep_off || n_frames < 0 || !sweep_row_off || !frame_rows ||
        (!quads_out && cap_rows > 0) || ((uintptr_t)quads_out & 15))
        return CM3D_RD_ERR_ARG;
    if (bad_index) *bad_index = -1;
    try {
        if (frame_sweep_off[0] != 0 || frame_sweep_off[n_frames] != n_files) return CM3D_RD_ERR_ARG;
        const int64_t row_bytes = (int64_t)file_stride * 4;
        std::vector<int64_t> file_rows((size_t)n_files);
        for (int i = 0; i < n_files; ++i) {
            struct stat st;
            if (!paths[i] || stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode)) { if (bad_index) *bad_index = i; return CM3D_RD_ERR_IO; }
            if (st.st_size % row_bytes) { if (bad_index) *bad_index = i; return CM3D_RD_ERR_FORMAT; }
            file_rows[i] = st.st_size / row_bytes;
        }
        // padded numbering: sweeps of a frame back to back, the next frame on the next multiple of 4
        int64_t rows = 0;
        for (int f = 0; f < n_frames; ++f) {
            if (frame_sweep_off[f + 1] < frame_sweep_off[f]) return CM3D_RD_ERR_ARG;
            int64_t fr = 0;
            for (int i = frame_sweep_off[f]; i < frame_sweep_off[f + 1]; ++i) {
                sweep_row_off[i] = (int32_t)(rows + fr);
                fr += file_rows[i];
                if (rows + fr > 0x7FFFFFF0) { if (bad_index) *bad_index = i; return CM3D_RD_ERR_CAPACITY; }
            }
            frame_rows[f] = (int32_t)fr;
            rows += (fr + 3) & ~(int64_t)3;
        }
        sweep_row_off[n_files] = (int32_t)rows;
        // (a frame's padding belongs to its last sweep: sweep_row_off of the next frame's first sweep is the padded start, set above)
        if (rows > cap_rows) return CM3D_RD_ERR_CAPACITY;
        const float qnan = std::numeric_limits<float>::quiet_NaN();
        for (int f = 0; f < n_frames; ++f) {                            // the padding rows (at most 3 per frame)
            const int i1 = frame_sweep_off[f + 1];
            if (i1 == frame_sweep_off[f]) continue;
            const int64_t end = (int64_t)sweep_row_off[frame_sweep_off[f]] + frame_rows[f], pend = (end + 3) & ~(int64_t)3;
            for (int64_t q = end; q < pend; ++q) {
                float *d = quads_out + (q >> 2) * 12 + (q & 3);
                d[0] = qnan; d[4] = qnan; d[8] = qnan;
                if (intensity_out) intensity_out[q] = 0.f;
            }
        }
        std::atomic<int> bad{-1};
        constexpr int64_t PIECE = 4096;                                 // rows per read: 80 KB of a 5-column file
        r->pool.run(n_files, [&](int i) {
            const int64_t a = sweep_row_off[i], n = file_rows[i];
            if (n == 0) return;
            const int fd = open(paths[i], O_RDONLY);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0 || st.st_size != n * row_bytes) { bad = i; if (fd >= 0) close(fd); return; }
            auto scatter = [&](const float *src, int64_t r0, int64_t m) {
                int64_t q = a + r0;
                for (int64_t k = 0; k < m; ++k, ++q, src += file_stride) {
                    float *d = quads_out + (q >> 2) * 12 + (q & 3);
                    d[0] = src[0]; d[4] = src[1]; d[8] = src[2];
                    if (intensity_out) intensity_out[q] = src[3];
                }
            };
            // CM3D_READER_MMAP=1: the file's pages mapped (one pass, page cache -> batch, no copy into a buffer of ours first).  Off by
            // default: measured on the GPU box (16 cores, 128 reader threads) the map / unmap of every file costs more than the copy it
            // saves -- sweeps of a 256-frame batch 11.6 ms mapped against 6.9 ms read in pieces (address-space lock, TLB shootdowns)
            static const bool use_mmap = getenv("CM3D_READER_MMAP") && atoi(getenv("CM3D_READER_MMAP")) != 0;
            void *mp = use_mmap ? mmap(nullptr, (size_t)(n * row_bytes), PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0) : MAP_FAILED;
            if (mp != MAP_FAILED) {
                scatter((const float *)mp, 0, n);
                munmap(mp, (size_t)(n * row_bytes));
                close(fd);
                return;
            }
            std::vector<float> buf((size_t)(PIECE * file_stride));
            for (int64_t r0 = 0; r0 < n; r0 += PIECE) {
                const int64_t m = std::min(PIECE, n - r0);
                size_t off = 0;
                const size_t want = (size_t)(m * row_bytes);
                while (off < want) {
                    const ssize_t got = pread(fd, (char *)buf.data() + off, want - off, (off_t)(r0 * row_bytes + (int64_t)off));
                    if (got <= 0) { bad = i; close(fd); return; }
                    off += (size_t)got;
                }
                scatter(buf.data(), r0, m);
            }
            close(fd);
        });
        if (bad >= 0) { if (bad_index) *bad_index = bad; return CM3D_RD_ERR_IO; }
        return CM3D_RD_OK;
    } catch (...) {
        return CM3D_RD_ERR_IO;
    }
}

extern "C" int cm3d_reader_load_masks(cm3d_reader *r, const char *const *paths, int32_t n_files, uint32_t *counts_out, int64_t cap_counts,
                                      int32_t *rle_off, int32_t *frame_mask_off, int32_t *mask_wh, int32_t cap_masks, int64_t *needed,
                                      int32_t *bad_index)
{
    if (!r || !paths || n_files < 0 || !frame_mask_off || !needed) return CM3D_RD_ERR_ARG;
    if (bad_index) *bad_index = -1;
    try {
        struct PerFile { std::vector<uint32_t> cnts; std::vector<int64_t> off; std::vector<int32_t> wh; int rc = CM3D_RD_OK; };
        std::vector<PerFile> pf((size_t)n_files);
        r->pool.run(n_files, [&](int i) {
            if (!paths[i] || !paths[i][0]) return;                 // a frame without detections has no file
            try {                                                  // (no exception may leave a pool thread or the C ABI)
                std::vector<uint8_t> buf;
                if (!read_file(paths[i], buf)) { pf[i].rc = CM3D_RD_ERR_IO; return; }
                pf[i].rc = parse_mask_file(buf, pf[i].cnts, pf[i].off, pf[i].wh);
            } catch (...) {
                pf[i].rc = CM3D_RD_ERR_FORMAT;                     // out of memory on a file-controlled size: treat the file as malformed
            }
        });
        int64_t total_counts = 0, total_masks = 0;
        for (int i = 0; i < n_files; ++i) {
            if (pf[i].rc != CM3D_RD_OK) { if (bad_index) *bad_index = i; return pf[i].rc; }
            total_counts += (int64_t)pf[i].cnts.size();
            total_masks += (int64_t)pf[i].off.size();
        }
        needed[0] = total_counts;
        needed[1] = total_masks;
        if (total_counts > cap_counts || total_masks > cap_masks || total_counts > 0x7FFFFFFF || !counts_out || !rle_off || !mask_wh)
            return CM3D_RD_ERR_CAPACITY;
        // exclusive offsets, then every file's share copied in parallel
        std::vector<int64_t> c0((size_t)n_files + 1, 0), m0((size_t)n_files + 1, 0);
        for (int i = 0; i < n_files; ++i) { c0[i + 1] = c0[i] + (int64_t)pf[i].cnts.size(); m0[i + 1] = m0[i] + (int64_t)pf[i].off.size(); }
        for (int i = 0; i <= n_files; ++i) frame_mask_off[i] = (int32_t)m0[i];
        rle_off[0] = 0;
        r->pool.run(n_files, [&](int i) {
            if (!pf[i].cnts.empty()) memcpy(counts_out + c0[i], pf[i].cnts.data(), pf[i].cnts.size() * sizeof(uint32_t));
            for (size_t k = 0; k < pf[i].off.size(); ++k) {
                rle_off[m0[i] + (int64_t)k + 1] = (int32_t)(c0[i] + pf[i].off[k]);
                mask_wh[2 * (m0[i] + (int64_t)k)] = pf[i].wh[2 * k];
                mask_wh[2 * (m0[i] + (int64_t)k) + 1] = pf[i].wh[2 * k + 1];
            }
        });
        return CM3D_RD_OK;
    } catch (...) {
        return CM3D_RD_ERR_FORMAT;
    }
}

// =====================================================================================================================
// nuScenes tables + frame manifests + result writer: the rest of the reference's host-side frame loop in native code.
//   reference (src/nuscenes/2d_to_3d.py): NuScenes(VER_NAME, INPUT_PATH) :382 (the tables), the per-frame table walk :415-441 and
//   :489-503 (sweep chain with its calibrated_sensor / ego_pose rows, the six cameras' records), json.load of <f>_data.json :423,
//   and json.dump of the result dict :929-930.  cm3d_amd/nusc_io.py + lifting.nuscenes_results_json are the Python forms of
//   the same steps; tests/test_reader.py holds the two against each other array for array and byte for byte.
// All arithmetic on the records is IEEE double in the operation order of cm3d_amd/geometry.py (this file is compiled
// with -ffp-contract=off), so the float32 records are bit-identical to the Python path's.
#include <charconv>
#include <cmath>
#include <memory>
#include <string_view>
#include <unordered_map>

namespace {

// ---------------------------------------------------------------------------------------------- a small JSON reader
struct JP {
    const char *p, *e;
    bool ok = true;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    std::string str()                                           // a JSON string, escapes decoded (UTF-8 out)
    {
        std::string out;
        ws();
        if (p >= e || *p != '"') { ok = false; return out; }
        ++p;
        while (p < e && *p != '"') {
            char c = *p++;
            if (c != '\\') { out.push_back(c); continue; }
            if (p >= e) { ok = false; return out; }
            c = *p++;
            switch (c) {
            case 'n': out.push_back('\n'); break; case 't': out.push_back('\t'); break; case 'r': out.push_back('\r'); break;
            case 'b': out.push_back('\b'); break; case 'f': out.push_back('\f'); break;
            case 'u': {
                if (e - p < 4) { ok = false; return out; }
                unsigned v = 0;
                for (int k = 0; k < 4; ++k) {
                    const char h = p[k];
                    const int d = (h >= '0' && h <= '9') ? h - '0' : ((h | 32) >= 'a' && (h | 32) <= 'f') ? (h | 32) - 'a' + 10 : -1;
                    if (d < 0) { ok = false; return out; }
                    v = v * 16 + (unsigned)d;
                }
                p += 4;
                if (v < 0x80) out.push_back((char)v);
                else if (v < 0x800) { out.push_back((char)(0xC0 | (v >> 6))); out.push_back((char)(0x80 | (v & 0x3F))); }
                else { out.push_back((char)(0xE0 | (v >> 12))); out.push_back((char)(0x80 | ((v >> 6) & 0x3F))); out.push_back((char)(0x80 | (v & 0x3F))); }
                break;
            }
            default: out.push_back(c); break;                   // \" \\ \/
            }
        }
        if (p >= e) { ok = false; return out; }
        ++p;
        return out;
    }
    double num()
    {
        ws();
        // strtod needs a terminated buffer: the table files are read with a trailing NUL (read_text)
        char *end = nullptr;
        const double v = strtod(p, &end);                      // correctly rounded, like Python's float(): the same double
        if (end == p || end > e) { ok = false; return 0.0; }
        p = end;
        return v;
    }
    bool boolean()
    {
        ws();
        if (e - p >= 4 && !memcmp(p, "true", 4)) { p += 4; return true; }
        if (e - p >= 5 && !memcmp(p, "false", 5)) { p += 5; return false; }
        ok = false;
        return false;
    }
    void skip()
    {
        ws();
        if (p >= e) { ok = false; return; }
        if (*p == '"') { (void)str(); return; }
        if (*p == '{' || *p == '[') {
            int depth = 0;
            while (p < e) {
                if (*p == '"') { (void)str(); if (!ok) return; continue; }
                if (*p == '{' || *p == '[') ++depth;
                else if ((*p == '}' || *p == ']') && --depth == 0) { ++p; return; }
                ++p;
            }
            ok = false;
            return;
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;   // number / true / false / null
    }
    int numbers(double *out, int cap)                         // a flat array of numbers; returns the count (-1: malformed / too many)
    {
        if (!eat('[')) { ok = false; return -1; }
        int n = 0;
        if (eat(']')) return 0;
        for (;;) {
            if (n >= cap) { ok = false; return -1; }
            out[n++] = num();
            if (!ok) return -1;
            if (eat(',')) continue;
            if (eat(']')) return n;
            ok = false;
            return -1;
        }
    }
};

bool read_text(const char *path, std::vector<uint8_t> &buf)
{
    if (!read_file(path, buf)) return false;
    buf.push_back(0);                                          // strtod's sentinel
    return true;
}

// walks `[ {..}, {..}, ... ]`: begin() per object, field(j, key) per key (must consume the value)
template <typename Begin, typename Field>
bool walk_table(const std::vector<uint8_t> &buf, Begin begin, Field field)
{
    JP j{(const char *)buf.data(), (const char *)buf.data() + buf.size() - 1};
    if (!j.eat('[')) return false;
    if (j.eat(']')) return true;
    for (;;) {
        if (!j.eat('{')) return false;
        begin();
        if (!j.eat('}')) {
            for (;;) {
                const std::string key = j.str();
                if (!j.ok || !j.eat(':')) return false;
                field(j, key);
                if (!j.ok) return false;
                if (j.eat(',')) continue;
                if (j.eat('}')) break;
                return false;
            }
        }
        if (j.eat(',')) continue;
        return j.eat(']');
    }
}

struct PoseRow { std::string token; double t[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}; };
struct CsRow { std::string token, sensor_tok; double t[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; bool has_k = false; int channel = -1; };
struct SdRow { std::string token, sample_tok, pose_tok, cs_tok, next_tok, filename; bool key = false; int sample = -1, pose = -1, cs = -1, next = -1; };
struct SampleRow { std::string token, next_tok, scene_tok; int next = -1; int data[7] = {-1, -1, -1, -1, -1, -1, -1}; };   // [0] LIDAR_TOP, [1..6] CAM_LIST
struct SceneRow { std::string token, name, log_tok, first_tok; int first = -1, log = -1, nbr = 0; };
struct NameRow { std::string token, value; };

const char *const kChannels[7] = {"LIDAR_TOP", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_RIGHT", "CAM_BACK", "CAM_BACK_LEFT", "CAM_FRONT_LEFT"};

// unit quaternion (w, x, y, z) -> row-major 3x3: the arithmetic of geometry.quat_to_rotmat, operation for operation
void quat_to_rotmat(const double *qin, double *R)
{
    double w = qin[0], x = qin[1], y = qin[2], z = qin[3];
    const double nrm = std::sqrt(w * w + x * x + y * y + z * z);
    w = w / nrm; x = x / nrm; y = y / nrm; z = z / nrm;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

}   // namespace

struct cm3d_tables {
    std::string dataroot, version;
    std::vector<SceneRow> scenes;
    std::vector<SampleRow> samples;
    std::vector<SdRow> sds;
    std::vector<PoseRow> poses;
    std::vector<CsRow> css;
    std::vector<NameRow> sensors, logs;
    std::unordered_map<std::string, int> scene_by_name;
};

extern "C" cm3d_tables *cm3d_tables_open(cm3d_reader *r, const char *dataroot, const char *version, int32_t *err)
{
    auto fail = [&](int code) { if (err) *err = code; return (cm3d_tables *)nullptr; };
    if (err) *err = CM3D_RD_OK;
    if (!r || !dataroot || !version) return fail(CM3D_RD_ERR_ARG);
    try {
        std::unique_ptr<cm3d_tables> t(new cm3d_tables);
        t->dataroot = dataroot; t->version = version;
        const std::string base = t->dataroot + "/" + t->version + "/";
        int rc[7] = {0, 0, 0, 0, 0, 0, 0};
        // the seven tables are parsed side by side (sample_data is by far the longest), token references resolved afterwards
        r->pool.run(7, [&](int which) {
            try {
                std::vector<uint8_t> buf;
                static const char *const names[7] = {"scene", "sample", "sample_data", "ego_pose", "calibrated_sensor", "sensor", "log"};
                if (!read_text((base + names[which] + ".json").c_str(), buf)) { rc[which] = CM3D_RD_ERR_IO; return; }
                bool ok = false;
                switch (which) {
                case 0: ok = walk_table(buf, [&] { t->scenes.emplace_back(); }, [&](JP &j, const std::string &k) {
                        SceneRow &s = t->scenes.back();
                        if (k == "token") s.token = j.str(); else if (k == "name") s.name = j.str(); else if (k == "log_token") s.log_tok = j.str();
                        else if (k == "first_sample_token") s.first_tok = j.str(); else if (k == "nbr_samples") s.nbr = (int)j.num(); else j.skip();
                    }); break;
                case 1: ok = walk_table(buf, [&] { t->samples.emplace_back(); }, [&](JP &j, const std::string &k) {
                        SampleRow &s = t->samples.back();
                        if (k == "token") s.token = j.str(); else if (k == "next") s.next_tok = j.str(); else if (k == "scene_token") s.scene_tok = j.str(); else j.skip();
                    }); break;
                case 2: ok = walk_table(buf, [&] { t->sds.emplace_back(); }, [&](JP &j, const std::string &k) {
                        SdRow &s = t->sds.back();
                        if (k == "token") s.token = j.str(); else if (k == "sample_token") s.sample_tok = j.str(); else if (k == "ego_pose_token") s.pose_tok = j.str();
                        else if (k == "calibrated_sensor_token") s.cs_tok = j.str(); else if (k == "filename") s.filename = j.str();
                        else if (k == "next") s.next_tok = j.str(); else if (k == "is_key_frame") s.key = j.boolean(); else j.skip();
                    }); break;
                case 3: ok = walk_table(buf, [&] { t->poses.emplace_back(); }, [&](JP &j, const std::string &k) {
                        PoseRow &s = t->poses.back();
                        if (k == "token") s.token = j.str();
                        else if (k == "translation") { if (j.numbers(s.t, 3) != 3) j.ok = false; }
                        else if (k == "rotation") { if (j.numbers(s.q, 4) != 4) j.ok = false; }
                        else j.skip();
                    }); break;
                case 4: ok = walk_table(buf, [&] { t->css.emplace_back(); }, [&](JP &j, const std::string &k) {
                        CsRow &s = t->css.back();
                        if (k == "token") s.token = j.str(); else if (k == "sensor_token") s.sensor_tok = j.str();
                        else if (k == "translation") { if (j.numbers(s.t, 3) != 3) j.ok = false; }
                        else if (k == "rotation") { if (j.numbers(s.q, 4) != 4) j.ok = false; }
                        else if (k == "camera_intrinsic") {                      // [] for a lidar, three rows of three for a camera
                            if (!j.eat('[')) { j.ok = false; return; }
                            if (j.eat(']')) return;
                            for (int row = 0; row < 3; ++row) {
                                if (j.numbers(s.K + 3 * row, 3) != 3) { j.ok = false; return; }
                                if (row < 2 && !j.eat(',')) { j.ok = false; return; }
                            }
                            if (!j.eat(']')) { j.ok = false; return; }
                            s.has_k = true;
                        }
                        else j.skip();
                    }); break;
                case 5: ok = walk_table(buf, [&] { t->sensors.emplace_back(); }, [&](JP &j, const std::string &k) {
                        if (k == "token") t->sensors.back().token = j.str(); else if (k == "channel") t->sensors.back().value = j.str(); else j.skip();
                    }); break;
                default: ok = walk_table(buf, [&] { t->logs.emplace_back(); }, [&](JP &j, const std::string &k) {
                        if (k == "token") t->logs.back().token = j.str(); else if (k == "location") t->logs.back().value = j.str(); else j.skip();
                    }); break;
                }
                if (!ok) rc[which] = CM3D_RD_ERR_FORMAT;
            } catch (...) {
                rc[which] = CM3D_RD_ERR_FORMAT;
            }
        });
        for (int k = 0; k < 7; ++k) if (rc[k] != CM3D_RD_OK) return fail(rc[k]);
        // token -> row
        std::unordered_map<std::string, int> sample_ix, sd_ix, pose_ix, cs_ix, sensor_ix, log_ix;
        auto index = [](auto &rows, std::unordered_map<std::string, int> &ix) { ix.reserve(rows.size() * 2); for (size_t i = 0; i < rows.size(); ++i) ix.emplace(rows[i].token, (int)i); };
        index(t->samples, sample_ix); index(t->sds, sd_ix); index(t->poses, pose_ix); index(t->css, cs_ix); index(t->sensors, sensor_ix); index(t->logs, log_ix);
        auto find = [](const std::unordered_map<std::string, int> &ix, const std::string &tok) { auto it = ix.find(tok); return it == ix.end() ? -1 : it->second; };
        for (auto &c : t->css) {
            const int s = find(sensor_ix, c.sensor_tok);
            if (s < 0) return fail(CM3D_RD_ERR_FORMAT);
            for (int ch = 0; ch < 7; ++ch) if (t->sensors[s].value == kChannels[ch]) c.channel = ch;
        }
        for (auto &s : t->samples) s.next = s.next_tok.empty() ? -1 : find(sample_ix, s.next_tok);
        for (size_t i = 0; i < t->sds.size(); ++i) {
            SdRow &d = t->sds[i];
            d.sample = find(sample_ix, d.sample_tok); d.pose = find(pose_ix, d.pose_tok); d.cs = find(cs_ix, d.cs_tok);
            d.next = d.next_tok.empty() ? -1 : find(sd_ix, d.next_tok);
            if (d.sample < 0 || d.pose < 0 || d.cs < 0) return fail(CM3D_RD_ERR_FORMAT);
            // sample['data'][channel] like the devkit builds it: key frames only (a later row of the same channel wins, like a dict)
            if (d.key && t->css[d.cs].channel >= 0) t->samples[d.sample].data[t->css[d.cs].channel] = (int)i;
        }
        for (size_t i = 0; i < t->scenes.size(); ++i) {
            SceneRow &s = t->scenes[i];
            s.first = find(sample_ix, s.first_tok); s.log = find(log_ix, s.log_tok);
            if (s.first < 0 || s.log < 0) return fail(CM3D_RD_ERR_FORMAT);
            t->scene_by_name.emplace(s.name, (int)i);
        }
        return t.release();
    } catch (...) {
        return fail(CM3D_RD_ERR_FORMAT);
    }
}

extern "C" void cm3d_tables_close(cm3d_tables *t) { delete t; }

extern "C" int32_t cm3d_tables_scene_samples(const cm3d_tables *t, const char *scene_name)
{
    if (!t || !scene_name) return CM3D_RD_ERR_ARG;
    auto it = t->scene_by_name.find(scene_name);
    if (it == t->scene_by_name.end()) return CM3D_RD_ERR_ARG;
    int n = 0;
    for (int s = t->scenes[it->second].first; s >= 0 && n <= (int)t->samples.size(); s = t->samples[s].next) ++n;   // (bounded: a cycle of `next` ends here)
    return n;
}

extern "C" int32_t cm3d_tables_scene_location(const cm3d_tables *t, const char *scene_name, char *out, int32_t cap)
{
    if (!t || !scene_name || !out || cap <= 0) return CM3D_RD_ERR_ARG;
    auto it = t->scene_by_name.find(scene_name);
    if (it == t->scene_by_name.end()) return CM3D_RD_ERR_ARG;
    const std::string &loc = t->logs[t->scenes[it->second].log].value;
    if ((int32_t)loc.size() + 1 > cap) return CM3D_RD_ERR_CAPACITY;
    memcpy(out, loc.c_str(), loc.size() + 1);
    return (int32_t)loc.size();
}

// all scene names, NUL-separated, in table order; returns the bytes needed
extern "C" int64_t cm3d_tables_scene_names(const cm3d_tables *t, char *out, int64_t cap)
{
    if (!t) return CM3D_RD_ERR_ARG;
    int64_t need = 0;
    for (const auto &s : t->scenes) need += (int64_t)s.name.size() + 1;
    if (out && cap >= need) {
        char *p = out;
        for (const auto &s : t->scenes) { memcpy(p, s.name.c_str(), s.name.size() + 1); p += s.name.size() + 1; }
    }
    return need;
}

// sample tokens of the given scenes in job order (scenes in the given order, samples in scene order), NUL-separated, and
// their rows in sample.json; returns the bytes the tokens need (negative: unknown scene)
extern "C" int64_t cm3d_tables_job_tokens(const cm3d_tables *t, const char *const *scene_names, int32_t n_scenes, char *out, int64_t cap,
                                          int32_t *rows_out, int64_t cap_rows)
{
    if (!t || !scene_names || n_scenes < 0) return CM3D_RD_ERR_ARG;
    int64_t need = 0, k = 0;
    char *p = out;
    for (int i = 0; i < n_scenes; ++i) {
        auto it = t->scene_by_name.find(scene_names[i] ? scene_names[i] : "");
        if (it == t->scene_by_name.end()) return CM3D_RD_ERR_ARG;
        int64_t walked = 0;                                      // a cycle of `next` in sample.json ends the walk with an error, not the process
        for (int s = t->scenes[it->second].first; s >= 0; s = t->samples[s].next) {
            if (++walked > (int64_t)t->samples.size()) return CM3D_RD_ERR_FORMAT;
            const std::string &tok = t->samples[s].token;
            need += (int64_t)tok.size() + 1;
            if (out && need <= cap) { memcpy(p, tok.c_str(), tok.size() + 1); p += tok.size() + 1; }
            if (rows_out && k < cap_rows) rows_out[k] = s;
            ++k;
        }
    }
    return need;
}

namespace {

// <f>_data.json: {"labels": [str], "detection_scores": [float], "cam_nums": [int]} (gen_2d_masks_detic.py:497-504)
struct FrameData { std::vector<std::string> labels; std::vector<double> scores; std::vector<int> cams; };
bool parse_data_json(const std::vector<uint8_t> &buf, FrameData &d)
{
    JP j{(const char *)buf.data(), (const char *)buf.data() + buf.size() - 1};
    if (!j.eat('{')) return false;
    if (j.eat('}')) return true;
    for (;;) {
        const std::string key = j.str();
        if (!j.ok || !j.eat(':')) return false;
        if (key == "labels" || key == "detection_scores" || key == "cam_nums") {
            if (!j.eat('[')) return false;
            if (!j.eat(']')) {
                for (;;) {
                    if (key == "labels") d.labels.push_back(j.str());
                    else if (key == "detection_scores") d.scores.push_back(j.num());
                    else d.cams.push_back((int)j.num());
                    if (!j.ok) return false;
                    if (j.eat(',')) continue;
                    if (j.eat(']')) break;
                    return false;
                }
            }
        } else j.skip();
        if (!j.ok) return false;
        if (j.eat(',')) continue;
        return j.eat('}');
    }
}

bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

}   // namespace

struct cm3d_manifest {
    int n_frames = 0;
    std::vector<int32_t> sample_index;               // row of every frame in sample.json (its token is the caller's to look up)
    std::vector<int32_t> frame_sweep_off, frame_mask_off;
    std::vector<float> sweep_xf, cams;
    std::vector<double> ego_xyz, score;
    std::vector<int32_t> mask_cam, class_id;
    std::string sweep_paths, mask_paths;             // NUL-separated
    std::vector<int32_t> sweep_path_off, mask_path_off;
    int bad_frame = -1;
    std::string bad_label;
};

// The table walk of a batch of scenes (reference :415-441, :489-503 per frame; nusc_io.scene_manifest).  class_names = the
// detection names in class-index order; labels are renamed like get_detection_name (:122-132) before the lookup.
extern "C" cm3d_manifest *cm3d_tables_manifest(const cm3d_tables *t, cm3d_reader *r, const char *const *scene_names, int32_t n_scenes,
                                               const char *mask_dir, int32_t n_sweeps, double ratio, const char *const *class_names,
                                               int32_t n_classes, int32_t missing_ok, int32_t *err)
{
    auto fail = [&](int code) { if (err) *err = code; return (cm3d_manifest *)nullptr; };
    if (err) *err = CM3D_RD_OK;
    if (!t || !r || !scene_names || n_scenes <= 0 || !mask_dir || n_sweeps <= 0 || !class_names || n_classes <= 0) return fail(CM3D_RD_ERR_ARG);
    try {
        std::unique_ptr<cm3d_manifest> m(new cm3d_manifest);
        // frames of the batch: (scene name, frame number inside the scene, sample row)
        struct Fr { const std::string *scene; int num, sample; };
        std::vector<Fr> frames;
        for (int k = 0; k < n_scenes; ++k) {
            auto it = t->scene_by_name.find(scene_names[k] ? scene_names[k] : "");
            if (it == t->scene_by_name.end()) return fail(CM3D_RD_ERR_ARG);
            int num = 0;
            for (int s = t->scenes[it->second].first; s >= 0 && num <= (int)t->samples.size(); s = t->samples[s].next)
                frames.push_back({&t->scenes[it->second].name, num++, s});
        }
        const int F = (int)frames.size();
        m->n_frames = F;
        struct PerFrame { std::vector<float> xf, cams; std::vector<std::string> sweeps; std::string mask; FrameData d; std::vector<int32_t> cls; double ego[3]; int rc = CM3D_RD_OK; std::string bad; };
        std::vector<PerFrame> pf((size_t)F);
        const float ratio32 = (float)ratio;
        const std::string mdir(mask_dir);
        r->pool.run(F, [&](int i) {
            PerFrame &o = pf[i];
            try {
                const SampleRow &smp = t->samples[frames[i].sample];
                // mask file + data file (a frame without detections has neither, gen_2d_masks_detic.py:490-491)
                const std::string stem = mdir + "/" + *frames[i].scene + "/" + std::to_string(frames[i].num);
                const std::string mp = stem + "_masks.pkl", dp = stem + "_data.json";
                if (file_exists(mp) && file_exists(dp)) {
                    o.mask = mp;
                    std::vector<uint8_t> buf;
                    if (!read_text(dp.c_str(), buf)) { o.rc = CM3D_RD_ERR_IO; return; }
                    if (!parse_data_json(buf, o.d)) { o.rc = CM3D_RD_ERR_FORMAT; return; }
                    if (o.d.labels.size() != o.d.scores.size() || o.d.labels.size() != o.d.cams.size()) { o.rc = CM3D_RD_ERR_FORMAT; return; }
                    for (const std::string &lab : o.d.labels) {
                        const char *name = lab == "trafficcone" ? "traffic_cone" : lab == "constructionvehicle" ? "construction_vehicle" : lab == "human" ? "pedestrian" : lab.c_str();
                        int ci = -1;
                        for (int c = 0; c < n_classes; ++c) if (!strcmp(class_names[c], name)) { ci = c; break; }
                        if (ci < 0) { o.rc = CM3D_RD_ERR_FORMAT; o.bad = lab; return; }
                        o.cls.push_back(ci);
                    }
                } else if (!missing_ok) { o.rc = CM3D_RD_ERR_IO; o.bad = mp; return; }
                // sweeps: the key frame's LIDAR_TOP sample_data and its `next` chain (:433-463)
                int sd = smp.data[0];
                if (sd < 0) { o.rc = CM3D_RD_ERR_FORMAT; return; }
                const PoseRow &key_pose = t->poses[t->sds[sd].pose];
                o.ego[0] = key_pose.t[0]; o.ego[1] = key_pose.t[1]; o.ego[2] = key_pose.t[2];
                for (int k = 0; k < n_sweeps && sd >= 0; ++k) {
                    const SdRow &d = t->sds[sd];
                    const CsRow &cs = t->css[d.cs];
                    const PoseRow &pose = t->poses[d.pose];
                    o.sweeps.push_back(t->dataroot + "/" + d.filename);
                    double R[9];
                    float xf[24];
                    quat_to_rotmat(cs.q, R);
                    for (int q = 0; q < 9; ++q) xf[q] = (float)R[q];
                    for (int q = 0; q < 3; ++q) xf[9 + q] = (float)cs.t[q];
                    quat_to_rotmat(pose.q, R);
                    for (int q = 0; q < 9; ++q) xf[12 + q] = (float)R[q];
                    for (int q = 0; q < 3; ++q) xf[21 + q] = (float)pose.t[q];
                    o.xf.insert(o.xf.end(), xf, xf + 24);
                    sd = d.next;
                }
                // the six cameras' records (:489-503 + :569-587): p += f32(-t_ego); p = f32(R_ego^T) p; p += f32(-t_cs); p = f32(R_cs^T) p; K' = f32(K) * f32(ratio)
                o.cams.assign(6 * 64, 0.0f);
                for (int c = 0; c < 6; ++c) {
                    const int csd = smp.data[1 + c];
                    if (csd < 0) { o.rc = CM3D_RD_ERR_FORMAT; return; }
                    const SdRow &d = t->sds[csd];
                    const CsRow &cs = t->css[d.cs];
                    const PoseRow &pose = t->poses[d.pose];
                    if (!cs.has_k) { o.rc = CM3D_RD_ERR_FORMAT; return; }
                    float *rec = o.cams.data() + 64 * c;
                    double R[9];
                    quat_to_rotmat(pose.q, R);
                    for (int q = 0; q < 3; ++q) rec[q] = (float)(-pose.t[q]);
                    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) rec[3 + 3 * a + b] = (float)R[3 * b + a];          // transposed
                    quat_to_rotmat(cs.q, R);
                    for (int q = 0; q < 3; ++q) rec[15 + q] = (float)(-cs.t[q]);
                    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) rec[18 + 3 * a + b] = (float)R[3 * b + a];
                    for (int q = 0; q < 9; ++q) rec[45 + q] = (float)cs.K[q] * ratio32;        // float32 product (geometry.scaled_intrinsic_f32)
                    rec[45 + 8] = 1.0f;
                    rec[54] = 2.0f;
                    rec[55] = 5.0f;                                                             // both stages translate first
                }
            } catch (...) {
                o.rc = CM3D_RD_ERR_FORMAT;
            }
        });
        m->frame_sweep_off.push_back(0);
        m->frame_mask_off.push_back(0);
        m->sweep_path_off.push_back(0);
        m->mask_path_off.push_back(0);
        for (int i = 0; i < F; ++i) {
            PerFrame &o = pf[i];
            if (o.rc != CM3D_RD_OK) {
                m->bad_frame = i; m->bad_label = o.bad;
                if (err) *err = o.rc;
                return m.release();                        // the caller reads bad_frame / bad_label, then closes it
            }
            m->sample_index.push_back(frames[i].sample);
            m->sweep_xf.insert(m->sweep_xf.end(), o.xf.begin(), o.xf.end());
            m->cams.insert(m->cams.end(), o.cams.begin(), o.cams.end());
            m->ego_xyz.insert(m->ego_xyz.end(), o.ego, o.ego + 3);
            for (const std::string &p : o.sweeps) { m->sweep_paths += p; m->sweep_paths.push_back('\0'); m->sweep_path_off.push_back((int32_t)m->sweep_paths.size()); }
            m->frame_sweep_off.push_back(m->frame_sweep_off.back() + (int32_t)o.sweeps.size());
            m->mask_paths += o.mask; m->mask_paths.push_back('\0'); m->mask_path_off.push_back((int32_t)m->mask_paths.size());
            m->mask_cam.insert(m->mask_cam.end(), o.d.cams.begin(), o.d.cams.end());
            m->class_id.insert(m->class_id.end(), o.cls.begin(), o.cls.end());
            m->score.insert(m->score.end(), o.d.scores.begin(), o.d.scores.end());
            m->frame_mask_off.push_back(m->frame_mask_off.back() + (int32_t)o.cls.size());
        }
        return m.release();
    } catch (...) {
        return fail(CM3D_RD_ERR_FORMAT);
    }
}

extern "C" void cm3d_manifest_close(cm3d_manifest *m) { delete m; }

// sizes: [0] frames, [1] sweeps, [2] masks (per the data files), [3] bytes of the sweep-path blob, [4] of the mask-path blob,
// [5] first frame that could not be read (-1: none)
extern "C" void cm3d_manifest_sizes(const cm3d_manifest *m, int64_t *out)
{
    if (!m || !out) return;
    out[0] = m->n_frames; out[1] = (int64_t)m->sweep_xf.size() / 24; out[2] = (int64_t)m->class_id.size();
    out[3] = (int64_t)m->sweep_paths.size(); out[4] = (int64_t)m->mask_paths.size(); out[5] = m->bad_frame;
}

extern "C" const char *cm3d_manifest_bad_label(const cm3d_manifest *m) { return m ? m->bad_label.c_str() : ""; }

// copies the manifest's arrays into caller buffers sized by cm3d_manifest_sizes (any pointer may be NULL to skip it)
extern "C" int cm3d_manifest_copy(const cm3d_manifest *m, int32_t *sample_index, int32_t *frame_sweep_off, float *sweep_xf, float *cams,
                                  double *ego_xyz, int32_t *frame_mask_off, int32_t *mask_cam, int32_t *class_id, double *score,
                                  char *sweep_paths, int32_t *sweep_path_off, char *mask_paths, int32_t *mask_path_off)
{
    if (!m || m->bad_frame >= 0) return CM3D_RD_ERR_ARG;
    auto cp = [](auto *dst, const auto &src) { if (dst && !src.empty()) memcpy(dst, src.data(), src.size() * sizeof(src[0])); };
    cp(sample_index, m->sample_index); cp(frame_sweep_off, m->frame_sweep_off); cp(sweep_xf, m->sweep_xf); cp(cams, m->cams);
    cp(ego_xyz, m->ego_xyz); cp(frame_mask_off, m->frame_mask_off); cp(mask_cam, m->mask_cam); cp(class_id, m->class_id); cp(score, m->score);
    cp(sweep_paths, m->sweep_paths); cp(sweep_path_off, m->sweep_path_off); cp(mask_paths, m->mask_paths); cp(mask_path_off, m->mask_path_off);
    return CM3D_RD_OK;
}

// the sweeps / mask files of a manifest through the batch loaders above (paths straight from the manifest: no path list
// crosses the language boundary)
extern "C" int cm3d_manifest_load_sweeps(cm3d_reader *r, const cm3d_manifest *m, int32_t stride, float *raw_out, int64_t cap_rows,
                                         int32_t *sweep_row_off, int32_t *bad_index)
{
    if (!r || !m || m->bad_frame >= 0) return CM3D_RD_ERR_ARG;
    try {
        const int n = (int)m->sweep_path_off.size() - 1;
        std::vector<const char *> paths((size_t)n);
        for (int i = 0; i < n; ++i) paths[i] = m->sweep_paths.data() + m->sweep_path_off[i];
        return cm3d_reader_load_sweeps(r, paths.data(), n, stride, raw_out, cap_rows, sweep_row_off, bad_index);
    } catch (...) {
        return CM3D_RD_ERR_IO;
    }
}

extern "C" int cm3d_manifest_load_sweeps_quads(cm3d_reader *r, const cm3d_manifest *m, int32_t file_stride, float *quads_out, float *intensity_out,
                                               int64_t cap_rows, int32_t *sweep_row_off, int32_t *frame_rows, int32_t *bad_index)
{
    if (!r || !m || m->bad_frame >= 0) return CM3D_RD_ERR_ARG;
    try {
        const int n = (int)m->sweep_path_off.size() - 1;
        std::vector<const char *> paths((size_t)n);
        for (int i = 0; i < n; ++i) paths[i] = m->sweep_paths.data() + m->sweep_path_off[i];
        return cm3d_reader_load_sweeps_quads(r, paths.data(), n, file_stride, m->frame_sweep_off.data(), (int)m->frame_sweep_off.size() - 1, quads_out,
                                             intensity_out, cap_rows, sweep_row_off, frame_rows, bad_index);
    } catch (...) {
        return CM3D_RD_ERR_IO;
    }
}

extern "C" int cm3d_manifest_load_masks(cm3d_reader *r, const cm3d_manifest *m, uint32_t *counts_out, int64_t cap_counts, int32_t *rle_off,
                                        int32_t *frame_mask_off, int32_t *mask_wh, int32_t cap_masks, int64_t *needed, int32_t *bad_index)
{
    if (!r || !m || m->bad_frame >= 0) return CM3D_RD_ERR_ARG;
    try {
        const int n = (int)m->mask_path_off.size() - 1;
        std::vector<const char *> paths((size_t)n);
        for (int i = 0; i < n; ++i) paths[i] = m->mask_paths.data() + m->mask_path_off[i];
        return cm3d_reader_load_masks(r, paths.data(), n, counts_out, cap_counts, rle_off, frame_mask_off, mask_wh, cap_masks, needed, bad_index);
    } catch (...) {
        return CM3D_RD_ERR_FORMAT;
    }
}

// ---------------------------------------------------------------------------------------------- result writer
namespace {

// repr(float) of Python 3 (float_repr_style 'short'): the shortest digit string that round-trips, fixed notation for
// 1e-4 <= |x| < 1e16 (with ".0" when there is no fraction), else d.ddde[+-]XX with at least two exponent digits.
// json.dumps spells the non-finite values Infinity / -Infinity / NaN.
void py_float_repr(double x, std::string &out)
{
    if (std::isnan(x)) { out += "NaN"; return; }
    if (std::isinf(x)) { out += x < 0 ? "-Infinity" : "Infinity"; return; }
    if (x == 0.0) { out += std::signbit(x) ? "-0.0" : "0.0"; return; }
    char buf[40];
    const auto res = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);      // shortest round-trip digits
    std::string_view sv(buf, (size_t)(res.ptr - buf));
    const bool neg = sv[0] == '-';
    if (neg) sv.remove_prefix(1);
    const size_t epos = sv.find('e');
    std::string digits;
    for (char c : sv.substr(0, epos)) if (c != '.') digits.push_back(c);
    int exp10 = 0;
    std::from_chars(sv.data() + epos + 1 + (sv[epos + 1] == '+' ? 1 : 0), sv.data() + sv.size(), exp10);
    const int decpt = exp10 + 1;                                // position of the decimal point relative to the digit string
    if (neg) out.push_back('-');
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) { out += "0."; out.append((size_t)(-decpt), '0'); out += digits; }
        else if ((size_t)decpt >= digits.size()) { out += digits; out.append((size_t)decpt - digits.size(), '0'); out += ".0"; }
        else { out.append(digits, 0, (size_t)decpt); out.push_back('.'); out.append(digits, (size_t)decpt, std::string::npos); }
    } else {
        out.push_back(digits[0]);
        if (digits.size() > 1) { out.push_back('.'); out.append(digits, 1, std::string::npos); }
        out.push_back('e');
        const int ex = decpt - 1;
        out.push_back(ex < 0 ? '-' : '+');
        const int a = ex < 0 ? -ex : ex;
        if (a < 10) out.push_back('0');
        out += std::to_string(a);
    }
}

}   // namespace

// The text json.dump writes for the reference's result dict (:808-817, :929-930), straight from the gathered box records:
//   records  double[n][10]: 0-2 translation, 3 qw, 4 qz, 5 index of the sample in `tokens`, 7 score, 8 class
//   tokens   n_tokens JSON-quoted sample tokens ("..." incl. the quotes), NUL-separated, in output order
//   cls_head / cls_tail  per class: the text between the translation and the rotation (', "size": [...], "rotation": [')
//            and the text behind the score (velocity, name, attribute, closing brace) -- rendered once by the caller
//   prefix   everything up to and including '"results": {'
// Returns the number of bytes written to `out` (needs cap), or the needed size as a negative number when cap is too small.
extern "C" int64_t cm3d_write_results_json(const double *records, int64_t n, const char *tokens, int32_t n_tokens, const char *const *cls_mid,
                                           const char *const *cls_score, const char *const *cls_tail, int32_t n_classes, const char *prefix,
                                           char *out, int64_t cap)
{
    if ((!records && n) || !tokens || n_tokens < 0 || !cls_mid || !cls_score || !cls_tail) return 0;
    try {
        std::vector<const char *> tok((size_t)n_tokens);
        const char *p = tokens;
        for (int i = 0; i < n_tokens; ++i) { tok[i] = p; p += strlen(p) + 1; }
        // stable bucket by sample
        std::vector<int64_t> first((size_t)n_tokens + 1, 0), order((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            // (range first, cast after: a NaN or an infinite column must not reach an integer conversion)
            const double td = records[10 * i + 5], cd = records[10 * i + 8];
            if (!(td >= 0.0 && td < (double)n_tokens) || !(cd >= 0.0 && cd < (double)n_classes)) return 0;
            const int64_t ti = (int64_t)td;
            ++first[ti + 1];
        }
        for (int i = 0; i < n_tokens; ++i) first[i + 1] += first[i];
        std::vector<int64_t> cur(first.begin(), first.end() - 1);
        for (int64_t i = 0; i < n; ++i) order[cur[(int64_t)records[10 * i + 5]]++] = i;
        std::string s;
        s.reserve((size_t)(n * 330 + n_tokens * 48 + 256));
        if (prefix) s += prefix;
        for (int ti = 0; ti < n_tokens; ++ti) {
            if (ti) s += ", ";
            s += tok[ti]; s += ": [";
            for (int64_t q = first[ti]; q < first[ti + 1]; ++q) {
                const double *r = records + 10 * order[q];
                const int ci = (int)r[8];
                if (q > first[ti]) s += ", ";
                s += "{\"sample_token\": "; s += tok[ti]; s += ", \"translation\": [";
                py_float_repr(r[0], s); s += ", "; py_float_repr(r[1], s); s += ", "; py_float_repr(r[2], s);
                s += cls_mid[ci];
                py_float_repr(r[3], s); s += ", 0.0, 0.0, "; py_float_repr(r[4], s);
                s += cls_score[ci];
                py_float_repr(r[7], s);
                s += cls_tail[ci];
            }
            s += "]";
        }
        if (prefix) s += "}}";                                  // (prefix == NULL: only the "token: [...]" entries, for a caller that streams)
        if ((int64_t)s.size() > cap || !out) return -(int64_t)s.size();
        memcpy(out, s.data(), s.size());
        return (int64_t)s.size();
    } catch (...) {
        return 0;
    }
}
