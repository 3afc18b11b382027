// libcm3d_reader.so: host-side loader of the lifting path (include/cm3d_reader.h).
//   reference: src/nuscenes/2d_to_3d.py:422-428 (pickle.load of <f>_masks.pkl + pycocotools decode) and :437-441 with
//   utils/pcd.py:246-257 (np.fromfile of every sweep) -- per frame, single-threaded Python there; per batch, on a thread
//   pool, straight into the caller's page-locked staging buffers here.
// Plain C++17 + pthreads; no HIP, no Python.
#include "../../include/cm3d_reader.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <mutex>
#include <string>
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
