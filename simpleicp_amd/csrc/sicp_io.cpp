// sicp_io.cpp -- host-side .xyz text I/O: the reference's callers read
// clouds with np.genfromtxt (tests/test_simpleicp.py:102-103) and write them with pandas/np.savetxt
// (pointcloud.py:219-226, corrpts.py:213-237); at 10M+ points that text handling dwarfs the ICP run.
// Multithreaded, mmap-based, correctly rounded (strtod / printf, so values and bytes are identical to
// the reference's Python I/O).  No GPU involved.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/simpleicp_hip.h"

#define SICP_EXPORT extern "C" __attribute__((visibility("default")))

extern int sicp_io_fail(int code, const char *fmt, ...);   // sets the library's last-error text (sicp_api.cpp)

namespace {

struct Mapped {
    const char *p = nullptr; size_t n = 0; int fd = -1;
    ~Mapped() { if (p && n) munmap((void *)p, n); if (fd >= 0) close(fd); }
};

int map_file(const char *path, Mapped &m)
{
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) return sicp_io_fail(SICP_ERR_INVALID, "cannot open %s", path);
    struct stat st;
    if (fstat(m.fd, &st) != 0) return sicp_io_fail(SICP_ERR_INVALID, "cannot stat %s", path);
    m.n = (size_t)st.st_size;
    if (m.n == 0) return SICP_OK;
    void *p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (p == MAP_FAILED) { m.n = 0; return sicp_io_fail(SICP_ERR_INVALID, "cannot mmap %s", path); }
    m.p = (const char *)p;
    return SICP_OK;
}

inline bool starts_number(const char *s, const char *e)
{
    while (s < e && (*s == ' ' || *s == '\t' || *s == '\r')) ++s;
    if (s >= e) return false;
    char c = *s;
    if ((c >= '0' && c <= '9') || c == '.') return true;
    if (c == '-' || c == '+') { if (++s >= e) return false; c = *s; if ((c >= '0' && c <= '9') || c == '.') return true; }
    // "nan" / "inf" rows are data for np.genfromtxt (and for strtod), not headers
    if (e - s >= 3) {
        const char a = (char)(s[0] | 0x20), b = (char)(s[1] | 0x20), d = (char)(s[2] | 0x20);
        if ((a == 'n' && b == 'a' && d == 'n') || (a == 'i' && b == 'n' && d == 'f')) return true;
    }
    return false;
}

// [b, e) snapped to whole lines of the mapping
void chunk(const Mapped &m, int t, int T, size_t &b, size_t &e)
{
    b = m.n * (size_t)t / T; e = m.n * (size_t)(t + 1) / T;
    if (t > 0) { while (b < m.n && m.p[b - 1] != '\n') ++b; }
    if (t < T - 1) { while (e < m.n && m.p[e - 1] != '\n') ++e; } else e = m.n;
    if (b > e) b = e;
}

int64_t count_rows(const Mapped &m, size_t b, size_t e)
{
    int64_t rows = 0;
    const char *s = m.p + b, *end = m.p + e;
    while (s < end) {
        const char *nl = (const char *)memchr(s, '\n', (size_t)(end - s));
        const char *le = nl ? nl : end;
        if (starts_number(s, le)) ++rows;
        s = le + 1;
    }
    return rows;
}

// parses the data rows of [b, e) into out (3 doubles per row); returns rows parsed or -1 on a short row
int64_t parse_rows(const Mapped &m, size_t b, size_t e, double *out)
{
    int64_t rows = 0;
    const char *s = m.p + b, *end = m.p + e;
    char buf[512];
    std::string longline;
    while (s < end) {
        const char *nl = (const char *)memchr(s, '\n', (size_t)(end - s));
        const char *le = nl ? nl : end;
        if (starts_number(s, le)) {
            // strtod needs a terminator: lines are short, copy (also protects the unterminated last line)
            const size_t len = (size_t)(le - s);
            char *p = buf;
            if (len < sizeof buf) { memcpy(buf, s, len); buf[len] = 0; }
            else { longline.assign(s, len); p = &longline[0]; }      // e.g. "%.3f" of 1e300: hundreds of digits
            for (int c = 0; c < 3; ++c) {
                char *q;
                const double v = strtod(p, &q);
                if (q == p) return -1;
                out[3 * rows + c] = v; p = q;
            }
            ++rows;
        }
        s = le + 1;
    }
    return rows;
}

int clamp_threads(int t, size_t bytes)
{
    if (t <= 0) t = (int)std::thread::hardware_concurrency();
    if (t <= 0) t = 1;
    const size_t by_size = bytes / (1u << 20) + 1;   // at least ~1 MiB per thread
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)t, std::min<size_t>(by_size, 256)));
}

}  // namespace

// Number of data rows (lines starting with a number; blank lines and `//...` / `#...` headers skipped).
SICP_EXPORT int sicp_xyz_count(const char *path, int64_t *rows_out)
{
    if (!path || !rows_out) return sicp_io_fail(SICP_ERR_INVALID, "null argument");
    Mapped m;
    int rc = map_file(path, m);
    if (rc != SICP_OK) return rc;
    const int T = clamp_threads(0, m.n);
    std::vector<int64_t> cnt(T, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t b, e; chunk(m, t, T, b, e); cnt[t] = count_rows(m, b, e); });
    for (auto &x : th) x.join();
    int64_t n = 0; for (auto c : cnt) n += c;
    *rows_out = n;
    return SICP_OK;
}

// Reads the first three columns of every data row into xyz_out (row-major, capacity_rows rows).
SICP_EXPORT int sicp_xyz_read(const char *path, double *xyz_out, int64_t capacity_rows, int64_t *rows_out, int threads)
{
    if (!path || !xyz_out || !rows_out) return sicp_io_fail(SICP_ERR_INVALID, "null argument");
    Mapped m;
    int rc = map_file(path, m);
    if (rc != SICP_OK) return rc;
    const int T = clamp_threads(threads, m.n);
    std::vector<int64_t> cnt(T, 0), off(T + 1, 0), got(T, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t b, e; chunk(m, t, T, b, e); cnt[t] = count_rows(m, b, e); });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; ++t) off[t + 1] = off[t] + cnt[t];
    if (off[T] > capacity_rows) return sicp_io_fail(SICP_ERR_INVALID, "%s has %lld rows, buffer holds %lld", path, (long long)off[T], (long long)capacity_rows);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] { size_t b, e; chunk(m, t, T, b, e); got[t] = parse_rows(m, b, e, xyz_out + 3 * off[t]); });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; ++t)
        if (got[t] != cnt[t]) return sicp_io_fail(SICP_ERR_INVALID, "%s: a data row has fewer than 3 numeric columns", path);
    *rows_out = off[T];
    return SICP_OK;
}

// Writes n rows of `cols` columns as "%.<decimals>f" (decimals >= 0) or "%.18e" (decimals < 0, np.savetxt's
// default) separated by single spaces; `header` (may be NULL) is written verbatim as the first line.
SICP_EXPORT int sicp_xyz_write(const char *path, const double *data, int64_t n, int cols, int decimals, const char *header, int threads)
{
    if (!path || (!data && n > 0) || cols < 1 || cols > 16 || decimals > 64) return sicp_io_fail(SICP_ERR_INVALID, "bad arguments");
    FILE *f = fopen(path, "wb");
    if (!f) return sicp_io_fail(SICP_ERR_INVALID, "cannot open %s for writing", path);
    if (header) { fputs(header, f); fputc('\n', f); }
    char fmt[16];
    if (decimals >= 0) snprintf(fmt, sizeof fmt, "%%.%df", decimals); else snprintf(fmt, sizeof fmt, "%%.18e");
    const int T = clamp_threads(threads, (size_t)n * cols * 8);
    const int64_t block = 1 << 16;                       // rows per formatting task
    for (int64_t base = 0; base < n; base += block * T) {
        std::vector<std::string> bufs(T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                const int64_t b = base + block * t, e = std::min<int64_t>(n, b + block);
                if (b >= e) return;
                std::string &s = bufs[t];
                s.reserve((size_t)(e - b) * cols * 14);
                // "%.3f" of 1e308 has 309 integer digits: the stack buffer covers every double at the
                // decimals the callers use; anything longer goes through a heap buffer of the exact size
                char tmp[384];
                std::vector<char> big;
                for (int64_t i = b; i < e; ++i)
                    for (int c = 0; c < cols; ++c) {
                        const double v = data[i * cols + c];
                        const int len = snprintf(tmp, sizeof tmp, fmt, v);
                        if (len < 0) continue;
                        if ((size_t)len < sizeof tmp) s.append(tmp, (size_t)len);
                        else {
                            big.resize((size_t)len + 1);
                            snprintf(big.data(), big.size(), fmt, v);
                            s.append(big.data(), (size_t)len);
                        }
                        s.push_back(c + 1 < cols ? ' ' : '\n');
                    }
            });
        for (auto &x : th) x.join();
        for (auto &s : bufs) if (!s.empty() && fwrite(s.data(), 1, s.size(), f) != s.size()) { fclose(f); return sicp_io_fail(SICP_ERR_INVALID, "short write to %s", path); }
    }
    if (fclose(f) != 0) return sicp_io_fail(SICP_ERR_INVALID, "cannot close %s", path);
    return SICP_OK;
}
