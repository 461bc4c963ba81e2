// Host-only: the text of the tracker's CSV (marker_detection.py:464-468, `DataFrame.to_csv(index=False)`), formatted by
// several threads.  86 k rows x 7 float columns per 512 frames: Python's per-cell repr() was 90 % of the drop-in's wall
// time.  Floats are written like Python's repr: shortest digits that round-trip (std::to_chars), fixed notation for
// decimal exponents -4 .. 16, otherwise d[.ddd]e+XX; NaN = empty cell, as pandas writes it.
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vbs.h"

static inline char* put_repr(double x, char* o) {
    if (x != x) return o;                                // NaN: empty cell
    if (std::isinf(x)) { const char* t = x > 0 ? "inf" : "-inf"; size_t l = strlen(t); memcpy(o, t, l); return o + l; }
    char t[40];
    const auto r = std::to_chars(t, t + sizeof t, x, std::chars_format::scientific);     // [-]d[.ddd]e[+-]XX, shortest
    const char* p = t;
    if (*p == '-') *o++ = *p++;
    char dig[24];
    int n = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') dig[n++] = *p;
    int e10 = 0;
    {
        ++p;                                             // 'e'
        const bool neg = *p == '-';
        ++p;
        for (; p < r.ptr; ++p) e10 = 10 * e10 + (*p - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;                           // value = 0.d1d2..dn x 10^decpt
    if (decpt <= -4 || decpt > 16) {                     // repr's exponent form
        *o++ = dig[0];
        if (n > 1) { *o++ = '.'; memcpy(o, dig + 1, n - 1); o += n - 1; }
        *o++ = 'e';
        int e = decpt - 1;
        *o++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int en = 0;
        do { eb[en++] = (char)('0' + e % 10); e /= 10; } while (e);
        if (en < 2) eb[en++] = '0';
        while (en) *o++ = eb[--en];
    } else if (decpt <= 0) {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -decpt; ++i) *o++ = '0';
        memcpy(o, dig, n); o += n;
    } else if (decpt >= n) {
        memcpy(o, dig, n); o += n;
        for (int i = 0; i < decpt - n; ++i) *o++ = '0';
        *o++ = '.'; *o++ = '0';
    } else {
        memcpy(o, dig, decpt); o += decpt;
        *o++ = '.';
        memcpy(o, dig + decpt, n - decpt); o += n - decpt;
    }
    return o;
}

extern "C" int64_t vbs_format_csv(const int64_t* frameno, const int64_t* row, const int64_t* col, const double* const* fcols,
                                  int nf, int64_t n, char* buf, int64_t cap, int threads) {
    if (!frameno || !row || !col || !fcols || nf < 0 || nf > 16 || n < 0 || !buf) return VBS_EINVAL;
    const int64_t per_row = 3 * 21 + (int64_t)nf * 26 + 2;       // upper bound of a row's text
    if (cap < n * per_row) return -(n * per_row);                // (negative: the capacity that is enough)
    if (threads < 1) threads = 1;
    const int64_t min_rows = 2048;
    if ((int64_t)threads > (n + min_rows - 1) / min_rows) threads = (int)((n + min_rows - 1) / min_rows);
    if (threads < 1) threads = 1;
    std::vector<int64_t> len(threads, 0);
    auto work = [&](int t) {
        const int64_t a = n * t / threads, b = n * (t + 1) / threads;
        char* o = buf + a * per_row;                     // every thread writes into its own slice, compacted afterwards
        char* const o0 = o;
        for (int64_t i = a; i < b; ++i) {
            o = std::to_chars(o, o + 21, frameno[i]).ptr; *o++ = ',';
            o = std::to_chars(o, o + 21, row[i]).ptr; *o++ = ',';
            o = std::to_chars(o, o + 21, col[i]).ptr;
            for (int c = 0; c < nf; ++c) { *o++ = ','; o = put_repr(fcols[c][i], o); }
            *o++ = '\n';
        }
        len[t] = o - o0;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    int64_t total = len[0];
    for (int t = 1; t < threads; ++t) {
        memmove(buf + total, buf + (n * t / threads) * per_row, (size_t)len[t]);
        total += len[t];
    }
    return total;
}
