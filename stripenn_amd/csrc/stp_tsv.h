// stp_tsv.h -- result tables as text (host code only: no context, no device).
// What it replaces: `df.to_csv(path, sep='\t', header=True, index=False)` of the reference's drivers (stripenn.py:156-157,
// score.py:60), whose float columns pandas writes with Python's repr(float): the shortest digits that round-trip (David Gay's
// mode 0, PyOS_double_to_string(x, 'r', 0, Py_DTSF_ADD_DOT_0)), in exponent form when the decimal point would sit before the
// fourth leading zero or behind the 16th digit (decpt <= -4 or decpt > 16: pystrtod.c, format_float_short), with ".0" after an
// integral value, NaN as the empty field.  std::to_chars (C++17, shortest round-trip, closest candidate on ties) supplies the
// digits; the layout is rebuilt here.  tests/test_tsv_native.py compares millions of values with repr().
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

static inline char* stp_tsv_i64(char* p, int64_t v)
{
    char b[24];
    int n = 0;
    uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = b[--n];
    return p;
}

// at most 25 characters ("-1.2345678901234567e-308")
static inline char* stp_tsv_repr(char* p, double v)
{
    if (v != v) return p;                                   // NaN: the empty field (pandas' na_rep)
    if (std::isinf(v)) { if (v < 0) *p++ = '-'; memcpy(p, "inf", 3); return p + 3; }
    if (v == 0.0) { if (std::signbit(v)) *p++ = '-'; memcpy(p, "0.0", 3); return p + 3; }
    char b[48];
    const std::to_chars_result r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX[X]
    const char* q = b;
    if (*q == '-') { *p++ = '-'; q++; }
    char dig[24];
    int nd = 0;
    dig[nd++] = *q++;
    if (*q == '.') { q++; while (*q != 'e') dig[nd++] = *q++; }
    q++;                                                    // 'e'
    const bool eneg = *q == '-';
    q++;
    int e = 0;
    while (q < r.ptr) e = e * 10 + (*q++ - '0');
    const int decpt = (eneg ? -e : e) + 1;                  // value = 0.d1 d2 ... x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *p++ = dig[0];
        if (nd > 1) { *p++ = '.'; memcpy(p, dig + 1, (size_t)nd - 1); p += nd - 1; }
        *p++ = 'e';
        int x = decpt - 1;
        *p++ = x < 0 ? '-' : '+';
        if (x < 0) x = -x;
        if (x >= 100) { *p++ = (char)('0' + x / 100); x %= 100; *p++ = (char)('0' + x / 10); *p++ = (char)('0' + x % 10); }
        else { *p++ = (char)('0' + x / 10); *p++ = (char)('0' + x % 10); }          // at least two exponent digits
    } else if (decpt <= 0) {
        *p++ = '0'; *p++ = '.';
        for (int k = 0; k < -decpt; k++) *p++ = '0';
        memcpy(p, dig, (size_t)nd); p += nd;
    } else if (decpt >= nd) {
        memcpy(p, dig, (size_t)nd); p += nd;
        for (int k = nd; k < decpt; k++) *p++ = '0';
        *p++ = '.'; *p++ = '0';
    } else {
        memcpy(p, dig, (size_t)decpt); p += decpt;
        *p++ = '.';
        memcpy(p, dig + decpt, (size_t)(nd - decpt)); p += nd - decpt;
    }
    return p;
}
