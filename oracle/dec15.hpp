// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker.
//
// dec15: decimal floating point with 15 significant digits and HALF_UP rounding after every single
// operation — the arithmetic the reference runs on: java.math.BigDecimal with
// MathContext(15, RoundingMode.HALF_UP) (reference LPState.java:18, used at :139,:144,:146,:157,:162,:164,
// :171,:172,:177,:297).  BigDecimal and MathContext live in the JDK (not vendored in the reference and
// no JVM exists in this image), so this file restates the published General Decimal Arithmetic rule
// they implement: "compute the exact result, then round once to `precision` significant digits".
//
// Representation: value = c * 10^e with c == 0 or 10^14 <= |c| < 10^15 (left-justified, so any value
// with <= 15 significant digits is exactly representable and compare() is a lexicographic test).
// BigDecimal's scale / trailing-zero bookkeeping is deliberately not modelled: every consumer in the
// reference uses compareTo()/arithmetic, which see values only.  HALF_UP means "round away from zero
// iff the first discarded digit is >= 5", so a truncated quotient needs no sticky bit.
//
// Pinned by tests/test_oracle_dec15.py against Python's stdlib decimal (prec=15, ROUND_HALF_UP), which
// implements the same specification, and through it by the reference's own Spock vectors.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace dec15 {

typedef unsigned __int128 u128;
typedef __int128 i128;

static const int64_t P14 = 100000000000000LL;   // 10^14
static const int64_t P15 = 1000000000000000LL;  // 10^15

struct Pow10Table {
  u128 tab[39];
  Pow10Table() {
    tab[0] = 1;
    for (int i = 1; i < 39; i++) tab[i] = tab[i - 1] * 10;
  }
};

inline u128 pow10_u128(int k) {
  static const Pow10Table t;  // C++11 magic static: thread-safe initialisation
  return t.tab[k];
}

inline int ndigits_u128(u128 x) {  // x > 0
  int lo = 1, hi = 38;             // number of digits d satisfies 10^(d-1) <= x < 10^d
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (x >= pow10_u128(mid)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

struct Dec {
  int64_t c;  // signed coefficient, 0 or 10^14 <= |c| < 10^15
  int32_t e;  // decimal exponent

  Dec() : c(0), e(0) {}
  Dec(int64_t cc, int32_t ee) : c(cc), e(ee) {}

  bool is_zero() const { return c == 0; }
  int sign() const { return (c > 0) - (c < 0); }
};

// Round the exact magnitude mag * 10^exp to 15 significant digits, HALF_UP, and left-justify.
inline Dec make_rounded(bool neg, u128 mag, int32_t exp) {
  if (mag == 0) return Dec(0, 0);
  int d = ndigits_u128(mag);
  if (d > 15) {
    int drop = d - 15;
    u128 p = pow10_u128(drop);
    u128 q = mag / p;
    u128 r = mag - q * p;
    if (r >= p / 2) q += 1;  // first discarded digit >= 5  (p/2 = 5*10^(drop-1) exactly)
    exp += drop;
    if (q == (u128)P15) { q = (u128)P14; exp += 1; }
    mag = q;
  } else if (d < 15) {
    mag *= pow10_u128(15 - d);
    exp -= (15 - d);
  }
  int64_t c = (int64_t)mag;
  return Dec(neg ? -c : c, exp);
}

inline Dec from_int(int64_t x) {
  bool neg = x < 0;
  u128 mag = neg ? (u128)(-(i128)x) : (u128)x;
  return make_rounded(neg, mag, 0);
}

// Parse a decimal literal ("-12.5", "1e-9", "0.000123").  Values with more than 15 significant digits
// are rounded HALF_UP at load (a BigDecimal would keep them exact until the first operation; the
// oracle's inputs are restricted to <= 15 digits).
inline Dec from_string(const char* s) {
  while (*s == ' ') s++;
  bool neg = false;
  if (*s == '+') s++; else if (*s == '-') { neg = true; s++; }
  u128 mag = 0;
  int32_t exp = 0;
  int digits = 0;  // significant digits accumulated so far (leading zeros do not count)
  bool seen_point = false, any = false;
  for (; *s; s++) {
    if (*s == '.') { if (seen_point) throw std::invalid_argument("bad decimal"); seen_point = true; continue; }
    if (*s < '0' || *s > '9') break;
    any = true;
    int dg = *s - '0';
    if (digits < 36) {
      mag = mag * 10 + dg;
      if (mag != 0) digits++;
      if (seen_point) exp--;
    } else if (!seen_point) {
      exp++;  // digits beyond the 36th are truncated: they cannot reach the 16th digit
    }
  }
  if (!any) throw std::invalid_argument("bad decimal");
  if (*s == 'e' || *s == 'E') {
    s++;
    exp += (int32_t)strtol(s, nullptr, 10);
  }
  return make_rounded(neg, mag, exp);
}

// Exact path: shortest round-trip decimal of the double (%.17g), rounded HALF_UP to 15 significant digits.
inline Dec from_double_exact(double x) {
  if (x == 0.0) return Dec(0, 0);
  char buf[64];
  snprintf(buf, sizeof buf, "%.17g", x);
  return from_string(buf);
}

// Fast path used for bulk loads: scale by a power of ten in long double and round to the nearest
// 15-digit integer.  For a double that was written from a decimal literal with <= 15 significant digits
// (every fixture and every parity input) the scaled value is within 1e-18 relative of that integer, so
// this returns exactly the literal, like the exact path; for arbitrary doubles it may differ from it in
// the 15th digit on near-ties (irrelevant: such inputs only occur in timing runs).
inline Dec from_double(double x) {
  if (x == 0.0) return Dec(0, 0);
  if (!(x == x) || x > 1.7e308 || x < -1.7e308) throw std::invalid_argument("non-finite input");
  const bool ng = x < 0;
  const long double ax = ng ? -(long double)x : (long double)x;
  int e10 = (int)floorl(log10l(ax));
  for (int attempt = 0; attempt < 3; attempt++) {
    const int sh = 14 - e10;
    const long double scaled = sh >= 0 ? ax * powl(10.0L, sh) : ax / powl(10.0L, -sh);
    const long long c = llroundl(scaled);
    if (c >= P15) { e10++; continue; }
    if (c < P14) { e10--; continue; }
    return Dec(ng ? -c : c, e10 - 14);
  }
  return from_double_exact(x);
}

inline double to_double(const Dec& a) {
  if (a.c == 0) return 0.0;
  char buf[64];
  snprintf(buf, sizeof buf, "%llde%d", (long long)a.c, (int)a.e);
  return strtod(buf, nullptr);  // correctly rounded decimal -> binary
}

// Canonical text: coefficient with trailing zeros stripped + exponent ("125e-1"); "0" for zero.
inline std::string to_string(const Dec& a) {
  if (a.c == 0) return "0";
  int64_t c = a.c;
  int32_t e = a.e;
  while (c % 10 == 0) { c /= 10; e++; }
  char buf[64];
  snprintf(buf, sizeof buf, "%llde%d", (long long)c, (int)e);
  return buf;
}

inline Dec neg(const Dec& a) { return Dec(-a.c, a.e); }   // BigDecimal.negate(): exact, no rounding
inline Dec abs(const Dec& a) { return Dec(a.c < 0 ? -a.c : a.c, a.e); }

inline int cmp(const Dec& a, const Dec& b) {  // BigDecimal.compareTo
  int sa = a.sign(), sb = b.sign();
  if (sa != sb) return sa < sb ? -1 : 1;
  if (sa == 0) return 0;
  int64_t ma = a.c < 0 ? -a.c : a.c, mb = b.c < 0 ? -b.c : b.c;
  int mag;  // compare magnitudes: both coefficients are left-justified, so exponent first
  if (a.e != b.e) mag = a.e < b.e ? -1 : 1;
  else mag = (ma < mb) ? -1 : (ma > mb ? 1 : 0);
  return sa > 0 ? mag : -mag;
}

inline Dec mul(const Dec& a, const Dec& b) {  // a.multiply(b, mc)
  if (a.c == 0 || b.c == 0) return Dec(0, 0);
  bool ng = (a.c < 0) != (b.c < 0);
  u128 ma = (u128)(a.c < 0 ? -a.c : a.c), mb = (u128)(b.c < 0 ? -b.c : b.c);
  return make_rounded(ng, ma * mb, a.e + b.e);
}

inline Dec div(const Dec& a, const Dec& b) {  // a.divide(b, mc)
  if (b.c == 0) throw std::domain_error("Division by zero");  // ArithmeticException in the JDK
  if (a.c == 0) return Dec(0, 0);
  bool ng = (a.c < 0) != (b.c < 0);
  u128 ma = (u128)(a.c < 0 ? -a.c : a.c), mb = (u128)(b.c < 0 ? -b.c : b.c);
  // ma*10^17/mb lies in (10^16, 10^18): 17..18 digits, so the 16th digit (first discarded) is exact.
  u128 q = (ma * pow10_u128(17)) / mb;
  return make_rounded(ng, q, a.e - b.e - 17);
}

inline Dec add(const Dec& a, const Dec& b) {  // a.add(b, mc)
  if (a.c == 0) return b;
  if (b.c == 0) return a;
  const Dec* hi = &a;
  const Dec* lo = &b;
  if (lo->e > hi->e) { const Dec* t = hi; hi = lo; lo = t; }
  int32_t d = hi->e - lo->e;
  // Both coefficients are left-justified 15-digit numbers.  For d >= 17 the low operand is below 1/100
  // of the high operand's last place (and below 1/10 of the last place of the decade underneath it), so
  // the correctly rounded sum or difference is the high operand itself.
  if (d >= 17) return *hi;
  i128 s = (i128)hi->c * (i128)pow10_u128(d) + (i128)lo->c;
  bool ng = s < 0;
  u128 mag = ng ? (u128)(-s) : (u128)s;
  return make_rounded(ng, mag, lo->e);
}

inline Dec sub(const Dec& a, const Dec& b) { return add(a, neg(b)); }  // a.subtract(b, mc)

// BigDecimal.setScale(6, HALF_UP) (reference LPSolver.java:113): returns the value rounded to 6
// decimal places as text with exactly 6 fractional digits.
inline std::string set_scale6(const Dec& a) {
  // value = c * 10^e ; want integer n = round_half_up(value * 10^6)
  if (a.c == 0) return "0.000000";
  bool ng = a.c < 0;
  u128 mag = (u128)(ng ? -a.c : a.c);
  int32_t sh = a.e + 6;
  u128 n;
  if (sh >= 0) {
    if (sh > 20) throw std::overflow_error("set_scale6: value too large");
    n = mag * pow10_u128(sh);
  } else if (-sh > 15) {
    n = 0;  // |value*10^6| < 10^15 * 10^-16 = 0.1 -> rounds to 0
  } else {
    u128 p = pow10_u128(-sh);
    n = mag / p;
    if (mag - n * p >= p / 2) n += 1;
  }
  // print n / 10^6
  char digs[64];
  int k = 0;
  u128 t = n;
  if (t == 0) digs[k++] = '0';
  while (t > 0) { digs[k++] = (char)('0' + (int)(t % 10)); t /= 10; }
  while (k < 7) digs[k++] = '0';
  std::string out;
  if (ng && n != 0) out.push_back('-');
  for (int i = k - 1; i >= 6; i--) out.push_back(digs[i]);
  out.push_back('.');
  for (int i = 5; i >= 0; i--) out.push_back(digs[i]);
  return out;
}

}  // namespace dec15
