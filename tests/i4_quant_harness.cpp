/* Host build of the sweep-vector quantisers of the pre-filters (beta_cores_amd/csrc/bc_i4_quant.h, bc_i8_quant.h), under
 * ASan + UBSan (tests/test_sanitized_cpu.py).  For random vectors (Gaussian, sparse, wide dynamic range, a NaN) it checks what
 * the sweeps' bounds rest on:
 *   4-bit record: every digit pair is a valid signed nibble pair (d0 in [-7, 7], d1 in [-8, 7]), 16 d0 + d1 = Q with
 *     |Q| <= 119 and |Q vstep - v_k| <= vstep / 2 (1 + 1e-11), padding nibbles are zero, and the header's bound
 *     fev >= || v^ - v ||_2 with fvn >= ||v||;
 *   int8 record: d0 in [-127, 127], d1 in [-64, 64], 128 d0 + d1 = Q, |Q vstep0 - v_k| <= vstep0 / 2 (1 + 1e-11), the single
 *     digit of the second vector likewise, fev0 / fev1 >= the quantisation error norms.
 * The device builtins the headers use are given host meanings here (a lane is the whole wave).  Prints "ok <vectors>". */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __device__
#define __forceinline__ inline
#define __restrict__
struct int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float __double2float_ru(double x) { float f = (float)x; return ((double)f < x) ? nextafterf(f, INFINITY) : f; }
static inline double bc_wave_max_all(double v) { return v; }
#include "bc_i8_quant.h"
#include "bc_i4_quant.h"

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }
static int nib(unsigned w, int j) { int v = (int)((w >> (4 * j)) & 0xfu); return v >= 8 ? v - 16 : v; }
static int byt(unsigned w, int j) { int v = (int)((w >> (8 * j)) & 0xffu); return v >= 128 ? v - 256 : v; }
#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } } while (0)

template <int MODE>
static void one(int S, int kind) {
  const int nv = MODE == 0 ? 2 : 1;
  std::vector<double> v((size_t)nv * S + 64, 0.0);
  double m[2] = {0., 0.}, n2[2] = {0., 0.};
  for (int k = 0; k < S; ++k)
    for (int c = 0; c < nv; ++c) {
      double x = nrand();
      if (kind == 1 && urand() < 0.8) x = 0.;
      if (kind == 2) x *= pow(10., -8. * urand());
      v[(size_t)nv * k + c] = x;
      m[c] = bc_i8q_absmax(m[c], x);
      n2[c] += x * x;
    }
  const double vn = sqrt(n2[0]);
  // ---- 4-bit record
  {
    const int g8 = (S + 7) / 8, sp8 = g8 + 3;                      // (three padding groups)
    const bc_i4q_scalars q = bc_i4q_steps(m[0], m[1]);
    CHECK(!q.vbad, "vbad on finite input");
    double e2[2] = {0., 0.};
    for (int g = 0; g < sp8; ++g) {
      unsigned w[4];
      bc_i4q_group<MODE>(v.data(), S, g, q, w);
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        for (int c = 0; c < 2; ++c) {
          const int d0 = nib(w[2 * c], j), d1 = nib(w[2 * c + 1], j);
          if (k >= S || c >= nv) { CHECK(d0 == 0 && d1 == 0, "padding digit not zero (S %d g %d j %d)", S, g, j); continue; }
          CHECK(d0 >= -7 && d0 <= 7 && d1 >= -8 && d1 <= 7, "nibble out of range: %d %d", d0, d1);
          const int Q = 16 * d0 + d1;
          CHECK(Q >= -BC_I4Q_QMAX && Q <= BC_I4Q_QMAX, "Q %d", Q);
          const double step = c == 0 ? q.vstep0 : q.vstep1, err = (double)Q * step - v[(size_t)nv * k + c];
          CHECK(fabs(err) <= 0.5 * step * (1. + 1e-11) + 1e-300, "4-bit digit off: S %d k %d err %g step %g", S, k, err, step);
          e2[c] += err * err;
        }
      }
    }
    const bc_i4q_hdr h = bc_i4q_header(q, S, vn);
    CHECK((double)h.fev0 >= sqrt(e2[0]) && (double)h.fev1 >= sqrt(e2[1]) && (double)h.fvn >= vn, "4-bit header bound too small");
    CHECK(fabs((double)h.fvs0 - q.vstep0) <= 6e-8 * q.vstep0 + 1e-300, "fvs0");
  }
  // ---- int8 record
  {
    const int g4 = (S + 3) / 4, sp4 = g4 + 2;
    const bc_i8q_scalars q = bc_i8q_steps(m[0], m[1]);
    double e2[2] = {0., 0.};
    for (int g = 0; g < sp4; ++g) {
      unsigned w[4];
      bc_i8q_group<MODE>(v.data(), S, g, q, w);
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * g + j;
        const int d0 = byt(w[0], j), d1 = byt(w[1], j), e = byt(w[2], j);
        if (k >= S) { CHECK(d0 == 0 && d1 == 0 && e == 0, "int8 padding digit not zero"); continue; }
        CHECK(d0 >= -127 && d0 <= 127 && d1 >= -64 && d1 <= 64, "int8 digits out of range: %d %d", d0, d1);
        const double err = (double)(128 * d0 + d1) * q.vstep0 - v[(size_t)nv * k];
        CHECK(fabs(err) <= 0.5 * q.vstep0 * (1. + 1e-11) + 1e-300, "int8 digit off");
        e2[0] += err * err;
        if (MODE == 0) {
          CHECK(e >= -127 && e <= 127, "second vector's digit %d", e);
          const double er1 = (double)e * q.vstep1 - v[(size_t)nv * k + 1];
          CHECK(fabs(er1) <= 0.5 * q.vstep1 * (1. + 1e-11) + 1e-300, "int8 second digit off");
          e2[1] += er1 * er1;
        } else {
          CHECK(e == 0, "dot mode: third word must stay empty");
        }
      }
    }
    const bc_i8q_hdr h = bc_i8q_header(q, S, vn);
    CHECK((double)h.fev0 >= sqrt(e2[0]) && (double)h.fev1 >= sqrt(e2[1]), "int8 header bound too small");
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 2000;
  srand(12345);
  long done = 0;
  for (long i = 0; i < n; ++i) {
    const int S = 1 + rand() % 260, kind = rand() % 3;
    one<0>(S, kind);
    one<1>(S, kind);
    done += 2;
  }
  // a NaN / inf in v makes the record "bad": every row then gets [-inf, inf]
  {
    const bc_i4q_scalars a = bc_i4q_steps(bc_i8q_absmax(0., NAN), 1.), b = bc_i4q_steps(1., INFINITY);
    const bc_i8q_scalars c = bc_i8q_steps(bc_i8q_absmax(0., NAN), 1.);
    CHECK(a.vbad && b.vbad && c.vbad, "NaN / inf not flagged");
    unsigned w[4];
    double v[16] = {NAN, 1., 2., 3., 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bc_i4q_group<1>(v, 4, 0, a, w);
    CHECK(w[0] == 0 && w[1] == 0, "bad record must carry no digits");
  }
  printf("ok %ld\n", done);
  return 0;
}
