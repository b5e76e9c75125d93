// K1: per-datapoint (beta-)log-likelihood projection, fused with row-centring (projector.py:26,55),
// row norms (giga.py:10) and per-tile column sums (K2, hilbert.py:17 / bcores.py:77).
//
//   Phi[i, s] = f(z_i, theta_s) - mean_s f(z_i, theta_.)
//
// The contraction P[s, i] = sum_d Theta[s, d] * Z[i, d] is a dense (S x D)(D x 128) product
// per 128-row tile; it runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) with Theta as the
// A operand and the Z tile as the B operand, so that the accumulator of a lane holds, for ONE
// data row, samples {g + 4*reg + 16*tile}: the model formula, the row mean and the row norm are
// then computed in registers with two cross-lane adds, and the tile is stored straight into
// the [S][128] layout the K3 sweep streams.  Z and Theta are staged through LDS in D-chunks
// (coalesced global loads, register prefetch of the next chunk while the MFMAs of the current
// one run).  Algorithmic traffic per tile: 8*128*Dz B read + 8*128*S B written.
//
// Formula sources (expression order kept, -ffp-contract=off):
//   model_linreg.py:4-10 / model_neurlinr.py:90-97,102-110 / model_lr.py:72-86 / gaussian.py:7-15,34-62
#include "bc_internal.h"
#include <cmath>
#include <cstring>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

struct ProjArgs {
  const double* z;        // [n_rows][dz]
  const double* theta;    // [nt*16][dk]  zero padded
  const double* saux;     // [nt*16] per-sample extra (gauss: theta^T Siginv theta) or null
  const double* rowaux;   // [n_rows] per-row extra (gauss: x^T Siginv x) or null
  double* tiles;
  double* norms;
  double* tile_part;
  long long n_rows;
  int dz, d, dk, s, model;
  double p[6];            // model constants, see bc_project()
};

__device__ __forceinline__ double bc_model_value(int model, double p, double ra, double sa, const double* c) {
  switch (model) {
    case BC_MODEL_LINREG_LL: {            // c0 - c1*(y^2 - 2*p*y + p^2)
      const double q = (ra * ra - (2. * p) * ra) + p * p;
      return c[0] - c[1] * q;
    }
    case BC_MODEL_LINREG_BETA: {          // k0*(k1*exp(k2*q) + k3)
      const double q = (ra * ra - (2. * p) * ra) + p * p;
      return c[0] * (c[1] * exp(c[2] * q) + c[3]);
    }
    case BC_MODEL_LOGISTIC_LL: {          // m = -z.th ; m < 100 ? -log1p(exp(m)) : -m
      const double m = -p;
      return (m < 100.) ? -log1p(exp(m)) : -m;
    }
    case BC_MODEL_LOGISTIC_BETA: {        // -( (b+1)/b*(1+e^m)^-b - ((1+e^m)^(-b-1) + (1+e^-m)^(-b-1)) )
      const double m = -p;
      const double em = exp(m), enm = exp(-m);
      return -((c[0] * pow(1. + em, c[1])) - (pow(1. + em, c[2]) + pow(1. + enm, c[2])));
    }
    case BC_MODEL_GAUSS_LL: {             // cc - 1/2*(xSx + tSt - 2*xSt)
      const double q = (ra + sa) - 2. * p;
      return c[0] - 1. / 2. * q;
    }
    case BC_MODEL_GAUSS_BETA: {           // 1/b*exp(-.5*b*q) - (1+b)^(-.5d-1)
      const double q = (ra + sa) - 2. * p;
      return c[0] * exp(c[1] * q) - c[2];
    }
    default: {                            // BC_MODEL_GAUSS_BETA_GRAD, gaussian.py:46-62
      const double q = (ra + sa) - 2. * p;
      const double gq = exp(c[1] * q);
      const double t1 = c[3] * (c[0] * gq - c[2]);
      const double t2 = c[4] * gq;
      const double t3 = c[5] * q * gq;
      return ((t1 - t2) - t3) - p_dummy_never_used(c);
    }
  }
}
