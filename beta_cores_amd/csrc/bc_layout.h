// bc_layout.h -- the index arithmetic of the HBM layouts, shared by the kernels, the host-side allocation code and the
// host model that walks every address the ragged-end kernels form (tests/layout_harness.c, built with
// -fsanitize=address,undefined by tests/test_sanitized_cpu.py).  Round 3's one GPU fault was an index error of exactly
// this kind (the one-pass int8 mirror builder read rows past the end of Phi when the number of 128-row tiles is odd):
// the expressions live here, once, so that the model cannot drift from the kernels.
//
//   Phi         row tiles of 128 rows, inside a tile [S][128]:   element (row r, sample k) at  (r/128)*S*128 + k*128 + r%128
//   int8 mirror tiles of 256 rows, inside a tile [SP4][256] dwords (one dword = 4 consecutive samples of one row),
//               SP4 = ceil(S/4) rounded up to a multiple of 5 (the sweep walks k-groups in batches of 5); (scale, delta)
//               halfs per row: [ptiles*256]
//   chunks      of a pipelined projection (bc_project_from_host): multiples of `unit` rows -- 512 tiles for the staged K1,
//               8 * n_cu 32-row groups for the Theta-resident one -- of about 128 MiB
#ifndef BC_LAYOUT_H
#define BC_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BC_LAY __host__ __device__ __forceinline__
#else
#define BC_LAY static inline
#endif

#define BC_LAY_TILE 128      /* == BC_TILE_ROWS */
#define BC_LAY_ITILE 256     /* == BC_ITILE */
#define BC_LAY_IU 5          /* == BC_IU */
#define BC_LAY_IMAXG 320     /* == BC_IMAXG: k-groups the sweep's digit table holds */

BC_LAY long long bc_lay_tiles(long long n_rows) { return (n_rows + BC_LAY_TILE - 1) / BC_LAY_TILE; }
BC_LAY size_t bc_lay_phi_doubles(long long tiles, int S) { return (size_t)tiles * (size_t)S * BC_LAY_TILE; }
BC_LAY size_t bc_lay_phi_elem(long long row, int k, int S) {
  return (size_t)(row >> 7) * (size_t)S * BC_LAY_TILE + (size_t)k * BC_LAY_TILE + (size_t)(row & (BC_LAY_TILE - 1));
}

BC_LAY int bc_lay_i8_sp4(int S) { return ((S + 3) / 4 + BC_LAY_IU - 1) / BC_LAY_IU * BC_LAY_IU; }
BC_LAY long long bc_lay_i8_tiles(long long n_rows) {
  const long long t = (n_rows + BC_LAY_ITILE - 1) / BC_LAY_ITILE;
  return t < 1 ? 1 : t;
}
BC_LAY size_t bc_lay_i8_words(long long ptiles, int sp4) { return (size_t)ptiles * (size_t)sp4 * BC_LAY_ITILE; }
BC_LAY size_t bc_lay_i8_word(long long ptile, int g, int row_in_tile, int sp4) {
  return (size_t)ptile * (size_t)sp4 * BC_LAY_ITILE + (size_t)g * BC_LAY_ITILE + (size_t)row_in_tile;
}
// The one-pass builder loads a row's S values unconditionally (all loads in flight, no per-lane condition): mirror rows past
// the end of the data have no Phi tile behind them when the number of 128-row tiles is odd, so they read row 0 (and are dead).
BC_LAY long long bc_lay_i8_src_row(long long r, long long n_rows) {
#ifdef BC_LAY_TEST_NO_CLAMP      /* tests/test_sanitized_cpu.py only: the host model must FAIL without the clamp */
  (void)n_rows;
  return r;
#else
  return r < n_rows ? r : 0;
#endif
}

// ---- two-level pre-filter (bc_prefilter_i4.h)
//   4-bit mirror  tiles of 256 rows, inside a tile [SP8][256] dwords (one dword = 8 consecutive samples of one row as signed
//                 nibbles, sample 8g+j in bits 4j..4j+3); SP8 = ceil(S/8) rounded up to a multiple of the sweep's batch U;
//                 per row one 16-bit word (scale code | delta code << 8): [ptiles*256]
//   int8 records  row-major copy of the int8 mirror for the second level's gather: record of RB bytes per row,
//                 ceil(S/4) dwords of digits, zeros, and the row's (scale, delta) halfs in its LAST dword; RB a multiple of
//                 128 so that a row is one (or a few whole) cache lines
BC_LAY int bc_lay_i4_groups(int S) { return (S + 7) / 8; }
BC_LAY int bc_lay_i4_sp8(int S, int U) { return (bc_lay_i4_groups(S) + U - 1) / U * U; }
// the batch size (loads a wave keeps in flight per buffer) with the least padding among the instantiated ones; ties: larger
BC_LAY int bc_lay_i4_batch(int S) {
  const int cand[5] = {13, 8, 7, 6, 5};
  int best = cand[0];
  for (int i = 1; i < 5; ++i)
    if (bc_lay_i4_sp8(S, cand[i]) < bc_lay_i4_sp8(S, best)) best = cand[i];
  return best;
}
BC_LAY size_t bc_lay_i4_words(long long ptiles, int sp8) { return (size_t)ptiles * (size_t)sp8 * BC_LAY_ITILE; }
BC_LAY size_t bc_lay_i4_word(long long ptile, int g, int row_in_tile, int sp8) {
  return (size_t)ptile * (size_t)sp8 * BC_LAY_ITILE + (size_t)g * BC_LAY_ITILE + (size_t)row_in_tile;
}
BC_LAY int bc_lay_r8_bytes(int S) { return (4 * ((S + 3) / 4) + 4 + 127) / 128 * 128; }

BC_LAY long long bc_lay_chunk_unit(int rgrid) { return rgrid > 0 ? (long long)rgrid * 8 * 32 : (long long)BC_LAY_TILE * 512; }
BC_LAY long long bc_lay_chunk_rows(int dz, long long unit, long long forced) {
  long long rows = (((long long)128 << 20) / ((long long)dz * 8) + unit - 1) / unit * unit;
  if (forced > 0) rows = (forced + unit - 1) / unit * unit;
  return rows;
}

#endif  // BC_LAYOUT_H
