/* Host model of the ragged-end index arithmetic (beta_cores_amd/csrc/bc_layout.h), built with
 * -fsanitize=address,undefined by tests/test_sanitized_cpu.py.
 *
 * For every row count on the command line (and every S in a fixed list) the buffers the library allocates on the device
 * are malloc'ed here at EXACTLY their device sizes (bc_phi_alloc, bc_pref_create: no alignment slack), and the launches
 * whose threads run past the end of the data are replayed thread by thread with the kernels' own index expressions,
 * doing real loads and stores: AddressSanitizer turns any address outside an allocation into a hard error.
 *   - k_build_i8_r / k_build_i8 (bc_prefilter_i8.h): the int8 mirror builders, grid = mirror tiles of 256 rows over a
 *     Phi of 128-row tiles (round 3's fault: an odd number of Phi tiles);
 *   - the chunk decomposition of bc_project_from_host (bc_project.hip: launch_chunk): every chunk's tile / norm /
 *     column-partial offsets, staged (one partial row per tile) and Theta-resident (one per wave, n_cu = 256) kernels.
 *   - the two-level pre-filter (bc_prefilter_i4.h, round 5): k_build_i4 (4-bit mirror + 16-bit row codes, reads the int8 mirror
 *     and the fp64 rows of LIVE rows only), k_build_r8 (row-major int8 records) and the sweep's two-buffer walk over the
 *     4-bit tiles (every batch it requests, for grids of 1 .. 8 waves per "CU" of a small model chip), with the S-dependent
 *     batch size the host picks (bc_lay_i4_batch).
 * Prints "ok <cases>" and exits 0. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bc_layout.h"

static const int S_LIST[] = {1, 4, 5, 20, 21, 100, 104, 105, 200, 208, 256, 300};

static void replay_builders(long long n, int S) {
  const long long tiles = bc_lay_tiles(n) < 1 ? 1 : bc_lay_tiles(n);          /* bc_phi_alloc: cap_tiles >= 1 */
  const size_t phi_n = bc_lay_phi_doubles(tiles, S), norm_n = (size_t)tiles * BC_LAY_TILE;
  const int sp4 = bc_lay_i8_sp4(S);
  const long long ptiles = bc_lay_i8_tiles(n);
  const size_t words = bc_lay_i8_words(ptiles, sp4), rq_n = (size_t)ptiles * BC_LAY_ITILE;
  double* phi = (double*)malloc(phi_n * sizeof(double));
  double* norms = (double*)malloc(norm_n * sizeof(double));
  int* u8 = (int*)malloc(words * sizeof(int));
  unsigned* rowq = (unsigned*)malloc(rq_n * sizeof(unsigned));             /* bc_hq2 = two halfs = 4 bytes */
  if (!phi || !norms || !u8 || !rowq) { fprintf(stderr, "malloc\n"); exit(2); }
  for (size_t i = 0; i < phi_n; ++i) phi[i] = 1.0;
  for (size_t i = 0; i < norm_n; ++i) norms[i] = (long long)i < n ? 1.0 : 0.0;
  volatile double sink = 0.;
  for (int one_pass = 0; one_pass < 2; ++one_pass) {
    if (one_pass && S > 104) continue;                                    /* k_build_i8_r<104> serves S <= 104 */
    for (long long t = 0; t < ptiles; ++t)
      for (int tid = 0; tid < BC_LAY_ITILE; ++tid) {
        const long long r = t * BC_LAY_ITILE + tid;
        const int live = r < n && norms[r < n ? r : 0] != 0.;
        if (live) sink += norms[r];
        if (one_pass) {
          const long long rr = bc_lay_i8_src_row(r, n);
          const double* p = phi + bc_lay_phi_elem(rr, 0, S);
          for (int k = 0; k < 104; ++k)
            if (k < S && n > 0) sink += p[(size_t)k * BC_LAY_TILE];        /* unconditional per lane: dead rows load too */
        } else if (live) {
          const double* p = phi + bc_lay_phi_elem(r, 0, S);
          for (int k = 0; k < S; ++k) sink += p[(size_t)k * BC_LAY_TILE];
        }
        int* q = u8 + bc_lay_i8_word(t, 0, tid, sp4);
        for (int g = 0; g < sp4; ++g) q[(size_t)g * BC_LAY_ITILE] = g;
        rowq[r] = 1u;
      }
  }
  (void)sink;
  free(phi); free(norms); free(u8); free(rowq);
}

static void replay_two_level(long long n, int S) {
  const long long tiles = bc_lay_tiles(n) < 1 ? 1 : bc_lay_tiles(n);
  const size_t phi_n = bc_lay_phi_doubles(tiles, S), norm_n = (size_t)tiles * BC_LAY_TILE;
  const int sp4 = bc_lay_i8_sp4(S), g4 = (S + 3) / 4, U = bc_lay_i4_batch(S), sp8 = bc_lay_i4_sp8(S, U), rb = bc_lay_r8_bytes(S);
  const long long ptiles = bc_lay_i8_tiles(n);
  if (sp8 % U != 0 || sp8 * 8 < S || rb % 128 != 0 || rb < 4 * g4 + 4) { fprintf(stderr, "i4 layout: S %d U %d sp8 %d rb %d\n", S, U, sp8, rb); exit(1); }
  const size_t w8 = bc_lay_i8_words(ptiles, sp4), w4 = bc_lay_i4_words(ptiles, sp8), rq_n = (size_t)ptiles * BC_LAY_ITILE;
  double* phi = (double*)malloc(phi_n * sizeof(double));
  double* norms = (double*)malloc(norm_n * sizeof(double));
  int* u8 = (int*)malloc(w8 * sizeof(int));
  unsigned* rowq = (unsigned*)malloc(rq_n * sizeof(unsigned));
  int* u4 = (int*)malloc(w4 * sizeof(int));
  unsigned short* rowq4 = (unsigned short*)malloc(rq_n * sizeof(unsigned short));
  unsigned char* r8 = (unsigned char*)malloc(rq_n * (size_t)rb);
  if (!phi || !norms || !u8 || !rowq || !u4 || !rowq4 || !r8) { fprintf(stderr, "malloc\n"); exit(2); }
  for (size_t i = 0; i < phi_n; ++i) phi[i] = 1.0;
  for (size_t i = 0; i < norm_n; ++i) norms[i] = (long long)i < n ? 1.0 : 0.0;
  for (size_t i = 0; i < w8; ++i) u8[i] = 0;
  for (size_t i = 0; i < rq_n; ++i) rowq[i] = (long long)i < n ? 1u : 0xffffffffu;      /* dead rows past the end */
  volatile double sink = 0.;
  /* k_build_i4 and k_build_r8: one block per mirror tile, thread = row */
  for (long long t = 0; t < ptiles; ++t)
    for (int tid = 0; tid < BC_LAY_ITILE; ++tid) {
      const long long r = t * BC_LAY_ITILE + tid;
      const int dead = rowq[r] == 0xffffffffu;
      int* q = u4 + bc_lay_i4_word(t, 0, tid, sp8);
      if (dead) {
        for (int g = 0; g < sp8; ++g) q[(size_t)g * BC_LAY_ITILE] = 0;
      } else {
        const int* src = u8 + bc_lay_i8_word(t, 0, tid, sp4);
        for (int g = 0; g < g4; ++g) sink += src[(size_t)g * BC_LAY_ITILE];
        sink += norms[r];
        const double* p = phi + bc_lay_phi_elem(r, 0, S);
        for (int g = 0; g < sp8; ++g) {
          for (int j = 0; j < 8; ++j) if (8 * g + j < S) sink += p[(size_t)(8 * g + j) * BC_LAY_TILE];
          q[(size_t)g * BC_LAY_ITILE] = g;
        }
      }
      rowq4[r] = 1;
      {
        const int* src = u8 + bc_lay_i8_word(t, 0, tid, sp4);
        int* dst = (int*)(r8 + (size_t)r * rb);
        for (int c = 0; c < rb / 16; ++c)
          for (int j = 0; j < 4; ++j) {
            const int g = 4 * c + j;
            dst[4 * c + j] = g < g4 ? src[(size_t)g * BC_LAY_ITILE] : (g == rb / 4 - 1 ? (int)rowq[r] : 0);
          }
      }
    }
  /* the sweep's walk: waves 4b + w of a grid sized as bc_pref_create does (wmax waves, every wave the same number of tiles),
   * two buffers, the batches it requests (k_sweep_i4: issue) -- every 1 KiB read must lie inside the mirror */
  for (int n_cu = 1; n_cu <= 3; n_cu += 2) {
    const long long wmax = (long long)n_cu * 8, rounds = (ptiles + wmax - 1) / wmax, waves = (ptiles + rounds - 1) / rounds;
    const long long grid = (waves + 3) / 4 < 1 ? 1 : (waves + 3) / 4, tstride = grid * 4;
    char* seen = (char*)calloc((size_t)ptiles, 1);
    for (long long wg = 0; wg < tstride; ++wg) {
      long long tc = wg;
      int gc = 0;
      while (tc < ptiles) {
        for (int u = 0; u < U; ++u) for (int lane = 0; lane < 64; lane += 63) {
          const int* p = u4 + (size_t)tc * sp8 * BC_LAY_ITILE + (size_t)(gc + u) * 256 + 4 * lane;      /* dwordx4 of lane */
          sink += p[0] + p[3];
        }
        if (gc == 0) { sink += rowq4[tc * BC_LAY_ITILE] + rowq4[tc * BC_LAY_ITILE + 255]; seen[tc] += 1; }
        gc += U;
        if (gc >= sp8) { gc = 0; tc += tstride; }
      }
    }
    for (long long t = 0; t < ptiles; ++t) if (seen[t] != 1) { fprintf(stderr, "tile %lld walked %d times (n = %lld, S = %d)\n", t, seen[t], n, S); exit(1); }
    free(seen);
  }
  (void)sink;
  free(phi); free(norms); free(u8); free(rowq); free(u4); free(rowq4); free(r8);
}

static void replay_chunks(long long n, int S, int dz, int rgrid, long long forced) {
  const long long tiles = bc_lay_tiles(n);
  if (rgrid > 0 && tiles < (long long)rgrid * 8) return;                   /* project_r_grid: the resident kernel needs >= 8 * n_cu tiles */
  const size_t phi_n = bc_lay_phi_doubles(tiles, S), norm_n = (size_t)tiles * BC_LAY_TILE;
  const size_t part_rows = rgrid > 0 ? (size_t)rgrid * 8 : (size_t)tiles;  /* tile_part holds `tiles` rows; the resident kernel uses 8 * rgrid of them */
  char* phi_touched = (char*)calloc(tiles, 1);                              /* one flag per tile instead of 8 GB of doubles */
  char* norms = (char*)malloc(norm_n);                                      /* one byte per row stands for one double */
  char* part = (char*)malloc((size_t)tiles * S);
  char* z_touched = (char*)calloc((size_t)n, 1);
  if (!phi_touched || !norms || !part || !z_touched) { fprintf(stderr, "malloc\n"); exit(2); }
  (void)phi_n;
  const long long unit = bc_lay_chunk_unit(rgrid);
  const long long chunk = bc_lay_chunk_rows(dz, unit, forced);
  if (chunk % unit != 0 || chunk <= 0) { fprintf(stderr, "chunk %lld is no multiple of %lld\n", chunk, unit); exit(1); }
  for (long long row0 = 0; row0 < n; row0 += chunk) {
    const long long rows = (n - row0) < chunk ? (n - row0) : chunk;
    if (row0 % BC_LAY_TILE != 0) { fprintf(stderr, "chunk start %lld not tile aligned\n", row0); exit(1); }
    if (rgrid > 0 && (row0 / 32) % ((long long)rgrid * 8) != 0) { fprintf(stderr, "chunk start %lld breaks the wave -> group map\n", row0); exit(1); }
    const long long tile0 = row0 / BC_LAY_TILE, nt = (rows + BC_LAY_TILE - 1) / BC_LAY_TILE;
    for (long long t = 0; t < nt; ++t) phi_touched[tile0 + t] = 1;          /* a.tiles + tile0 * S * 128, tile t of the launch */
    for (long long r = 0; r < rows; ++r) { norms[row0 + r] = 1; z_touched[row0 + r] = 1; }
    if (rgrid > 0) {
      for (size_t w = 0; w < part_rows; ++w) part[w * S + (S - 1)] = 1;   /* a.tile_part: one row per wave, shared by the chunks */
    } else {
      for (long long t = 0; t < nt; ++t) part[(size_t)(tile0 + t) * S + (S - 1)] = 1;   /* a.tile_part + tile0 * S */
    }
  }
  for (long long t = 0; t < tiles; ++t) if (!phi_touched[t]) { fprintf(stderr, "tile %lld never written (n = %lld)\n", t, n); exit(1); }
  for (long long r = 0; r < n; ++r) if (!z_touched[r]) { fprintf(stderr, "row %lld never projected\n", r); exit(1); }
  free(phi_touched); free(norms); free(part); free(z_touched);
}

int main(int argc, char** argv) {
  long cases = 0;
  for (int a = 1; a < argc; ++a) {
    const long long n = atoll(argv[a]);
    for (size_t i = 0; i < sizeof(S_LIST) / sizeof(S_LIST[0]); ++i) {
      if (n <= 2000 || (S_LIST[i] == 100 && n <= 70000)) { replay_builders(n, S_LIST[i]); ++cases; }
      if (S_LIST[i] <= 256 && (n <= 1500 || (S_LIST[i] == 100 && n <= 70000))) { replay_two_level(n, S_LIST[i]); ++cases; }
    }
    if (n > 0) {
      replay_chunks(n, 100, 129, 0, 65536); ++cases;
      replay_chunks(n, 100, 129, 0, 0); ++cases;
      replay_chunks(n, 100, 129, 256, 65536); ++cases;
      replay_chunks(n, 100, 129, 256, 0); ++cases;
      replay_chunks(n, 37, 7, 256, 100000); ++cases;
    }
  }
  printf("ok %ld\n", cases);
  return 0;
}
