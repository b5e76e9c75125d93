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
 * Prints "ok <cases>" and exits 0. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bc_layout.h"

static const int S_LIST[] = {1, 4, 5, 20, 21, 100, 104, 105, 300};

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
