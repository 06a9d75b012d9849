/* Offline model of what k_schur_items asks of an XCD's L2 (tools/l2sim/run.py writes the work lists): per XCD group the items are
 * taken in list order by W concurrent waves (one item each), every wave walks its item's pairs `step` at a time, all resident
 * waves advancing in turn; a pair touches the G block of k and of k2 (blk_lines lines of 128 B each, block k at line k *
 * blk_lines).  Cache: `mb` MiB, 16-way set associative, LRU, 128-byte lines.  Prints requests / hits for the k side and the k2
 * side and the lines that leave the cache hierarchy (misses x 128 B).
 * usage: l2sim <file> [W=640] [mb=4] [step=8] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int32_t *tag; uint32_t *age; uint32_t clock; int sets; } cache_t;
static int cache_access(cache_t* c, int64_t line) {
  const int s = (int)((uint64_t)(line * 0x9E3779B97F4A7C15ull >> 20) % (uint64_t)c->sets);
  int32_t* t = c->tag + (size_t)s * 16; uint32_t* a = c->age + (size_t)s * 16;
  int victim = 0; uint32_t oldest = 0xFFFFFFFFu;
  c->clock++;
  for (int w = 0; w < 16; ++w) {
    if (t[w] == (int32_t)line) { a[w] = c->clock; return 1; }
    if (a[w] < oldest) { oldest = a[w]; victim = w; }
  }
  t[victim] = (int32_t)line; a[victim] = c->clock;
  return 0;
}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  const int W = argc > 2 ? atoi(argv[2]) : 640, step = argc > 4 ? atoi(argv[4]) : 8;
  const double mb = argc > 3 ? atof(argv[3]) : 4.0;
  int64_t hdr[4];
  if (!f || fread(hdr, 8, 4, f) != 4) return 1;
  const int64_t n_pairs = hdr[0], n_items = hdr[1], blk_lines = hdr[2];
  int32_t *pk = malloc(n_pairs * 4), *pk2 = malloc(n_pairs * 4), *ib = malloc(n_items * 4), *ie = malloc(n_items * 4), *xi = malloc(n_items * 4), xp[9];
  if (fread(pk, 4, n_pairs, f) != (size_t)n_pairs || fread(pk2, 4, n_pairs, f) != (size_t)n_pairs || fread(ib, 4, n_items, f) != (size_t)n_items ||
      fread(ie, 4, n_items, f) != (size_t)n_items || fread(xp, 4, 9, f) != 9 || fread(xi, 4, n_items, f) != (size_t)n_items) return 2;
  int64_t req[2] = {0, 0}, hit[2] = {0, 0};
  for (int x = 0; x < 8; ++x) {
    cache_t c; c.sets = (int)(mb * 1048576.0 / 128 / 16); c.clock = 0;
    c.tag = malloc((size_t)c.sets * 16 * 4); c.age = calloc((size_t)c.sets * 16, 4);
    memset(c.tag, 0xFF, (size_t)c.sets * 16 * 4);
    int next = xp[x]; const int end = xp[x + 1];
    int* cur = malloc(W * 4); int* pos = malloc(W * 4);
    int live = 0;
    for (int w = 0; w < W; ++w) { cur[w] = -1; }
    for (;;) {
      int active = 0;
      for (int w = 0; w < W; ++w) {
        if (cur[w] < 0 || pos[w] >= ie[cur[w]]) { if (next < end) { cur[w] = xi[next++]; pos[w] = ib[cur[w]]; } else { cur[w] = -1; continue; } }
        active++;
        const int e = pos[w] + step < ie[cur[w]] ? pos[w] + step : ie[cur[w]];
        for (int p = pos[w]; p < e; ++p) {
          for (int side = 0; side < 2; ++side) {
            const int64_t blk = side ? pk2[p] : pk[p];
            if (side && pk2[p] == pk[p]) continue;                    /* self-pair: one gather serves both operands */
            for (int l = 0; l < blk_lines; ++l) { req[side]++; hit[side] += cache_access(&c, blk * blk_lines + l); }
          }
        }
        pos[w] = e;
      }
      if (!active) break;
      (void)live;
    }
    free(c.tag); free(c.age); free(cur); free(pos);
  }
  const double tot = (double)(req[0] + req[1]), h = (double)(hit[0] + hit[1]);
  printf("W %d, %.1f MiB, step %d: k side %lld requests %.1f %% hits; k2 side %lld requests %.1f %% hits; all %.1f %% hits; leaves L2: %.3f GB\n", W, mb, step,
         (long long)req[0], 100.0 * hit[0] / (double)req[0], (long long)req[1], 100.0 * hit[1] / (double)req[1], 100.0 * h / tot, (tot - h) * 128 / 1e9);
  return 0;
}
