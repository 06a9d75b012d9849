// Brute-force kNN (k=2) + Lowe ratio test for gfx950 (MI355X).
//
// Replaces cv2.BFMatcher.knnMatch(desc1, desc2, k=2) and the ratio loop of
// ImageMatcher.match_features (/root/reference/utils/find_matches.py:141-155).
//
// L2 on uint8 descriptors (SIFT, dim 32/64/128) is exact integer arithmetic on the matrix cores:
//   a = x - 128 (= x ^ 0x80), c = 127 - y (= y ^ 0x7F) are int8, and
//   d^2 = sum (x-y)^2 = sum (a + c + 1)^2 = TN_i + QN_j + 2 * dot(a_i, c_j),
//   TN_i = sum (a+1)^2 - dim,  QN_j = sum (c+1)^2           (all int32, d^2 <= 128*255^2 < 2^23)
// with dot() from v_mfma_i32_32x32x32_i8 (A = 32 train rows, B = 32 queries: the accumulator puts
// the query on the lane and 16 train rows in the registers, so the running top-2 of a query never
// leaves its lane).  The accumulator starts at TN_i >> 1, so that 2 acc + (TN_i & 1) = d^2 - QN_j; a candidate filter
// on the accumulators (k_knn2_u8) skips what cannot enter a top-2, the rest is ranked by ONE int32 key
// = ((d^2 - QN_j) << 8) | row-in-window, i.e. v_lshl_add_u32 + v_min_i32 + v_med3_i32 per candidate; keys are unpacked
// every 256 train rows.  Two distance kernels: k_knn2_u8 (train rows staged through LDS; batched image pairs, small
// dims) and k_knn2_u8_direct (one dim-128 pair of >= 8e6 distances: operands straight from L2, no LDS, no barrier).
// Ranking rule = the reference's: OpenCV compares the float32 distances sqrtf(d^2), lowest train index first
// on ties (matcher_oracle.py).  sqrtf is monotone and injective on integers below 2^22, so ranking on the integer
// d^2 gives the same two neighbours whenever the second-best d^2 is below 2^22 (always, for SIFT descriptors:
// rows of norm <= 512 are at most 2^20 apart).  From 2^22 on distinct d^2 can round to the same float32 and the
// lower index must win among them; queries whose second-best d^2 reaches 2^22 are therefore re-ranked on the
// float32 value over the whole train set by k_knn2_u8_rerank (k_merge_splits_u8 lists them).
#include "common.h"
#include "match_plan.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <type_traits>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define D2_MAX_VALID 8323200   // 128 * 255^2
#define SENT_TH 0x3FFFFF       // accumulator start of a padding row: 2 * SENT_TH = 8388606 > every valid d^2, key below 2^31

// Correctly rounded float32 square root (what sqrtf / OpenCV's std::sqrt give on the CPU): the fp64 root
// (correctly rounded, 53 >= 2*24+2 bits) rounded once to float32.  __fsqrt_rn is NOT correctly rounded here.
__device__ __forceinline__ float sqrt_rn_f32(float x) { return (float)sqrt((double)x); }

struct Cand { float d; int i; };   // distance (already sqrtf'ed / popcount), train index; i < 0 = empty

// Batched (segmented) matching: one launch covers every image pair of a preprocessing step
// (find_matches.py:329-350 calls match_features once per pair, serially).  A segment = one pair = a range of query
// rows against a range of train rows; the host cuts the segments into workgroup-sized pieces (plan_segments) and
// every workgroup of the distance kernels reads its piece from this record.  wg == nullptr = one segment covering
// the whole arrays, pieces computed from blockIdx as before.
struct MatchSegs {            // device copies of the segment table, for the kernels that work per output row
  const int64_t *q_beg, *t_beg, *t_end, *out_ptr;
  int32_t n_seg;
};

__device__ __forceinline__ bool cand_less(float d, int i, float d2, int i2) {
  return (d < d2) || (d == d2 && (unsigned)i < (unsigned)i2);   // i = -1 (empty) sorts last
}
__device__ __forceinline__ void top2_insert(float d, int i, float& b1d, int& b1i, float& b2d, int& b2i) {
  if (i < 0) return;
  if (cand_less(d, i, b1d, b1i)) { b2d = b1d; b2i = b1i; b1d = d; b1i = i; }
  else if (cand_less(d, i, b2d, b2i)) { b2d = d; b2i = i; }
}

// ------------------------------------------------------------------------------------ row norms (uint8)
// out[i] = sum (v+1)^2 - sub over the row, v = (byte ^ flip) as int8.  train: flip 0x80, sub = dim;
// query: flip 0x7F, sub = 0.  dim / 16 lanes per row (16-byte loads: a row is read as one coalesced run), partial sums
// combined by shuffles; dim is 32, 64 or 128.
__global__ __launch_bounds__(256) void k_row_norm_u8(const uint8_t* __restrict__ x, int64_t n, int dim, int flip,
                                                     int sub, int* __restrict__ out) {
  const int parts = dim >> 4;                                  // 2, 4 or 8 lanes per row
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = t / parts;
  const int part = (int)(t - r * parts);
  int s = 0;
  if (r < n) {
    const uint4 v = *(const uint4*)(x + r * dim + part * 16);
    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int sh = 0; sh < 32; sh += 8) {
        const int b = (int)(int8_t)(((wds[q] >> sh) & 0xFFu) ^ (uint32_t)flip) + 1;
        s += b * b;
      }
  }
  for (int o = parts >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);      // the lanes of a row are adjacent and aligned
  if (r < n && part == 0) out[r] = s - sub;
}

// ------------------------------------------------------------------------------------ L2 / uint8 on i8 MFMA
// Train-side pre-pass (one launch per call): per row i the accumulator start TH_i = TN_i >> 1, the key parity bit
// PAR_i = (TN_i & 1) << 8 and the row with its bytes flipped to int8 (x ^ 0x80), so that the distance kernel can
// bring all three into LDS with direct global->LDS loads (no registers, no ds_write).  Slot nt of every array is
// the padding row: zero bytes, TH = SENT_TH, PAR = 0.
__global__ __launch_bounds__(256) void k_train_prep_u8(const uint8_t* __restrict__ x, int64_t n, int dim,
                                                       uint8_t* __restrict__ xf, int* __restrict__ th, int* __restrict__ par,
                                                       int* __restrict__ fix_cnt) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *fix_cnt = 0;    // the re-rank list of this call starts empty (filled by k_merge_splits_u8)
  const int parts = dim >> 4;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = t / parts;
  const int part = (int)(t - r * parts);
  int s = 0;
  if (r < n) {
    uint4 v = *(const uint4*)(x + r * dim + part * 16);
    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    *(uint4*)(xf + r * dim + part * 16) = v;
    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int sh = 0; sh < 32; sh += 8) {
        const int b = (int)(int8_t)((wds[q] >> sh) & 0xFFu) + 1;
        s += b * b;
      }
  } else if (r == n) {
    *(uint4*)(xf + r * dim + part * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  for (int o = parts >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);      // the lanes of a row are adjacent and aligned
  if (part == 0) {
    if (r < n) { const int tn = s - dim; th[r] = tn >> 1; par[r] = (tn & 1) << 8; }
    else if (r == n) { th[r] = SENT_TH; par[r] = 0; }
  }
}

// loop with a compile-time index (arrays indexed by it stay in registers whatever the unroller decides)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// grid.x = n_qblocks * nsplit: split = blockIdx.x % nsplit, the workgroups of one query block run side by side.
// 256 threads = 4 waves, each wave owns QB blocks of 32 queries; the train split is streamed through LDS in chunks of 128
// rows (XOR-swizzled 16-byte granules: conflict-free ds_read_b128), double buffered through registers: the loads of chunk
// ch + 1 are issued before chunk ch is worked on and written to LDS after it.  (Direct global->LDS loads into a ring of
// three buffers, two chunks in flight, measured 8-13 % slower here: ~100 cycles of issue per 1 KiB piece and wave.)
//
// Candidate filter (FILTER): the accumulator starts at TH_i = TN_i >> 1, so that after the MFMAs
// 2 acc + parity = d^2 - QN_j, and a candidate can only matter if acc < (U2 - QN_j) / 2 + 1, U2 = an upper bound of
// the query's final second-best d^2: the second-best so far over both half-waves AND over the other train splits of
// the same queries, refreshed every 256 train rows.  A 32 x 32 tile whose 16 x 64 accumulators all fail that test
// (10 min operations + 1 compare per lane instead of 48 ranking operations) is skipped, and in a tile that is not, only
// the groups of four rows that hold a candidate are ranked.  Skipping is exact, not approximate: the test keeps every
// candidate with d^2 <= U2, and ties are still resolved on (d^2, index) by the ranking and the merge.
#ifndef KNN_WAVES
#define KNN_WAVES 2
#endif
// dim 256 (KS = 8: Hamming-256 / ORB over unpacked bits) holds twice the query and operand fragments and spills 7-15 registers
// at two waves per SIMD.  Measured (round 3, 30,000 x 30,000 x 256 bits): one wave per SIMD, no spills (-DSFM_KNN_KS8_WAVES=1)
// 535 us = 1.68e12 pairs/s; two waves with the spills 429 us = 2.10e12 - the second wave is worth more than the spills cost.
// Pairs of 8e6 distances and more no longer come here: they take k_knn2_u8_direct<2, 8> (no LDS, 230 registers, no spill in the
// loop): 30,000 x 30,000: 461 -> 289 us per call = 3.1e12 pairs/s, 50,000 x 50,000: 1,130 -> 660 us = 3.8e12.
#ifndef SFM_KNN_KS8_WAVES
#define SFM_KNN_KS8_WAVES KNN_WAVES
#endif
template <int KS, int QB, bool FILTER>   // KS = dim / 32
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KS == 8 ? SFM_KNN_KS8_WAVES : KNN_WAVES, KS == 8 ? SFM_KNN_KS8_WAVES : KNN_WAVES))) void k_knn2_u8(
    const uint8_t* __restrict__ q, int64_t nq, const uint8_t* __restrict__ tf, int64_t nt,
    const int* __restrict__ th_g, const int* __restrict__ par_g, const int* __restrict__ qn,
    int nsplit, int64_t rows_per_split, const MatchWG* __restrict__ wg, int64_t total_out, Cand* __restrict__ part, int* u2g) {
  constexpr int DIM = KS * 32;
  constexpr int GPR = DIM / 16;                 // 16-byte granules per row
  constexpr int CHUNK = 128;                   // rows per LDS chunk
  constexpr int GPT = CHUNK * GPR / 256;        // granules staged per thread
  __shared__ uint4 s_tb[2][CHUNK * GPR];
  __shared__ int s_thb[2][CHUNK];     // TH_i = TN_i >> 1: where the accumulator of train row i starts
  __shared__ int s_parb[2][CHUNK];    // PAR_i = (TN_i & 1) << 8: the parity bit of the key
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
  int split;
  int64_t t_beg, t_end, q0, t_seg = 0, out_shift = 0;     // out_shift: output row = query row + out_shift
  if (wg) {
    const MatchWG r = wg[blockIdx.x];
    split = r.split; t_beg = r.t_first; t_end = r.t_end; t_seg = r.t_seg;
    q0 = r.q_first + (int64_t)w * (QB * 32);
    nq = r.q_end;
    out_shift = r.out_first - r.q_first;
  } else {
    split = blockIdx.x % nsplit;
    const int64_t qblock = blockIdx.x / nsplit;
    t_beg = (int64_t)split * rows_per_split;
    t_end = (t_beg + rows_per_split) < nt ? (t_beg + rows_per_split) : nt;
    q0 = qblock * (4 * QB * 32) + (int64_t)w * (QB * 32);
    total_out = nq;
  }

  // query fragments: lane holds bytes [32 ks + 16 half, +16) of query q0 + 32 qb + l31
  v4i bq[QB][KS];
  int qnv[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int64_t qi = q0 + qb * 32 + l31;
    const bool ok = qi < nq;
    qnv[qb] = ok ? qn[qi] : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      v4i v = {0, 0, 0, 0};
      if (ok) v = *(const v4i*)(q + qi * DIM + ks * 32 + half * 16);
      bq[qb][ks] = v ^ 0x7F7F7F7F;
    }
  }
  int g1i[QB], g2i[QB];
  int g1k[QB], g2k[QB];   // exact integer d^2 of the running best two
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { g1k[qb] = 0x7FFFFFFF; g2k[qb] = 0x7FFFFFFF; g1i[qb] = -1; g2i[qb] = -1; }

  const int64_t n_rows = t_end > t_beg ? (t_end - t_beg) : 0;
  const int n_chunks = (int)((n_rows + CHUNK - 1) / CHUNK);

  // rows past the split read the padding row (slot nt: zero bytes, TH = SENT_TH)
  uint4 stage[GPT];
  int stage_th = 0, stage_par = 0;
  auto load_chunk = [&](int ch) {
    const int64_t base = t_beg + (int64_t)ch * CHUNK;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const int g = tid + 256 * i;
      const int row = g / GPR, slot = g % GPR;
      int64_t tr = base + row;
      if (tr >= t_end) tr = nt;
      stage[i] = *(const uint4*)(tf + tr * DIM + slot * 16);
    }
    if (tid < CHUNK) {
      int64_t tr = base + tid;
      if (tr >= t_end) tr = nt;
      stage_th = th_g[tr]; stage_par = par_g[tr];
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const int g = tid + 256 * i;
      const int row = g / GPR, slot = g % GPR;
      s_tb[buf][row * GPR + (slot ^ ((row >> 1) & (GPR - 1)))] = stage[i];
    }
    if (tid < CHUNK) { s_thb[buf][tid] = stage_th; s_parb[buf][tid] = stage_par; }
  };

  // two independent (best, second) key pairs per query block - even / odd accumulator registers - halve the
  // serial min/med3 dependency chain; they are merged when the 256-row window is flushed
  int m1[QB][2], m2[QB][2];
  int thr[QB];                       // FILTER: a candidate matters only if its accumulator is below this
  int u2_seen[QB], u2_sent[QB];      // FILTER: what the other splits reported one window ago / what this one has reported
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    thr[qb] = (q0 + qb * 32 + l31 < nq) ? 0x7FFFFFFF : (int)0x80000000;
    u2_seen[qb] = u2_sent[qb] = 0x7FFFFFFF;
  }
  if (n_chunks > 0) { load_chunk(0); store_chunk(0); }
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int buf = ch & 1;
    const uint4* s_t = s_tb[buf];
    const int* s_th = s_thb[buf];
    const int* s_par = s_parb[buf];
    if (ch + 1 < n_chunks) load_chunk(ch + 1);
    if ((ch & 1) == 0) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) { m1[qb][0] = m1[qb][1] = 0x7FFFFFFF; m2[qb][0] = m2[qb][1] = 0x7FFFFFFF; }
    }
    const int wbase = ((ch & 1) << 7) | (4 * half);      // row inside the 256-row window of register 0 of tile 0
    // Software pipeline over the (tile, query block) steps of the chunk: the matrix unit works on step s while the vector
    // unit ranks the accumulators of step s - 1 in the gaps between its MFMAs, and the LDS operands of the next tile are
    // fetched a tile ahead into a second register set.
    constexpr int TILES = CHUNK / 32, STEPS = TILES * QB;
    // with four query blocks a tile is 16 MFMAs long: one operand set, refilled behind the tile's last chain (the other
    // wave of the SIMD covers the LDS round trip), leaves the registers to the query fragments
    constexpr int SETS = (QB * KS >= 16 || KNN_WAVES >= 3) ? 1 : 2;
    v4i at[SETS][KS];
    v16i th[SETS];       // accumulator start values of the 16 train rows a lane's registers hold: rows (r&3) + 8 (r>>2) + 4 half
    int4 kp[4];          // parity bits of the rows of the tile being ranked (one set: fetched right after the previous tile's last ranking)
    v16i acc[2];
    auto load_ops = [&](int tile, int set) {
      const int row = tile * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int slot = ks * 2 + half;
        const int sw = (slot ^ ((row >> 1) & (GPR - 1)));
        const uint4 v = s_t[row * GPR + sw];
        at[set][ks] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w};
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 kk = *(const int4*)&s_th[tile * 32 + 8 * g + 4 * half];
        th[set][4 * g] = kk.x; th[set][4 * g + 1] = kk.y; th[set][4 * g + 2] = kk.z; th[set][4 * g + 3] = kk.w;
      }
    };
    auto load_kp = [&](int tile) {
#pragma unroll
      for (int g = 0; g < 4; ++g) kp[g] = *(const int4*)&s_par[tile * 32 + 8 * g + 4 * half];
    };
    int gm[4];                                            // FILTER: smallest accumulator of each group of four rows
    auto rank_group_min = [&](const v16i& a, int g) {
      gm[g] = min(min(min(a[4 * g], a[4 * g + 1]), a[4 * g + 2]), a[4 * g + 3]);
    };
    auto rank_finish = [&](const v16i& a, int tile, int qb) {
      if (FILTER) {
        const int mn = min(min(min(gm[0], gm[1]), gm[2]), gm[3]);
        if (__builtin_amdgcn_ballot_w64(mn < thr[qb]) == 0ull) return;               // wave-uniform: nothing in this tile can enter a top-2
      }
      if (KNN_WAVES >= 3) load_kp(tile);                    // lean build: the parity bits only when a tile is ranked at all
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // past the first windows a tile that is not skipped has one or two candidates: only their groups are ranked
        if (FILTER && __builtin_amdgcn_ballot_w64(gm[g] < thr[qb]) == 0ull) continue;
        const int kpr[4] = {kp[g].x, kp[g].y, kp[g].z, kp[g].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          // ((2 acc + parity) << 8) | row in window = ((d^2 - QN) << 8) | row
          const int key = (a[r] << 9) + (kpr[j] + (wbase + tile * 32 + 8 * g + j));
          int nm2;
          asm("v_med3_i32 %0, %1, %2, %3" : "=v"(nm2) : "v"(m1[qb][r & 1]), "v"(m2[qb][r & 1]), "v"(key));
          m2[qb][r & 1] = nm2;
          m1[qb][r & 1] = min(m1[qb][r & 1], key);
        }
      }
    };
    load_ops(0, 0);
    if (KNN_WAVES < 3) load_kp(0);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int tile = st / QB, qb = st % QB, set = tile & (SETS - 1);
      const v16i& prev = acc[(st - 1) & 1];
      if (SETS == 2 && qb == 0 && tile + 1 < TILES) load_ops(tile + 1, set ^ 1);     // that set's last MFMA was issued a step ago
      acc[st & 1] = th[set];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        acc[st & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(at[set][ks], bq[qb][ks], acc[st & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (FILTER && st > 0) {                                          // KS gaps for the four group minima
          for (int g = ks * 4 / KS; g < (ks + 1) * 4 / KS; ++g) rank_group_min(prev, g);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (SETS == 1 && qb == QB - 1 && tile + 1 < TILES) load_ops(tile + 1, 0);      // behind the last chain that reads the set
      if (st > 0) rank_finish(prev, (st - 1) / QB, (st - 1) % QB);
      if (KNN_WAVES < 3 && qb == 0 && tile > 0) load_kp(tile);          // after the last ranking that read the previous tile's
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) rank_group_min(acc[(STEPS - 1) & 1], g);
    rank_finish(acc[(STEPS - 1) & 1], TILES - 1, QB - 1);
    // every second chunk (and at the end): unpack the window's best two and merge into the running pair
    // (later windows = higher indices, so strict '<' keeps the lower index on equal d^2)
    const int cbase = (int)(t_beg - t_seg + (int64_t)(ch & ~1) * CHUNK);      // train index inside the segment
    if ((ch & 1) == 1 || ch + 1 == n_chunks) {
    static_for<0, QB>([&](auto qbc) __attribute__((always_inline)) {
      constexpr int qb = decltype(qbc)::value;
      // keys carry the row index, so they are totally ordered: best two of the four
      const int lo = min(m1[qb][0], m1[qb][1]), hi = max(m1[qb][0], m1[qb][1]);
      const int second = min(hi, min(m2[qb][0], m2[qb][1]));
      static_for<0, 2>([&](auto sc) __attribute__((always_inline)) {
        const int m = decltype(sc)::value == 0 ? lo : second;
        const bool ok = (m >> 8) < 2 * SENT_TH;            // not a padding row or an empty slot
        const int d2 = (m >> 8) + qnv[qb];
        const int idx = cbase + (m & 0xFF);
        // selects, not branches: hipcc merges the stores of an if / else-if through a selected ADDRESS, which puts the
        // four running values in scratch memory - and every scratch access drains the chunk prefetch (vmcnt)
        const bool c1 = ok && d2 < g1k[qb], c2 = ok && !c1 && d2 < g2k[qb];
        g2k[qb] = c1 ? g1k[qb] : (c2 ? d2 : g2k[qb]);
        g2i[qb] = c1 ? g1i[qb] : (c2 ? idx : g2i[qb]);
        g1k[qb] = c1 ? d2 : g1k[qb];
        g1i[qb] = c1 ? idx : g1i[qb];
      });
      if (FILTER) {
        // second-best d^2 of the query over BOTH half-waves (they hold disjoint train rows of the same query) ...
        const int o1 = __shfl_xor(g1k[qb], 32, 64), o2 = __shfl_xor(g2k[qb], 32, 64);
        int u2 = min(min(g2k[qb], o2), max(g1k[qb], o1));
        // ... and over the other train splits of the same queries, which run side by side on other XCDs: every
        // split posts its bound with an atomic minimum and picks up the combined one a window later (the returned value
        // is not waited for here).  Any stale or missing value only leaves the threshold looser: a bound is the
        // second-best d^2 over SOME train rows, so no row of the final top-2 lies above it.
        u2 = min(u2, u2_seen[qb]);
        if (half == 0 && thr[qb] != (int)0x80000000 && u2 < u2_sent[qb]) {
          u2_seen[qb] = atomicMin(u2g + (q0 + qb * 32 + l31 + out_shift), u2);
          u2_sent[qb] = u2;
        }
        u2 = min(u2, __shfl_xor(u2, 32, 64));
        if (thr[qb] != (int)0x80000000) thr[qb] = (u2 == 0x7FFFFFFF) ? 0x7FFFFFFF : ((u2 - qnv[qb]) >> 1) + 1;
      }
    });
    }
    if (ch + 1 < n_chunks) store_chunk(buf ^ 1);
    __syncthreads();
  }
  // the two halves of the wave hold the same query with disjoint train rows: merge, lower index on ties
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int o1k = __shfl_xor(g1k[qb], 32, 64), o1i = __shfl_xor(g1i[qb], 32, 64);
    const int o2k = __shfl_xor(g2k[qb], 32, 64), o2i = __shfl_xor(g2i[qb], 32, 64);
    float b1d = 3.0e38f, b2d = 3.0e38f; int b1i = -1, b2i = -1;
    top2_insert((float)g1k[qb], g1i[qb], b1d, b1i, b2d, b2i);      // exact: d^2 < 2^24
    top2_insert((float)g2k[qb], g2i[qb], b1d, b1i, b2d, b2i);
    top2_insert((float)o1k, o1i, b1d, b1i, b2d, b2i);
    top2_insert((float)o2k, o2i, b1d, b1i, b2d, b2i);
    const int64_t qi = q0 + qb * 32 + l31;
    if (half == 0 && qi < nq) {
      Cand* o = part + ((int64_t)split * total_out + qi + out_shift) * 2;
      o[0].d = b1i >= 0 ? b1d : 0.0f; o[0].i = b1i;      // d = d^2 (an exact integer below 2^24); k_merge_splits_u8 takes the root
      o[1].d = b2i >= 0 ? b2d : 0.0f; o[1].i = b2i;
    }
  }
}

// ------------------------------------------------------------------------------------ L2 / uint8, operands straight from L2
// The large single-pair case (dim 128, tens of thousands of queries).  k_knn2_u8 above stages the train rows through
// LDS for the four waves of a workgroup and pays for it with a barrier and an exposed load per 128 rows (44 % of the
// wave time waiting at 2 waves per SIMD).  Here the pre-pass stores the train set in MFMA operand order - per
// 32-row tile and 32-byte k-slice the 64 lanes' 16-byte pieces back to back, 1 KiB per load instruction = 8 full
// lines - and every wave fetches its A operands with plain global loads one tile ahead: no LDS, no barrier, the waves
// of a workgroup are independent (they hit the same lines in the CU's L1).  TN_i travels with the tile in register
// order (16 values per half-wave); TH = TN >> 1 and the parity bit are taken from it in registers.
// Two widths: KS = 4 (dim 128: SIFT, 128-bit strings) and KS = 8 (256 unpacked bits: ORB, find_matches.py:144 - twice the
// MFMAs per step for the same filter and ranking work, two query blocks per wave).
// The same launch also does the two other things the distance kernel waits for - the query rows' norms (the workgroups past
// the last tile: k_row_norm_u8's arithmetic with flip 0x7F) and "no bound yet" in the shared bounds u2 (every thread a few words)
// - which were a launch and a fill of their own in front of every call: ~7 us of a 0.39-ms call at 50k x 50k.
template <int KS>   // KS = dim / 32: 4 (SIFT) or 8 (256 unpacked bits)
__global__ __launch_bounds__(64 * KS) void k_train_tile_u8(const uint8_t* __restrict__ x, int64_t n, uint8_t* __restrict__ xt,
                                                           int* __restrict__ th_t, int* __restrict__ pb_t, int* __restrict__ fix_cnt,
                                                           int n_tile_blocks, const uint8_t* __restrict__ q, int64_t nq, int* __restrict__ qn,
                                                           int* __restrict__ u2, int64_t n_u2) {
  constexpr int PPR = 2 * KS, DIM = 32 * KS;               // 16-byte pieces per row; a workgroup = 64 KS threads = the 32 rows of ONE tile
  {
    const int64_t total = (int64_t)gridDim.x * (64 * KS);
    for (int64_t i = (int64_t)blockIdx.x * (64 * KS) + threadIdx.x; i < n_u2; i += total) u2[i] = 0x7F7F7F7F;
  }
  if ((int)blockIdx.x >= n_tile_blocks) {
    // query norms: QN_j = sum (v + 1)^2 over the row, v = (byte ^ 0x7F) as int8; 2 KS lanes per row, adjacent and aligned
    const int64_t t = (int64_t)(blockIdx.x - n_tile_blocks) * (64 * KS) + threadIdx.x;
    const int64_t r = t / PPR;
    const int part = (int)(t - r * PPR);
    int s = 0;
    if (r < nq) {
      const uint4 v = *(const uint4*)(q + r * DIM + part * 16);
      const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int sh = 0; sh < 32; sh += 8) {
          const int b = (int)(int8_t)(((wds[qq] >> sh) & 0xFFu) ^ 0x7Fu) + 1;
          s += b * b;
        }
    }
#pragma unroll
    for (int o = PPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (r < nq && part == 0) qn[r] = s;
    return;
  }
  __shared__ int s_pb[2];
  if (threadIdx.x < 2) s_pb[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *fix_cnt = 0;    // the re-rank list of this call starts empty (filled by k_merge_splits_u8)
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * (64 * KS) + threadIdx.x;
  const int64_t r = t / PPR;                              // row (up to the end of the last 32-row tile)
  const int gi = (int)(t - r * PPR);                      // 16-byte piece of the row: k-slice gi >> 1, half gi & 1
  // the grid covers whole tiles (n rounded up to 32 rows): no thread leaves before the barrier below
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  int s = 0;
  if (r < n) {
    v = *(const uint4*)(x + r * DIM + gi * 16);
    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int sh = 0; sh < 32; sh += 8) {
        const int b = (int)(int8_t)((wds[q] >> sh) & 0xFFu) + 1;
        s += b * b;
      }
  }
  const int64_t tile = r >> 5;
  const int l31 = (int)(r & 31);
  *(uint4*)(xt + (((tile * KS + (gi >> 1)) * 64) + (gi & 1) * 32 + l31) * 16) = v;
#pragma unroll
  for (int o = 1; o < PPR; o <<= 1) s += __shfl_xor(s, o, 64);     // the 2 KS lanes of a row are adjacent and aligned
  const int tn = (r < n) ? (s - DIM) : 2 * SENT_TH;
  // accumulator register rr of half-wave hh holds row (rr & 3) + 8 (rr >> 2) + 4 hh of the tile
  const int hh = (l31 >> 2) & 1, rr = (l31 & 3) + 4 * (l31 >> 3);
  if (gi == 0) th_t[(tile * 2 + hh) * 16 + rr] = tn >> 1;
  // parity bits: bit rr of word (tile, hh), collected in LDS
  if (gi == 0 && (tn & 1)) atomicOr(&s_pb[hh], 1 << rr);
  __syncthreads();
  if (threadIdx.x < 2) pb_t[tile * 2 + threadIdx.x] = s_pb[threadIdx.x];
}

// train rows per workgroup, first : second on a CU (k_knn2_u8_direct).  Measured at 50k x 50k, distance kernel by HIP events:
// even 378 us, 128:100 370, 135:100 367-368, 150:100 367 (the wave end stamps then lie within 306-354 us instead of 283-375)
constexpr int MATCH_W_FIRST = 135, MATCH_W_SECOND = 100;
#ifndef SFM_MATCH_STAMPS
#define SFM_MATCH_STAMPS 0
#endif
#ifndef SFM_MATCH_TH_AHEAD
#define SFM_MATCH_TH_AHEAD 1
#endif
#if SFM_MATCH_STAMPS
__device__ unsigned long long g_match_stamps[4 * 4096 * 4];
extern "C" int sfm_debug_match_stamps(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_match_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#define MATCH_STAMP_BEGIN() const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime()
#define MATCH_STAMP_END() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) { \
    unsigned long long* o = g_match_stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; \
    o[0] = stamp_t0; o[1] = __builtin_amdgcn_s_memrealtime(); \
    o[2] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); o[3] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20); } } while (0)
#else
#define MATCH_STAMP_BEGIN() do {} while (0)
#define MATCH_STAMP_END() do {} while (0)
#endif
template <int QB, int KS = 4>   // KS = dim / 32: 4 (SIFT), 8 (256 unpacked bits: ORB)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((QB == 2 && KS == 4) ? 3 : 2, (QB == 2 && KS == 4) ? 3 : 2))) void k_knn2_u8_direct(
    const uint8_t* __restrict__ q, int64_t nq, const uint8_t* __restrict__ xt, int64_t nt, const int* __restrict__ th_t,
    const int* __restrict__ pb_t, const int* __restrict__ qn, int nsplit, int64_t rows_per_split, Cand* __restrict__ part, int* u2g,
    int w_first, int w_second) {
  constexpr int DIM = 32 * KS;
  static_assert(QB % 2 == 0, "the two accumulators alternate by step across tiles");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
  // Workgroup -> (split, query block), split-major over the workgroups of one XCD (round-robin placement: XCD = b % 8):
  // an XCD then streams one or two train splits (<= 2.6 MB at 50k rows) out of its own 4 MB L2 instead of all of them
  // out of the Infinity Cache.  Speed only - any mapping gives the same result.
  const int nb = gridDim.x, xcd = blockIdx.x & 7, per = nb >> 3, rem = nb & 7;
  const int id = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int n_qblocks = nb / nsplit;
  const int split = id / n_qblocks;
  const int64_t qblock = id - split * n_qblocks;
  int64_t t_beg = (int64_t)split * rows_per_split;                       // a multiple of 128
  int64_t t_end = (t_beg + rows_per_split) < nt ? (t_beg + rows_per_split) : nt;
  if (w_first > 0) {
    // Uneven splits (single-round grids, two workgroups per CU).  Measured with begin / end stamps per wave (tools/
    // exp_matcher_lifetimes.sh, 50k x 50k): all 490 workgroups start within 0.4 us, the workgroup that reached its CU FIRST
    // finishes after 300 us, the one that came second after 362 (vector issue between the two waves of a SIMD is arbitrated
    // by age: MI355X_MICROARCH.md, Two waves per SIMD, item 2), and for the last sixth of the launch every SIMD runs one wave.
    // Which of the two a workgroup is could be told from its place in its XCD's dispatch sequence in 458 of 458 cases:
    // (blockIdx >> 3) < 32 CUs <=> first.  So the train rows of a query block are cut in proportion to the rate its
    // workgroups are expected to run at (w_first : w_second), in units of 128 rows.  Speed only: any cut gives the same
    // neighbours (the merge is exact), and a placement other than the assumed one just leaves the old imbalance.
    const int64_t chunks = (nt + 127) >> 7;
    int cum = 0, tot = 0, mine = 0;
    for (int s2 = 0; s2 < nsplit; ++s2) {
      const int idu = s2 * n_qblocks + (int)qblock;
      int x = 0;
      while (x < 7 && (x + 1) * per + ((x + 1) < rem ? (x + 1) : rem) <= idu) ++x;
      const int j = idu - (x * per + (x < rem ? x : rem));
      const int wt = j < 32 ? w_first : w_second;
      if (s2 < split) cum += wt;
      if (s2 == split) mine = wt;
      tot += wt;
    }
    t_beg = (chunks * cum / tot) << 7;
    t_end = split == nsplit - 1 ? nt : ((chunks * (cum + mine) / tot) << 7);
    if (t_end > nt) t_end = nt;
  }
  const int64_t q0 = qblock * (4 * QB * 32) + (int64_t)w * (QB * 32);
  if (q0 >= nq) return;                                                  // waves are independent: no barrier below
  MATCH_STAMP_BEGIN();

  v4i bq[QB][KS];
  // What only the window flush touches (the running best two, the posted bound, QN) lives in LDS between flushes - the
  // kernel has no other use for LDS, and the registers pay for a second tile of operands in flight.
  enum { ST_G1K = 0, ST_G2K = QB, ST_G1I = 2 * QB, ST_G2I = 3 * QB, ST_SENT = 4 * QB, ST_QN = 5 * QB, ST_N = 6 * QB };
  __shared__ int s_st[ST_N][256];
  int m1[QB][2], m2[QB][2];
  int thr[QB];
  int u2_seen[QB], u2_next[QB];      // the other splits' bounds: in use / on their way
  // the bounds of a wave's QB x 32 queries are kept query-in-block major ([l31][qb]: this kernel is their only user), so
  // that a lane fetches its QB values with QB / 2 eight-byte loads; lanes past the last query read values nobody posts
  const int* u2_mine = u2g + q0 + l31 * QB;
  const __amdgpu_buffer_rsrc_t u2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)u2g, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int64_t qi = q0 + qb * 32 + l31;
    const bool ok = qi < nq;
    s_st[ST_QN + qb][tid] = ok ? qn[qi] : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      v4i v = {0, 0, 0, 0};
      if (ok) v = *(const v4i*)(q + qi * DIM + ks * 32 + half * 16);
      bq[qb][ks] = v ^ 0x7F7F7F7F;
    }
    s_st[ST_G1K + qb][tid] = 0x7FFFFFFF; s_st[ST_G2K + qb][tid] = 0x7FFFFFFF; s_st[ST_G1I + qb][tid] = -1; s_st[ST_G2I + qb][tid] = -1;
    s_st[ST_SENT + qb][tid] = 0x7FFFFFFF;
    thr[qb] = ok ? 0x7FFFFFFF : (int)0x80000000;
    u2_seen[qb] = u2_next[qb] = 0x7FFFFFFF;
    m1[qb][0] = m1[qb][1] = 0x7FFFFFFF; m2[qb][0] = m2[qb][1] = 0x7FFFFFFF;
  }
  const int64_t tile0 = t_beg >> 5;
  const int n_tiles = t_end > t_beg ? (int)((t_end - t_beg + 31) >> 5) : 0;

  v4i at[2][KS];
  int4 thq[2];          // one quarter of the accumulator start values TN >> 1 of the tile's rows (see load_tile)
  int pb[2];            // the rows' parity bits (bit r = TN & 1 of register r)
  // quad_perm [c,c,c,c]: every lane of a quad reads the value of the quad's lane c
#define QUAD_BCAST(x, c) __builtin_amdgcn_mov_dpp((x), (c) | ((c) << 2) | ((c) << 4) | ((c) << 6), 0xF, 0xF, true)
  auto spread_th = [&](const int4& qv) {
    v16i o;
    o[0] = QUAD_BCAST(qv.x, 0); o[1] = QUAD_BCAST(qv.y, 0); o[2] = QUAD_BCAST(qv.z, 0); o[3] = QUAD_BCAST(qv.w, 0);
    o[4] = QUAD_BCAST(qv.x, 1); o[5] = QUAD_BCAST(qv.y, 1); o[6] = QUAD_BCAST(qv.z, 1); o[7] = QUAD_BCAST(qv.w, 1);
    o[8] = QUAD_BCAST(qv.x, 2); o[9] = QUAD_BCAST(qv.y, 2); o[10] = QUAD_BCAST(qv.z, 2); o[11] = QUAD_BCAST(qv.w, 2);
    o[12] = QUAD_BCAST(qv.x, 3); o[13] = QUAD_BCAST(qv.y, 3); o[14] = QUAD_BCAST(qv.z, 3); o[15] = QUAD_BCAST(qv.w, 3);
    return o;
  };
  // the same, four registers at a time (piece g = registers 4 g .. 4 g + 3 = the quarter of quad lane g)
  auto spread_th_piece = [&](v16i& o, const int4& qv, int g) {
    switch (g) {
      case 0: o[0] = QUAD_BCAST(qv.x, 0); o[1] = QUAD_BCAST(qv.y, 0); o[2] = QUAD_BCAST(qv.z, 0); o[3] = QUAD_BCAST(qv.w, 0); break;
      case 1: o[4] = QUAD_BCAST(qv.x, 1); o[5] = QUAD_BCAST(qv.y, 1); o[6] = QUAD_BCAST(qv.z, 1); o[7] = QUAD_BCAST(qv.w, 1); break;
      case 2: o[8] = QUAD_BCAST(qv.x, 2); o[9] = QUAD_BCAST(qv.y, 2); o[10] = QUAD_BCAST(qv.z, 2); o[11] = QUAD_BCAST(qv.w, 2); break;
      default: o[12] = QUAD_BCAST(qv.x, 3); o[13] = QUAD_BCAST(qv.y, 3); o[14] = QUAD_BCAST(qv.z, 3); o[15] = QUAD_BCAST(qv.w, 3); break;
    }
  };
#undef QUAD_BCAST
  auto load_tile = [&](int64_t tile, int set) {
#if SFM_MATCH_TH_AHEAD
    // (first: it is used one step before the operands - see th_carry - and loads return in order)
    thq[set] = *(const int4*)(th_t + (tile * 2 + half) * 16 + 4 * (lane & 3));
#endif
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) at[set][ks] = *(const v4i*)(xt + ((tile * KS + ks) * 64 + lane) * 16);
    // the 16 start values of a half-wave: every lane fetches ONE quarter (the quarter its position in its quad names), the
    // quads then pass the quarters round with DPP moves (spread_th).  Four 16-byte loads per lane instead of one would give
    // every lane all 16 directly - and cost the CU's single texture-address path 64 cycles per wave and tile instead of 16;
    // with 13 vector loads per tile that path, not the matrix unit, bounded the kernel (8 waves x 13 x 16 cycles against
    // 1,024 cycles of MFMAs per tile and SIMD).
#if !SFM_MATCH_TH_AHEAD
    thq[set] = *(const int4*)(th_t + (tile * 2 + half) * 16 + 4 * (lane & 3));
#endif
    pb[set] = pb_t[tile * 2 + half];
  };
  // The ranking of a step's accumulators (see k_knn2_u8) is issued in the gaps between the MFMAs of the NEXT step, across
  // tile boundaries too: what it needs from its own tile are the parity word and the window position, one register each.
  int gm[4];
  auto rank_group_min = [&](const v16i& a, int g) {
    gm[g] = min(min(min(a[4 * g], a[4 * g + 1]), a[4 * g + 2]), a[4 * g + 3]);
  };
  auto rank_finish = [&](const v16i& a, int qb, int pbits, int wbase) {
    const int mn = min(min(min(gm[0], gm[1]), gm[2]), gm[3]);
    if (__builtin_amdgcn_ballot_w64(mn < thr[qb]) == 0ull) return;                 // wave-uniform: nothing in this tile can enter a top-2
    int wb = wbase;
    asm volatile("" : "+v"(wb));            // keeps the 16 row numbers from being formed ahead of time for every tile
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (__builtin_amdgcn_ballot_w64(gm[g] < thr[qb]) == 0ull) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j;
        // ((2 acc + parity) << 8) | row in window = ((d^2 - QN) << 8) | row
        const int key = (a[r] << 9) + ((((pbits >> r) & 1) << 8) + (wb + 8 * g + j));
        // both updates IN PLACE (tied operands): with a separate result register the compiler kept every query block's four
        // values in two homes and copied them from one to the other after every step, ranked or not (4 of ~16 vector
        // instructions of a step nothing was ranked in)
        asm("v_med3_i32 %0, %1, %0, %2" : "+v"(m2[qb][r & 1]) : "v"(m1[qb][r & 1]), "v"(key));
        asm("v_min_i32 %0, %0, %1" : "+v"(m1[qb][r & 1]) : "v"(key));
      }
    }
  };
  // merge the finished 256-row window whose first tile is tw into the running best two, post / pick up the bound
  auto flush = [&](int tw) __attribute__((always_inline)) {
    const int cbase = (int)(t_beg + (int64_t)tw * 32);                   // train index of the window's first row
    int g1k[QB], g2k[QB], g1i[QB], g2i[QB], qnq[QB], sent[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {                                    // all LDS reads first: one round trip, not one per value
      g1k[qb] = s_st[ST_G1K + qb][tid]; g2k[qb] = s_st[ST_G2K + qb][tid]; g1i[qb] = s_st[ST_G1I + qb][tid];
      g2i[qb] = s_st[ST_G2I + qb][tid]; qnq[qb] = s_st[ST_QN + qb][tid]; sent[qb] = s_st[ST_SENT + qb][tid];
    }
    int u2[QB];
    static_for<0, QB>([&](auto qbc) __attribute__((always_inline)) {
      constexpr int qb = decltype(qbc)::value;
      // keys carry the row index, so they are totally ordered: best two of the four
      const int lo = min(m1[qb][0], m1[qb][1]), hi = max(m1[qb][0], m1[qb][1]);
      const int second = min(hi, min(m2[qb][0], m2[qb][1]));
      static_for<0, 2>([&](auto sc) __attribute__((always_inline)) {
        const int m = decltype(sc)::value == 0 ? lo : second;
        const bool ok = (m >> 8) < 2 * SENT_TH;            // not a padding row or an empty slot
        const int d2 = (m >> 8) + qnq[qb];
        const int idx = cbase + (m & 0xFF);
        // selects, not branches (see k_knn2_u8)
        const bool c1 = ok && d2 < g1k[qb], c2 = ok && !c1 && d2 < g2k[qb];
        g2k[qb] = c1 ? g1k[qb] : (c2 ? d2 : g2k[qb]);
        g2i[qb] = c1 ? g1i[qb] : (c2 ? idx : g2i[qb]);
        g1k[qb] = c1 ? d2 : g1k[qb];
        g1i[qb] = c1 ? idx : g1i[qb];
      });
      m1[qb][0] = m1[qb][1] = 0x7FFFFFFF; m2[qb][0] = m2[qb][1] = 0x7FFFFFFF;
    });
    int o1[QB], o2[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { o1[qb] = __shfl_xor(g1k[qb], 32, 64); o2[qb] = __shfl_xor(g2k[qb], 32, 64); }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      s_st[ST_G1K + qb][tid] = g1k[qb]; s_st[ST_G2K + qb][tid] = g2k[qb]; s_st[ST_G1I + qb][tid] = g1i[qb]; s_st[ST_G2I + qb][tid] = g2i[qb];
      // second-best d^2 of the query over both half-waves and over the other train splits (see k_knn2_u8)
      u2[qb] = min(min(min(g2k[qb], o2[qb]), max(g1k[qb], o1[qb])), u2_seen[qb]);
      if (half == 0 && thr[qb] != (int)0x80000000 && u2[qb] < sent[qb]) {
        // posted with an atomic the compiler does not see (nothing waits for it), picked up by the loads at the top of
        // every tile: a returning atomic here, on one path only, makes the compiler wait with vmcnt(0) in every tile
        asm volatile("global_atomic_smin %0, %1, off" ::"v"(u2_mine + qb), "v"(u2[qb]) : "memory");
        s_st[ST_SENT + qb][tid] = u2[qb];
      }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int u = min(u2[qb], __shfl_xor(u2[qb], 32, 64));
      if (thr[qb] != (int)0x80000000) thr[qb] = (u == 0x7FFFFFFF) ? 0x7FFFFFFF : ((u - qnq[qb]) >> 1) + 1;
    }
  };
  v16i acc[2];
#if SFM_MATCH_TH_AHEAD
  v16i th_carry;
#endif
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[1][r] = SENT_TH;         // "previous step" of the first one: padding rows, ranked and ignored
  pb[1] = 0;
  auto do_tile = [&](int t, int set) __attribute__((always_inline)) {
    // one tile ahead (two measured the same), into the set the previous tile has just finished with.  Always issued (the
    // last tile fetches itself again): with the loads under a condition the compiler has to pick ONE s_waitcnt count for
    // both paths and picks one that also waits for some of the loads just issued
    const int pset = set ^ 1;
    const int pb_prev = pb[pset];                          // the previous tile's parity word, before its set is refilled
    // the bounds the other splits have posted: fetched in EVERY tile (same number of loads on every path, see below), past
    // the XCD's own L2 (device-scope load), used from the next tile on
    if (QB == 4) {
      // the lane's four bounds are 16 contiguous, 16-byte aligned bytes: ONE device-scope load (a buffer load with sc1, which the
      // compiler counts like the others) instead of two 8-byte ones
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) u2_seen[qb] = u2_next[qb];
      const v4i v = __builtin_amdgcn_raw_buffer_load_b128(u2_rsrc, (int)((q0 + (int64_t)l31 * QB) * 4), 0, 16);
      u2_next[0] = v[0]; u2_next[1] = v[1]; u2_next[2 % QB] = v[2]; u2_next[3 % QB] = v[3];
    } else {
#pragma unroll
    for (int qb = 0; qb < QB; qb += 2) {
      u2_seen[qb] = u2_next[qb]; u2_seen[qb + 1] = u2_next[qb + 1];
      const unsigned long long v = __hip_atomic_load((const unsigned long long*)(u2_mine + qb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      u2_next[qb] = (int)(unsigned)v; u2_next[qb + 1] = (int)(unsigned)(v >> 32);
    }
    }
    load_tile(tile0 + (t + 1 < n_tiles ? t + 1 : t), pset);
    const int wb_prev = (((t - 1) & 7) << 5) | (4 * half), wb_cur = ((t & 7) << 5) | (4 * half);
#if SFM_MATCH_TH_AHEAD
    // The 16 DPP moves that spread a tile's start values used to stand between the last MFMA of one tile and the first of the
    // next (the first MFMA takes them as its C operand).  They now run in the MFMA gaps of the PREVIOUS tile's last step, four
    // per gap, into th_carry - whose registers are free by then (the last step has copied its start values already).
    const v16i th = th_carry;
#else
    const v16i th = spread_th(thq[set]);
#endif
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const v16i& prev = acc[(qb - 1) & 1];                // the very first step ranks the padding values acc[1] starts with
      acc[qb & 1] = th;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        acc[qb & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(at[set][ks], bq[qb][ks], acc[qb & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks < 4) rank_group_min(prev, ks);              // the four group minima of the previous step: one per MFMA gap
#if SFM_MATCH_TH_AHEAD
        if (qb == QB - 1 && ks < 4) spread_th_piece(th_carry, thq[pset], ks);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      if (qb > 0) rank_finish(prev, qb - 1, pb[set], wb_cur);
      else {
        rank_finish(prev, QB - 1, pb_prev, wb_prev);
        if ((t & 7) == 0 && t > 0) flush(t - 8);           // the window that ended with the previous tile
      }
    }
  };
  if (n_tiles > 0) load_tile(tile0, 0);
#if SFM_MATCH_TH_AHEAD
  th_carry = spread_th(thq[0]);
#endif
  // pairs of tiles in a loop with ONE exit, an odd last tile after it: with `if (t + 1 >= n_tiles) break;` between the two
  // halves of the body the compiler kept the running keys of every query block in two register homes, one per exit, and
  // copied them from one to the other after every step (4 of the ~16 vector instructions of a step nothing is ranked in)
  for (int t = 0; t + 1 < n_tiles; t += 2) {
    do_tile(t, 0);
    do_tile(t + 1, 1);
  }
  if (n_tiles & 1) do_tile(n_tiles - 1, 0);
  if (n_tiles > 0) {                                       // the last step's ranking and the last window
    const int t = n_tiles - 1, set = t & 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) rank_group_min(acc[(QB - 1) & 1], g);
    rank_finish(acc[(QB - 1) & 1], QB - 1, pb[set], ((t & 7) << 5) | (4 * half));
    flush(t & ~7);
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int g1k = s_st[ST_G1K + qb][tid], g2k = s_st[ST_G2K + qb][tid], g1i = s_st[ST_G1I + qb][tid], g2i = s_st[ST_G2I + qb][tid];
    const int o1k = __shfl_xor(g1k, 32, 64), o1i = __shfl_xor(g1i, 32, 64);
    const int o2k = __shfl_xor(g2k, 32, 64), o2i = __shfl_xor(g2i, 32, 64);
    float b1d = 3.0e38f, b2d = 3.0e38f; int b1i = -1, b2i = -1;
    top2_insert((float)g1k, g1i, b1d, b1i, b2d, b2i);      // exact: d^2 < 2^24
    top2_insert((float)g2k, g2i, b1d, b1i, b2d, b2i);
    top2_insert((float)o1k, o1i, b1d, b1i, b2d, b2i);
    top2_insert((float)o2k, o2i, b1d, b1i, b2d, b2i);
    const int64_t qi = q0 + qb * 32 + l31;
    if (half == 0 && qi < nq) {
      Cand* o = part + ((int64_t)split * nq + qi) * 2;
      o[0].d = b1i >= 0 ? b1d : 0.0f; o[0].i = b1i;        // d = d^2 (an exact integer below 2^24); k_merge_splits_u8 takes the root
      o[1].d = b2i >= 0 ? b2d : 0.0f; o[1].i = b2i;
    }
  }
  MATCH_STAMP_END();
}

// ------------------------------------------------------------------------------------ generic VALU kernel
// METRIC 1: float32, d^2 accumulated over k ascending with separately rounded multiply and add (the
// oracle's stated order), distance = sqrtf(d^2).  METRIC 2: Hamming popcount over dim bytes (dim % 4 == 0).
// One thread = one query (held in registers), train rows broadcast from LDS, 64-row tiles.
template <int METRIC, int DIMW>   // DIMW = 32-bit words per row
__global__ __launch_bounds__(256) void k_knn2_valu(const uint32_t* __restrict__ q, int64_t nq,
                                                   const uint32_t* __restrict__ t, int64_t nt, int nsplit,
                                                   int64_t rows_per_split, const MatchWG* __restrict__ wg,
                                                   int64_t total_out, Cand* __restrict__ part) {
  constexpr int TILE = 32;
  __shared__ uint32_t s_t[TILE * DIMW];
  const int tid = threadIdx.x;
  int split;
  int64_t qi, t_beg, t_end, t_seg = 0, out_shift = 0;
  if (wg) {
    const MatchWG r = wg[blockIdx.x];
    split = r.split; t_beg = r.t_first; t_end = r.t_end; t_seg = r.t_seg;
    qi = r.q_first + tid;
    nq = r.q_end;
    out_shift = r.out_first - r.q_first;
  } else {
    split = blockIdx.x % nsplit;
    qi = (int64_t)(blockIdx.x / nsplit) * 256 + tid;
    t_beg = (int64_t)split * rows_per_split;
    t_end = (t_beg + rows_per_split) < nt ? (t_beg + rows_per_split) : nt;
    total_out = nq;
  }
  uint32_t qr[DIMW];
#pragma unroll
  for (int k = 0; k < DIMW; ++k) qr[k] = (qi < nq) ? q[qi * DIMW + k] : 0u;
  float b1d = 3.0e38f, b2d = 3.0e38f; int b1i = -1, b2i = -1;
  for (int64_t base = t_beg; base < t_end; base += TILE) {
    const int cnt = (int)((t_end - base) < TILE ? (t_end - base) : TILE);
    __syncthreads();
    for (int i = tid; i < cnt * DIMW; i += 256) s_t[i] = t[base * DIMW + i];
    __syncthreads();
    for (int r = 0; r < cnt; ++r) {
      float dist;
      if (METRIC == 1) {
        // hipcc defaults to -ffp-contract=fast and would fuse the multiply and the add into one FMA;
        // the stated order (oracle/matcher_oracle.py: sq_l2) rounds them separately
#pragma clang fp contract(off)
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < DIMW; ++k) {
          const float diff = __uint_as_float(qr[k]) - __uint_as_float(s_t[r * DIMW + k]);
          const float prod = diff * diff;      // plain operators: the pragma is lexical
          acc = acc + prod;
        }
        dist = sqrt_rn_f32(acc);
      } else {
        int pc = 0;
#pragma unroll
        for (int k = 0; k < DIMW; ++k) pc += __popc(qr[k] ^ s_t[r * DIMW + k]);
        dist = (float)pc;
      }
      top2_insert(dist, (int)(base - t_seg + r), b1d, b1i, b2d, b2i);
    }
  }
  if (qi < nq) {
    Cand* o = part + ((int64_t)split * total_out + qi + out_shift) * 2;
    o[0].d = b1i >= 0 ? b1d : 0.0f; o[0].i = b1i;
    o[1].d = b2i >= 0 ? b2d : 0.0f; o[1].i = b2i;
  }
}

// merge the per-split candidates of each query: order by (distance, index)
__global__ __launch_bounds__(256) void k_merge_splits(int64_t nq, int nsplit, const Cand* __restrict__ part,
                                                      int* __restrict__ idx1, int* __restrict__ idx2,
                                                      float* __restrict__ d1, float* __restrict__ d2) {
  const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (qi >= nq) return;
  float b1d = 3.0e38f, b2d = 3.0e38f; int b1i = -1, b2i = -1;
  for (int s = 0; s < nsplit; ++s) {
    const Cand* c = part + ((int64_t)s * nq + qi) * 2;
    top2_insert(c[0].d, c[0].i, b1d, b1i, b2d, b2i);
    top2_insert(c[1].d, c[1].i, b1d, b1i, b2d, b2i);
  }
  idx1[qi] = b1i; idx2[qi] = b2i; d1[qi] = b1d; d2[qi] = b2d;
}

// Hamming on the int8 matrix path: over bits unpacked to bytes b in {127, 128} the uint8 L2 d^2 = sum (b_q - b_t)^2 IS the
// number of differing bits, with the same tie rule (lowest train index) - so ORB descriptors of 128 / 256 bits run through
// k_knn2_u8 / k_knn2_u8_direct as "dim 128 / 256 uint8 rows", and the merge reports d^2 itself (take_root = 0) where L2 takes
// the root.  One thread per input byte -> 8 output bytes.
__global__ __launch_bounds__(256) void k_unpack_bits(const uint8_t* __restrict__ x, int64_t n_bytes, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_bytes) return;
  const uint32_t v = x[i];
  uint2 o;
  o.x = 0x7F7F7F7Fu + ((v & 1u) | ((v & 2u) << 7) | ((v & 4u) << 14) | ((v & 8u) << 21));
  o.y = 0x7F7F7F7Fu + (((v >> 4) & 1u) | ((v & 32u) << 3) | ((v & 64u) << 10) | ((v & 128u) << 17));
  *(uint2*)(out + i * 8) = o;
}

// L2 / uint8: the per-split candidates carry the exact integer d^2 (as float32).  Merge by (d^2, index), take the
// correctly rounded float32 root, and list the queries whose second neighbour lies where sqrtf stops being
// injective (d^2 >= 2^22): those are re-ranked on the float32 value by k_knn2_u8_rerank.
#define D2_SQRT_INJECTIVE_BELOW 4194304.0f
__global__ __launch_bounds__(256) void k_merge_splits_u8(int64_t nq, int nsplit, const Cand* __restrict__ part,
                                                         int* __restrict__ idx1, int* __restrict__ idx2,
                                                         float* __restrict__ d1, float* __restrict__ d2,
                                                         int* __restrict__ fix_cnt, int* __restrict__ fix_list, int take_root) {
  const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (qi >= nq) return;
  float b1d = 3.0e38f, b2d = 3.0e38f; int b1i = -1, b2i = -1;
  for (int s = 0; s < nsplit; ++s) {
    const Cand* c = part + ((int64_t)s * nq + qi) * 2;
    top2_insert(c[0].d, c[0].i, b1d, b1i, b2d, b2i);
    top2_insert(c[1].d, c[1].i, b1d, b1i, b2d, b2i);
  }
  idx1[qi] = b1i; idx2[qi] = b2i;
  d1[qi] = take_root ? sqrt_rn_f32(b1d) : b1d; d2[qi] = take_root ? sqrt_rn_f32(b2d) : b2d;
  if (take_root && b2d >= D2_SQRT_INJECTIVE_BELOW) fix_list[atomicAdd(fix_cnt, 1)] = (int)qi;     // list order is irrelevant: one writer per query
}

// One workgroup per listed query (grid-stride over the list, whose length is only known on the device): every
// train row's exact integer d^2, its float32 root, top-2 by (root, index) - the rule of the oracle - over the
// whole train set.  Rare by construction (see the header), so it is written for clarity, not speed.
__global__ __launch_bounds__(256) void k_knn2_u8_rerank(const uint8_t* __restrict__ q, const uint8_t* __restrict__ t,
                                                        int64_t nt, int dim, MatchSegs segs, const int* __restrict__ fix_cnt,
                                                        const int* __restrict__ fix_list, int* __restrict__ idx1,
                                                        int* __restrict__ idx2, float* __restrict__ d1,
                                                        float* __restrict__ d2) {
  __shared__ uint32_t s_q[32];
  __shared__ Cand s_c[256][2];
  const int tid = threadIdx.x;
  const int n_fix = *fix_cnt;
  const int words = dim >> 2;
  for (int f = blockIdx.x; f < n_fix; f += gridDim.x) {
    const int qi = fix_list[f];                        // output row
    int64_t q_row = qi, t0 = 0, t1 = nt;
    if (segs.n_seg > 0) {                              // segment of this output row: last s with out_ptr[s] <= qi
      int lo = 0, hi = segs.n_seg - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs.out_ptr[mid] <= qi) lo = mid; else hi = mid - 1; }
      q_row = segs.q_beg[lo] + (qi - segs.out_ptr[lo]);
      t0 = segs.t_beg[lo]; t1 = segs.t_end[lo];
    }
    __syncthreads();
    if (tid < words) s_q[tid] = ((const uint32_t*)(q + q_row * dim))[tid];
    __syncthreads();
    float b1d = 3.0e38f, b2d = 3.0e38f; int b1i = -1, b2i = -1;
    for (int64_t r = tid; r < t1 - t0; r += 256) {
      const uint32_t* tr = (const uint32_t*)(t + (t0 + r) * dim);
      int d2i = 0;
      for (int k = 0; k < words; ++k) {
        const uint32_t a = s_q[k], b = tr[k];
#pragma unroll
        for (int sft = 0; sft < 32; sft += 8) {
          const int df = (int)((a >> sft) & 0xFFu) - (int)((b >> sft) & 0xFFu);
          d2i += df * df;
        }
      }
      top2_insert(sqrt_rn_f32((float)d2i), (int)r, b1d, b1i, b2d, b2i);
    }
    s_c[tid][0].d = b1d; s_c[tid][0].i = b1i; s_c[tid][1].d = b2d; s_c[tid][1].i = b2i;
    __syncthreads();
    if (tid == 0) {
      float m1d = 3.0e38f, m2d = 3.0e38f; int m1i = -1, m2i = -1;
      for (int u = 0; u < 256; ++u) {
        top2_insert(s_c[u][0].d, s_c[u][0].i, m1d, m1i, m2d, m2i);
        top2_insert(s_c[u][1].d, s_c[u][1].i, m1d, m1i, m2d, m2i);
      }
      idx1[qi] = m1i; idx2[qi] = m2i; d1[qi] = m1d; d2[qi] = m2d;
    }
  }
}

// ------------------------------------------------------------------------------------ ratio test + compaction
__global__ __launch_bounds__(256) void k_ratio_count(int64_t nq, const float* __restrict__ d1,
                                                     const float* __restrict__ d2, double ratio,
                                                     int* __restrict__ blk_cnt) {
  __shared__ int s_c[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool keep = (i < nq) && ((double)d1[i] < ratio * (double)d2[i]);
  const unsigned long long b = __ballot(keep);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
// exclusive scan of the per-block match counts: ONE workgroup, every thread a contiguous run of blocks (a single thread
// walking all of them took 42-67 us of dependent loads per call - more than any other kernel of a batched call but the
// distances)
__global__ __launch_bounds__(256) void k_ratio_scan(int nblk, const int* __restrict__ blk_cnt, int* __restrict__ blk_off,
                                                    int64_t* __restrict__ total) {
  __shared__ int64_t s_sum[256];
  const int tid = threadIdx.x;
  const int per = (nblk + 255) / 256;
  const int beg = tid * per, end = (beg + per) < nblk ? (beg + per) : nblk;
  int64_t mine = 0;
  for (int i = beg; i < end; ++i) mine += blk_cnt[i];
  s_sum[tid] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {                       // inclusive Hillis-Steele scan of the 256 run sums
    const int64_t v = tid >= o ? s_sum[tid - o] : 0;
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  int64_t run = s_sum[tid] - mine;
  for (int i = beg; i < end; ++i) { blk_off[i] = (int)run; run += blk_cnt[i]; }
  if (tid == 255) *total = s_sum[255];
}
__global__ __launch_bounds__(256) void k_ratio_scatter(int64_t nq, const int* __restrict__ idx1,
                                                       const float* __restrict__ d1, const float* __restrict__ d2,
                                                       double ratio, const int* __restrict__ blk_off,
                                                       int* __restrict__ query_idx, int* __restrict__ train_idx,
                                                       float* __restrict__ dist) {
  __shared__ int s_c[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool keep = (i < nq) && ((double)d1[i] < ratio * (double)d2[i]);
  const unsigned long long b = __ballot(keep);
  if (lane == 0) s_c[w] = __popcll(b);
  __syncthreads();
  int off = blk_off[blockIdx.x];
  for (int k = 0; k < w; ++k) off += s_c[k];
  off += __popcll(b & ((1ull << lane) - 1ull));
  if (keep) { query_idx[off] = (int)i; train_idx[off] = idx1[i]; dist[off] = d1[i]; }
}

__global__ __launch_bounds__(256) void k_f32_to_u8(const float* __restrict__ src, int64_t n, uint8_t* __restrict__ dst,
                                                   int* __restrict__ all_integral) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = src[i];
  const bool ok = (v >= 0.0f) && (v <= 255.0f) && (v == rintf(v));
  dst[i] = ok ? (uint8_t)v : 0;
  if (!ok) *all_integral = 0;
}

// ------------------------------------------------------------------------------------ host

// workspace carve (bytes): per-split candidates | train norms | query norms | ratio scratch | re-rank list + counter |
// (batched only) workgroup records + segment table
struct MatchWs {
  Cand* part; uint8_t* tf; uint8_t* qbits; uint8_t* tbits; int* th; int* par; int* qn; int* u2; int* fix_list; int* fix_cnt; char* plan; int64_t total;
};
static MatchWs match_ws_carve(char* ws, int64_t n_out, int64_t nt_rows, int64_t nq_rows, int64_t plan_bytes, bool bits = false) {
  MatchWs w;
  int64_t off = 0;
  w.part = (Cand*)(ws + off); off += align_up(8 * n_out * 2 * (int64_t)sizeof(Cand), 256);   // nsplit <= 8
  w.tf = (uint8_t*)(ws + off); off += align_up((nt_rows + 32) * (bits ? 256 : 128), 256);       // int8 copy of the train rows + padding (dim <= 128; 256 for unpacked bits)
  w.qbits = w.tbits = nullptr;
  if (bits) {                                                                                  // Hamming: rows unpacked to one byte per bit (<= 256 bits)
    w.qbits = (uint8_t*)(ws + off); off += align_up(nq_rows * 256, 256);
    w.tbits = (uint8_t*)(ws + off); off += align_up(nt_rows * 256, 256);
  }
  w.th = (int*)(ws + off); off += align_up((nt_rows + 32) * 4, 256);
  w.par = (int*)(ws + off); off += align_up((nt_rows + 1) * 4, 256);
  w.qn = (int*)(ws + off); off += align_up(nq_rows * 4, 256);
  w.u2 = (int*)(ws + off); off += align_up((n_out + 128) * 4, 256);                             // shared candidate thresholds (whole 128-query wave blocks)
  off += align_up(((n_out + 255) / 256) * 8 + 64, 256);                                        // sfm_match_ratio's scratch may alias here
  w.fix_list = (int*)(ws + off); off += align_up(n_out * 4, 256);
  w.fix_cnt = (int*)(ws + off); off += 256;
  w.plan = ws + off; off += align_up(plan_bytes, 256);
  w.total = off + 1024;
  return w;
}

extern "C" int sfm_match_workspace_bytes(int metric, int64_t nq, int64_t nt, int dim, int64_t* bytes) {
  if (!bytes || nq < 0 || nt < 0 || dim <= 0) return SFM_ERR_ARG;
  (void)metric;
  *bytes = match_ws_carve(nullptr, nq, nt, nq, 0, metric == SFM_METRIC_HAMMING && dim <= 32).total;
  return SFM_OK;
}

// The distance + top-2 stage for one segment (wg == nullptr) or a planned batch.  n_out = output rows.
static int match_launch(sfm_ctx* h, int metric, const void* q, int64_t nq_rows, const void* t, int64_t nt_rows, int dim,
                        int nsplit, int64_t rps, int64_t qpw /* queries per workgroup */, int64_t filter_rows /* train rows of the longest segment */,
                        const MatchWG* wg, unsigned grid, int64_t n_out, MatchSegs segs,
                        const MatchWs& w, int32_t* idx1, int32_t* idx2, float* d1, float* d2, bool take_root = true) {
  const uint8_t* q8 = (const uint8_t*)q; const uint8_t* t8 = (const uint8_t*)t;
  const uint32_t* qq = (const uint32_t*)q; const uint32_t* tt = (const uint32_t*)t;
  if (wg)   // segments choose their own number of splits: slots a segment does not use must read as "empty" (i = -1)
    SFM_HIP(h, hipMemsetAsync(w.part, 0xFF, (size_t)8 * n_out * 2 * sizeof(Cand), h->stream));
  if (metric == SFM_METRIC_L2_U8) {
    const bool qb4 = qpw == 512;
    const char* d_env = getenv("SFM_MATCH_DIRECT");        // test / tuning knob: "0" = the LDS kernel also for the large case
    // one pair at dim 128: the LDS-free kernel, with 4 query blocks per wave from 28,672 queries on and 2 below (8-20 % faster
    // than the LDS kernel at 2,000 .. 11,000 queries against 4,000 .. 50,000 train rows); the LDS kernel serves the batched
    // form, the smaller dims and the smallest pairs
    // dim 256 (ORB over unpacked bits; SFM_MATCH_DIRECT256=0 keeps the LDS kernel): two query blocks per wave, eight MFMAs per step
    const char* d256_env = getenv("SFM_MATCH_DIRECT256");
    const bool direct256 = !wg && dim == 256 && !qb4 && (double)nq_rows * (double)nt_rows >= 8e6 && !(d256_env && d256_env[0] == '0');
    const bool direct2 = !wg && (dim == 128 || direct256) && !qb4 && (double)nq_rows * (double)nt_rows >= 8e6;      // below: launch-bound, the LDS kernel's lighter pre-pass wins by ~3 us
    const bool direct = (qb4 || direct2) && !(d_env && d_env[0] == '0');
    if (direct)
      {
      // one pre-pass launch: train tiles, then the query norms, and the bounds' initial value on the side
      const unsigned tb = (unsigned)((nt_rows + 31) >> 5);
      const int64_t n_u2 = n_out + 128;
      if (dim == 256) hipLaunchKernelGGL(k_train_tile_u8<8>, dim3(tb + cdiv(nq_rows * 16, 512)), dim3(512), 0, h->stream, t8, nt_rows, w.tf, w.th, w.par, w.fix_cnt,
                                         (int)tb, q8, nq_rows, w.qn, w.u2, n_u2);
      else hipLaunchKernelGGL(k_train_tile_u8<4>, dim3(tb + cdiv(nq_rows * 8, 256)), dim3(256), 0, h->stream, t8, nt_rows, w.tf, w.th, w.par, w.fix_cnt,
                              (int)tb, q8, nq_rows, w.qn, w.u2, n_u2);
    }
    else {
      hipLaunchKernelGGL(k_train_prep_u8, dim3(cdiv((nt_rows + 1) * (dim >> 4), 256)), dim3(256), 0, h->stream, t8, nt_rows, dim, w.tf, w.th, w.par, w.fix_cnt);
      hipLaunchKernelGGL(k_row_norm_u8, dim3(cdiv(nq_rows * (dim >> 4), 256)), dim3(256), 0, h->stream, q8, nq_rows, dim, 0x7F, 0, w.qn);
    }
    // the candidate filter pays once the queries see a few thousand train rows (see k_knn2_u8); below that it is 10
    // operations per tile for nothing
    const char* f_env = getenv("SFM_MATCH_FILTER");        // test / tuning knob: "0" off, "1" on
    const bool filter = f_env ? f_env[0] == '1' : filter_rows >= 2048;
    if (filter && !direct) SFM_HIP(h, hipMemsetAsync(w.u2, 0x7F, (size_t)(n_out + 128) * sizeof(int), h->stream));      // "no bound yet" (the direct path's pre-pass writes it)
    sfm_prof_begin(h, SFM_PROF_KNN);
#define KNN_LAUNCH(KS, QB, F) hipLaunchKernelGGL((k_knn2_u8<KS, QB, F>), dim3(grid), dim3(256), 0, h->stream, q8, nq_rows, w.tf, nt_rows, w.th, w.par, w.qn, nsplit, rps, wg, n_out, w.part, w.u2)
    if (direct) {
      // uneven train splits for the two workgroups of a CU (see the kernel): grids of one round at two workgroups per CU only
      // (256 < grid <= 512, four query blocks per wave), splits long enough to cut.  SFM_MATCH_SPLIT_W="first:second" overrides,
      // "0" = even splits
      int w_first = 0, w_second = 0;
      if (!direct2 && grid > 256 && grid <= 512 && nsplit > 1 && nt_rows / nsplit >= 2048) { w_first = MATCH_W_FIRST; w_second = MATCH_W_SECOND; }
      if (const char* we = getenv("SFM_MATCH_SPLIT_W")) {
        int a = 0, b = 0;
        if (sscanf(we, "%d:%d", &a, &b) == 2 && a > 0 && b > 0 && a <= 1024 && b <= 1024 && nsplit > 1) { w_first = a; w_second = b; }
        else { w_first = w_second = 0; }
      }
      if (direct256) hipLaunchKernelGGL((k_knn2_u8_direct<2, 8>), dim3(grid), dim3(256), 0, h->stream, q8, nq_rows, w.tf, nt_rows, w.th, w.par, w.qn, nsplit, rps, w.part, w.u2, w_first, w_second);
      else if (direct2) hipLaunchKernelGGL((k_knn2_u8_direct<2>), dim3(grid), dim3(256), 0, h->stream, q8, nq_rows, w.tf, nt_rows, w.th, w.par, w.qn, nsplit, rps, w.part, w.u2, w_first, w_second);
      else hipLaunchKernelGGL((k_knn2_u8_direct<4>), dim3(grid), dim3(256), 0, h->stream, q8, nq_rows, w.tf, nt_rows, w.th, w.par, w.qn, nsplit, rps, w.part, w.u2, w_first, w_second);
    } else if (filter) {
      if (qb4) KNN_LAUNCH(4, 4, true); else if (dim == 256) KNN_LAUNCH(8, 2, true); else if (dim == 128) KNN_LAUNCH(4, 2, true); else if (dim == 64) KNN_LAUNCH(2, 2, true); else KNN_LAUNCH(1, 2, true);
    } else {
      if (qb4) KNN_LAUNCH(4, 4, false); else if (dim == 256) KNN_LAUNCH(8, 2, false); else if (dim == 128) KNN_LAUNCH(4, 2, false); else if (dim == 64) KNN_LAUNCH(2, 2, false); else KNN_LAUNCH(1, 2, false);
    }
#undef KNN_LAUNCH
    sfm_prof_end(h, SFM_PROF_KNN);
    hipLaunchKernelGGL(k_merge_splits_u8, dim3(cdiv(n_out, 256)), dim3(256), 0, h->stream, n_out, wg ? 8 : nsplit, w.part, idx1, idx2,
                       d1, d2, w.fix_cnt, w.fix_list, take_root ? 1 : 0);
    hipLaunchKernelGGL(k_knn2_u8_rerank, dim3(n_out < 2048 ? (unsigned)n_out : 2048u), dim3(256), 0, h->stream, q8, t8, nt_rows, dim,
                       segs, w.fix_cnt, w.fix_list, idx1, idx2, d1, d2);
  } else {
#define VALU_LAUNCH(M, W) hipLaunchKernelGGL((k_knn2_valu<M, W>), dim3(grid), dim3(256), 0, h->stream, qq, nq_rows, tt, nt_rows, nsplit, rps, wg, n_out, w.part)
    if (metric == SFM_METRIC_L2_F32) {
      if (dim == 128) VALU_LAUNCH(1, 128);
      else if (dim == 64) VALU_LAUNCH(1, 64);
      else VALU_LAUNCH(1, 32);
    } else {
      if (dim == 32) VALU_LAUNCH(2, 8);
      else if (dim == 64) VALU_LAUNCH(2, 16);
      else VALU_LAUNCH(2, 4);
    }
#undef VALU_LAUNCH
    hipLaunchKernelGGL(k_merge_splits, dim3(cdiv(n_out, 256)), dim3(256), 0, h->stream, n_out, wg ? 8 : nsplit, w.part, idx1, idx2, d1, d2);
  }
  SFM_LAUNCH_CHECK(h, "sfm_match_knn2");
  return SFM_OK;
}

static int match_check_metric(sfm_ctx* h, int metric, int dim, const char* what) {
  if (metric == SFM_METRIC_L2_U8) {
    if (dim != 32 && dim != 64 && dim != 128) return sfm_fail(h, SFM_ERR_ARG, what, "L2_U8 needs dim 32, 64 or 128");
  } else if (metric == SFM_METRIC_L2_F32) {
    if (dim != 32 && dim != 64 && dim != 128) return sfm_fail(h, SFM_ERR_ARG, what, "L2_F32 supports dim 32, 64, 128");
  } else if (metric == SFM_METRIC_HAMMING) {
    if (dim != 16 && dim != 32 && dim != 64) return sfm_fail(h, SFM_ERR_ARG, what, "HAMMING supports dim 16, 32, 64 bytes");
  } else {
    return sfm_fail(h, SFM_ERR_ARG, what, "unknown metric");
  }
  return SFM_OK;
}
// queries one workgroup of the distance kernel takes
// rows of train data per split and queries per workgroup of the kernel a metric uses

extern "C" int sfm_match_knn2(sfm_handle h, int metric, const void* q, int64_t nq, const void* t, int64_t nt,
                              int dim, int32_t* idx1, int32_t* idx2, float* d1, float* d2, void* workspace,
                              int64_t workspace_bytes) {
  if (!h) return SFM_ERR_ARG;
  if (!q || !t || !idx1 || !idx2 || !d1 || !d2 || !workspace) return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2", "null pointer");
  if (nq < 1 || nt < 2) return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2", "needs nq >= 1 and nt >= 2");
  if (nt > 0x7FFFFF00LL || nq > 0x7FFFFF00LL) return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2", "too many rows");
  int rc = match_check_metric(h, metric, dim, "sfm_match_knn2"); if (rc) return rc;
  const bool bits = metric == SFM_METRIC_HAMMING && dim <= 32;      // 128 / 256 bits: on the int8 kernels (k_unpack_bits)
  const MatchWs w = match_ws_carve((char*)workspace, nq, nt, nq, 0, bits);
  if (workspace_bytes < w.total) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_match_knn2", "workspace too small");
  if (bits) {
    hipLaunchKernelGGL(k_unpack_bits, dim3(cdiv(nq * dim, 256)), dim3(256), 0, h->stream, (const uint8_t*)q, nq * dim, w.qbits);
    hipLaunchKernelGGL(k_unpack_bits, dim3(cdiv(nt * dim, 256)), dim3(256), 0, h->stream, (const uint8_t*)t, nt * dim, w.tbits);
    q = w.qbits; t = w.tbits; metric = SFM_METRIC_L2_U8; dim *= 8;
  }
  int nsplit; int64_t rps;
  const int64_t qpw = match_qpw(metric, dim, nq, false);
  match_tiling(metric, nq, qpw, nt, &nsplit, &rps);
  const unsigned grid = cdiv(nq, qpw) * nsplit;
  MatchSegs none = {nullptr, nullptr, nullptr, nullptr, 0};
  return match_launch(h, metric, q, nq, t, nt, dim, nsplit, rps, qpw, nt, nullptr, grid, nq, none, w, idx1, idx2, d1, d2, !bits);
}

// ---- batched: every image pair of a preprocessing step in one launch

extern "C" int sfm_match_batched_workspace_bytes(int metric, int32_t n_seg, const int64_t* q_beg_host, const int64_t* q_end_host,
                                                 const int64_t* t_beg_host, const int64_t* t_end_host, int64_t nq_rows,
                                                 int64_t nt_rows, int64_t* n_out_host, int64_t* bytes_host) {
  if (n_seg < 1 || !q_beg_host || !q_end_host || !t_beg_host || !t_end_host || !n_out_host || !bytes_host) return SFM_ERR_ARG;
  std::vector<MatchWG> wgs; std::vector<int64_t> out_ptr;
  plan_segments(metric, 32 /* the descriptor size is not known here: the smallest pieces, i.e. the longest plan */, n_seg, q_beg_host,
                q_end_host, t_beg_host, t_end_host, &wgs, &out_ptr);
  *n_out_host = out_ptr[n_seg];
  const int64_t plan_bytes = (int64_t)wgs.size() * sizeof(MatchWG) + 4 * ((int64_t)n_seg + 1) * 8 + 256;
  *bytes_host = match_ws_carve(nullptr, out_ptr[n_seg] > 0 ? out_ptr[n_seg] : 1, nt_rows, nq_rows, plan_bytes, metric == SFM_METRIC_HAMMING).total;
  return SFM_OK;
}

extern "C" int sfm_match_knn2_batched(sfm_handle h, int metric, const void* q, int64_t nq_rows, const void* t, int64_t nt_rows,
                                      int dim, int32_t n_seg, const int64_t* q_beg_host, const int64_t* q_end_host,
                                      const int64_t* t_beg_host, const int64_t* t_end_host, int32_t* idx1, int32_t* idx2,
                                      float* d1, float* d2, int64_t* out_ptr_device, void* workspace, int64_t workspace_bytes) {
  if (!h) return SFM_ERR_ARG;
  if (!q || !t || !idx1 || !idx2 || !d1 || !d2 || !workspace || !q_beg_host || !q_end_host || !t_beg_host || !t_end_host || n_seg < 1)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2_batched", "null pointer / no segments");
  int rc = match_check_metric(h, metric, dim, "sfm_match_knn2_batched"); if (rc) return rc;
  if (nq_rows > 0x7FFFFF00LL || nt_rows > 0x7FFFFF00LL) return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2_batched", "too many rows");
  for (int s = 0; s < n_seg; ++s) {
    if (q_beg_host[s] < 0 || q_end_host[s] < q_beg_host[s] || q_end_host[s] > nq_rows || t_beg_host[s] < 0 || t_end_host[s] > nt_rows)
      return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2_batched", "segment outside the descriptor arrays");
    if (q_end_host[s] > q_beg_host[s] && t_end_host[s] - t_beg_host[s] < 2)
      return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2_batched", "a segment with queries needs at least 2 train rows");
  }
  const bool bits = metric == SFM_METRIC_HAMMING && dim <= 32;      // 128 / 256 bits: on the int8 kernels (k_unpack_bits)
  const int plan_metric = bits ? SFM_METRIC_L2_U8 : metric, plan_dim = bits ? dim * 8 : dim;
  std::vector<MatchWG> wgs; std::vector<int64_t> out_ptr;
  plan_segments(plan_metric, plan_dim, n_seg, q_beg_host, q_end_host, t_beg_host, t_end_host, &wgs, &out_ptr);
  const int64_t n_out = out_ptr[n_seg];
  if (n_out < 1) return sfm_fail(h, SFM_ERR_ARG, "sfm_match_knn2_batched", "no query rows");
  const int64_t wg_bytes = (int64_t)wgs.size() * sizeof(MatchWG), seg_bytes = ((int64_t)n_seg + 1) * 8;
  const MatchWs w = match_ws_carve((char*)workspace, n_out, nt_rows, nq_rows, wg_bytes + 4 * seg_bytes + 256, metric == SFM_METRIC_HAMMING);
  if (workspace_bytes < w.total) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_match_knn2_batched", "workspace too small");
  if (bits) {
    const bool same = q == t && nq_rows == nt_rows;                  // match_pairs hands one array for both sides
    hipLaunchKernelGGL(k_unpack_bits, dim3(cdiv(nt_rows * dim, 256)), dim3(256), 0, h->stream, (const uint8_t*)t, nt_rows * dim, w.tbits);
    if (!same) hipLaunchKernelGGL(k_unpack_bits, dim3(cdiv(nq_rows * dim, 256)), dim3(256), 0, h->stream, (const uint8_t*)q, nq_rows * dim, w.qbits);
    q = same ? w.tbits : w.qbits; t = w.tbits; metric = SFM_METRIC_L2_U8; dim *= 8;
  }
  // plan -> device: workgroup records, then q_beg | t_beg | t_end | out_ptr
  char* dp = w.plan;
  int64_t* d_seg = (int64_t*)(dp + align_up(wg_bytes, 256));
  SFM_HIP(h, hipMemcpyAsync(dp, wgs.data(), (size_t)wg_bytes, hipMemcpyHostToDevice, h->stream));
  SFM_HIP(h, hipMemcpyAsync(d_seg, q_beg_host, (size_t)n_seg * 8, hipMemcpyHostToDevice, h->stream));
  SFM_HIP(h, hipMemcpyAsync(d_seg + (n_seg + 1), t_beg_host, (size_t)n_seg * 8, hipMemcpyHostToDevice, h->stream));
  SFM_HIP(h, hipMemcpyAsync(d_seg + 2 * (n_seg + 1), t_end_host, (size_t)n_seg * 8, hipMemcpyHostToDevice, h->stream));
  SFM_HIP(h, hipMemcpyAsync(d_seg + 3 * (n_seg + 1), out_ptr.data(), (size_t)seg_bytes, hipMemcpyHostToDevice, h->stream));
  if (out_ptr_device)
    SFM_HIP(h, hipMemcpyAsync(out_ptr_device, out_ptr.data(), (size_t)seg_bytes, hipMemcpyHostToDevice, h->stream));
  SFM_HIP(h, hipStreamSynchronize(h->stream));          // the host vectors are pageable: the copies must have left them
  MatchSegs segs = {d_seg, d_seg + (n_seg + 1), d_seg + 2 * (n_seg + 1), d_seg + 3 * (n_seg + 1), n_seg};
  int64_t longest = 0;
  for (int s = 0; s < n_seg; ++s) longest = (t_end_host[s] - t_beg_host[s]) > longest ? (t_end_host[s] - t_beg_host[s]) : longest;
  return match_launch(h, metric, q, nq_rows, t, nt_rows, dim, 1, 0, 256, longest, (const MatchWG*)dp, (unsigned)wgs.size(), n_out, segs, w,
                      idx1, idx2, d1, d2, !bits);
}

extern "C" int sfm_match_ratio(sfm_handle h, int64_t nq, const int32_t* idx1, const float* d1, const float* d2,
                               double ratio, int32_t* query_idx, int32_t* train_idx, float* dist,
                               int64_t* n_matches, void* workspace, int64_t workspace_bytes) {
  if (!h) return SFM_ERR_ARG;
  if (!idx1 || !d1 || !d2 || !query_idx || !train_idx || !dist || !n_matches || !workspace || nq < 1)
    return sfm_fail(h, SFM_ERR_ARG, "sfm_match_ratio", "bad argument");
  const int nblk = (int)cdiv(nq, 256);
  if (workspace_bytes < (int64_t)nblk * 8) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_match_ratio", "workspace too small");
  int* blk_cnt = (int*)workspace;
  int* blk_off = blk_cnt + nblk;
  hipLaunchKernelGGL(k_ratio_count, dim3(nblk), dim3(256), 0, h->stream, nq, d1, d2, ratio, blk_cnt);
  hipLaunchKernelGGL(k_ratio_scan, dim3(1), dim3(256), 0, h->stream, nblk, blk_cnt, blk_off, n_matches);
  hipLaunchKernelGGL(k_ratio_scatter, dim3(nblk), dim3(256), 0, h->stream, nq, idx1, d1, d2, ratio, blk_off,
                     query_idx, train_idx, dist);
  SFM_LAUNCH_CHECK(h, "sfm_match_ratio");
  return SFM_OK;
}

// after the compaction of a batch: where each segment's matches begin, and query indices relative to their segment
__global__ __launch_bounds__(256) void k_seg_split(int64_t n_out, int n_seg, const int64_t* __restrict__ out_ptr,
                                                   const int64_t* __restrict__ n_matches, int* __restrict__ query_idx,
                                                   int64_t* __restrict__ seg_match_ptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t M = *n_matches;
  if (i <= n_seg) {                       // first match whose output row is >= out_ptr[i]   (query_idx is ascending)
    const int64_t key = out_ptr[i];
    int64_t lo = 0, hi = M;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (query_idx[mid] < key) lo = mid + 1; else hi = mid; }
    seg_match_ptr[i] = lo;
  }
}
__global__ __launch_bounds__(256) void k_seg_localise(int n_seg, const int64_t* __restrict__ out_ptr,
                                                      const int64_t* __restrict__ n_matches, int* __restrict__ query_idx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= *n_matches) return;
  const int64_t row = query_idx[i];
  int lo = 0, hi = n_seg - 1;             // last segment with out_ptr[s] <= row
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (out_ptr[mid] <= row) lo = mid; else hi = mid - 1; }
  query_idx[i] = (int)(row - out_ptr[lo]);
}

extern "C" int sfm_match_ratio_batched(sfm_handle h, int64_t n_out, int32_t n_seg, const int64_t* out_ptr, const int32_t* idx1,
                                       const float* d1, const float* d2, double ratio, int32_t* query_idx, int32_t* train_idx,
                                       float* dist, int64_t* seg_match_ptr, void* workspace, int64_t workspace_bytes) {
  if (!h) return SFM_ERR_ARG;
  if (!out_ptr || !seg_match_ptr || n_seg < 1) return sfm_fail(h, SFM_ERR_ARG, "sfm_match_ratio_batched", "bad argument");
  const int64_t need = (int64_t)cdiv(n_out, 256) * 8 + 64;
  if (workspace_bytes < need) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_match_ratio_batched", "workspace too small");
  int64_t* n_matches = (int64_t*)((char*)workspace + align_up((int64_t)cdiv(n_out, 256) * 8, 8));
  int rc = sfm_match_ratio(h, n_out, idx1, d1, d2, ratio, query_idx, train_idx, dist, n_matches, workspace, (int64_t)cdiv(n_out, 256) * 8);
  if (rc) return rc;
  // the split reads the global (output-row) query indices: it must run before they are made segment-relative
  hipLaunchKernelGGL(k_seg_split, dim3(cdiv((int64_t)n_seg + 1, 256)), dim3(256), 0, h->stream, n_out, n_seg, out_ptr, n_matches,
                     query_idx, seg_match_ptr);
  hipLaunchKernelGGL(k_seg_localise, dim3(cdiv(n_out, 256)), dim3(256), 0, h->stream, n_seg, out_ptr, n_matches, query_idx);
  SFM_LAUNCH_CHECK(h, "sfm_match_ratio_batched");
  return SFM_OK;
}

extern "C" int sfm_match_f32_to_u8(sfm_handle h, const float* src, int64_t n_elems, uint8_t* dst, int32_t* all_integral) {
  if (!h) return SFM_ERR_ARG;
  if (!src || !dst || !all_integral || n_elems < 1) return sfm_fail(h, SFM_ERR_ARG, "sfm_match_f32_to_u8", "bad argument");
  SFM_HIP(h, hipMemsetD32Async((hipDeviceptr_t)all_integral, 1, 1, h->stream));
  hipLaunchKernelGGL(k_f32_to_u8, dim3(cdiv(n_elems, 256)), dim3(256), 0, h->stream, src, n_elems, dst, all_integral);
  SFM_LAUNCH_CHECK(h, "sfm_match_f32_to_u8");
  return SFM_OK;
}
