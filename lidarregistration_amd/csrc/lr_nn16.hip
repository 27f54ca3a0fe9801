// Fast exact nearest / second-nearest neighbour search: f16 matrix-core filter + exact fp32 verification.
//
// Contract: reference Experiments/algorithms/matching.py:22-65 with the arithmetic of oracle/oracle.c -- the RESULT is
// bit-identical to the fp32 fma-chain definition.  What is free is how the N0 x N1 candidates are pruned.  On gfx950 the
// fp32-input MFMA runs at the fp32 VALU rate (round 1, profiles/r01a_*: 1024 + ~800 cycles per 32x32 tile), so the pruning
// runs in f16:
//
//   sample  f16 MFMA (v_mfma_f32_16x16x32_f16, the real matrix pipe) over every `sstride`-th column tile; per query row
//           the 2nd smallest (1st for top-1) approximate value u' = n1[j] - 2 dot16(i,j) of that subset: U_i
//   thresh  tau_i = U_i + 2 E_i (+ a sqrt-rounding band), E_i a rigorous bound on |d2_exact - (n0_i + u')|
//   walk    f16 MFMA over ALL column tiles; a column is a candidate of row i iff u' <= tau_i.  The accumulator starts
//           at tau_i/2 (minus the smallest column term when all column norms are alike), so the test of a lane's 16
//           accumulator registers is the AND of their sign bits (or, with spread-out norms, their maximum against n1[j]/2):
//           9 VALU ops per 1024 elements, interleaved with the next tile's MFMAs; candidates are ~1e-4 of the elements;
//           a hit only parks { column, lane group }, the rows and filter values are re-derived later by the same MFMA on
//           the gathered columns, and the thresholds tighten as the walk finds better neighbours (sample, thresholds and
//           walk are ONE kernel, nn16_passb_kernel)
//   reverse (mutual filter) thresholds from the forward result, rows / columns ordered by forward NN distance so that
//           each row block walks only a prefix of the column tiles (lr_nn16_reverse)
//   exact   per row, the fp32 fma-chain distance of its few candidates, ordered by (sqrt value, index) --
//           this is exactly torch.min's "first minimal value" order, so no separate tie-break path is needed;
//           rows whose candidate list overflowed (duplicate-heavy inputs) or could not be filled (non-finite
//           f16 conversions) are re-done in place by an exact scan of all columns.
//
// Why the candidate set is a superset of what the exact order needs: let S be the sampled columns and j1, j2 in S the
// two with the smallest u'.  Their exact distances are <= n0_i + U_i + E_i, hence so is the exact 2nd smallest x2 of
// the whole row.  Every j whose sqrt value ties with or beats the 2nd smallest has d2(j) <= x2 (1 + 2^-21), therefore
// u'(j) <= d2(j) - n0_i + E_i <= U_i + 2 E_i + 2^-21 (n0_i + U_i + E_i)  -- the threshold used below.
//
// Error bound (u = 2^-24; n0, n1 squared norms; all terms worst case):
//   exact chain vs real arithmetic      <= (2 g32 + 3u)(n0 + n1),  g32 = 32u/(1-32u)          ~  67 u (n0+n1)
//   f16 input rounding (RN, 2^-11 rel.) <= (2^-10 + 2^-22) 2 sqrt(n0 n1) <= (2^-10 + 2^-22)(n0+n1)
//   f16 underflow (|x| < 2^-14)         <= 2 * 2^-25 (|a|_1 + |b|_1) <= 3.4e-7 (1 + (n0+n1)/2)
//   MFMA fp32 accumulation of exact f16 products, norms, final fma             <= ~162 u (n0+n1) (generous)
//   => E_ij <= 9.91e-4 (n0+n1) + 3.4e-7 ;  used: 1.05e-3 (n0_i + max_j n1_j) + 4e-7.
#include "lr_internal.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LR_INF __builtin_huge_valf()
#define LR_IMAX 0x7fffffff

// order-preserving image of a float in an unsigned integer (smaller float <=> smaller integer; NaN never travels)
__device__ __forceinline__ uint32_t lr_ord_enc(float v) { const uint32_t b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float lr_ord_dec(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
#define LR_ORD_INF 0xff800000u      // lr_ord_enc(+inf)

// ------------------------------------------------------------------ prep: norms + f16 copy
// Eight threads per row (coalesced 16-byte loads).  The norm is the sequential fp32 fma chain over k = 0..31 of the
// arithmetic contract: thread t continues the chain over its four values from where thread t-1 stopped.
// H[row] (64 B) = f16 of k0..31 in order: lane group kb of an MFMA operand reads bytes [16 kb, 16 kb + 16).
__global__ void __launch_bounds__(256)
nn16_prep_kernel(const float *__restrict__ Fa, int na, _Float16 *__restrict__ Ha, float *__restrict__ nrma, float *__restrict__ bmaxa, float *__restrict__ bmina,
                 const float *__restrict__ Fb, int nb, _Float16 *__restrict__ Hb, float *__restrict__ nrmb, float *__restrict__ bmaxb, float *__restrict__ bminb,
                 uint32_t *__restrict__ seed_b, unsigned long long *__restrict__ seed64_b, int32_t *__restrict__ counters, int zero_counters, int nblk_a,
                 uint32_t *__restrict__ yshare_a, lr_zargs z)
{
    __shared__ float s_m[4], s_n[4];
    if (z.descs) { const lr_pair_desc d = z.descs[blockIdx.z]; Fa = d.F0; na = d.n0; Fb = d.F1; nb = d.n1; }
    lr_z(bmina, z, blockIdx.z); lr_z(bminb, z, blockIdx.z);
    lr_z(Ha, z, blockIdx.z); lr_z(nrma, z, blockIdx.z); lr_z(bmaxa, z, blockIdx.z); lr_z(Hb, z, blockIdx.z); lr_z(nrmb, z, blockIdx.z);
    lr_z(bmaxb, z, blockIdx.z); lr_z(seed_b, z, blockIdx.z); lr_z(seed64_b, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(yshare_a, z, blockIdx.z);
    // first kernel of a pair: the counter block starts from zero (lr_register_pair) and the distance range of
    // lr_nn16_reverse from { 0x7f7f7f7f, 0 }
    if (counters && blockIdx.x == 0 && (int)threadIdx.x < LR_CNT_TOTAL) {
        const int k = threadIdx.x;
        if (k == LR_CNT_RLO) counters[k] = 0x7f7f7f7f;
        else if (k == LR_CNT_RHI || k == LR_CNT_FORM_MISS_F || k == LR_CNT_FORM_MISS_R || zero_counters) counters[k] = 0;
    }
    // blocks [0, nblk_a) prepare cloud a, the rest cloud b (one launch for the pair; nblk_a = ceil(na/32) of the largest
    // cloud of a batch: blocks past a pair's own rows only write a zero maximum)
    const bool second = (int)blockIdx.x >= nblk_a;
    const float *__restrict__ F = second ? Fb : Fa;
    _Float16 *__restrict__ H = second ? Hb : Ha;
    float *__restrict__ nrm = second ? nrmb : nrma;
    float *__restrict__ block_max = second ? bmaxb : bmaxa;
    float *__restrict__ block_min = second ? bminb : bmina;
    const int n = second ? nb : na;
    const int blk = second ? blockIdx.x - nblk_a : blockIdx.x;
    const int gid = blk * 256 + threadIdx.x;
    const int row = gid >> 3, t = gid & 7, lane = threadIdx.x & 63;
    const bool live = row < n;
    f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (live) v = reinterpret_cast<const f32x4 *>(F + (size_t)row * 32)[t];
    float run = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float in = __shfl(run, (lane & ~7) | (s > 0 ? s - 1 : 0));
        if (s == 0) in = 0.0f;
        float out = __builtin_fmaf(v.x, v.x, in);
        out = __builtin_fmaf(v.y, v.y, out);
        out = __builtin_fmaf(v.z, v.z, out);
        out = __builtin_fmaf(v.w, v.w, out);
        if (t == s) run = out;
    }
    const float norm = __shfl(run, (lane & ~7) | 7);
    if (live) {
        if (t == 0) {
            nrm[row] = norm;
            if (!second && yshare_a) yshare_a[row] = LR_ORD_INF;      // "no strip of the forward filter pass has a threshold for this row yet"
            if (second && seed_b) { seed_b[row] = 0x7f7f7f7fu; seed64_b[row] = ~0ull; }      // "no query points at this row yet" (lr_nn16_reverse)
        }
        const int pos = 4 * t;
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hv = { (_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w };
        *reinterpret_cast<f16x4 *>(H + (size_t)row * 32 + pos) = hv;
    }
    // largest norm of the block (32 rows) -> block_max[blockIdx.x]; the threshold kernel reduces that short array
    // (no same-address atomics: thousands of them serialise at ~12 ns each)
    // (a NaN norm drops out of fmaxf: such rows are re-done exactly anyway.  For the minimum a norm that is not finite -- NaN, or an
    // overflowing sum -- counts as -1: the range kernel's minimum is then negative, which tells the filter pass not to select the sign
    // form of its test (it needs all norms alike) and the exact kernel to re-do EVERY row of the other cloud by the full scan: a NaN
    // column is every row's nearest neighbour under the contract -- fmaxf(NaN, 1e-30) is 1e-30, torch.min returns the NaN -- and no
    // filter value says so)
    float m = live ? norm : 0.0f, mn = live ? ((norm - norm == 0.0f) ? norm : -1.0f) : LR_INF;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { m = fmaxf(m, __shfl_xor(m, k)); mn = fminf(mn, __shfl_xor(mn, k)); }
    if (lane == 0) { s_m[threadIdx.x >> 6] = m; s_n[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_max[blk] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (block_min) block_min[blk] = fminf(fminf(s_n[0], s_n[1]), fminf(s_n[2], s_n[3]));
    }
}

// largest / smallest squared norm of both clouds from the per-32-row values of the prep kernel: range[0..1] cloud a, range[2..3] cloud b
// (one block per cloud and pair; the filter-pass blocks then read two floats instead of reducing ~1000 each)
__global__ void __launch_bounds__(256)
nn16_range_kernel(int na, const float *__restrict__ bmaxa, const float *__restrict__ bmina, int nb, const float *__restrict__ bmaxb,
                  const float *__restrict__ bminb, float *__restrict__ range, int32_t *__restrict__ form_out, lr_zargs z)
{
    __shared__ float s_m[4], s_n[4];
    if (z.descs) { na = z.descs[blockIdx.z].n0; nb = z.descs[blockIdx.z].n1; }
    lr_z(bmaxa, z, blockIdx.z); lr_z(bmina, z, blockIdx.z); lr_z(bmaxb, z, blockIdx.z); lr_z(bminb, z, blockIdx.z); lr_z(range, z, blockIdx.z);
    const bool second = blockIdx.x == 1;
    const float *__restrict__ bmax = second ? bmaxb : bmaxa, *__restrict__ bmin = second ? bminb : bmina;
    const int nblk = ((second ? nb : na) + 31) >> 5;
    float mx = 0.0f, mn = LR_INF;
    for (int b = threadIdx.x; b < nblk; b += 256) { mx = fmaxf(mx, bmax[b]); mn = fminf(mn, bmin[b]); }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, k)); mn = fminf(mn, __shfl_xor(mn, k)); }
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = mx; s_n[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float hi = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])), lo = fminf(fminf(s_n[0], s_n[1]), fminf(s_n[2], s_n[3]));
        range[2 * second] = hi;
        range[2 * second + 1] = lo;
        // (single-pair calls: which form of the filter pass this cloud's norms ask for when it is the column cloud -- nn16_passb_kernel's
        // own rule --, left in pinned host memory for the NEXT call on the workspace, which then launches that form alone)
        if (form_out) form_out[second] = (lo > 0.0f && hi < LR_INF && hi - lo <= 1e-4f * hi) ? 1 : 2;
    }
}

// ------------------------------------------------------------------ filter pass: sample phase + candidate walk in one kernel
// Block = 4 waves x 64 query rows = 256 rows of one column strip.  Column tiles (32 columns) are staged once per block through
// LDS in chunks of LR_PB_CH tiles (register-staged, double-buffered, one barrier per chunk) and shared by the four waves.  LDS
// image: one 64-byte row per column, its four 16-byte pieces XOR-swizzled (lr_lds_off) so that the walk's ds_read_b128 are
// bank-conflict-free for the instruction's lane groups.  Both phases use v_mfma_f32_16x16x32_f16: lane (c, kb) = (lane % 16,
// lane / 16) supplies the 8 K values 8 kb.. of row / column c of a 16-block and receives rows 4 kb + 0..3 of column c.
//
// Phase 1 (forward direction; the reverse direction gets its thresholds from the forward result): the block walks every
// `sstride`-th tile of its strip with the operand roles SWAPPED -- the column fragment is the MFMA's first operand, the query
// rows the second, so a LANE holds one query row per 16-row block and its 4 + 4 accumulator registers of a tile are 8 different
// columns.  The accumulator starts at zero and holds dot16; the lane's value of the tile is
//     b = max_j dot16(i, j) - max_j n1[j]/2      (in-lane maximum tree over the 8 values, one subtraction)
// a LOWER bound of the best g = dot16 - n1[j]/2 among those columns, attained up to the spread of the column norms inside the tile
// (zero for unit-norm descriptors such as FCGF's) -- no per-column operand is read.  Three more ops merge b into the running two
// largest values of the lane; they belong to different tiles, hence to different columns, and the four lanes that share a row see
// disjoint columns, so the second largest after merging the four is a valid lower bound of the row's 2nd largest g (u' = -2 g: an
// upper bound of the 2nd smallest u' -- any valid bound keeps the result exact, a looser one only admits more candidates).  Round 2
// ran this phase as a kernel of its own over every 4th tile (16 us per pair, 13 % of the pair); here it shares the staging buffers
// and the row fragments with the walk, samples every 16th tile (~3 us), and the thresholds it yields are only the START of the walk:
//
// Phase 2: f16 MFMA over ALL tiles of the strip, columns on the lanes.  The accumulator is started at y_i = tau_i / 2 instead of 0,
// so the candidate test u' <= tau_i  <=>  dot16 + y_i >= n1[j]/2 needs no per-element arithmetic: the lane's 16 accumulator
// registers of one 16-column block (the wave's 64 rows) are tested at once.  Two forms, two instantiations (SIGN; both are launched,
// the blocks of the one the pair's column norms do not ask for return at their first branch): with all column norms alike the
// smallest column term is folded into the start values and the test is the AND of 16 sign bits -- 7 v_bitop3_b32 + 1 v_and_b32,
// full rate, no per-column operand --, otherwise the largest of the 16 registers against x_j = n1[j]/2 (7 v_max3 + 1 v_max, half
// rate); + 1 v_cmp + 1 scalar branch.  There is ONE set of 32 accumulator registers: a group of 16 is tested (it holds the previous
// tile) and then handed to the four MFMAs that overwrite it with the current tile, so every test reads its registers four MFMAs
// after they were issued (the hardware does not interlock a vector read of an MFMA result; scheduling barriers keep the MFMAs
// behind the test).  The vector ops of one wave run while the matrix pipe works on another's MFMAs.  What bounds the kernel is the
// board's power limit: a stall that is removed comes back as a lower clock, an instruction that is removed from the fast path does
// not.  The column fragments of tile t+1 are read from LDS one step ahead, which moves the chunk barrier one step
// forward.  Hits are parked in a wave-private LDS list whose fill count lives in a scalar register: no atomics and no LDS round trip
// in the loop.
//
// Thresholds tighten while the walk runs (derive(), below): whenever LR_PB_TIGHTEN new entries have gathered in a list, the four
// waves of the block go through their new entries TOGETHER (they exchange what their lists want at the chunk barrier -- a wave
// in a round keeps its siblings waiting there, so the rounds are taken at the same time), 16 entries per instruction group: the
// entries' columns are gathered and the walk's own MFMA repeated on them with the thresholds of now -- same instruction, same
// operands, same bits --, which yields each entry's row mask and, for single-row entries, the filter value g = dot16 - x_j; the two
// largest g of every row (of the walk: distinct columns) are kept in LDS with two float atomics per entry, y_row <- min(y_row,
// E_row - g2 + ...), and the lanes reload their 16 threshold registers.  The number of hits of a row then grows like
// 2 + 2 ln(tiles / sampled tiles) instead of 2 tiles / sampled tiles, which is what lets the sample be small.  Several strips of one
// row block (single-pair calls) pool their thresholds through atomicMin on an order-preserving integer image (yshare).
#define LR_LDS_ROW 64
// byte offset of 16-byte piece p (K 8p..8p+7) of staged column j of a chunk: rows of 64 bytes, the piece index XOR-swizzled with bits
// 1..2 of the column.  ds_read_b128 is served in four fixed groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...); with
// the walk's lane map (column = lane % 16, piece = lane / 16) every group then reads 16 different 16-byte slots of the 256-byte bank
// row -- conflict-free (unswizzled 64- or 80-byte rows: 2-way).  Both phases read with that lane map.
__device__ __forceinline__ int lr_lds_off(int j, int p) { return j * LR_LDS_ROW + ((p ^ ((j >> 1) & 3)) << 4); }
#define LR_BLOCK_ROWS 256        // rows per block: 4 waves x 64 rows (four 16-row MFMA blocks per wave)
#define LR_PB_CH 4               // column tiles per staged chunk
#define LR_PB_WLIST 512          // entries per wave (8 bytes each)
// Entry of a hit list / of the candidate store (8 bytes): x = column (22 bits) | kb << 22 | LR_PB_HASG; y = 16-bit row mask | g16 << 16.
// kb = lane / 16 of the lane that saw the hit: bit b = 4 rbk + g of the mask <-> row 16 rbk + 4 kb + g of the wave (the lane's 16
// accumulator registers of one 16-column block).  The walk parks entries with an empty mask; derive() fills it in.
#define LR_PB_COLMASK 0x3fffffu
#define LR_PB_HASG 0x1000000u    // entry flag (in x): exactly one row, and y carries its filter value g rounded up to 16 bits
#define LR_PB_DGROUPS 3          // groups of 16 entries derive() has in flight at once (registers: 9 per group)
#define LR_PB_TIGHTEN 48         // new entries of a wave that trigger a tightening round
// (The variants measured against this kernel -- no staging / no tests / no tightening, geometric tightening schedule, per-wave rounds,
// the per-tile fold of phase 1 for the sign form, four waves per SIMD, 8-tile chunks -- live in tools/pb_variants/ as patches applied to a
// copy of this file by tools/pb_variant.sh; none of them is compiled into the library.  profiles/r05_pb_ablation.txt has their numbers.)

// row (0..63 of the wave) of mask bit b (0..15) of an entry of lane group kb
__device__ __forceinline__ int lr_pb_row(int kb, int b) { return 16 * (b >> 2) + 4 * kb + (b & 3); }
__device__ __forceinline__ int lr_pb_kb(unsigned x) { return (int)(x >> 22) & 3; }

#define LR_RS_BUCKETS 4096

// monotone map distance -> bucket: linear over the pair's range of forward NN distances [lo, hi] (well spread keys keep
// the sort's atomics apart); any monotone map is valid, a coarser one only prunes less
__device__ __forceinline__ int rs_bucket(float v, float lo, float scale)
{
    const int b = (int)((v - lo) * scale);
    return min(max(b, 0), LR_RS_BUCKETS - 1);
}
__device__ __forceinline__ float rs_scale(float lo, float hi) { return hi > lo ? (float)LR_RS_BUCKETS / (hi - lo) : 0.0f; }

// what the thresholds are made of: squared norms of the query rows (by data row), per-32-row maxima of the column cloud's squared
// norms, how many neighbours are wanted, and the sampling stride of phase 1 (forward direction)
struct lr_thr_in {
    const float *nQ;
    const float *range_c;          // { largest, smallest } squared norm of the column cloud (nn16_range_kernel), or nullptr
    int need, sstride;
};
// grid shape of a 1-D XCD-aware launch + direction (0: rows = cloud 0, columns = cloud 1; 1: the reverse pass)
struct lr_pb_grid { int gx, gy, total, dir, only; };      // only != 0: the other instantiation is not launched (single-pair calls, see lr_nn16_forms)
// Where the exact stage (nn16_exact_kernel) leaves its results: the two neighbours by data row, and (forward direction of a pair) the
// seeds of the reverse pass
struct lr_ex_out {
    int32_t *idx1, *idx2;
    float *s1o, *s2o;
    uint32_t *seed_out;              // [columns] smallest forward distance pointing at each column (bit pattern), or nullptr
    float *seed_s1;                  // [rows] column key of the reverse pass (2nd-NN distance, or the NN distance when need == 1)
    uint32_t *seed_range;            // { smallest, largest } of the seeds and keys
    unsigned long long *seed64;      // forward: [columns] (distance bits << 32) | smallest row at that distance; reverse: the seeds to start from
};
// The filter pass' FIRST kernel parameter, read only at the END of the kernel and through the kernel-argument segment pointer behind an
// opaque asm: as an ordinary parameter the pointer would sit in two scalar registers across the walk, and the kernel is at its scalar
// register limit (round 6: a fifteen-pointer version of this struct -- the fused verification, docs/HISTORY.md -- cost 17 scalar spills).
struct lr_pb_tail {
    unsigned long long *clk;         // { shader cycles, 100 MHz ticks } summed over the blocks of the launch (lr_workspace_clock), or nullptr
};
__device__ __forceinline__ lr_pb_tail lr_pb_tail_args()
{
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();          // (explicit arguments start at offset 0 of the segment)
    asm volatile("" : "+s"(ka) :: "memory");                   // keeps the compiler from hoisting the load to the top of the kernel
    typedef const unsigned long long __attribute__((opencl_constant)) *cw_t;
    static_assert(sizeof(lr_pb_tail) == 8, "read as one 64-bit word");
    lr_pb_tail t;
    const unsigned long long w = *(cw_t)ka;
    __builtin_memcpy(&t, &w, sizeof t);
    return t;
}

// ------------------------------------------------------------------ exact stage: shared pieces
#define LR_EX_EMPTY 0xffffffffffffffffull
// The two best candidates of a row under the (sqrt value, index) order -- torch.min's "first minimal value" -- are kept as 64-bit keys (value
// bits << 32 | index) with two LDS atomics per candidate:  old = atomicMin(best, key);  atomicMin(second, max(old, key)).  Whatever the
// order of arrival, `best` ends as the smallest key and `second` as the second smallest (every loser max(old, key) is at least the second
// smallest, and the second smallest itself loses exactly once).
__device__ __forceinline__ void ex_offer(unsigned long long *best, unsigned long long *second, float sv, int j)
{
    const unsigned long long key = ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)j;       // sv > 0: bits order like values
    const unsigned long long old = atomicMin(best, key);
    atomicMin(second, old > key ? old : key);
}
// results of one row (the calling lane's): neighbours and distances by data row; forward direction of a pair: seed the reverse pass (what
// nn16_rev_seed_kernel would recompute bit for bit): best forward distance per target, the row's own column key.  sv / key: the lane's
// contribution to the range the ordering buckets of the reverse pass span (ex_range reduces them over the wave)
__device__ __forceinline__ void ex_write_row(const lr_ex_out &o, int rowd, unsigned long long kb, unsigned long long ks, int nb, int need, float &sv, float &key)
{
    const int i1 = kb == LR_EX_EMPTY ? -1 : (int)(unsigned)kb;
    const float b1 = kb == LR_EX_EMPTY ? LR_INF : __uint_as_float((unsigned)(kb >> 32));
    o.idx1[rowd] = i1;
    if (o.idx2) o.idx2[rowd] = ks == LR_EX_EMPTY ? (nb > 1 ? LR_IMAX : -1) : (int)(unsigned)ks;
    if (o.s1o) o.s1o[rowd] = b1;
    if (o.s2o) o.s2o[rowd] = ks == LR_EX_EMPTY ? LR_INF : __uint_as_float((unsigned)(ks >> 32));
    if (o.seed_out) {
        sv = b1;
        if (!(sv < 3.0e38f)) sv = 3.0e38f;
        if (i1 >= 0 && i1 < nb) {
            atomicMin(&o.seed_out[i1], __float_as_uint(sv));
            atomicMin(&o.seed64[i1], ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)rowd);
        }
        // column key of the reverse pass: a lower bound of the row's distance to every point it does NOT point at -- its exact
        // second-smallest distance when the second neighbour was asked for (the thresholds then guarantee it), else the smallest
        key = sv;
        if (need >= 2) { key = ks == LR_EX_EMPTY ? 3.0e38f : __uint_as_float((unsigned)(ks >> 32)); if (!(key < 3.0e38f)) key = 3.0e38f; }
        o.seed_s1[rowd] = key;
    }
}
// (one wave, every lane: writer = the lane wrote a row) the range of the seeds and keys
__device__ __forceinline__ void ex_range(const lr_ex_out &o, bool writer, float sv, float key, int lane)
{
    float lo = writer ? sv : 3.0e38f, hi = writer ? fmaxf(sv, key) : 0.0f;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { lo = fminf(lo, __shfl_xor(lo, k)); hi = fmaxf(hi, __shfl_xor(hi, k)); }
    if (lane == 0) {
        atomicMin(&o.seed_range[0], __float_as_uint(lo));
        atomicMax(&o.seed_range[1], __float_as_uint(hi));
    }
}
// a row whose f16 copy is not finite never produced a meaningful filter value (u, v: eight of its fp32 values)
__device__ __forceinline__ bool ex_bad8(const f32x4 &u, const f32x4 &v)
{
    const float big = fmaxf(fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))), fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    return !(big <= 65504.0f) || u.x != u.x || u.y != u.y || u.z != u.z || u.w != u.w || v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
}

// The one development switch of this file: -DLR_PB_PROBE compiles the hit statistics of tools/pb_micro.hip into the filter pass (waves,
// tests, slow-path visits, hits, derive() rounds, 16-entry groups, 10 ns ticks per phase).  Off in the library: LR_PROBE(...) is empty.
#ifdef LR_PB_PROBE
__device__ unsigned long long lr_pb_stat[16];
#define LR_PROBE(...) __VA_ARGS__
#else
#define LR_PROBE(...)
#endif
#define LR_PB_WAVES 3            // waves per SIMD the filter pass is compiled for
template <bool SIGN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LR_PB_WAVES, LR_PB_WAVES)))
nn16_passb_kernel(lr_pb_tail tail_first /* read at the end only, through lr_pb_tail_args() */, unsigned long long *clk_on,
                  const _Float16 *__restrict__ Hq, int na_host, const int32_t *__restrict__ rowmap, const int32_t *__restrict__ na_dev,
                  const _Float16 *__restrict__ Hc, const float *__restrict__ nC, int nb,
                  int tiles_per_strip, const float *__restrict__ tau, int32_t *__restrict__ cand_cnt, int32_t *__restrict__ cand,
                  const int32_t *__restrict__ colmap, const float *__restrict__ tile_min, const uint32_t *__restrict__ row_bound,
                  const int32_t *__restrict__ rev_offs, const uint32_t *__restrict__ rev_range, float *__restrict__ yfin, int yfin_stride,
                  uint32_t *__restrict__ yshare, int32_t *__restrict__ form_miss, lr_thr_in thr, lr_pb_grid pg, lr_zargs z)
{
    // 1-D XCD-aware grid -> (row block, strip, pair): the blocks one XCD receives are consecutive row blocks of the same
    // (strip, pair), i.e. they stream the same columns through that XCD's L2
    int logical;
    if (!lr_xcd_block(pg.total, logical)) return;
    const int bx = logical % pg.gx, by = (logical / pg.gx) % pg.gy, pair = logical / (pg.gx * pg.gy);
    if (z.descs) {
        const lr_pair_desc d = z.descs[pair];
        na_host = pg.dir ? d.n1 : d.n0; nb = pg.dir ? d.n0 : d.n1;
    }
    lr_z(Hq, z, pair); lr_z(rowmap, z, pair); lr_z(na_dev, z, pair); lr_z(Hc, z, pair); lr_z(nC, z, pair); lr_z(tau, z, pair);
    lr_z(cand_cnt, z, pair); lr_z(cand, z, pair); lr_z(colmap, z, pair); lr_z(tile_min, z, pair); lr_z(row_bound, z, pair); lr_z(rev_offs, z, pair); lr_z(rev_range, z, pair);
    lr_z(thr.nQ, z, pair); lr_z(thr.range_c, z, pair); lr_z(yfin, z, pair); lr_z(yshare, z, pair);
    // rows: either 0..na_host-1, or (reverse direction) the ordered list rowmap[0..*na_dev-1]; tau, cand_cnt and cand
    // are indexed by the position in that list.  Columns: Hc/nC as they lie; with colmap they are a permuted copy and
    // a candidate's column id is colmap[position].
    const int na = na_dev ? *na_dev : na_host;
    if (bx * LR_BLOCK_ROWS >= na) return;
    // The candidate test of the walk, "some accumulator y_i + dot16 >= x_j", takes two forms -- two instantiations of this kernel, both
    // launched, of which the one the pair's column norms do not ask for returns here.  When the norms are (nearly) all the same -- FCGF
    // descriptors are L2-normalised -- the smallest column term xhat = min_j x_j is folded into the accumulators' start values:
    // y_i - xhat + dot16 >= 0 is then NECESSARY for a hit (x_j >= xhat), a sign test: the AND of 8 sign bits, three v_bitop3_b32 + one
    // v_and_b32 at the full vector rate instead of three v_max3_f32 + one v_max_f32 at half rate, and no per-column operand.  It admits
    // a few pairs more (those between xhat and x_j); derive() applies the exact x_j to every hit anyway.  With spread-out norms that
    // slack would flood the hit lists, so SIGN = false compares with x_j itself (xhat = 0).  (A NaN, infinite, overflowing or
    // non-positive norm anywhere in the column cloud: the plain test -- the prep kernel reports such a norm as a minimum of 0, and
    // an infinite maximum fails `max_nc < inf`; with `inf - min <= 1e-4 inf` alone one overflowing column selected the sign form.)
    float max_nc = 0.0f, min_nc = 0.0f;
    if (thr.range_c) { max_nc = thr.range_c[0]; min_nc = thr.range_c[1]; }
    const bool sign_ok = thr.range_c != nullptr && min_nc > 0.0f && max_nc < LR_INF && max_nc - min_nc <= 1e-4f * max_nc && true;
    if (sign_ok != SIGN) {
        // (launched alone on the strength of the previous call's norms, and this call's ask for the other form: say so -- the exact
        // kernel then re-does every row by the full scan, and the next call launches the right one)
        if (pg.only && form_miss && bx == 0 && by == 0 && threadIdx.x == 0) { lr_z(form_miss, z, pair); *form_miss = 1; }
        return;
    }
    const float xhat = SIGN ? 0.5f * min_nc : 0.0f;
    __shared__ int s_limit[4];
    // (LR_OPT_CLOCK_PROBE: shader cycles and 100 MHz ticks of every block that walks, summed per workspace -> lr_workspace_clock: the
    // clock the power-limited walk really ran at, the one thing that tells a 3 % change of the kernel from a 3 % slower box)
    __shared__ unsigned long long s_clk[2];
    if (clk_on && threadIdx.x == 0) { s_clk[0] = __builtin_amdgcn_s_memtime(); s_clk[1] = __builtin_amdgcn_s_memrealtime(); }
    constexpr int CH = LR_PB_CH;
    constexpr int XOFF = CH * 32 * LR_LDS_ROW;
    constexpr int BUF = XOFF + (SIGN ? 0 : CH * 32 * 4);      // (the sign form stages no per-column operand, in either phase)
    // The two chunk buffers are SEPARATE arrays on purpose: the walk fills one by LDS-direct loads while it reads the other, and the
    // compiler's wait-count pass puts a vmcnt(0) in front of every LDS read that MAY alias a pending LDS-direct load -- with one array
    // that was every fragment read behind the loads, i.e. the prefetch became synchronous (round 5: found in the ISA).
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[BUF];
    __shared__ __attribute__((aligned(16))) unsigned char lds_b[BUF];
    // (with 8-tile chunks the plain form, which also stages x_j, gives up an eighth of its list to stay at three blocks per CU)
    constexpr int WL = (LR_PB_CH >= 8 && !SIGN) ? (LR_PB_WLIST * 7) / 8 : LR_PB_WLIST;
    __shared__ uint2 wlist[4][WL];   // per wave: entries as above (mask empty until derive() has seen the entry)
    // per row of the block: y = tau/2 (what the accumulators start from), the row's error term, the two largest g of the walk
    __shared__ __attribute__((aligned(16))) float s_Y[LR_BLOCK_ROWS];
    __shared__ float s_D[LR_BLOCK_ROWS], s_N1[LR_BLOCK_ROWS], s_N2[LR_BLOCK_ROWS];
    __shared__ unsigned s_att[2];      // [0]: what the four waves' hit lists want (one byte per wave)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, kb = lane >> 4;
    const int row0 = bx * LR_BLOCK_ROWS + wave * 64;
    LR_PROBE(int n_visits = 0, n_hits = 0, n_rounds = 0, n_groups = 0, n_tk_derive = 0, n_tk_flush = 0, n_tk_a = 0, n_tk_b = 0;
             const unsigned long long tk_start = __builtin_amdgcn_s_memrealtime(); unsigned long long tk_walk = 0;)
    int ntiles = (nb + 31) >> 5;
    int my_strips = pg.gy;           // strips this row block really uses (the ordered reverse pass: as many as its column prefix is worth)
    if (tile_min) {
        // Ordered reverse pass (lr_nn16_reverse): a column can only win a row if its key (a lower bound of its distance to every row
        // it does not point at) is <= the row's bound.  Rows lie by descending bucket of their bound, columns by ascending bucket of
        // their key (one monotone map, nn16_rev_order_kernel), so the block's FIRST row has its largest bucket b and the columns
        // that can matter to the block are the first offs[b + 1] of the order (the end of bucket b): three
        // dependent loads by one thread instead of 256 bounds + all tile minima reduced by the block -- most of the (row block, strip)
        // blocks of this launch only find out here that they are not needed.
        if (tid == 0) {
            const float lo = __uint_as_float(rev_range[0]), scale = rs_scale(lo, __uint_as_float(rev_range[1]));
            const float sv = __uint_as_float(row_bound[rowmap[bx * LR_BLOCK_ROWS]]);
            const int b = rs_bucket(sv, lo, scale);        // (bucket b of the column order ends where bucket b + 1 starts)
            s_limit[0] = ((b + 1 < LR_RS_BUCKETS ? rev_offs[b + 1] : nb) + 31) >> 5;
        }
        __syncthreads();
        ntiles = min(ntiles, s_limit[0]);
        // as many of the offered strips as the prefix is worth (a full-length row block uses all of them)
        my_strips = min(pg.gy, (ntiles + tiles_per_strip - 1) / tiles_per_strip);
        if (by >= my_strips) return;
        tiles_per_strip = (ntiles + my_strips - 1) / my_strips;
    }
    const int t_begin = by * tiles_per_strip;
    const int t_end = min(ntiles, t_begin + tiles_per_strip);
    const int nchunks = t_end > t_begin ? (t_end - t_begin + CH - 1) / CH : 0;

    // Both phases run on v_mfma_f32_16x16x32_f16 (all 32 K in one instruction; under the board's power limit it sustains ~20 % more
    // flops than 32x32x16, tools/mfma_clock.hip): lane (c, kb) = (lane % 16, lane / 16) supplies K bytes 16 kb.. of row / column c of
    // a 16-block and receives rows 4 kb + 0..3, column c of the 16 x 16 result.  Row fragments of the wave's four 16-row blocks -- the
    // same registers serve as the MFMA's first operand (phase 2: rows x columns) and as its second (phase 1: columns x rows):
    f16x8 a16[4];
#pragma unroll
    for (int rbk = 0; rbk < 4; ++rbk) {
        int row = min(row0 + 16 * rbk + c16, na - 1);
        if (rowmap) row = rowmap[row];
        a16[rbk] = *reinterpret_cast<const f16x8 *>(Hq + (size_t)row * 32 + 8 * kb);
    }

    f32x4 stage[CH / 2];
    float stage_n = 0.0f;        // raw norm of the staged column: NOT touched until store_chunk, so that the compiler's
                                 // s_waitcnt for it lands after the tile loop instead of right behind the prefetch

    // ---------------------------------------------------------------- thresholds of the block's 256 rows -> LDS
    // given (reverse direction), or made here by phase 1: U = need-th smallest sampled u' = -2 * (need-th largest g);
    // tau = U + 2E + sqrt band 2^-21 (n0 + U + E) + rounding slop; y = tau/2; +inf when fewer than `need` tiles were sampled
    {
        const int rw = bx * LR_BLOCK_ROWS + tid;
        // error term of the row for thresholds made from a filter value found during the walk: y = E' - g (+ 2e-6 |g|)
        float Dv = LR_INF;
        if (thr.nQ && rw < na) {
            const float scale = thr.nQ[rowmap ? rowmap[rw] : rw] + max_nc;
            Dv = (1.05e-3f * scale + 4e-7f) + 8e-6f * scale;
        }
        s_D[tid] = Dv; s_N1[tid] = -LR_INF; s_N2[tid] = -LR_INF;
        if (tau) s_Y[tid] = rw < na ? 0.5f * tau[rw] : -LR_INF;          // rows past the end never pass the test
        else {
            // ---- phase 1: every sstride-th tile of the strip, rows on the lanes
            const int sstride = thr.sstride;
            const int nsamp = t_end > t_begin ? (t_end - t_begin + sstride - 1) / sstride : 0;     // tiles this block samples
            const int nsch = (nsamp + CH - 1) / CH;
            constexpr bool P1E = SIGN;      // (see fold() below)
            auto tile_s = [&](int c, int k) { return t_begin + (c * CH + k) * sstride; };
            auto load_s = [&](int c) {
#pragma unroll
                for (int q = 0; q < CH / 2; ++q) {
                    const int p = tid + 256 * q;
                    const int col = tile_s(c, p >> 7) * 32 + ((p >> 2) & 31);
                    stage[q] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(Hc) + (size_t)min(col, nb - 1) * 64 + (p & 3) * 16);
                }
                const int lc = tid & (CH * 32 - 1);
                const int col = tile_s(c, lc >> 5) * 32 + (lc & 31);
                if constexpr (!P1E) stage_n = nC[min(col, nb - 1)];      // (columns past the end repeat the last one: same tile, same norm)
            };
            auto store_s = [&](int buf) {
#pragma unroll
                for (int q = 0; q < CH / 2; ++q) {
                    const int p = tid + 256 * q;
                    *reinterpret_cast<f32x4 *>(&(buf ? lds_b : lds_a)[lr_lds_off(p >> 2, p & 3)]) = stage[q];
                }
                if constexpr (P1E) return;
                // largest x_j = n1[j]/2 of every tile of the chunk: the threads tid < CH*32 hold one column each, 32 per tile
                float xm = 0.5f * stage_n;
#pragma unroll
                for (int k = 16; k >= 1; k >>= 1) xm = fmaxf(xm, __shfl_xor(xm, k));
                if (tid < CH * 32 && (tid & 31) == 0) *reinterpret_cast<float *>(&(buf ? lds_b : lds_a)[XOFF + (tid >> 5) * 4]) = xm;
            };
            float m1[4], m2[4];           // running two largest (lane, tile) maxima of the lane's four rows (one per 16-row block)
#pragma unroll
            for (int rbk = 0; rbk < 4; ++rbk) { m1[rbk] = -LR_INF; m2[rbk] = -LR_INF; }
            // The phase is software-pipelined like the walk: two accumulator sets, the MFMAs of tile k are issued into one while tile k - 1
            // is folded out of the other (the fold is inline asm -- no canonicalising v_max x, x, x -- and reads its registers at least
            // eight MFMAs after they were issued: the hardware does not interlock a vector read of an MFMA result and the compiler does
            // not see inside the asm, so scheduling barriers pin the order); the LDS fragments of tile k + 1 are requested before the
            // MFMAs of tile k.  Unpipelined (round 3) the phase took 1.75x the walk's time per tile: 9 % of a block's lifetime.
            // P1E (all column norms alike): the lane keeps ELEMENTWISE running maxima of its accumulators instead -- register
            // g of row block rbk collects columns 4 kb + g and 16 + 4 kb + g of every sampled tile, one v_max3 per register and tile (16
            // per tile instead of 32 half-rate ops); the 4 registers x 4 lanes of a row are 16 disjoint column classes, so the two largest
            // class maxima belong to different columns, and g >= dot16 - max_nc / 2 bounds the filter value from below
            f32x4 runm[4];
#pragma unroll
            for (int rbk = 0; rbk < 4; ++rbk) runm[rbk] = f32x4{ -LR_INF, -LR_INF, -LR_INF, -LR_INF };
            auto fold = [&](const f32x4 &lo4, const f32x4 &hi4, int rbk, float xmax) {
                if constexpr (P1E) {
                    asm("v_max3_f32 %0, %0, %4, %8\n\tv_max3_f32 %1, %1, %5, %9\n\tv_max3_f32 %2, %2, %6, %10\n\tv_max3_f32 %3, %3, %7, %11"
                        : "+v"(runm[rbk][0]), "+v"(runm[rbk][1]), "+v"(runm[rbk][2]), "+v"(runm[rbk][3])
                        : "v"(lo4[0]), "v"(lo4[1]), "v"(lo4[2]), "v"(lo4[3]), "v"(hi4[0]), "v"(hi4[1]), "v"(hi4[2]), "v"(hi4[3]));
                    return;
                }
                float t, lo;
                asm("v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %0, %0, %7, %8\n\tv_max_f32 %0, %0, %9\n\t"
                    "v_sub_f32 %0, %0, %10\n\tv_min_f32 %1, %11, %0"
                    : "=&v"(t), "=&v"(lo)
                    : "v"(lo4[0]), "v"(lo4[1]), "v"(lo4[2]), "v"(lo4[3]), "v"(hi4[0]), "v"(hi4[1]), "v"(hi4[2]), "v"(hi4[3]), "v"(xmax), "v"(m1[rbk]));
                asm("v_max_f32 %0, %0, %1" : "+v"(m1[rbk]) : "v"(t));
                asm("v_max_f32 %0, %0, %1" : "+v"(m2[rbk]) : "v"(lo));
            };
            const f32x4 zero4 = { 0, 0, 0, 0 };
            const int frag16s = lr_lds_off(c16, kb);
            f32x4 sA[8], sB[8];           // [rbk]: columns 0..15 of the tile, [4 + rbk]: columns 16..31
            bool pend = false;            // the set that was filled last awaits its fold (false: its tile does not count)
            float pend_x = 0.0f;
            if (nsch > 0) {
                load_s(0); store_s(0);
                __syncthreads();
                for (int c = 0; c < nsch; ++c) {
                    const int buf = c & 1;
                    if (c + 1 < nsch) load_s(c + 1);
                    // fragments of tile k: lane (c, kb) reads piece kb of columns c and 16 + c; in the result the lane holds query row c
                    // of a row block and registers g <-> columns 4 kb + g of the column block
                    const unsigned char *lbuf = buf ? lds_b : lds_a;
                    const unsigned char *bp = &lbuf[frag16s];
                    f16x8 b0 = *reinterpret_cast<const f16x8 *>(bp), b1 = *reinterpret_cast<const f16x8 *>(bp + 16 * LR_LDS_ROW);
                    float xmax = *reinterpret_cast<const float *>(&lbuf[XOFF]);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        f16x8 n0 = b0, n1 = b1; float nx = xmax;
                        if (k + 1 < CH) {
                            n0 = *reinterpret_cast<const f16x8 *>(bp + (k + 1) * 32 * LR_LDS_ROW);
                            n1 = *reinterpret_cast<const f16x8 *>(bp + (k + 1) * 32 * LR_LDS_ROW + 16 * LR_LDS_ROW);
                            nx = *reinterpret_cast<const float *>(&lbuf[XOFF + (k + 1) * 4]);
                        }
                        f32x4 (&cur)[8] = (k & 1) ? sB : sA;
                        const f32x4 (&prev)[8] = (k & 1) ? sA : sB;
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rbk = 0; rbk < 4; ++rbk) {
                            cur[rbk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, a16[rbk], zero4, 0, 0, 0);
                            cur[4 + rbk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, a16[rbk], zero4, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (pend) {
#pragma unroll
                            for (int rbk = 0; rbk < 4; ++rbk) fold(prev[rbk], prev[4 + rbk], rbk, pend_x);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // Tiles past the end of the strip (the last chunk may reach beyond it) are not counted -- and neither is a
                        // PARTIAL last tile: its staged image repeats the cloud's last column in the padding, so a row whose best
                        // column is that one would get the same column as its best and its second best from two of the four lanes
                        // that share the row (the merge below relies on the lanes seeing disjoint columns) -- a threshold one
                        // neighbour too tight (found by tools/soak_fr.py, round 3; the walk of phase 2 still visits that tile).
                        pend = tile_s(c, k) < t_end && (tile_s(c, k) + 1) * 32 <= nb;
                        pend_x = xmax;
                        b0 = n0; b1 = n1; xmax = nx;
                    }
                    if (c + 1 < nsch) store_s(buf ^ 1);
                    __syncthreads();
                }
                if (pend) {      // the last tile sits in set B (CH is even); an explicit wait covers the MFMA write latency
                    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
                    for (int rbk = 0; rbk < 4; ++rbk) fold(sB[rbk], sB[4 + rbk], rbk, pend_x);
                }
            }
            if constexpr (P1E) {
                // the two largest of the lane's four class maxima per row block; x_j <= max_nc / 2 for every column
                const float xg = 0.5f * max_nc;
#pragma unroll
                for (int rbk = 0; rbk < 4; ++rbk) {
                    const float a = fmaxf(runm[rbk][0], runm[rbk][1]), b = fminf(runm[rbk][0], runm[rbk][1]);
                    const float c = fmaxf(runm[rbk][2], runm[rbk][3]), d = fminf(runm[rbk][2], runm[rbk][3]);
                    m1[rbk] = fmaxf(a, c) - xg;
                    m2[rbk] = fmaxf(fminf(a, c), fmaxf(b, d)) - xg;
                }
            }
            // the four lanes of a row (kb = 0..3: disjoint columns) merge their pairs in two exchange rounds; lane kb = 0 writes the row's
            // start value
#pragma unroll
            for (int rbk = 0; rbk < 4; ++rbk) {
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float c1 = __shfl_xor(m1[rbk], o), c2 = __shfl_xor(m2[rbk], o);
                    const float hi = fmaxf(m1[rbk], c1), lo = fminf(m1[rbk], c1);
                    m2[rbk] = fmaxf(lo, fmaxf(m2[rbk], c2));
                    m1[rbk] = hi;
                }
                const int rl = wave * 64 + 16 * rbk + c16, row = bx * LR_BLOCK_ROWS + rl;
                if (kb == 0) {
                    float yv = -LR_INF;
                    if (row < na) {
                        const float U = -2.0f * (thr.need >= 2 ? m2[rbk] : m1[rbk]);
                        const float scale = thr.nQ[rowmap ? rowmap[row] : row] + max_nc;
                        const float E = 1.05e-3f * scale + 4e-7f;
                        yv = 0.5f * (U + 2.0f * E + 6e-6f * scale + 2e-6f * fabsf(U));
                    }
                    s_Y[rl] = yv;
                }
            }
        }
        __syncthreads();
    }
    // Several column strips per row block (single-pair and small-batch calls): the strips' blocks run at the same time and each finds
    // thresholds for the same rows from its own columns.  A threshold is a property of the ROW -- any valid upper bound of its need-th
    // smallest u' holds for every column -- so the strips pool them: one atomicMin per row on an order-preserving integer image of y
    // (yshare, set to +inf by the prep kernel), here after the sample phase and then once per tightening round.  The returned value
    // is what the other strips had found by then.  (A rendezvous of the strips after the sample phase -- a bounded spin on an arrival
    // counter -- was measured and dropped: the strips of a row block are dispatched too far apart, 127 against 110 us for one pair.)
    const bool pooled = yshare != nullptr && thr.nQ != nullptr && my_strips > 1;
    if (pooled) {
        const int rw = bx * LR_BLOCK_ROWS + tid;
        if (rw < na) {
            const float yv = s_Y[tid];
            if (yv == yv) { const float other = lr_ord_dec(atomicMin(&yshare[rw], lr_ord_enc(yv))); if (other < yv) s_Y[tid] = other; }
        }
        __syncthreads();
    }

    // the lane's 4 x 4 threshold registers: register g of row block rbk <-> row 16 rbk + 4 kb + g of the wave
    f32x4 y4[4];
    auto load_y = [&]() {
#pragma unroll
        for (int rbk = 0; rbk < 4; ++rbk) y4[rbk] = *reinterpret_cast<const f32x4 *>(&s_Y[wave * 64 + 16 * rbk + 4 * kb]);
#pragma unroll
        for (int rbk = 0; rbk < 4; ++rbk) y4[rbk] -= xhat;          // (xhat = 0 unless SIGN: x - 0 is x)
    };
    load_y();
    LR_PROBE(tk_walk = __builtin_amdgcn_s_memrealtime();)
    // Staging of the walk: buffer loads STRAIGHT INTO LDS (buffer_load_dwordx4 ... lds: no staging registers, no ds_write; the chunk's
    // position is a scalar offset; rows past the end of the cloud read as zeros through the range check of the buffer descriptor, and
    // whatever such a column -- or a column past the end of the strip -- makes of the test is masked by derive() and flush()).
    // A wave instruction of 64 lanes x 16 bytes fills 1 KB of LDS from M0 onwards in lane order = 16 columns; the XOR swizzle of the
    // LDS image is folded into the lane's GLOBAL address: lane L fetches piece (L & 3) ^ ((L >> 3) & 3) of column L >> 2 of its group.
    // Wave w owns the groups w NQ .. w NQ + NQ - 1 of a chunk; consecutive groups are 1 KB apart in both address spaces (the
    // instruction's offset field applies to both).  The loads of chunk c + 2 are issued right behind the barrier of chunk c (every read
    // of the buffer they go to was complete before that barrier) and awaited -- vmcnt(0) -- before the barrier of chunk c + 1.
    static_assert(CH % 4 == 0, "the staging maps 4 waves onto groups of 16 columns");
    constexpr int NQ = CH * 2 / 4;          // 1 KB groups per wave and chunk
    const __amdgpu_buffer_rsrc_t rsrcH = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(Hc), 0, nb * 64, 0x27000);
    const __amdgpu_buffer_rsrc_t rsrcN = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(nC), 0, nb * 4, 0x27000);
    const int ld_voff = wave * NQ * 1024 + (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) << 4);
    const int st_noff = (tid & (CH * 32 - 1)) * 4;
    auto load_chunk = [&](auto bufc, int c) {
        constexpr int buf = decltype(bufc)::value;
        const int col0 = (t_begin + c * CH) * 32;       // wave-uniform
        __attribute__((address_space(3))) void *dst = (__attribute__((address_space(3))) void *)((buf ? lds_b : lds_a) + wave * NQ * 1024);
        static_assert(NQ == 2 || NQ == 4, "chunks of 4 or 8 tiles");      // (the offset field must be a literal)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcH, dst, 16, ld_voff, col0 * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcH, dst, 16, ld_voff, col0 * 64, 1024, 0);
        if constexpr (NQ == 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcH, dst, 16, ld_voff, col0 * 64, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcH, dst, 16, ld_voff, col0 * 64, 3072, 0);
        }
        if constexpr (!SIGN) stage_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcN, st_noff, col0 * 4, 0));
    };
    // the chunk's loads have landed (caller: vmcnt(0)); what is left is the plain form's x_j = n1[j]/2 (+inf masks columns past the end
    // of the cloud or of the strip)
    auto finish_chunk = [&](auto bufc, int c) {
        constexpr int buf = decltype(bufc)::value;
        if constexpr (!SIGN) {
            const int col = (t_begin + c * CH) * 32 + (tid & (CH * 32 - 1));
            const bool ok = col < nb && (col >> 5) < t_end;
            if (tid < CH * 32) *reinterpret_cast<float *>(&(buf ? lds_b : lds_a)[XOFF + tid * 4]) = ok ? 0.5f * stage_n : LR_INF;
        }
    };
    typedef std::integral_constant<int, 0> c0_t;
    typedef std::integral_constant<int, 1> c1_t;
    // per-lane byte offsets of the fragment / x_j of tile 0 of the current and the other buffer; tile k adds a constant.  The lane
    // reads piece kb of columns c and 16 + c of the tile, and their x_j
    const int frag16 = lr_lds_off(c16, kb), x_lane = XOFF + c16 * 4;
    auto read_b = [&](auto bufc, int k, f16x8 &b0, f16x8 &b1, f32x2 &xj) {
        constexpr int buf = decltype(bufc)::value;
        const unsigned char *lbuf = buf ? lds_b : lds_a;          // (buf is a compile-time value)
        b0 = *reinterpret_cast<const f16x8 *>(&lbuf[frag16 + k * 32 * LR_LDS_ROW]);
        b1 = *reinterpret_cast<const f16x8 *>(&lbuf[frag16 + k * 32 * LR_LDS_ROW + 16 * LR_LDS_ROW]);
        if constexpr (!SIGN) {      // (the sign form of the test reads no per-column operand)
            xj.x = *reinterpret_cast<const float *>(&lbuf[x_lane + k * 32 * 4]);
            xj.y = *reinterpret_cast<const float *>(&lbuf[x_lane + k * 32 * 4 + 64]);
        }
    };
    int wcnt = 0;            // entries in this wave's list (wave-uniform: lives in a scalar register)
    const unsigned lanecode = (unsigned)c16 | ((unsigned)kb << 22);      // the lane's part of an entry: column within the 16-block | lane group
    // The slow paths (derive(), flush(); a hit in the plain form) work out the lane's coordinates afresh: values that only they use would
    // otherwise be kept in registers across the hot loop (or spilled to scratch memory and fetched back on every visit: +4 us per pair)
    auto cold_lane = [&]() { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; };
    int wdone = 0;           // ... of which the tightening has seen this many
    int trig = LR_PB_TIGHTEN;      // new entries that make the wave ask for a tightening round
    // what the wave's list wants (bit 0: a tightening round, bit 1: to be emptied): recomputed where wcnt / wdone change (the slow path of
    // a hit, derive(), flush()) -- not at every chunk -- and posted to the block only when it differs from what was posted last
    int wposted = 0, wnext = 0;      // the wish posted last; the fill count from which the list wants something
    // The wave owns one segment of the candidate store: seg[(row block, wave, strip)][seg_cap] entries { column, code | mask }
    // exactly as they lie in its LDS list.  Emptying the list is a compacting copy with plain stores -- no atomics, nothing
    // to wait for; nn16_exact_kernel expands the masks and bins the entries by row.  seg_fill < 0: the segment overflowed
    // (duplicate-heavy input), the wave's 64 rows go through the exact full-row scan.
    // (the LR_NN16_SEG entries of the wave's rows are divided among the strips the row block really uses)
    const int seg_cap = lr_seg_cap(my_strips);
    uint2 *__restrict__ seg = reinterpret_cast<uint2 *>(cand) + (size_t)(bx * 4 + wave) * LR_NN16_SEG + (size_t)by * seg_cap;
    int seg_fill = 0;
    const bool tightening = thr.nQ != nullptr;
    auto my_wish = [&]() { return (wcnt >= WL / 2 ? 2 : 0) | ((tightening && wcnt - wdone >= trig) ? 1 : 0); };
    auto set_next = [&]() { wnext = tightening ? min(wdone + trig, WL / 2) : WL / 2; };      // (my_wish() != 0 <=> wcnt >= wnext)
    set_next();
    // One round over the new entries [wdone, wcnt) of the wave's list (wave-local).  The slow path of the walk only parks { column,
    // register group } of a hit; WHICH of the group's 8 rows passed, and with what filter value, is worked out here, 16 entries at a
    // time, by the matrix pipe itself: lane (c, kb) fetches K slice kb of the column of entry e0 + c (one 16-byte gather), four
    // v_mfma_f32_16x16x32_f16 with the row fragments and the thresholds of NOW give it y_i + dot16 of that column against rows
    // 16 rbk + 4 kb + 0..3 -- the same instruction on the same operands as the walk, so the same bits -- and the lane whose kb is the
    // entry's compares the group's 8 registers with x_j: 2 vector instructions per entry and register instead of 16 per hit and wave.
    // Thresholds only ever decrease, so a mask taken against the thresholds of now is a subset of the one the walk saw and still a
    // superset of what the final thresholds admit; an entry that no longer passes any row is dropped on the spot.
    // update: (forward direction, and the reverse one when it is given the rows' error terms) the two largest g of every row ->
    // y -> threshold registers.  An entry with a single row keeps its g = dot16 - x_j for the drop tests of flush() and
    // nn16_exact_kernel.
    auto derive = [&](bool update) {
        const int lane = cold_lane(), c16 = lane & 15, kb = lane >> 4;      // (shadow the kernel's)
        const int nlist = min(wcnt, WL);
        __builtin_amdgcn_s_setprio(3);      // the wave's three siblings wait for it at the next chunk barrier (60.1 -> 59.3 us per pair)
        LR_PROBE(++n_rounds; n_groups += (nlist - wdone + 15) >> 4; const unsigned long long tk0 = __builtin_amdgcn_s_memrealtime();)
        constexpr int DG = SIGN ? LR_PB_DGROUPS : (LR_PB_DGROUPS > 2 ? 2 : LR_PB_DGROUPS);      // (the plain form holds the x_j operands too: a third group would spill)
        for (int e0 = wdone; e0 < nlist; e0 += 16 * DG) {
            // a few groups of 16 entries per pass: all their gathers are in flight before the first MFMA (one L2 latency per pass)
            uint2 v[DG]; f16x8 bf[DG]; float xn[DG];
#pragma unroll
            for (int g = 0; g < DG; ++g) {
                v[g] = wlist[wave][min(e0 + 16 * g + c16, WL - 1)];
                const int col = (int)(v[g].x & LR_PB_COLMASK);
                bf[g] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcH, col * 64 + kb * 16, 0, 0));
                xn[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcN, col * 4, 0, 0));
            }
            LR_PROBE({ const unsigned long long ta = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); n_tk_a += (int)(__builtin_amdgcn_s_memrealtime() - ta); })
#pragma unroll
            for (int g = 0; g < DG; ++g) {
                if (e0 + 16 * g >= nlist) break;              // (wave-uniform)
                const int e = e0 + 16 * g + c16;
                const bool valid = e < nlist;
                const int col = (int)(v[g].x & LR_PB_COLMASK);
                // padding columns (past the end of the cloud or of the strip) pass the walk's test only when the threshold is +inf
                const float x = (valid && col < nb && (col >> 5) < t_end) ? 0.5f * xn[g] - xhat : LR_INF;      // (xhat = 0 unless SIGN)
                f32x4 d[4];
#pragma unroll
                for (int rbk = 0; rbk < 4; ++rbk) d[rbk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[rbk], bf[g], y4[rbk], 0, 0, 0);
                if (valid && lr_pb_kb(v[g].x) == kb) {
                    // bit 4 rbk + k <-> register k of row block rbk; set unless acc < x (a NaN accumulator -- non-finite f16 operands --
                    // stays a candidate: rows with such operands are re-done by the exact full-row scan, columns only add candidates
                    // the exact stage orders)
                    unsigned mask = 0u;
#pragma unroll
                    for (int rbk = 0; rbk < 4; ++rbk)
#pragma unroll
                        for (int k = 0; k < 4; ++k) mask |= (d[rbk][k] < x) ? 0u : (1u << (4 * rbk + k));
                    uint2 w = make_uint2(v[g].x, mask);              // mask 0: the entry is dropped when the list is emptied
                    if (update && mask != 0u && (mask & (mask - 1u)) == 0u) {
                        const int b = __builtin_ctz(mask);
                        const int rl = wave * 64 + lr_pb_row(kb, b);
                        // (the one register that passed is the largest of the 16)
                        float m = fmaxf(fmaxf(fmaxf(d[0][0], d[0][1]), fmaxf(d[0][2], d[0][3])), fmaxf(fmaxf(d[1][0], d[1][1]), fmaxf(d[1][2], d[1][3])));
                        m = fmaxf(m, fmaxf(fmaxf(fmaxf(d[2][0], d[2][1]), fmaxf(d[2][2], d[2][3])), fmaxf(fmaxf(d[3][0], d[3][1]), fmaxf(d[3][2], d[3][3]))));
                        // the register holds y_row + dot16, so g = dot16 - x_j = (register - x_j) - y_row
                        const float gv = (m - x) - s_Y[rl];
                        const float old = atomicMax(&s_N1[rl], gv);
                        atomicMax(&s_N2[rl], fminf(old, gv));
                        // (kept rounded UP to 16 bits next to the mask: the form the candidate store carries; the drop tests -- flush() now, the
                        // exact kernel against the final threshold -- then only ever keep more than the exact value would)
                        const unsigned gb = __float_as_uint(gv);
                        const unsigned up16 = (gb & 0x80000000u) ? (gb >> 16) : ((gb + 0xffffu) >> 16);      // towards +inf
                        w.y |= up16 << 16;
                        w.x |= LR_PB_HASG;
                    }
                    wlist[wave][e] = w;
                }
            }
        }
        LR_PROBE(const unsigned long long tb = __builtin_amdgcn_s_memrealtime();)
        wdone = nlist;
        if (false && update) trig = min(2 * trig, 192);
        set_next();
        if (update) {
            // lane = row: y <- min(y, E' - g_need + 2e-6 |g_need|)   (g_need: the need-th largest g of the walk so far; -inf: no change)
            const int rl = wave * 64 + lane;
            const float gn = thr.need >= 2 ? s_N2[rl] : s_N1[rl];
            const float yn = (s_D[rl] - gn) + 2e-6f * fabsf(gn);
            float yv = s_Y[rl];
            if (yn < yv) yv = yn;                           // (NaN compares false: the row keeps its threshold)
            // (the exchange as a plain load at the start of the round + an atomic nobody waits for: no change, 96.4 us either way -- round 5)
            if (pooled && row0 + lane < na && yv == yv) {   // pool with the row block's other strips
                const float other = lr_ord_dec(atomicMin(&yshare[row0 + lane], lr_ord_enc(yv)));
                if (other < yv) yv = other;
            }
            s_Y[rl] = yv;
            load_y();
        }
        LR_PROBE(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); n_tk_b += (int)(__builtin_amdgcn_s_memrealtime() - tb); n_tk_derive += (int)(__builtin_amdgcn_s_memrealtime() - tk0);)
        __builtin_amdgcn_s_setprio(0);
    };
    auto flush = [&]() {
        if (wdone < wcnt) derive(tightening);
        const int lane = cold_lane();
        LR_PROBE(const unsigned long long tkf = __builtin_amdgcn_s_memrealtime();)
        if (wcnt > WL) seg_fill = -1;       // more hits between two chunk boundaries than the list holds
        else if (seg_fill >= 0) {
            for (int e0 = 0; e0 < wcnt; e0 += 64) {
                const int e = e0 + lane;
                uint2 v = make_uint2(0u, 0u);
                if (e < wcnt) v = wlist[wave][e];
                const int col = (int)(v.x & LR_PB_COLMASK);
                // padding columns pass the test only when tau is +inf
                bool keep = e < wcnt && (v.y & 0xffffu) != 0u && col < nb && (col >> 5) < t_end;
                if (v.x & LR_PB_HASG) {
                    // g of the entry against the row's threshold of NOW (candidate <=> g >= -y): what an earlier, looser threshold let
                    // in is dropped here (g as derive() left it: rounded UP to 16 bits); the exact kernel repeats the test against the final threshold
                    const float gv = __uint_as_float(v.y & 0xffff0000u);
                    const int rl = wave * 64 + lr_pb_row(lr_pb_kb(v.x), __builtin_ctz((v.y & 0xffffu) | 0x8000u));
                    const float yr = s_Y[rl];
                    if (gv + 1e-5f * (fabsf(gv) + fabsf(yr)) < -yr) keep = false;
                }
                const unsigned long long kb = __builtin_amdgcn_ballot_w64(keep);
                const int pos = seg_fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u));
                const int nk = __builtin_popcountll(kb);
                if (seg_fill + nk > seg_cap) { seg_fill = -1; break; }
                if (keep) { if (colmap) v.x = (unsigned)colmap[col] | (v.x & ~LR_PB_COLMASK); seg[pos] = v; }
                seg_fill += nk;
            }
        }
        wcnt = 0; wdone = 0; set_next();
        LR_PROBE(n_tk_flush += (int)(__builtin_amdgcn_s_memrealtime() - tkf);)
    };
    // candidate test of the lane's 16 accumulator registers of column block cb (16 columns x the wave's 64 rows).
    // A hit only parks { column, lane group } -- a handful of vector instructions and one LDS write; everything else about it is worked
    // out later, 16 entries per instruction group (derive()).
    auto check = [&](const f32x4 &r0, const f32x4 &r1, const f32x4 &r2, const f32x4 &r3, float x, int tile, int cb) {
        bool mine_hit;
        if constexpr (SIGN) {
            // two interleaved chains of three-input ANDs over the 16 sign bits (one asm block: no hazard padding between the halves)
            int sa, sb;
            asm("v_bitop3_b32 %0, %2, %3, %4 bitop3:0x80\n\tv_bitop3_b32 %1, %10, %11, %12 bitop3:0x80\n\t"
                "v_bitop3_b32 %0, %0, %5, %6 bitop3:0x80\n\tv_bitop3_b32 %1, %1, %13, %14 bitop3:0x80\n\t"
                "v_bitop3_b32 %0, %0, %7, %8 bitop3:0x80\n\tv_bitop3_b32 %1, %1, %15, %16 bitop3:0x80\n\t"
                "v_bitop3_b32 %0, %0, %1, %9 bitop3:0x80\n\tv_and_b32 %0, %0, %17"
                : "=&v"(sa), "=&v"(sb)
                : "v"(r0[0]), "v"(r0[1]), "v"(r0[2]), "v"(r0[3]), "v"(r1[0]), "v"(r1[1]), "v"(r1[2]), "v"(r1[3]),
                  "v"(r2[0]), "v"(r2[1]), "v"(r2[2]), "v"(r2[3]), "v"(r3[0]), "v"(r3[1]), "v"(r3[2]), "v"(r3[3]));
            mine_hit = sa >= 0;                        // the AND of the sign bits is clear: some register is not negative
        } else {
            float ma, mb;
            asm("v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %10, %11, %12\n\t"
                "v_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %13, %14\n\t"
                "v_max3_f32 %0, %0, %7, %8\n\tv_max3_f32 %1, %1, %15, %16\n\t"
                "v_max3_f32 %0, %0, %1, %9\n\tv_max_f32 %0, %0, %17"
                : "=&v"(ma), "=&v"(mb)
                : "v"(r0[0]), "v"(r0[1]), "v"(r0[2]), "v"(r0[3]), "v"(r1[0]), "v"(r1[1]), "v"(r1[2]), "v"(r1[3]),
                  "v"(r2[0]), "v"(r2[1]), "v"(r2[2]), "v"(r2[3]), "v"(r3[0]), "v"(r3[1]), "v"(r3[2]), "v"(r3[3]));
            mine_hit = ma >= x;
        }
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(mine_hit);
        if (__builtin_expect(hit != 0ull, 0)) {       // rare: keeps the common path a fall-through
            if (mine_hit) {
                const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hit >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hit, 0u));
                // (the column is worked out here, not on the fast path: tile is wave-uniform)
                unsigned lc = lanecode;
                if constexpr (!SIGN) { const int ln = cold_lane(); lc = (unsigned)(ln & 15) | ((unsigned)(ln >> 4) << 22); }      // (the plain form has no register to spare for it)
                if (pos < WL) wlist[wave][pos] = make_uint2((unsigned)(tile * 32 + cb * 16) | lc, 0u);
            }
            wcnt += __builtin_popcountll(hit);
            LR_PROBE(++n_visits; n_hits += __builtin_popcountll(hit);)
        }
    };

    f16x8 b0, b1;
    f32x2 xN = { LR_INF, LR_INF }, xC = { LR_INF, LR_INF };
    // ONE set of accumulators (32 registers): a group of 8 is tested -- it holds the previous tile -- and then handed to the two MFMAs
    // that overwrite it with the current tile, so every test reads its registers six MFMAs after they were issued and nothing is
    // double-buffered (the 32 registers this saves are what lets a fourth wave onto the SIMD)
    f32x4 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[q][cb][g] = -LR_INF;

    // one pipeline step: tests of the previous tile group by group, each followed by the MFMAs of tile (c, k) into the registers just
    // tested; LDS read of the next tile (par = c & 1: the buffer chunk c lies in -- a compile-time value, the loop below is unrolled by
    // two chunks so that no buffer offset is computed, swapped or selected at run time)
    auto step = [&](int c, int k, auto parc) {
        constexpr int par = decltype(parc)::value;
        f16x8 n0, n1; f32x2 nx;
        if (k + 1 < CH) read_b(std::integral_constant<int, par>{}, k + 1, n0, n1, nx);
        else read_b(std::integral_constant<int, par ^ 1>{}, 0, n0, n1, nx);
        const int tileC = t_begin + c * CH + k - 1;
#define MF(rbk, cb, b) acc[rbk][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[rbk], b, y4[rbk], 0, 0, 0)
        // (the scheduling barriers keep the MFMAs behind the test of the registers they overwrite: hoisted above it they would need
        // a second set of accumulators -- and the kernel its row fragments from scratch memory)
        check(acc[0][0], acc[1][0], acc[2][0], acc[3][0], xC.x, tileC, 0);
        __builtin_amdgcn_sched_barrier(0);
        MF(0, 0, b0); MF(1, 0, b0); MF(2, 0, b0); MF(3, 0, b0);
        __builtin_amdgcn_sched_barrier(0);
        check(acc[0][1], acc[1][1], acc[2][1], acc[3][1], xC.y, tileC, 1);
        __builtin_amdgcn_sched_barrier(0);
        MF(0, 1, b1); MF(1, 1, b1); MF(2, 1, b1); MF(3, 1, b1);
        __builtin_amdgcn_sched_barrier(0);
#undef MF
        b0 = n0; b1 = n1; xC = xN; xN = nx;
    };
    auto drain = [&]() {
        const int tileC = t_begin + nchunks * CH - 1;
        check(acc[0][0], acc[1][0], acc[2][0], acc[3][0], xC.x, tileC, 0);
        check(acc[0][1], acc[1][1], acc[2][1], acc[3][1], xC.y, tileC, 1);
    };
    if (nchunks > 0) {
        load_chunk(c0_t{}, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): chunk 0 has landed (and everything older: row fragments, thresholds)
        finish_chunk(c0_t{}, 0);
        if (nchunks > 1) load_chunk(c1_t{}, 1);
        if (tid == 0) s_att[0] = 0u;
        __syncthreads();
        read_b(c0_t{}, 0, b0, b1, xN);
        // The walk is a loop nest: the inner loop is the hot one and contains no tightening code (the compiler then keeps the threshold
        // registers loop-invariant and its wait counts exact); it is left whenever a hit list wants attention.
        // The four waves of a block attend to their lists TOGETHER: a wave in derive() keeps its three siblings waiting
        // at the next chunk barrier, so 4 x ~11 rounds per block, one wave at a time, stall the block four times as often as ~12 rounds
        // that all four take at once.  Every wave keeps what its list wants in one byte of an LDS word (written only when it changes),
        // reads all four wishes behind the chunk barrier and acts at the end of the chunk.  (The decision need not be block-uniform: a wave
        // acts on its own list only, and neither derive() nor flush() contains a barrier.)
        // What a chunk boundary costs is the wave's own instruction stream (round 5, tools/r5_pmc_micro.sh: a wave issues one instruction
        // per ~5 cycles whatever its class, and at three waves per SIMD it has 48 cycles per MFMA): per 32 MFMAs the boundary used to
        // be 34 scalar + 12 vector instructions -- bounds tests, buffer parity, exec save / restore around the one-lane write, materialised
        // booleans.  Now: the two chunks of a buffer pair are separate code (offsets are immediates), the last two chunks of the strip
        // run as a tail with the bounds tests, and a wish is posted only when it changes.
        int c = 0;
        int wish = 0;
        auto chunk = [&](auto parc, auto tailc) {
            constexpr int par = decltype(parc)::value;
            constexpr bool tail = decltype(tailc)::value != 0;
            unsigned wishes = 0u;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (k == CH - 1) {
                    // the last step of a chunk reads the first fragment of the next one: make that chunk visible now.  All
                    // reads of the buffer it goes to were issued before the previous barrier (the step above read this
                    // chunk's own last tile), so one barrier per chunk still orders everything.
                    if (!tail || c + 1 < nchunks) { __builtin_amdgcn_s_waitcnt(0x0F70); finish_chunk(std::integral_constant<int, par ^ 1>{}, c + 1); }      // vmcnt(0): chunk c + 1 has landed
                    // (one scalar comparison per chunk; the wish is worked out and written when the list reaches the mark; a round
                    // follows at the end of this very chunk -- the wave reads its own byte behind the barrier -- and takes it back)
                    if (__builtin_expect(wcnt >= wnext, 0)) {
                        const int wm = my_wish();
                        if (lane == 0) reinterpret_cast<unsigned char *>(&s_att[0])[wave] = (unsigned char)wm;
                        wposted = wm; wnext = 0x7fffffff;
                    }
                    __syncthreads();
                    // (requested here, looked at behind the chunk's last step)
                    wishes = s_att[0];
                    if (!tail || c + 2 < nchunks) load_chunk(std::integral_constant<int, par>{}, c + 2);      // (into the buffer chunk c has just been read out of)
                }
                step(c, k, parc);
            }
            // (the four bytes as they are: which bit is set is looked at outside the loop)
            wish = (int)__builtin_amdgcn_readfirstlane(wishes);
            ++c;
        };
        typedef std::integral_constant<int, 0> body_t;
        typedef std::integral_constant<int, 1> tail_t;
        while (c < nchunks) {
            wish = 0;
            // The hot loop takes two chunks per iteration, buffer 0 then buffer 1, as straight-line code (`body`: chunks c + 1 and c + 2
            // exist, nothing is tested).  An odd chunk on its own -- the walk resumes at one after a round, or it is the strip's last --
            // and the last two chunks of the strip run through the `tail` form with the bounds tests.
            if (c & 1) chunk(c1_t{}, tail_t{});
            else if (c + 3 < nchunks) {
                do { chunk(c0_t{}, body_t{}); if (wish) break; chunk(c1_t{}, body_t{}); } while (!wish && c + 3 < nchunks);
            } else chunk(c0_t{}, tail_t{});
            wish = (wish & 0x02020202) ? 2 : (wish ? 1 : 0);
            if (wish && wposted) { if (lane == 0) reinterpret_cast<unsigned char *>(&s_att[0])[wave] = 0; wposted = 0; }
            if (wish & 2) flush();
            else if (wish) { if (wdone < wcnt) derive(true); }
        }
        // drain: the last tile of the last chunk sits in the accumulators.  The inline-asm tests below read MFMA results the
        // compiler cannot see them read (no automatic wait states): inside the loop every such read is at least two
        // MFMAs behind its producer; here an explicit wait covers the 8-pass MFMA write latency.
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        drain();
        flush();           // (with a last tightening round: the entries get their g, the rows their final thresholds)
    }
    if (lane == 0) {
        int32_t *cw = cand_cnt + (bx * 4 + wave) * (pg.gy + 1);
        cw[by] = seg_fill;                     // entries in this wave's segment (< 0: overflow)
        if (by == 0) cw[pg.gy] = my_strips;    // how the wave's store is divided (read by nn16_exact_kernel)
    }
    LR_PROBE(if (lane == 0) {
        atomicAdd(&lr_pb_stat[0], 1ull); atomicAdd(&lr_pb_stat[1], (unsigned long long)nchunks * CH * 4); atomicAdd(&lr_pb_stat[2], (unsigned long long)n_visits);
        atomicAdd(&lr_pb_stat[3], (unsigned long long)n_hits); atomicAdd(&lr_pb_stat[4], (unsigned long long)n_rounds); atomicAdd(&lr_pb_stat[5], (unsigned long long)n_groups);
        const unsigned long long tk_end = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&lr_pb_stat[6], (unsigned long long)n_tk_derive); atomicAdd(&lr_pb_stat[7], (unsigned long long)n_tk_flush);
        atomicAdd(&lr_pb_stat[8], tk_walk - tk_start); atomicAdd(&lr_pb_stat[9], tk_end - tk_start);
        atomicAdd(&lr_pb_stat[10], (unsigned long long)n_tk_a); atomicAdd(&lr_pb_stat[11], (unsigned long long)n_tk_b);
    })
    // the rows' final thresholds (wave-local: every wave writes its own 64 rows): nn16_exact_kernel drops the entries they exclude
    if (yfin) {
        const int rw = row0 + lane;
        if (rw < na) yfin[(size_t)by * yfin_stride + rw] = tightening ? s_Y[wave * 64 + lane] : LR_INF;
    }
    if (threadIdx.x == 0) {
        const lr_pb_tail tl = lr_pb_tail_args();
        if (tl.clk) {
            atomicAdd(&tl.clk[0], __builtin_amdgcn_s_memtime() - s_clk[0]);
            atomicAdd(&tl.clk[1], __builtin_amdgcn_s_memrealtime() - s_clk[1]);
        }
    }
}

// ------------------------------------------------------------------ exact verification of the candidates
// A block takes the 64 query rows of one pass-B wave.  Their fp32 descriptors are staged in LDS once; then every thread
// takes ENTRIES of that wave's segments (one segment per column strip): it gathers the entry's column row (128 B) once and,
// for every register bit of the mask, forms the fp32 fma-chain distance to that query row.  The two best candidates of a row
// under the (sqrt value, index) order -- torch.min's "first minimal value" -- are kept as 64-bit keys (value bits << 32 |
// index) with two LDS atomics per candidate:  old = atomicMin(best, key);  atomicMin(second, max(old, key)).  Whatever the
// order of arrival, `best` ends as the smallest key and `second` as the second smallest (every loser max(old, key) is at
// least the second smallest, and the second smallest itself loses exactly once).  The work is spread over the threads by
// entry, so a row with many candidates does not stall its neighbours, and nothing is binned or sorted.
// Rows whose segment overflowed, whose list is too short, or whose f16 copy is not finite are re-done by the whole block
// with an exact scan of all columns (slow, rare, and by construction the reference answer).
#define LR_EX_ROWS 64
#define LR_EX_STRIDE 33          // floats per staged query row (odd: conflict-free column-wise reads)

__global__ void __launch_bounds__(256)
nn16_exact_kernel(const float *__restrict__ Fq, const float *__restrict__ nQ, int na,
                  const float *__restrict__ Fc, const float *__restrict__ nC, int nb,
                  const int32_t *__restrict__ cand_cnt, const int32_t *__restrict__ cand, int nstrips, int need,
                  const float *__restrict__ yfin, int yfin_stride,
                  const int32_t *__restrict__ rowmap, const int32_t *__restrict__ na_dev, lr_ex_out out,
                  int32_t *__restrict__ counters, const float *__restrict__ range_c, int dir, int gx, int total, lr_zargs z)
{
    // 1-D XCD-aware grid -> (row block, pair): the blocks one XCD receives are consecutive row blocks of the same pairs, so the
    // column cloud they gather from (3.84 MB of fp32 rows at 30k points) stays in that XCD's L2
    int logical;
    if (!lr_xcd_block(total, logical)) return;
    const int bxi = logical % gx, pair = logical / gx;
    __shared__ float s_a[LR_EX_ROWS * LR_EX_STRIDE];
    __shared__ float s_nq[LR_EX_ROWS];
    __shared__ unsigned long long s_best[LR_EX_ROWS], s_second[LR_EX_ROWS];
    __shared__ int s_cnt[LR_EX_ROWS], s_rowd[LR_EX_ROWS], s_badrow[LR_EX_ROWS];
    __shared__ int s_bad, s_skip, s_nredo, s_redo[LR_EX_ROWS];
    if (z.descs) {      // dir 0: rows = cloud 0 against cloud 1; 1: the reverse direction
        const lr_pair_desc d = z.descs[pair];
        Fq = dir ? d.F1 : d.F0; na = dir ? d.n1 : d.n0; Fc = dir ? d.F0 : d.F1; nb = dir ? d.n0 : d.n1;
    }
    lr_z(nQ, z, pair); lr_z(nC, z, pair); lr_z(cand_cnt, z, pair); lr_z(cand, z, pair); lr_z(rowmap, z, pair); lr_z(yfin, z, pair);
    lr_z(na_dev, z, pair); lr_z(out.idx1, z, pair); lr_z(out.idx2, z, pair); lr_z(out.s1o, z, pair); lr_z(out.s2o, z, pair);
    lr_z(counters, z, pair); lr_z(out.seed_out, z, pair); lr_z(out.seed_s1, z, pair); lr_z(out.seed_range, z, pair); lr_z(out.seed64, z, pair); lr_z(range_c, z, pair);
    unsigned long long *__restrict__ const seed64 = out.seed64;
    if (na_dev) na = *na_dev;                  // compacted row list (reverse direction): lists by position, data by rowmap[]
    const int row0 = bxi * LR_EX_ROWS;
    if (row0 >= na) return;
    const int tid = threadIdx.x;
    // a filter pass that was launched in one form only while this call's norms asked for the other walked nothing -- and then NOBODY wrote
    // this call's candidate counts: the store is not looked at (s_skip below), every row goes through the full scan
    const int miss = counters[LR_CNT_FORM_MISS_F + (dir ? 1 : 0)] != 0 ? 1 : 0;
    // ---- stage the query rows: thread t moves floats [8 (t & 3) .. +8) of row t >> 2
    {
        const int rl = tid >> 2, part = tid & 3;
        const int rowc = min(row0 + rl, na - 1);
        const int rowd = rowmap ? rowmap[rowc] : rowc;
        const f32x4 *pa = reinterpret_cast<const f32x4 *>(Fq + (size_t)rowd * 32 + 8 * part);
        const f32x4 u = pa[0], v = pa[1];
        float *dst = &s_a[rl * LR_EX_STRIDE + 8 * part];
        dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w; dst[4] = v.x; dst[5] = v.y; dst[6] = v.z; dst[7] = v.w;
        const bool bad = ex_bad8(u, v);          // a query row whose f16 copy is not finite never produced a meaningful filter value
        if (tid < LR_EX_ROWS) { s_best[tid] = LR_EX_EMPTY; s_second[tid] = LR_EX_EMPTY; s_cnt[tid] = 0; s_badrow[tid] = 0; }
        // (a column cloud with a norm that is not finite -- the prep kernel's minimum is negative then --: every row by the full scan)
        // ... and so does a form miss (above)
        if (tid == 0) {
            s_bad = ((range_c != nullptr && range_c[1] < 0.0f) || miss) ? 1 : 0; s_skip = miss; s_nredo = 0;
        }
        if (part == 0) { s_nq[rl] = nQ[rowd]; s_rowd[rl] = rowd; }
        __syncthreads();
        if (bad) s_badrow[rl] = 1;
        // reverse direction seeded by this library's forward pass: the best of the points that POINT AT the row is known (distance and
        // smallest index at that distance); the scan below only adds points that do not point at it (see lr_nn16_reverse)
        if (dir == 1 && seed64 && tid < LR_EX_ROWS && row0 + tid < na) { s_best[tid] = seed64[s_rowd[tid]]; s_cnt[tid] = 1; }
    }
    __syncthreads();
    // exact distance of staged row rl to a column row held in registers, the arithmetic contract's fma chain
    auto dist = [&](int rl, const f32x4 (&t)[8], float ncj) {
        const float *a = &s_a[rl * LR_EX_STRIDE];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc = __builtin_fmaf(a[4 * k], t[k].x, acc);
            acc = __builtin_fmaf(a[4 * k + 1], t[k].y, acc);
            acc = __builtin_fmaf(a[4 * k + 2], t[k].z, acc);
            acc = __builtin_fmaf(a[4 * k + 3], t[k].w, acc);
        }
        const float d2 = __builtin_fmaf(-2.0f, acc, s_nq[rl] + ncj);
        return __builtin_sqrtf(fmaxf(d2, 1e-30f));
    };
    // ---- the candidates: entries { column | kb << 22 | LR_PB_HASG, 16-bit row mask | g16 << 16 } of the filter-pass wave (LR_PB_* above)
    if (!s_skip) {
        const int32_t *cw = cand_cnt + bxi * (nstrips + 1);
        const int used = min(max(cw[nstrips], 1), nstrips);        // strips the pass-B row block really used
        const int seg_cap = lr_seg_cap(used);
        // the rows' final thresholds of the filter pass (the tightest over the strips: a threshold is a property of the row, any valid
        // one applies to all of its entries): an entry that carries its filter value g and fails g >= -y is not a candidate any more
        __shared__ float s_yf[LR_EX_ROWS];
        if (tid < LR_EX_ROWS) {
            float yv = LR_INF;
            if (yfin && row0 + tid < na) for (int sidx = 0; sidx < used; ++sidx) yv = fminf(yv, yfin[(size_t)sidx * yfin_stride + row0 + tid]);
            s_yf[tid] = yv;
        }
        __syncthreads();
        const uint2 *__restrict__ segs = reinterpret_cast<const uint2 *>(cand) + (size_t)bxi * LR_NN16_SEG;
        for (int sidx = 0; sidx < used; ++sidx) {
            const int c = cw[sidx];
            if (c < 0) { if (tid == 0) s_bad = 1; continue; }      // segment overflowed: all 64 rows are re-done
            // (two entries in flight per thread measured slower: 118 VGPRs, 4 instead of 6 waves per SIMD)
            for (int e = tid; e < c; e += 256) {
                const uint2 v = segs[(size_t)sidx * seg_cap + e];
                const int j = (int)(v.x & LR_PB_COLMASK), ekb = lr_pb_kb(v.x);
                if (v.x & LR_PB_HASG) {       // (single-row entry with its g, rounded up to 16 bits)
                    const int rl = lr_pb_row(ekb, __builtin_ctz((v.y & 0xffffu) | 0x8000u));
                    const float gv = __uint_as_float(v.y & 0xffff0000u), yr = s_yf[rl];
                    if (gv + 1e-5f * (fabsf(gv) + fabsf(yr)) < -yr) continue;
                }
                f32x4 t[8];
                const f32x4 *pb = reinterpret_cast<const f32x4 *>(Fc + (size_t)j * 32);
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = pb[k];
                const float ncj = nC[j];
                unsigned m = v.y & 0xffffu;
                while (m) {
                    const int bit = __builtin_ctz(m);
                    m &= m - 1;
                    const int rl = lr_pb_row(ekb, bit);
                    ex_offer(&s_best[rl], &s_second[rl], dist(rl, t, ncj), j);
                    atomicAdd(&s_cnt[rl], 1);
                }
            }
        }
    }
    __syncthreads();
    // ---- rows to re-do by an exact scan of all columns
    if (tid < LR_EX_ROWS && row0 + tid < na && (s_bad || s_badrow[tid] || s_cnt[tid] < min(need, nb))) s_redo[atomicAdd(&s_nredo, 1)] = tid;
    __syncthreads();
    const int nredo = s_nredo;
    for (int r = 0; r < nredo; ++r) {
        const int rl = s_redo[r];
        if (tid == 0) { s_best[rl] = LR_EX_EMPTY; s_second[rl] = LR_EX_EMPTY; atomicAdd(&counters[LR_CNT_FIX_TOTAL], 1); }
        __syncthreads();
        for (int j = tid; j < nb; j += 256) {
            f32x4 t[8];
            const f32x4 *pb = reinterpret_cast<const f32x4 *>(Fc + (size_t)j * 32);
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = pb[k];
            ex_offer(&s_best[rl], &s_second[rl], dist(rl, t, nC[j]), j);
        }
    }
    __syncthreads();
    // ---- results: one thread per row
    float sv = 3.0e38f, key = 0.0f;
    const bool writer = tid < LR_EX_ROWS && row0 + tid < na;
    if (writer) ex_write_row(out, s_rowd[tid], s_best[tid], s_second[tid], nb, need, sv, key);
    if (out.seed_out && tid < 64) ex_range(out, writer, sv, key, tid);          // (the rows live in wave 0)
}

// ------------------------------------------------------------------ host side
int lr_nn16_prep(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, hipStream_t st, bool zero_counters)
{
    hipLaunchKernelGGL(nn16_prep_kernel, dim3(lr_cdiv(n0, 32) + lr_cdiv(n1, 32), 1, ws->zP), dim3(256), 0, st, F0, n0, ws->H0, ws->nrm0, ws->bmax0, ws->bmin0,
                       F1, n1, ws->H1, ws->nrm1, ws->bmax1, ws->bmin1, ws->rev_seed, ws->rev_seed64, ws->counters, zero_counters ? 1 : 0, lr_cdiv(n0, 32), ws->yshare, ws->z);
    hipLaunchKernelGGL(nn16_range_kernel, dim3(2, 1, ws->zP), dim3(256), 0, st, n0, (const float *)ws->bmax0, (const float *)ws->bmin0, n1, (const float *)ws->bmax1,
                       (const float *)ws->bmin1, ws->nn_range, ws->zP == 1 ? ws->form_dev : (int32_t *)nullptr, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// which form of the filter pass a single-pair call launches alone: what the range kernel of an earlier call on this workspace found
// for the column cloud (0: cloud 0 = reverse pass, 1: cloud 1 = forward pass); 0 = unknown (first call, batched call): launch both.
// The hint is only trusted when TWO calls in a row read the same value: a workspace that alternates between unit-norm and other
// clouds (the harness' calibration call next to user calls) would otherwise guess wrong on every call and pay the exact kernel's
// full scan each time, silently -- it launches both forms instead, the cost of an empty launch.  A miss that still
// happens (the first call of a changed workload) is reported in lr_pair_result.reserved[2].
static int lr_nn16_form_hint(lr_workspace *ws, int col_cloud)
{
    if (ws->zP != 1 || !ws->form_host) return 0;
    const int h = *reinterpret_cast<const volatile int32_t *>(&ws->form_host[col_cloud]);
    const int prev = ws->form_prev[col_cloud];
    ws->form_prev[col_cloud] = h;
    return ((h == 1 || h == 2) && h == prev) ? h : 0;
}

int lr_nn16_run(lr_workspace *ws, const float *Fq, const _Float16 *Hq, const float *nQ, int na,
                const float *Fc, const _Float16 *Hc, const float *nC, const float *range_c, int nb,
                int need, int32_t *idx1, int32_t *idx2, float *s1, float *s2, hipStream_t st, bool seed_reverse)
{
    const int ntiles = lr_cdiv(nb, 32);
    const int row_blocks = lr_cdiv(na, LR_BLOCK_ROWS);
    // strips: enough blocks (over all pairs of a batched call) to fill 256 CUs a few times over, at least 64 tiles per strip
    // (a single pair: as many strips as keep ALL blocks resident at once -- the strips of a row block pool their thresholds, see the
    // kernel, so a shorter strip no longer means a looser one, and a second round of blocks would be a tail)
    int strips = ws->zP > 1 ? lr_cdiv(ws->nn_blocks_batch, row_blocks * ws->zP) : ws->nn_blocks_target / row_blocks;
    int smax = ntiles / 64;
    if (strips > smax) strips = smax;
    if (strips > LR_NN_MAX_STRIPS) strips = LR_NN_MAX_STRIPS;
    if (strips < 1) strips = 1;
    const int tps = lr_cdiv(ntiles, strips);
    // phase 1 of the filter pass samples every `sstride`-th tile of a strip (any subset gives a valid, if looser, start; the walk
    // tightens it).  Its cost is ~ tiles / stride per row, the extra hits of a looser start ~ 2 ln(stride) per row: about 32
    // sampled tiles per strip, a stride of at most 32.
    // With several strips per row block the start thresholds are pooled, so the budget is per ROW: about 64 sampled tiles over all strips.
    int sstride = ws->nn_sample_stride > 0 ? ws->nn_sample_stride : (strips > 1 ? ntiles / 64 : tps / 32);
    if (ws->nn_sample_stride <= 0) { const int cap = strips > 1 ? 16 : 32; if (sstride > cap) sstride = cap; }      // (pipeline, 30k points: 16 / 24 / 32 / 48 / 64 -> 11 750 / 11 820 / 11 910 / 11 880 / 11 920 pairs/s)
    if (sstride < 1) sstride = 1;
    if (sstride > tps) sstride = tps;          // (one sampled tile per strip at least; keeps phase 1's column index in range)
    // 1-D XCD-aware grid over (row block, strip, pair), padded to a multiple of 8
    const int total = row_blocks * strips * ws->zP;
    dim3 grid(8 * lr_cdiv(total, 8));
    if (ws->timing && !ws->ev_pending) { LR_HIP(hipEventRecord(ws->ev[0], st)); }
    lr_thr_in thr = { nQ, range_c, need, sstride };
    // Both forms of the walk's candidate test; the blocks of the one the column norms do not ask for return at once.  A single-pair call
    // launches only the form the column cloud's norms asked for in the PREVIOUS call on this workspace (lr_nn16_form_hint: a flag the range
    // kernel leaves in pinned host memory; no synchronisation, any value is safe): if this call's norms ask for the other form, the
    // launched kernel flags the miss and the exact kernel re-does every row by the full scan -- correct, slow once, and the next call
    // launches the right form.  (An empty launch costs 4.7 us: 9.4 of a single pair's 346 us of kernels.)
    const int only = lr_nn16_form_hint(ws, 1);
    int32_t *miss = ws->counters + LR_CNT_FORM_MISS_F;
    const lr_ex_out eo = { idx1, idx2, s1, s2, seed_reverse ? ws->rev_seed : (uint32_t *)nullptr, ws->rev_s1, reinterpret_cast<uint32_t *>(ws->counters + LR_CNT_RLO),
                           ws->rev_seed64 };
    unsigned long long *clk = ws->clock_probe ? ws->clk_dev : (unsigned long long *)nullptr;
    const lr_pb_tail tl = { clk };
    if (only != 2)
        hipLaunchKernelGGL(nn16_passb_kernel<true>, grid, dim3(256), 0, st, tl, clk, Hq, na, (const int32_t *)nullptr, (const int32_t *)nullptr, Hc, nC, nb,
                           tps, (const float *)nullptr, ws->cand_cnt, ws->cand, (const int32_t *)nullptr, (const float *)nullptr,
                           (const uint32_t *)nullptr, (const int32_t *)nullptr, (const uint32_t *)nullptr, ws->yfin, ws->max_n, ws->yshare, miss, thr,
                           lr_pb_grid{ row_blocks, strips, total, 0, only }, ws->z);
    if (only != 1)
        hipLaunchKernelGGL(nn16_passb_kernel<false>, grid, dim3(256), 0, st, tl, clk, Hq, na, (const int32_t *)nullptr, (const int32_t *)nullptr, Hc, nC, nb,
                           tps, (const float *)nullptr, ws->cand_cnt, ws->cand, (const int32_t *)nullptr, (const float *)nullptr,
                           (const uint32_t *)nullptr, (const int32_t *)nullptr, (const uint32_t *)nullptr, ws->yfin, ws->max_n, ws->yshare, miss, thr,
                           lr_pb_grid{ row_blocks, strips, total, 0, only }, ws->z);
    if (ws->timing && !ws->ev_pending) { LR_HIP(hipEventRecord(ws->ev[1], st)); ws->ev_pending = 1; }
    const int ex_gx = lr_cdiv(na, LR_EX_ROWS), ex_total = ex_gx * ws->zP;
    hipLaunchKernelGGL(nn16_exact_kernel, dim3(8 * lr_cdiv(ex_total, 8)), dim3(256), 0, st, Fq, nQ, na, Fc, nC, nb, ws->cand_cnt, ws->cand,
                       strips, need, (const float *)ws->yfin, ws->max_n, (const int32_t *)nullptr, (const int32_t *)nullptr, eo, ws->counters,
                       range_c, 0, ex_gx, ex_total, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ reverse NN seeded by the forward result
// The mutual test only asks, for a cloud-1 point j that some query i points at, whether any other cloud-0 point beats
// that pair.  So the reverse direction needs no sampling pass: the exact distance s*_j of the best forward pair (i*, j)
// IS an upper bound of j's minimum, and pass B looks for points with u' <= s*^2 (1 + 4e-7) - n_j + E.
//   - points j nobody points at are left out (their reverse NN is reported as -1; the reference does not compute it
//     either, matching.py:224-225);
//   - a cloud-0 point i' that does NOT point at j is at least as far from j as from its own second neighbour: d(i', j) >= s2(i')
//     holds exactly in fp32 because both directions form bit-identical distances (key = s2, or s1 when no second neighbour was
//     asked for; a by-product of the forward exact kernel).  The points that DO point at j need no search: the forward exact kernel
//     leaves the best of them per target, with the smallest index among equals, in a 64-bit seed.  The rows (pointed-at j) are
//     therefore ordered by descending s*, the columns (all i') by ascending key (a counting sort on LR_RS_BUCKETS linear buckets;
//     any order is valid, a better one only prunes more), and each 256-row block walks only the prefix of column tiles that can
//     matter to it: the columns in the buckets up to the bucket of its first row's bound (nn16_passb_kernel).
// seeds: seed_bits[j] = min over i with idx1[i] == j of the exact distance; s1[i] = that distance; range = their min / max
__global__ void __launch_bounds__(256)
nn16_rev_seed_kernel(const float *__restrict__ F0, const float *__restrict__ n0v, int n0, const float *__restrict__ F1,
                     const float *__restrict__ n1v, const int32_t *__restrict__ idx1, uint32_t *__restrict__ seed_bits,
                     float *__restrict__ s1, uint32_t *__restrict__ range)
{
    __shared__ float s_lo[4], s_hi[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float sv = 0.0f;
    if (i < n0) {
    const int j = idx1[i];
    const f32x4 *pa = reinterpret_cast<const f32x4 *>(F0 + (size_t)i * 32);
    const f32x4 *pb = reinterpret_cast<const f32x4 *>(F1 + (size_t)j * 32);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x4 a = pa[k], b = pb[k];
        acc = __builtin_fmaf(a.x, b.x, acc);
        acc = __builtin_fmaf(a.y, b.y, acc);
        acc = __builtin_fmaf(a.z, b.z, acc);
        acc = __builtin_fmaf(a.w, b.w, acc);
    }
    // same value the reverse direction forms for (j, i): the sum n1 + n0 and the products commute bit for bit
    const float d2 = __builtin_fmaf(-2.0f, acc, n0v[i] + n1v[j]);
    sv = __builtin_sqrtf(fmaxf(d2, 1e-30f));
    if (!(sv < 3.0e38f)) sv = 3.0e38f;                  // non-finite features: keep the bit pattern below the "empty" fill
    atomicMin(&seed_bits[j], __float_as_uint(sv));      // sv > 0: bit patterns order like the values
    s1[i] = sv;
    }
    float lo = i < n0 ? sv : 3.0e38f, hi = i < n0 ? sv : 0.0f;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { lo = fminf(lo, __shfl_xor(lo, k)); hi = fmaxf(hi, __shfl_xor(hi, k)); }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&range[0], __float_as_uint(fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]))));
        atomicMax(&range[1], __float_as_uint(fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]))));
    }
}

// The ordering of the reverse pass in ONE launch (round 6; rounds 2-5: histogram + scan in a one-block kernel, ranks in a second, the
// permuted f16 copy in a third -- 15 + 39 + 41 us per call of 32 pairs and two dependent launch boundaries between the forward result
// and the reverse walk).  A counting sort on LR_RS_BUCKETS linear buckets per side: cloud-0 points (the columns) by ascending key, cloud-1
// points with a seed (the rows) by descending seed.  `parts` blocks per pair (a power of two), and NO exchange between them:
//   1. every block builds the histograms of ALL keys of its pair in LDS and scans them (redundantly: 4 bytes per key from L2; what the
//      one-block kernel did in 15 us, now without a launch of its own); part 0 writes the offsets the filter pass reads (bucket b ends
//      where bucket b + 1 starts) and the row count;
//   2. a block OWNS the buckets whose first position falls into its share [part, part + 1) n / parts of the order -- a contiguous range
//      per side, found from the block's own offsets;
//   3. it walks all keys again, 8192 per round: the rank of an owned element inside its bucket is an LDS atomic (any order inside a
//      bucket is valid); owned cloud-0 elements are queued as (point, position) in LDS and then moved by FOUR threads per row -- the f16
//      row (64 B) to its position in the permuted copy the reverse walk streams, with colmap and the norm: full lanes and whole 64-byte
//      segments, where a thread that ranks and copies its own element had 1 lane in 8 at work; owned cloud-1 elements get their position
//      in the row list and their threshold at once (4 + 4 bytes).
// Rows nobody points at get rev = -1; the segment counters of the reverse filter pass start from zero (row blocks that use fewer strips
// than offered leave the others untouched).
#define LR_RO_ROUND 8192         // keys per round (1024 threads x 8 keys, all eight loads in flight)
#define LR_RO_QUEUE 3072         // capacity of the round's copy queue: a block owns 1/parts of the keys (1/8 .. 1/32), three times the expected share
                                 // of a round; an element that finds the queue full is moved by its own thread (slow, correct: one bucket holding
                                 // most of a cloud -- hundreds of identical descriptors)
__global__ void __launch_bounds__(1024)
nn16_rev_order_kernel(int n0, int n1, const float *__restrict__ s1, const uint32_t *__restrict__ seed_bits, const uint32_t *__restrict__ range,
                      int32_t *__restrict__ offs, int32_t *__restrict__ n_rows, const float *__restrict__ block_max_c, int nblk_c,
                      const float *__restrict__ nrm1, const _Float16 *__restrict__ H0, const float *__restrict__ nrm0, int32_t *__restrict__ colmap,
                      _Float16 *__restrict__ H0s, float *__restrict__ nrm0s, int32_t *__restrict__ rowmap, float *__restrict__ tau,
                      int32_t *__restrict__ cand_cnt, int32_t *__restrict__ rev_out, int seg_counters, lr_zargs z)
{
    __shared__ int s_pos[2 * LR_RS_BUCKETS];     // histograms, then offsets, then the next free position of every bucket
    __shared__ int2 s_q[LR_RO_QUEUE];            // copy queue of the round: (cloud-0 point, its position)
    __shared__ float s_m[16];
    __shared__ int s_w[16], s_own[4], s_nq, s_nrows;
    if (z.descs) { n0 = z.descs[blockIdx.z].n0; n1 = z.descs[blockIdx.z].n1; nblk_c = (n0 + 31) >> 5; }
    lr_z(s1, z, blockIdx.z); lr_z(seed_bits, z, blockIdx.z); lr_z(range, z, blockIdx.z); lr_z(offs, z, blockIdx.z); lr_z(n_rows, z, blockIdx.z);
    lr_z(block_max_c, z, blockIdx.z); lr_z(nrm1, z, blockIdx.z); lr_z(H0, z, blockIdx.z); lr_z(nrm0, z, blockIdx.z); lr_z(colmap, z, blockIdx.z);
    lr_z(H0s, z, blockIdx.z); lr_z(nrm0s, z, blockIdx.z); lr_z(rowmap, z, blockIdx.z); lr_z(tau, z, blockIdx.z); lr_z(cand_cnt, z, blockIdx.z); lr_z(rev_out, z, blockIdx.z);
    const int tid = threadIdx.x, lane = tid & 63, part = blockIdx.x, parts = gridDim.x;
    for (int k = part * 1024 + tid; k < seg_counters; k += parts * 1024) cand_cnt[k] = 0;
    const float lo = __uint_as_float(range[0]), scale = rs_scale(lo, __uint_as_float(range[1]));
    for (int k = tid; k < 2 * LR_RS_BUCKETS; k += 1024) s_pos[k] = 0;
    if (tid < 4) s_own[tid] = (tid & 1) ? 0 : 0x7fffffff;
    if (tid == 0) s_nq = 0;
    // largest norm of cloud 0 (the columns): part of the rows' thresholds
    float mx = 0.0f;
    for (int b = tid; b < nblk_c; b += 1024) mx = fmaxf(mx, block_max_c[b]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k));
    if (lane == 0) s_m[tid >> 6] = mx;
    __syncthreads();
    float max_nc = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) max_nc = fmaxf(max_nc, s_m[w]);
    // ---- 1. histograms of both sides (sixteen independent loads in flight per thread -- eight keys of each cloud: the loop is bound by load
    //         latency, not by the atomics)
    for (int i0 = tid; i0 < max(n0, n1); i0 += 8 * 1024) {
        float v[8]; uint32_t u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = s1[min(i0 + 1024 * k, n0 - 1)]; u[k] = seed_bits[min(i0 + 1024 * k, n1 - 1)]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i0 + 1024 * k < n0) atomicAdd(&s_pos[rs_bucket(v[k], lo, scale)], 1);
            const float sv = __uint_as_float(u[k]);
            if (i0 + 1024 * k < n1 && sv <= 3.0e38f) atomicAdd(&s_pos[LR_RS_BUCKETS + (LR_RS_BUCKETS - 1 - rs_bucket(sv, lo, scale))], 1);
        }
    }
    __syncthreads();
    // exclusive prefix sums, in place; part 0 publishes them
    for (int side = 0; side < 2; ++side) {
        int *hp = s_pos + side * LR_RS_BUCKETS;
        int v[LR_RS_BUCKETS / 1024], sum = 0;
#pragma unroll
        for (int k = 0; k < LR_RS_BUCKETS / 1024; ++k) { v[k] = hp[tid * (LR_RS_BUCKETS / 1024) + k]; sum += v[k]; }
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) s_w[tid >> 6] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_w[w];
        int run = base + inc - sum;
#pragma unroll
        for (int k = 0; k < LR_RS_BUCKETS / 1024; ++k) {
            hp[tid * (LR_RS_BUCKETS / 1024) + k] = run;
            if (part == 0) offs[side * LR_RS_BUCKETS + tid * (LR_RS_BUCKETS / 1024) + k] = run;
            run += v[k];
        }
        if (side == 1 && tid == 1023) { s_nrows = run; if (part == 0) *n_rows = run; }
        __syncthreads();
    }
    // ---- 2. the buckets this block owns (contiguous per side: the two threads that sit on a range's ends write them -- no atomics)
    {
        const long long nrows = s_nrows;
        for (int k = tid; k < 2 * LR_RS_BUCKETS; k += 1024) {
            const int side = k >= LR_RS_BUCKETS, kb = k & (LR_RS_BUCKETS - 1);
            const long long n = side ? nrows : (long long)n0, lo_p = (long long)part * n, hi_p = (long long)(part + 1) * n;
            auto own = [&](int idx) { const long long sp = (long long)s_pos[idx] * parts; return sp >= lo_p && sp < hi_p; };
            if (own(k)) {
                if (kb == 0 || !own(k - 1)) s_own[2 * side] = k;
                if (kb == LR_RS_BUCKETS - 1 || !own(k + 1)) s_own[2 * side + 1] = k + 1;
            }
        }
    }
    __syncthreads();
    const int c_lo = s_own[0], c_len = max(s_own[1] - s_own[0], 0), r_lo = s_own[2], r_len = max(s_own[3] - s_own[2], 0);
    // ---- 3a. cloud 0: rank, queue, move (rounds of LR_RO_ROUND keys; every thread runs every round: the barriers are uniform)
    constexpr int U = LR_RO_ROUND / 1024;
    auto move_row = [&](int i, int p, int piece) {
        reinterpret_cast<f32x4 *>(H0s + (size_t)p * 32)[piece] = reinterpret_cast<const f32x4 *>(H0 + (size_t)i * 32)[piece];
        if (piece == 0) { colmap[p] = i; nrm0s[p] = nrm0[i]; }
    };
    for (int i0 = 0; i0 < n0; i0 += LR_RO_ROUND) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = s1[min(i0 + tid + 1024 * k, n0 - 1)];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = i0 + tid + 1024 * k, b = rs_bucket(v[k], lo, scale);
            if (i < n0 && (unsigned)(b - c_lo) < (unsigned)c_len) {
                const int p = atomicAdd(&s_pos[b], 1), slot = atomicAdd(&s_nq, 1);
                if (slot < LR_RO_QUEUE) s_q[slot] = make_int2(i, p);
                else { for (int piece = 0; piece < 4; ++piece) move_row(i, p, piece); }
            }
        }
        __syncthreads();
        const int nq = min(s_nq, LR_RO_QUEUE);
        __syncthreads();                 // (everybody has read the count before it is reset for the next round)
        if (tid == 0) s_nq = 0;
        for (int e = tid >> 2; e < nq; e += 256) { const int2 ip = s_q[e]; move_row(ip.x, ip.y, tid & 3); }
        __syncthreads();                 // (the queue is consumed and the reset visible before the next round pushes)
    }
    // ---- 3b. cloud 1: position in the row list + threshold, sixteen keys in flight per thread
    constexpr int U1 = 16;
    for (int j0 = tid; j0 < n1; j0 += U1 * 1024) {
        uint32_t v[U1];
#pragma unroll
        for (int k = 0; k < U1; ++k) v[k] = seed_bits[min(j0 + 1024 * k, n1 - 1)];
#pragma unroll
        for (int k = 0; k < U1; ++k) {
            const int row = j0 + 1024 * k;
            if (row >= n1) continue;
            const float sv = __uint_as_float(v[k]);
            if (!(sv <= 3.0e38f)) {      // still the 0x7f7f7f7f fill: no query points at this row
                if ((row & (parts - 1)) == part) rev_out[row] = -1;
                continue;
            }
            const int b = LR_RS_BUCKETS + (LR_RS_BUCKETS - 1 - rs_bucket(sv, lo, scale));
            if ((unsigned)(b - r_lo) >= (unsigned)r_len) continue;
            const int p = atomicAdd(&s_pos[b], 1);
            const float nj = nrm1[row], scl = nj + max_nc;
            const float E = 1.05e-3f * scl + 4e-7f;
            const float d2hi = sv * sv * (1.0f + 6e-7f);               // every d2 whose sqrt rounds to <= sv lies below this
            rowmap[p] = row;
            tau[p] = (d2hi - nj) + E + 6e-6f * scl + 2e-6f * d2hi;
        }
    }
}

int lr_nn16_reverse(lr_workspace *ws, const float *F0, const _Float16 *H0, const float *nrm0, const float *bmax0, int n0,
                    const float *F1, const _Float16 *H1, const float *nrm1, int n1, const int32_t *fwd_idx1,
                    int32_t *rev, hipStream_t st, bool seeded)
{
    // rows = cloud 1 (the columns of the forward direction), columns = cloud 0
    const int na = n1, nb = n0;
    const int ntiles = lr_cdiv(nb, 32);
    const int row_blocks = lr_cdiv(na, LR_BLOCK_ROWS);
    // the grid offers every row block the maximum number of strips; a block uses as many as its column prefix is worth
    // strips offered to every row block (no pass-A partial arrays on this path: not bound by LR_NN_MAX_STRIPS).  A block that is
    // not needed leaves after three loads, but tens of thousands of them still cost: 8 for one pair (parallelism for the row blocks
    // with long prefixes), fewer the more pairs a batched call brings (2 at 32 pairs: 7 990 against 7 740 pairs/s with 16)
    int strips = ws->rev_strips > 0 ? ws->rev_strips : 48 / ws->zP;
    if (ws->rev_strips <= 0) { if (strips > 8) strips = 8; if (strips < 2) strips = 2; }
    int smax = ntiles / 8;
    if (strips > smax) strips = smax;
    if (strips < 1) strips = 1;
    const int tps = lr_cdiv(ntiles, strips);
    uint32_t *seed = ws->rev_seed;          // filled with 0x7f7f7f7f, and range reset to { 0x7f7f7f7f, 0 }, by the prep kernel of this pair
    uint32_t *range = reinterpret_cast<uint32_t *>(ws->counters + LR_CNT_RLO);
    int32_t *n_rows = ws->counters + LR_CNT_NREV;
    if (!seeded)     // (the forward pass of lr_register_pair seeds from its exact kernel: same values, one launch less)
        hipLaunchKernelGGL(nn16_rev_seed_kernel, dim3(lr_cdiv(n0, 256)), dim3(256), 0, st, F0, nrm0, n0, F1, nrm1, fwd_idx1, seed, ws->rev_s1, range);
    const int order_parts = ws->zP >= 16 ? 8 : ws->zP >= 4 ? 16 : 32;      // (a power of two; few pairs: more, smaller shares)
    hipLaunchKernelGGL(nn16_rev_order_kernel, dim3(order_parts, 1, ws->zP), dim3(1024), 0, st, n0, n1, (const float *)ws->rev_s1, (const uint32_t *)seed,
                       (const uint32_t *)range, ws->rev_hist, n_rows, bmax0, lr_cdiv(nb, 32), nrm1, H0, nrm0, ws->rev_cols, ws->Hs, ws->nrms,
                       ws->rev_rows, ws->tau, ws->cand_cnt, rev, row_blocks * 4 * (strips + 1), ws->z);
    // grids are sized for all rows; blocks past the compacted count (or past their row block's strips) leave at once.
    // The column-prefix pruning (flagged by a non-null tile_min) rests on the keys being true lower bounds: that holds when
    // the list comes from this library's own forward pass (`seeded`, lr_register_pair).  A caller-supplied list (lr_nn_to_mutual,
    // lr_gpf*) may be anything -- the reference's nn_to_mutual accepts any corres_idx1 -- so every row block walks all column
    // tiles there: the seeds are still upper bounds of the row minima, only the prefix cut is given up.
    const int total = row_blocks * strips * ws->zP;
    dim3 grid(8 * lr_cdiv(total, 8));
    const bool timed = ws->timing && ws->ev_pending == 1 && !ws->rev_recorded;
    if (timed) { LR_HIP(hipEventRecord(ws->ev[4], st)); }
    const int only = lr_nn16_form_hint(ws, 0);          // (the columns of the reverse pass are cloud 0)
    int32_t *miss = ws->counters + LR_CNT_FORM_MISS_R;
    const lr_thr_in rthr = { nullptr, (const float *)ws->nn_range, 1, 0 };      // (no tightening: the column norms' range selects the form of the walk's test)
    const lr_ex_out reo = { rev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, seeded ? ws->rev_seed64 : (unsigned long long *)nullptr };
    unsigned long long *clk = ws->clock_probe ? ws->clk_dev : (unsigned long long *)nullptr;
    const lr_pb_tail tl = { clk };
    if (only != 2)
        hipLaunchKernelGGL(nn16_passb_kernel<true>, grid, dim3(256), 0, st, tl, clk, H1, na, (const int32_t *)ws->rev_rows, (const int32_t *)n_rows,
                           (const _Float16 *)ws->Hs, (const float *)ws->nrms, nb, tps, ws->tau, ws->cand_cnt, ws->cand,
                           (const int32_t *)ws->rev_cols, seeded ? (const float *)ws->rev_tmin : (const float *)nullptr, (const uint32_t *)seed,
                           (const int32_t *)ws->rev_hist, (const uint32_t *)range, (float *)nullptr, 0, (uint32_t *)nullptr, miss, rthr,
                           lr_pb_grid{ row_blocks, strips, total, 1, only }, ws->z);
    if (only != 1)
        hipLaunchKernelGGL(nn16_passb_kernel<false>, grid, dim3(256), 0, st, tl, clk, H1, na, (const int32_t *)ws->rev_rows, (const int32_t *)n_rows,
                           (const _Float16 *)ws->Hs, (const float *)ws->nrms, nb, tps, ws->tau, ws->cand_cnt, ws->cand,
                           (const int32_t *)ws->rev_cols, seeded ? (const float *)ws->rev_tmin : (const float *)nullptr, (const uint32_t *)seed,
                           (const int32_t *)ws->rev_hist, (const uint32_t *)range, (float *)nullptr, 0, (uint32_t *)nullptr, miss, rthr,
                           lr_pb_grid{ row_blocks, strips, total, 1, only }, ws->z);
    if (timed) { LR_HIP(hipEventRecord(ws->ev[5], st)); ws->rev_recorded = 1; }
    const int ex_gx = lr_cdiv(na, LR_EX_ROWS), ex_total = ex_gx * ws->zP;
    hipLaunchKernelGGL(nn16_exact_kernel, dim3(8 * lr_cdiv(ex_total, 8)), dim3(256), 0, st, F1, nrm1, na, F0, nrm0, nb, ws->cand_cnt, ws->cand,
                       strips, 1, (const float *)nullptr, 0, (const int32_t *)ws->rev_rows, (const int32_t *)n_rows, reo, ws->counters,
                       (const float *)ws->nn_range, 1, ex_gx, ex_total, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}
