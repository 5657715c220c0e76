// DistMult, ONE pass over node-sorted incidence lists with d rel_emb on the matrix cores (round 6).
//
// reference: biomedkg/kge_module.py:121-128 (decode + binary_cross_entropy_with_logits) + biomedkg/model/decoder.py:45-49
// (DistMult) and their autograd, for ONE label's triplet set (the step's fresh NEGATIVES).
//
// The two-pass form reads every triplet's two rows twice: the relation-ordered scorer (kge_fused.hip: scores, g,
// d rel_emb) and the node-ordered gradient (kge_bwd_nodes.hip: dz) -- 4 T row reads, 1.32 ms per 4 M negatives on the
// bench graph, both launches at the rate the fabric delivers uniform-random rows (7.3 TB/s of L2 misses).  Walking the
// 2 T incidences sorted by (node, role) instead, the node's own row z_n stays in registers and each entry's OTHER row o
// gives everything at once (DistMult is symmetric: the same score from either end):
//     s = sum_c z_n[c] rel_r[c] o[c],   g = sigmoid(s) - y,   dz_n += (g / total) rel_r (*) o
// -- 2 T row reads.  What the positives' run-based walk (kge_onepass_kernel) solves with (node, role, relation)-sorted
// lists and one record row per run does not carry over to fresh negatives (their runs are single entries): here
//     d rel_emb[r] += g * (z_n (*) o)       over the entries whose node is the HEAD
// is a segmented sum over a RANDOM relation id per entry.  It goes to the matrix pipe as a product with a one-hot matrix:
// per 16 head entries,  C[32 relations, 256 columns] += A[32, 16] B[16, 256]  with A[r, k] = (rel_k == r) (exact in bf16)
// and B[k, :] = bf16(g_k z_n (*) o_k) staged through LDS and read back k-major by ds_read_b64_tr_b16 (the hardware
// transpose read) -- eight v_mfma_f32_32x32x16_bf16 per 16 entries, no atomics, no relation order, no records.  B is an
// MFMA operand and therefore bf16 in this mode ("every tensor that is only ever an MFMA operand is stored as
// bf16"; RNE, unbiased: the relative error of a relation's sum falls with the square root of its entries).
//
// Parallelisation: ONE lane group (16 lanes x 2 pieces = a 512-byte bf16 row) walks ONE node's list, an entry per round;
// the four groups of a wave walk four consecutive nodes in lockstep, so a round is four entries and a node's dz is complete
// in its own group (no cross-group reduce).  The walk itself keeps ~220 vector registers live, and an accumulator C of
// 32 x 256 floats would be 128 more: so the four waves of a workgroup SHARE the product -- each wave stages its own 16
// entries, after a workgroup barrier wave w multiplies ALL FOUR staged chunks into columns 64 w .. 64 w + 63 (32
// accumulator registers), a second barrier frees the staging area.  256 registers per wave, 69 KB of LDS per workgroup:
// two workgroups per CU, two waves per SIMD.  A wave hides most of the fabric latency itself -- the rows of the NEXT 8
// rounds are in flight while the current 8 are consumed (an 8-slot row ring per group), the index vectors are loaded a
// step ahead of that, the next node quad's list bounds a quad ahead, its z_n when its first row loads are issued.  The
// waves of a workgroup run the same number of steps (a wave whose stream is done stages zeros).
//
// Only what the full-graph bf16 step uses: bf16 tables of width 256, <= 32 relations, label / total as host floats.
// Everything else keeps the two-pass form (biomedkg_amd/ops.py decides through bkg_kge_onepass_mx_ok).
#include "common.h"
#include "rowpiece.h"

namespace bkg {

typedef float km_acc_t __attribute__((ext_vector_type(16)));
typedef __bf16 km_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 km_bf16x2_t __attribute__((ext_vector_type(2)));
typedef short km_s4_t __attribute__((ext_vector_type(4)));
typedef short km_s8_t __attribute__((ext_vector_type(8)));
typedef float km_f2 __attribute__((ext_vector_type(2)));

constexpr int KM_D = 256;                    // row width
constexpr int KM_NR = 32;                    // relation rows of the one-hot product
constexpr int KM_STEP = 8;                   // rounds per step (= slots of a group's row ring: 32 rows in flight per wave)
constexpr int KM_ROW = 576;                  // bytes of a staged row: 512 + 64 pad (the transposed reads of a 32-lane half
                                             // then fall on 32 distinct bank pairs: rows 16 banks apart, 16-column blocks 8)
constexpr int KM_STAGE = 16 * KM_ROW + 64 + 1024;   // a wave's staging area: 16 rows + relk[16] (padded) + its one-hot operand
constexpr int KM_FLAGS = 64;                 // the waves' "my stream goes on" words of a step (two sets)
constexpr int KM_LDS = KM_NR * KM_D * 4 + 4 * KM_STAGE + KM_FLAGS;

struct KmArgs {
  const __bf16* z; int64_t ldz;
  const float* rel_emb; int64_t ldr; int n_rel;
  const int32_t* rowptr2;                    // [2 n_nodes + 1]: first entry of key 2 * node + role (role 0: the node is the head)
  const int32_t* packed;                     // [n_inc] list order: the entry's other endpoint | its relation << 24
  const int32_t* perm;                       // [n_inc] its position in cat(head, tail) (read only with scores)
  int n_inc, n_nodes;
  float label, inv_t;
  float* score;                              // nullable [T]: written from the head end
  float* loss_part;                          // [gridDim.x * 4] unscaled BCE sums
  float* dz; int64_t lddz;
  float* drel_part;                          // [gridDim.x][32][256] unscaled per-workgroup sums of g z_n (*) o by relation
};

template <int CTRL>
__device__ __forceinline__ float km_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float km_sum16(float v) {       // the sum over a 16-lane row, in every lane of it
  v += km_dpp<0x128>(v);      // row_ror:8
  v += km_dpp<0x141>(v);      // row_half_mirror
  v += km_dpp<0x4E>(v);       // quad_perm [2,3,0,1]
  v += km_dpp<0xB1>(v);       // quad_perm [1,0,3,2]
  return v;
}
__device__ __forceinline__ int km_pick(int v, int src_lane) {   // v of lane `src_lane` (any lane may ask for any lane)
  return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
__device__ __forceinline__ int km_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// one step of the stream a wave walks: 8 rounds of a node quad (lane group g: entries e0 .. e0 + 7 of ITS node's list)
struct KmStep {
  int e0, mid, end, node;      // per lane group
  int valid, first, last;      // wave-uniform: a real step / the first / the last step of its quad
};

template <bool SCORES>
__global__ __launch_bounds__(256, 2) void kge_onepass_mx_kernel(const KmArgs a) {
  constexpr int Q = 2, VQ = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char km_lds[];
  float* rel_lds = reinterpret_cast<float*>(km_lds);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, lig = lane & 15;
  unsigned char* stage0 = km_lds + KM_NR * KM_D * 4;                  // [4 waves] staging areas
  // (plain LDS words: a volatile generic pointer compiles to a flat store with system scope and a vmcnt(0) behind it)
  int* flags = reinterpret_cast<int*>(km_lds + KM_NR * KM_D * 4 + 4 * KM_STAGE);
  for (int i = threadIdx.x; i < KM_NR * (KM_D / 4); i += 256) {
    const int r = i / (KM_D / 4), c = i % (KM_D / 4);
    reinterpret_cast<float4*>(rel_lds)[i] =
        r < a.n_rel ? *reinterpret_cast<const float4*>(a.rel_emb + (int64_t)r * a.ldr + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = threadIdx.x; i < (4 * KM_STAGE + KM_FLAGS) / 16; i += 256)   // (a stale NaN under a zero of the one-hot would be a NaN)
    reinterpret_cast<uint4*>(km_lds + KM_NR * KM_D * 4)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  const int n_quads = (a.n_nodes + 3) >> 2;
  const int wave_id = blockIdx.x * 4 + wave, n_waves = gridDim.x * 4;
  const int last_entry = a.n_inc - 1;

  // ---- the quad whose list bounds are being fetched (one quad ahead of the index cursor).  The three bounds are loaded
  // UNCONDITIONALLY every step into the same registers (in-bounds addresses; masked when consumed): a load under a
  // condition ends in a copy at the join, and hipcc waits for it there with vmcnt(0) -- draining the row ring every step
  // (starts one stride BEFORE the wave's first quad, invalid: the first pass of the loop steps onto the real one and
  // issues its bounds' loads -- no load of the walk is issued ahead of the loop, see below)
  int d_quad = wave_id - n_waves, d_valid = 0;
  int d_b0 = 0, d_b1 = 0, d_b2 = 0;
  auto load_bounds = [&]() {
    const int nodec = min(max(4 * d_quad + grp, 0), a.n_nodes - 1);
    d_b0 = a.rowptr2[2 * nodec]; d_b1 = a.rowptr2[2 * nodec + 1]; d_b2 = a.rowptr2[2 * nodec + 2];
  };

  // ---- the index cursor: the step whose index vectors are being fetched
  KmStep I{0, 0, 0, 0, 0, 0, 0};
  int i_beg = 0, i_c = 0, i_nst = 0;                 // its quad's first entry (per group), step number, steps of the quad
  int i_vo = 0, i_vp = 0;                            // lane lig: entry e0 + lig % 8 -- other endpoint | relation << 24, position
  auto advance_I = [&]() {
    if (I.valid && i_c + 1 < i_nst) {
      ++i_c;
    } else {                                         // into the next quad (its bounds have been on their way for a quad)
      const int node = 4 * d_quad + grp;
      const bool ok = d_valid && node < a.n_nodes;
      i_beg = ok ? d_b0 : 0; I.mid = ok ? d_b1 : 0; I.end = ok ? d_b2 : 0; I.node = node; I.valid = d_valid; i_c = 0;
      const int len = I.end - i_beg;
      const int l0 = __builtin_amdgcn_readlane(len, 0), l1 = __builtin_amdgcn_readlane(len, 16);
      const int l2 = __builtin_amdgcn_readlane(len, 32), l3 = __builtin_amdgcn_readlane(len, 48);
      const int lmax = max(max(l0, l1), max(l2, l3));
      i_nst = max((lmax + KM_STEP - 1) / KM_STEP, 1);
      d_quad += n_waves;
      d_valid = d_quad < n_quads;
    }
    load_bounds();
    I.e0 = i_beg + KM_STEP * i_c;
    I.first = i_c == 0;
    I.last = i_c + 1 == i_nst;
    {                                                // (unconditional too: a finished stream re-reads entry 0)
      const int e = min(max(I.e0 + (lig & (KM_STEP - 1)), 0), last_entry);
      i_vo = a.packed[e];
      if (SCORES) i_vp = a.perm[e];
    }
  };

  // ---- row ring: slot u holds round u of the step being consumed, then of the step behind it
  Piece<__bf16> val[KM_STEP][2];
  int rid_s[KM_STEP], pm_s[KM_STEP];
  KmStep R{0, 0, 0, 0, 0, 0, 0}, C{0, 0, 0, 0, 0, 0, 0};
  int r_vo = 0, r_vp = 0;
  // the row of round u of step R: its entry is picked (an LDS crossbar round trip) at the TOP of the round that frees the
  // slot, the loads are issued at its end
  auto pick = [&](int u, int& nv, int& np) {
    const int src = (lane & 48) | u;
    nv = km_pick(r_vo, src);
    if (SCORES) np = km_pick(r_vp, src);
  };
  auto issue = [&](int u, int nv, int np) {
    rid_s[u] = (int)((unsigned)nv >> 24);
    if (SCORES) pm_s[u] = np;
    // (a slot past the end of this group's list -- padding: the quad's four lists run to the longest, rounded up to 8 --
    // loads row 0, a cache hit, instead of whatever entry the clamped position holds: 0.6 GB less off the fabric per launch)
    const unsigned row = (R.e0 + u < R.end) ? (unsigned)(nv & 0xffffff) : 0u;
    const __bf16* p = a.z + row * (unsigned)a.ldz + 8 * lig;                              // (< 2^31 elements: checked)
    val[u][0].load(p);
    val[u][1].load(p + 128);
  };

  // ---- per-quad row: z_n is fetched when the quad's first rows are issued (needed as soon as it is entered)
  Piece<__bf16> zn_next[2];
  float4 zn[VQ], acc[VQ];
  zero4(acc); zero4(zn);
  zn_next[0].v = make_uint4(0u, 0u, 0u, 0u); zn_next[1].v = make_uint4(0u, 0u, 0u, 0u);
  auto fetch_zn = [&](const KmStep& s) {
    const int node = min(s.node, a.n_nodes - 1);
    const __bf16* p = a.z + (int64_t)node * a.ldz + 8 * lig;
    zn_next[0].load(p);
    zn_next[1].load(p + 128);
  };

  km_acc_t cacc[2];                                  // columns 64 wave + 32 b .. + 31 of the workgroup's 32 x 256 sums
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) cacc[b][i] = 0.f;
  float loss_acc = 0.f;

  // transposed-read address of this lane inside a staged 16-row chunk: rows 8 (lane / 32) + q (+ 4 for the second read),
  // columns 16 ((lane & 31) / 16) + 4 p .. + 3 of a 32-column block (cdna_hip_programming.md T10)
  const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * KM_ROW + (16 * ((lane & 31) >> 4) + 4 * (lane & 3)) * 2 +
                     128 * wave;                                      // (this wave's 64 columns of every staged chunk)

  // ---- the stream starts inside the loop: its first three passes consume nothing (C invalid) -- the first fetches the first
  // quad's bounds, the second step 0's index vectors, the third issues its rows -- so that EVERY load of the walk is issued from one loop body (hipcc counts
  // a wait's vmcnt over the loads it can see on every path into the use: loads issued ahead of the loop made it assume
  // that almost nothing younger is in flight, and it drained the row ring at every step)
  for (int step = 0;; ++step) {
    C = R;
    R = I; r_vo = i_vo; r_vp = i_vp;
    advance_I();
    const bool c_on = km_uniform(C.valid) != 0;
    if (!c_on) { C.e0 = C.mid = C.end = 0; }                           // (every round dead: this wave only keeps the others company)
    if (lane == 0) flags[4 * (step & 1) + wave] = c_on ? 1 : 0;        // (read behind this step's first barrier)
    if (c_on && km_uniform(C.first)) {
      zn_next[0].expand(zn);
      zn_next[1].expand(zn + Q);
    }
    const bool r_on = km_uniform(R.valid) != 0;
    if (r_on && km_uniform(R.first)) fetch_zn(R);
    // does any group still have head entries in this step?  (wave-uniform: the one-hot product runs only then)
#ifdef KM_ABL
    const bool step_heads = (KM_ABL & 1) ? false : __builtin_amdgcn_ballot_w64(C.e0 < min(C.mid, C.end)) != 0;
#else
    const bool step_heads = __builtin_amdgcn_ballot_w64(C.e0 < min(C.mid, C.end)) != 0;
#endif
#pragma unroll
    for (int u = 0; u < KM_STEP; ++u) {
      unsigned char* stage = stage0 + wave * KM_STAGE;
      unsigned char* relk = stage + 16 * KM_ROW;
      const int e = C.e0 + u;
      const bool live = e < C.end, head = e < C.mid;
      int nv = 0, np = 0;
      pick(u, nv, np);
      if (__builtin_amdgcn_ballot_w64(live) != 0) {                   // (a round past every group's list costs nothing)
        float4 o[VQ];
        val[u][0].expand(o);
        val[u][1].expand(o + Q);
        const int rid = rid_s[u];
        const float* rrow = rel_lds + rid * KM_D;
        // t = rel (*) o serves both the score (s = z_n . t) and the gradient (dz_n += g t)
        float4 t4[VQ];
        km_f2 s2 = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          const float4 r = *reinterpret_cast<const float4*>(rrow + piece_col<__bf16, 16>(lig, q));
          t4[q] = make_float4(r.x * o[q].x, r.y * o[q].y, r.z * o[q].z, r.w * o[q].w);
          s2 = __builtin_elementwise_fma(km_f2{zn[q].x, zn[q].y}, km_f2{t4[q].x, t4[q].y}, s2);
          s2 = __builtin_elementwise_fma(km_f2{zn[q].z, zn[q].w}, km_f2{t4[q].z, t4[q].w}, s2);
        }
        const float sc = km_sum16(s2[0] + s2[1]);
        // BCE with logits (torch formula): max(s,0) - s*y + log1p(exp(-|s|));  d/ds = sigmoid(s) - y
        const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(sc));
        const float rc = __builtin_amdgcn_rcpf(1.0f + ex);
        const float sig = sc >= 0.f ? rc : ex * rc;
        const float gs = live ? sig - a.label : 0.f;
        const float gi = gs * a.inv_t;
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          acc[q].x = fmaf(gi, t4[q].x, acc[q].x); acc[q].y = fmaf(gi, t4[q].y, acc[q].y);
          acc[q].z = fmaf(gi, t4[q].z, acc[q].z); acc[q].w = fmaf(gi, t4[q].w, acc[q].w);
        }
        if (head && live && lig == 0) {
          if (SCORES) a.score[pm_s[u]] = sc;                            // (head end: its position IS the triplet id)
          loss_acc += fmaxf(sc, 0.f) - sc * a.label + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + ex);
        }
        if (step_heads) {
          const float gh = (head && live) ? gs : 0.f;
          const int k = 4 * (u & 3) + grp;
          unsigned char* dst = stage + k * KM_ROW + 16 * lig;
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const float4 za = zn[2 * v], zb = zn[2 * v + 1], oa = o[2 * v], ob = o[2 * v + 1];
            km_bf16x8_t h;
            h[0] = (__bf16)(gh * za.x * oa.x); h[1] = (__bf16)(gh * za.y * oa.y);
            h[2] = (__bf16)(gh * za.z * oa.z); h[3] = (__bf16)(gh * za.w * oa.w);
            h[4] = (__bf16)(gh * zb.x * ob.x); h[5] = (__bf16)(gh * zb.y * ob.y);
            h[6] = (__bf16)(gh * zb.z * ob.z); h[7] = (__bf16)(gh * zb.w * ob.w);
            *reinterpret_cast<uint4*>(dst + 256 * v) = __builtin_bit_cast(uint4, h);
          }
          if (lig == 0) relk[k] = (unsigned char)((head && live) ? rid : 255);
        }
      } else if (step_heads) {                                        // keep the chunk's rows defined: a dead round is zeros
        const int k = 4 * (u & 3) + grp;
        unsigned char* dst = stage + k * KM_ROW + 16 * lig;
        *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(dst + 256) = make_uint4(0u, 0u, 0u, 0u);
        if (lig == 0) relk[k] = 255;
      }
      issue(u, nv, np);   // this slot's next tenant: round u of the step behind (unconditional: past the stream's end entry 0 again)
      if ((u & 3) == 0 && !step_heads && lane < 16) relk[lane] = 255;   // (no head entry left: a zero one-hot over stale rows)
#if defined(KM_ABL) && (KM_ABL & 4)
      if (false) {
#else
      if ((u & 3) == 3) {
#endif
        // d rel_emb of the 4 x 16 entries the workgroup just staged: C[rel, col] += onehot[rel, k] * B[k, col], this wave
        // its 64 columns of all four chunks (unconditional: a product under a branch made hipcc keep two copies of the
        // accumulators and move them back after every chunk)
        {   // this wave's one-hot operand A[rel = lane % 32][k = 8 (lane / 32) + j] = (rel_k == rel), built ONCE and left in
            // LDS for the four consumers (each of them building all four cost 90 vector instructions per chunk)
          const unsigned me = lane & 31;
          const uint2 rk = *reinterpret_cast<const uint2*>(relk + 8 * (lane >> 5));
          uint4 aw;
          aw.x = (((rk.x) & 255u) == me ? 0x3F80u : 0u) | (((rk.x >> 8) & 255u) == me ? 0x3F800000u : 0u);
          aw.y = (((rk.x >> 16) & 255u) == me ? 0x3F80u : 0u) | ((rk.x >> 24) == me ? 0x3F800000u : 0u);
          aw.z = (((rk.y) & 255u) == me ? 0x3F80u : 0u) | (((rk.y >> 8) & 255u) == me ? 0x3F800000u : 0u);
          aw.w = (((rk.y >> 16) & 255u) == me ? 0x3F80u : 0u) | ((rk.y >> 24) == me ? 0x3F800000u : 0u);
          *reinterpret_cast<uint4*>(stage + 16 * KM_ROW + 64 + 16 * lane) = aw;
        }
#if !defined(KM_ABL) || !(KM_ABL & 2)
        __syncthreads();
#endif
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          const unsigned char* st = stage0 + w2 * KM_STAGE;
          const km_bf16x8_t af = __builtin_bit_cast(km_bf16x8_t, *reinterpret_cast<const uint4*>(st + 16 * KM_ROW + 64 + 16 * lane));
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const unsigned char* src = st + tr_off + 64 * b;
            const km_s4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) km_s4_t*)src);
            const km_s4_t b1 =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) km_s4_t*)(src + 4 * KM_ROW));
            const km_s8_t bb = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            cacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(km_bf16x8_t, bb), cacc[b], 0, 0, 0);
          }
        }
#if !defined(KM_ABL) || !(KM_ABL & 2)
        __syncthreads();                               // (one staging area: nobody stages the next chunk before all have read)
#endif
      }
    }
    if (c_on && km_uniform(C.last)) {
      if (C.node < a.n_nodes) {
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          float4 v = acc[q];
          *reinterpret_cast<float4*>(a.dz + (int64_t)C.node * a.lddz + piece_col<__bf16, 16>(lig, q)) = v;
        }
      }
      zero4(acc);
    }
    // does any wave of the workgroup go on?  (the flags were written before this step's barriers: the same answer everywhere)
    const int fb = 4 * (step & 1);
    if (step > 2 && !(flags[fb] | flags[fb + 1] | flags[fb + 2] | flags[fb + 3])) break;
  }

  // ---- the workgroup's sums leave, this wave its 64 columns: C[b][4 j + i] = row 8 j + 4 (lane / 32) + i, column 32 b + lane % 32
  float* out = a.drel_part + (int64_t)blockIdx.x * KM_NR * KM_D + 64 * wave;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      out[(8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)) * KM_D + 32 * b + (lane & 31)] = cacc[b][i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o, 64);
  if (lane == 0) a.loss_part[wave_id] = loss_acc;
}

}  // namespace bkg

using namespace bkg;

#ifndef KM_BLOCKS_PER_CU
#define KM_BLOCKS_PER_CU 2
#endif

// May the (node, role)-sorted one-pass walk take this triplet set?  (bf16 table of width 256, <= 32 relations.)
extern "C" int32_t bkg_kge_onepass_mx_ok(int32_t d, int32_t z_is_bf16, int64_t n_rel) {
  return d == KM_D && z_is_bf16 && n_rel > 0 && n_rel <= KM_NR;
}
// Workgroups of the launch (four waves each; a wave leaves loss_part[1], a workgroup drel_part[32][256]).
extern "C" int32_t bkg_kge_onepass_mx_blocks(int64_t n_nodes) {
  static int cus = 0;                               // (two workgroups per CU, persistent: two waves per SIMD)
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  const int64_t quads = (n_nodes + 3) / 4;
  return (int32_t)std::max<int64_t>(1, std::min<int64_t>((quads + 3) / 4, (int64_t)cus * KM_BLOCKS_PER_CU));
}

// The DistMult training pass of ONE label's triplets in one walk over their incidences sorted by (node, role):
// rowptr2 [2 n_nodes + 1] = first entry of key 2 * node + role (role 0: the node is the triplet's head), packed / perm
// [n_inc] the entries' (other endpoint | relation id << 24) and position in cat(head, tail), list order.  Writes score[T] (nullable),
// loss_part[4 * blocks] (unscaled BCE sums), dz rows = the list sums and
// drel_part[blocks][32][256] (unscaled: the caller sums over the workgroups and multiplies by inv_total).
extern "C" int bkg_kge_onepass_mx_f32(const void* z, int64_t ldz, int64_t n_nodes, const float* rel_emb, int64_t ldr,
                                      int64_t n_rel, const int32_t* rowptr2, const int32_t* packed, const int32_t* perm,
                                      int64_t n_inc, float label, float inv_total, float* score,
                                      float* loss_part, float* dz, int64_t lddz, float* drel_part, int32_t blocks,
                                      void* stream) {
  BKG_ARG(z && rel_emb && rowptr2 && loss_part && dz && drel_part && n_rel > 0 && n_rel <= KM_NR && n_nodes >= 0 && n_inc >= 0,
          "bkg_kge_onepass_mx_f32: needs the table, the relation rows (<= 32), the list bounds and the output buffers");
  BKG_ARG(n_inc > 0 && packed && (!score || perm), "bkg_kge_onepass_mx_f32: needs the entries' streams (n_inc > 0)");
  BKG_ARG(n_inc < (1ll << 31) && n_nodes <= (1ll << 24) && n_nodes * ldz < (1ll << 31),
          "bkg_kge_onepass_mx_f32: 32-bit entry positions, 24-bit node ids, a table below 2^31 elements");
  BKG_ARG(((uintptr_t)z % 16 == 0) && (ldz % 8 == 0) && ((uintptr_t)rel_emb % 16 == 0) && (ldr % 4 == 0) &&
              ((uintptr_t)dz % 16 == 0) && (lddz % 4 == 0) && ((uintptr_t)drel_part % 16 == 0),
          "bkg_kge_onepass_mx_f32: alignment");
  BKG_ARG(blocks > 0, "bkg_kge_onepass_mx_f32: blocks = bkg_kge_onepass_mx_blocks(n_nodes)");
  if (n_nodes == 0) return 0;
  KmArgs a{(const __bf16*)z, ldz, rel_emb, ldr, (int)n_rel, rowptr2, packed, perm, (int)n_inc, (int)n_nodes, label,
           inv_total, score, loss_part, dz, lddz, drel_part};
  hipStream_t s = (hipStream_t)stream;
  static bool lds_set = false;                      // (69,888 bytes of dynamic LDS: above the 64 KB a kernel gets unasked)
  if (!lds_set) {
    hipFuncSetAttribute((const void*)kge_onepass_mx_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_LDS);
    hipFuncSetAttribute((const void*)kge_onepass_mx_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_LDS);
    lds_set = true;
  }
  if (score) hipLaunchKernelGGL((kge_onepass_mx_kernel<true>), dim3(blocks), dim3(256), KM_LDS, s, a);
  else hipLaunchKernelGGL((kge_onepass_mx_kernel<false>), dim3(blocks), dim3(256), KM_LDS, s, a);
  return check_launch("bkg_kge_onepass_mx_f32");
}
