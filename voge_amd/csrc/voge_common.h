// Shared device helpers for libvoge_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "voge_hip.h"

#define VOGE_WAVE 64

// Sentinels of the fine trace outputs (ray_trace_voge.cu:244-247).
#define VOGE_SENT_LEN 1e10f
#define VOGE_SENT_ACT 1e10f

namespace voge {

// ---- per-Gaussian derived records (written by prep_kernel, read by the sweep) ----------
// cull  : centre (camera-centred, the fp32 `mus` row) + conservative reach R: a ray whose
//         line passes farther than R from the centre cannot satisfy act < thr_act.
//         R = +inf -> never cull; R < 0 -> never a candidate (behind-camera skip).
// eval  : coefficients of the three quadratic forms, arranged so that one (ray, Gaussian)
//         pair costs ~30 flops and is free of the reference formula's cancellation:
//           dsd = s00 dx^2 + s11 dy^2 + s22 dz^2 + s01 dx dy + s02 dx dz + s12 dy dz
//                 (sXY = A_XY + A_YX, i.e. exactly d^T A d for any 3x3 A)
//           len = (b . d) / dsd,  b = A^T mu            (== mu^T A d / d^T A d)
//           v   = mu - len * d                            (fma: one rounding per component)
//           act = v^T A v + len * (k . d),  k = (A - A^T) mu
//                 (== mu^T A mu - (mu^T A d)^2 / dsd identically; k = 0 for symmetric A)
struct EvalRec {
  float s00, s11, s22, s01, s02, s12;
  float bx, by, bz;
  float kx, ky, kz;
};
static_assert(sizeof(EvalRec) == 48, "EvalRec must be 3 x float4");

// a / b with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the ~12-instruction IEEE
// division sequence; relative error ~1.5e-7, three orders below the 1e-4 parity tolerance.
__device__ __forceinline__ float fast_div(const float a, const float b) { return a * __builtin_amdgcn_rcpf(b); }

// s01 of an isotropic record carries this NaN bit pattern (never produced by arithmetic)
constexpr uint32_t kIsoFlag = 0x7fc0a150u;

// How the isotropic "view" entry points read their Gaussians: centre = verts[shared ? n : b*N+n] - origin[b],
// a = sigma (mode 0), 2 sigma (mode 1) or 2 / sigma (mode 2).  origin == nullptr: plain (mus, a) arrays.
struct IsoView {
  const float *origin;
  int shared, mode;
  int cam_origin = 0;      // (round 6) the centre comes from the kernel's CamView instead of `origin`
  // (round 6, the general entry's frame form) how prep_kernel reads its `isg` argument: 0: A [P][9] as given; 1: the user's
  // per-axis sigmas [N | B*N][3], A = 2 diag(s); 2: the user's [N | B*N][3][3], A = 2 S (Renderer.py:133 + Aggregation.py:144-175);
  // sigma_shared: one [N, ...] set seen by every view.  `shared` above then says the same for the centres.
  int gen_kind = 0, sigma_shared = 0;
};
__device__ __forceinline__ float iso_view_a(const float s, const int mode) {
  return mode == 1 ? 2.0f * s : (mode == 2 ? 2.0f / s : s);
}

struct PairOut {
  float len, act, dsd;
};

// One (ray, Gaussian) evaluation.  q* are the ray's quadratic features (dx*dx, ... dy*dz).
// Written with explicit fmaf so that every call site (sweep, epilogue, list kernel) produces
// bit-identical results; the library is built with -ffp-contract=off.
// Isotropic A = a*I (the converters' output): dsd = a|d|^2, len = mu.d/|d|^2, act = a|v|^2.
__device__ __forceinline__ PairOut pair_eval_iso(const float mx, const float my, const float mz, const float a,
                                                 const float dx, const float dy, const float dz,
                                                 const float qxx, const float qyy, const float qzz) {
  PairOut o;
  const float dn2 = qxx + qyy + qzz;
  const float md = fmaf(mz, dz, fmaf(my, dy, mx * dx));
  const float t = fast_div(md, dn2) + 0.0f;  // +0 canonicalises -0
  const float vx = fmaf(-t, dx, mx);
  const float vy = fmaf(-t, dy, my);
  const float vz = fmaf(-t, dz, mz);
  o.len = t;
  o.act = a * fmaf(vz, vz, fmaf(vy, vy, vx * vx));
  o.dsd = a * dn2;
  return o;
}

// The same act / dsd from an already known len (the sweep's epilogue has it in the key): the identical
// fma chain after t, so the values are bit-identical to pair_eval_iso's.
__device__ __forceinline__ PairOut pair_eval_iso_at(const float mx, const float my, const float mz, const float a,
                                                    const float t, const float dx, const float dy, const float dz,
                                                    const float dn2) {
  PairOut o;
  const float vx = fmaf(-t, dx, mx);
  const float vy = fmaf(-t, dy, my);
  const float vz = fmaf(-t, dz, mz);
  o.len = t;
  o.act = a * fmaf(vz, vz, fmaf(vy, vy, vx * vx));
  o.dsd = a * dn2;
  return o;
}

__device__ __forceinline__ PairOut pair_eval_gen(const float mx, const float my, const float mz,
                                                 const EvalRec &e, const float dx, const float dy,
                                                 const float dz, const float qxx, const float qyy,
                                                 const float qzz, const float qxy, const float qxz,
                                                 const float qyz) {
  PairOut o;
  float ksk = e.s00 * qxx;
  ksk = fmaf(e.s11, qyy, ksk);
  ksk = fmaf(e.s22, qzz, ksk);
  ksk = fmaf(e.s01, qxy, ksk);
  ksk = fmaf(e.s02, qxz, ksk);
  ksk = fmaf(e.s12, qyz, ksk);
  float msk = e.bx * dx;
  msk = fmaf(e.by, dy, msk);
  msk = fmaf(e.bz, dz, msk);
  const float t = fast_div(msk, ksk) + 0.0f;  // +0 canonicalises -0
  const float vx = fmaf(-t, dx, mx);
  const float vy = fmaf(-t, dy, my);
  const float vz = fmaf(-t, dz, mz);
  float a = e.s00 * (vx * vx);
  a = fmaf(e.s11, vy * vy, a);
  a = fmaf(e.s22, vz * vz, a);
  a = fmaf(e.s01, vx * vy, a);
  a = fmaf(e.s02, vx * vz, a);
  a = fmaf(e.s12, vy * vz, a);
  float kd = e.kx * dx;
  kd = fmaf(e.ky, dy, kd);
  kd = fmaf(e.kz, dz, kd);
  a = fmaf(t, kd, a);
  o.len = t;
  o.act = a;
  o.dsd = ksk;
  return o;
}

// A = diag(a0, a1, a2) (the user's per-axis sigmas, Aggregation.py:169-172): pair_eval's result on such a form, bit for bit,
// without the nine coefficients that are zero -- the general chain's fmaf(0, q, x) / fmaf(t, 0, x) steps return x exactly for
// finite operands, b = A mu is (a0 mx, a1 my, a2 mz) exactly, and three equal axes take pair_eval_iso like make_eval's flag says.
__device__ __forceinline__ PairOut pair_eval_diag(const float mx, const float my, const float mz, const float a0, const float a1,
                                                  const float a2, const float dx, const float dy, const float dz) {
  const float qxx = dx * dx, qyy = dy * dy, qzz = dz * dz;
  if (a0 == a1 && a0 == a2) return pair_eval_iso(mx, my, mz, a0, dx, dy, dz, qxx, qyy, qzz);
  PairOut o;
  float ksk = a0 * qxx;
  ksk = fmaf(a1, qyy, ksk);
  ksk = fmaf(a2, qzz, ksk);
  float msk = (a0 * mx) * dx;
  msk = fmaf(a1 * my, dy, msk);
  msk = fmaf(a2 * mz, dz, msk);
  const float t = fast_div(msk, ksk) + 0.0f;  // +0 canonicalises -0
  const float vx = fmaf(-t, dx, mx);
  const float vy = fmaf(-t, dy, my);
  const float vz = fmaf(-t, dz, mz);
  float a = a0 * (vx * vx);
  a = fmaf(a1, vy * vy, a);
  a = fmaf(a2, vz * vz, a);
  o.len = t;
  o.act = a;
  o.dsd = ksk;
  return o;
}

__device__ __forceinline__ bool is_iso(const EvalRec &e) { return __float_as_uint(e.s01) == kIsoFlag; }

__device__ __forceinline__ PairOut pair_eval(const float mx, const float my, const float mz,
                                             const EvalRec &e, const float dx, const float dy,
                                             const float dz, const float qxx, const float qyy,
                                             const float qzz, const float qxy, const float qxz,
                                             const float qyz) {
  if (is_iso(e)) return pair_eval_iso(mx, my, mz, e.s00, dx, dy, dz, qxx, qyy, qzz);
  return pair_eval_gen(mx, my, mz, e, dx, dy, dz, qxx, qyy, qzz, qxy, qxz, qyz);
}

// Derive the eval record from raw (mu, A).  Same code in prep_kernel and the list kernel.
__device__ __forceinline__ EvalRec make_eval(const float mx, const float my, const float mz,
                                             const float *A) {
  EvalRec e;
  e.s00 = A[0];
  e.s11 = A[4];
  e.s22 = A[8];
  e.s01 = A[1] + A[3];
  e.s02 = A[2] + A[6];
  e.s12 = A[5] + A[7];
  e.bx = fmaf(A[6], mz, fmaf(A[3], my, A[0] * mx));
  e.by = fmaf(A[7], mz, fmaf(A[4], my, A[1] * mx));
  e.bz = fmaf(A[8], mz, fmaf(A[5], my, A[2] * mx));
  e.kx = fmaf(A[2] - A[6], mz, (A[1] - A[3]) * my);
  e.ky = fmaf(A[5] - A[7], mz, (A[3] - A[1]) * mx);
  e.kz = fmaf(A[7] - A[5], my, (A[6] - A[2]) * mx);
  if (A[1] == 0.0f && A[2] == 0.0f && A[3] == 0.0f && A[5] == 0.0f && A[6] == 0.0f && A[7] == 0.0f &&
      A[0] == A[4] && A[0] == A[8])
    e.s01 = __uint_as_float(kIsoFlag);
  return e;
}

// Monotone map float -> uint32 (all non-NaN values; -0 must be canonicalised by the caller).
__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ (uint32_t)(((int32_t)u >> 31) | 0x80000000);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
  return __uint_as_float(u);
}

// Per-lane sorted top-K list of 64-bit keys (ord(len) << 32 | idx) in LDS, slot s of the
// calling thread at keys[s * stride].  Precondition: key < worst (worst = ~0 while cnt < K).
// Implements "K lexicographically smallest (len, idx)", which is what the reference's
// in-place insertion (ray_trace_voge.cu:197-212: strict '<', candidates in ascending index)
// computes.  `tail` mirrors the largest stored key in a register (0 while empty): with a
// front-to-back candidate stream most insertions are appends and touch LDS with one write only.
__device__ __forceinline__ void topk_insert(uint64_t *keys, const int stride, const int K,
                                            int &cnt, uint64_t &worst, uint64_t &tail, const uint64_t key) {
  if (cnt < K) {
    if (key >= tail) {  // append
      keys[cnt * stride] = key;
      tail = key;
      if (++cnt == K) worst = key;
      return;
    }
    int pos = cnt;      // somewhere in the middle: everything above moves up, the tail stays the tail
    while (pos > 0) {
      const uint64_t prev = keys[(pos - 1) * stride];
      if (prev <= key) break;
      keys[pos * stride] = prev;
      --pos;
    }
    keys[pos * stride] = key;
    if (++cnt == K) worst = tail;
    return;
  }
  // full: the current tail (== worst) drops out
  int pos = K - 1;
  uint64_t new_tail = key;
  while (pos > 0) {
    const uint64_t prev = keys[(pos - 1) * stride];
    if (prev <= key) break;
    if (pos == K - 1) new_tail = prev;
    keys[pos * stride] = prev;
    --pos;
  }
  keys[pos * stride] = key;
  tail = new_tail;
  worst = new_tail;
}

// Same contract, written for the sweep's common case: `take` lanes whose key is >= their tail and
// whose list is not full APPEND; they do so without a branch (the other lanes store into the
// spare row K of their column, which is never read).  Only lanes that must shift stored keys
// (out-of-order arrival, or a full list) enter the loop-carrying slow path, and the wave skips
// that path entirely when no lane needs it.  keys has K + 1 rows.
template <bool ROWS4 = true>
__device__ __forceinline__ void topk_commit(uint64_t *keys, const int stride, const int K, int &cnt,
                                            uint64_t &worst, uint64_t &tail, const uint64_t key,
                                            const bool take /* key < worst */) {
  // (cnt == K implies tail == worst, and take implies key < worst: a full list never appends, so the
  // count needs no test of its own)
  const bool app = take && (key >= tail);
  // 32-bit element offset (24-bit multiply): a 64-bit multiply-add per candidate is quarter rate
  const unsigned off = __umul24((unsigned)(app ? cnt : K), (unsigned)stride);
  keys[off] = key;
  cnt += app ? 1 : 0;
  tail = app ? key : tail;
  worst = (app && cnt == K) ? key : worst;
#ifdef VOGE_NO_SLOW
  const bool slow = false;
#else
  const bool slow = take && !app;
#endif
  // (a divergent `if` is already "skip unless some lane needs it": s_and_saveexec + s_cbranch_execz)
  if (!ROWS4) {
    // row by row: the better form where nearly every out-of-order arrival sits one or two rows deep (scalar-sigma scenes,
    // whose lists arrive depth-sorted by centre: 47.5 -> 46.5 us at cfg3 against the four-row form below)
    if (slow) {
      if (cnt < K) {   // somewhere in the middle: everything above moves up, the tail stays the tail
        int pos = cnt;
        while (pos > 0) {
          const uint64_t prev = keys[(pos - 1) * stride];
          if (prev <= key) break;
          keys[pos * stride] = prev;
          --pos;
        }
        keys[pos * stride] = key;
        if (++cnt == K) worst = tail;
      } else {         // full: the current tail (== worst) drops out
        int pos = K - 1;
        uint64_t new_tail = key;
        while (pos > 0) {
          const uint64_t prev = keys[(pos - 1) * stride];
          if (prev <= key) break;
          if (pos == K - 1) new_tail = prev;
          keys[pos * stride] = prev;
          --pos;
        }
        keys[pos * stride] = key;
        tail = new_tail;
        worst = new_tail;
      }
    }
    return;
  }
  if (slow) {
    // Somewhere in the middle (or a full list, whose current tail drops out: the same insertion into its first K - 1
    // entries).  Everything above the new key moves up one row.  FOUR rows per round trip: the four keys below the
    // gap are read together, the prefix of them that is larger than the new key is moved (the list is sorted: it IS a
    // prefix), and only a lane that moved all four goes round again -- a quarter of the dependent LDS round trips of
    // the row-by-row walk (general scenes: 47 of 151 candidates per wave arrive out of order, 4.3 rows deep).
    const bool full = (cnt == K);
    int pos = full ? K - 1 : cnt;
    uint64_t top = key;      // (full lists) the key that ends up in row K - 1
    bool first = true;
    for (;;) {
      const uint64_t p0 = keys[(unsigned)max(pos - 1, 0) * (unsigned)stride], p1 = keys[(unsigned)max(pos - 2, 0) * (unsigned)stride],
                     p2 = keys[(unsigned)max(pos - 3, 0) * (unsigned)stride], p3 = keys[(unsigned)max(pos - 4, 0) * (unsigned)stride];
      const bool c0 = (pos >= 1) && (p0 > key);
      const bool c1 = c0 && (pos >= 2) && (p1 > key);
      const bool c2 = c1 && (pos >= 3) && (p2 > key);
      const bool c3 = c2 && (pos >= 4) && (p3 > key);
      if (first) { top = c0 ? p0 : key; first = false; }
      if (c0) keys[(unsigned)pos * (unsigned)stride] = p0;
      if (c1) keys[(unsigned)(pos - 1) * (unsigned)stride] = p1;
      if (c2) keys[(unsigned)(pos - 2) * (unsigned)stride] = p2;
      if (c3) keys[(unsigned)(pos - 3) * (unsigned)stride] = p3;
      pos -= (c0 ? 1 : 0) + (c1 ? 1 : 0) + (c2 ? 1 : 0) + (c3 ? 1 : 0);
      if (!c3) break;
    }
    keys[(unsigned)pos * (unsigned)stride] = key;
    if (full) {
      tail = top;
      worst = top;
    } else if (++cnt == K) {
      worst = tail;
    }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Wave-wide max / min / sum without a chain of six dependent LDS-crossbar shuffles: four DPP steps (VALU operands, a few
// cycles each) reduce every aligned row of 16 lanes, four v_readlane + three operations combine the rows.  Max and min do
// not depend on the order (same bits as the shuffle form); the sum's association differs from wave_sum's, so
// wave_sum_dpp is used only where the value is not part of a result (bounding-cone axes, counts).
// 16-byte stores of the trace's outputs.  NT = non-temporal: write-once data that nothing in the same chain of launches
// reads again -- the all-sentinel tiles (never read at all: consumers go by the hit counts) and the stand-alone entry
// point's (idx, len, act, dsd), 168 MB at cfg3 against 32 MB of L2.  As plain stores they evict the tile lists and records
// the sweep is reading and leave the kernel's end waiting for their write-back: trace entry 84.1 -> 76.5 us at cfg3 with
// both, sentinels alone 81.8 (tools/ab_bench.sh, interleaved).  The renderer's form keeps plain stores in its epilogue:
// its (idx, len) are read by the composite that follows from the Infinity Cache (NT there: trace - 2.4 us, frame - 1 %);
// the unfused pipeline (voge_fragments_fwd: trace with act / dsd, then the composite) is unchanged either way (118 us).
#ifndef VOGE_NT_STORES
#define VOGE_NT_STORES 3      // bit 0: sentinel tiles, bit 1: the epilogue with act / dsd (A/B builds: 0)
#endif
typedef float voge_v4f __attribute__((ext_vector_type(4)));
typedef int voge_v4i __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void st16f(float *p, const float a, const float b, const float c, const float d) {
  const voge_v4f v = {a, b, c, d};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<voge_v4f *>(p)); else *reinterpret_cast<voge_v4f *>(p) = v;
}
// 16-byte stores of what the NEXT kernel reads (the renderer-form trace's (idx, len) for the composite): agent-scope
// write-through (`sc1` on a buffer store).  The data leaves the XCD's L2 as it is written -- no write-back for the kernel's
// end to wait for -- and, unlike a non-temporal store, still allocates in the memory-side cache the consumer reads from:
// renderer-form trace 72.4 -> 70.9 us, frame + 0.5 % (plain stores / non-temporal: - 2.4 us but frame - 1 %).
// base: wave-uniform; byte offsets below 2^31.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(void *base, const unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st16f_wt(const __amdgpu_buffer_rsrc_t r, const unsigned byte_off, const float a, const float b,
                                         const float c, const float d) {
  const voge_v4f v = {a, b, c, d};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void st16i_wt(const __amdgpu_buffer_rsrc_t r, const unsigned byte_off, const int a, const int b,
                                         const int c, const int d) {
  const voge_v4i v = {a, b, c, d};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16 /* sc1 */);
}
template <bool NT>
__device__ __forceinline__ void st16i(int32_t *p, const int a, const int b, const int c, const int d) {
  const voge_v4i v = {a, b, c, d};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<voge_v4i *>(p)); else *reinterpret_cast<voge_v4i *>(p) = v;
}
#define VOGE_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
#define VOGE_LANE(v, l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l))
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, VOGE_DPP(v, 0xB1));     // quad_perm [1,0,3,2]
  v = fmaxf(v, VOGE_DPP(v, 0x4E));     // quad_perm [2,3,0,1]
  v = fmaxf(v, VOGE_DPP(v, 0x141));    // row_half_mirror
  v = fmaxf(v, VOGE_DPP(v, 0x140));    // row_mirror
  return fmaxf(fmaxf(VOGE_LANE(v, 0), VOGE_LANE(v, 16)), fmaxf(VOGE_LANE(v, 32), VOGE_LANE(v, 48)));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, VOGE_DPP(v, 0xB1));
  v = fminf(v, VOGE_DPP(v, 0x4E));
  v = fminf(v, VOGE_DPP(v, 0x141));
  v = fminf(v, VOGE_DPP(v, 0x140));
  return fminf(fminf(VOGE_LANE(v, 0), VOGE_LANE(v, 16)), fminf(VOGE_LANE(v, 32), VOGE_LANE(v, 48)));
}
// inclusive prefix sum of one int per lane over the wave (DPP row shifts, then the row totals)
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += VOGE_DPP(v, 0xB1);
  v += VOGE_DPP(v, 0x4E);
  v += VOGE_DPP(v, 0x141);
  v += VOGE_DPP(v, 0x140);
  return (VOGE_LANE(v, 0) + VOGE_LANE(v, 16)) + (VOGE_LANE(v, 32) + VOGE_LANE(v, 48));
}

// A bounding cone of a set of unit directions: axis a, cos/sin of the half angle with the
// rounding pushed to the conservative side.  ok == false -> cone too wide / degenerate,
// nothing may be culled against it.
struct Cone {
  float ax, ay, az, cs, sn;
  bool ok;
};

// Conservative "this Gaussian cannot be hit by any line through the origin whose direction
// lies in the cone".  c = (centre, reach).  Distance from the centre to the double cone is
// >= q*cos - |p|*sin (p, q = axial / radial parts of the centre).
__device__ __forceinline__ bool cone_keep(const float4 c, const Cone &k) {
  if (c.w < 0.0f) return false;
  const float p = fmaf(c.z, k.az, fmaf(c.y, k.ay, c.x * k.ax));
  const float rx = fmaf(-p, k.ax, c.x), ry = fmaf(-p, k.ay, c.y), rz = fmaf(-p, k.az, c.z);
  const float q = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
  const float gap = fmaf(q, k.cs, -fabsf(p) * k.sn);
  return !k.ok || !(gap > c.w);
}

constexpr int kST = 32;            // super-tile edge (pixels): the unit of the bounding cones

// ------------------------------------------------------------------------------------------
// Bounding cone of a set of rays (used by the bin kernels and the sweep).
// ------------------------------------------------------------------------------------------
struct RayDir {
  float ux, uy, uz;
  bool ok;      // finite, non-zero direction
  bool unit;    // |d| == 1 within 1e-4 (the depth bound of the early exit assumes unit rays)
};
__device__ __forceinline__ RayDir ray_dir(const float dx, const float dy, const float dz) {
  RayDir r;
  const float dn2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  const float inv = 1.0f / sqrtf(dn2);
  r.ok = (dn2 > 0.0f) && (inv < 3e38f) && (inv == inv);
  r.unit = fabsf(dn2 - 1.0f) < 1e-4f;
  r.ux = dx * inv; r.uy = dy * inv; r.uz = dz * inv;
  return r;
}
// Partial (per-lane) extrema of one ray w.r.t. a given axis; finish with cone_finish().
__device__ __forceinline__ void cone_partial(const RayDir &u, const float ax, const float ay, const float az,
                                             float &smax, float &cmin) {
  const float cl = fmaf(u.uz, az, fmaf(u.uy, ay, u.ux * ax));
  const float rx = fmaf(-cl, ax, u.ux), ry = fmaf(-cl, ay, u.uy), rz = fmaf(-cl, az, u.uz);
  const float sl = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
  smax = fmaxf(smax, u.ok ? sl : 2.0f);
  cmin = fminf(cmin, u.ok ? cl : -1.0f);
}
__device__ __forceinline__ Cone cone_finish(const float ax, const float ay, const float az, const float n,
                                            const float smax, const float cmin, const bool all_ok) {
  Cone c;
  c.ax = ax; c.ay = ay; c.az = az;
  c.sn = smax * (1.0f + 1e-5f) + 1e-7f;
  c.cs = cmin - 1e-6f;
  c.ok = all_ok && (n > 1e-3f) && (cmin > 0.05f) && (c.sn == c.sn);
  return c;
}

// ok: 1 = usable, 0 = nothing may be culled against it, -1 = no such super-tile (outside the image)
struct ConeRec {
  float ax, ay, az, cs, sn, ok, pad0, pad1;
};
__device__ __forceinline__ Cone load_cone(const ConeRec &r) {
  Cone c;
  c.ax = r.ax; c.ay = r.ay; c.az = r.az; c.cs = r.cs; c.sn = r.sn; c.ok = r.ok > 0.f;
  return c;
}

// ------------------------------------------------------------------------------------------
// Round 6: the CAMERA as a kernel argument.  The renderer's frame used to start with a launch of its own for the ray bundle
// (rays.hip: 3 MB written, read back by four kernels, plus the cone hierarchy) -- 7 us of launch floor per frame and the first
// of three autograd nodes on the host.  The frame's kernels now take (R, T, focal, principal point) and make what they need:
//   * a pixel's ray (cam_ray) -- the SAME operations in the same order as rays_fwd_kernel, which is built from these
//     functions too (-ffp-contract=off: identical bits wherever it is evaluated; tests/test_gpu_frame.py asserts it);
//   * the camera centre (cam_origin) and the view axis (cam_axis: column 2 of R, RayTracing.py's z > 0 rule);
//   * the bounding cone of any pixel rectangle (cam_rect_cone), ANALYTICALLY from its four corner rays: the image-plane
//     points whose ray lies within an angle theta of an axis form the inside of a conic section, a convex set, so a
//     rectangle is inside the cone as soon as its corners are -- the extrema over a block's rays are attained at corners,
//     which is what block_cones_hier256 finds by reducing over all 1024 rays of a super-tile.
// A CamView with R == nullptr means "no camera": the kernel reads rays / cones / origin from memory as before.
// ------------------------------------------------------------------------------------------
struct Mat3 {
  float m[9];
};
// inverse by adjugate; fp32 (R is a rotation in practice: cond ~ 1)
__device__ __forceinline__ Mat3 inv3(const float *R) {
  Mat3 o;
  const float a = R[0], b = R[1], c = R[2], d = R[3], e = R[4], f = R[5], g = R[6], h = R[7], i = R[8];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + b * B + c * C;
  const float id = 1.0f / det;
  o.m[0] = A * id;               o.m[1] = -(b * i - c * h) * id;  o.m[2] = (b * f - c * e) * id;
  o.m[3] = B * id;               o.m[4] = (a * i - c * g) * id;   o.m[5] = -(a * f - c * d) * id;
  o.m[6] = C * id;               o.m[7] = -(a * h - b * g) * id;  o.m[8] = (a * e - b * d) * id;
  return o;
}
struct CamView {
  const float *R, *T, *focal, *pp;      // [B,3,3], [B,3], [B,2], [B,2] (contiguous fp32); R == nullptr: no camera
  int row0, stripe_h, pitch;            // stacked row i of the rendered band is image row row0 + (i / stripe_h) * pitch + i % stripe_h
  int h, W;                             // the band: h stacked rows of W pixels
  int behind;                           // != 0: Gaussians behind the camera plane are no candidates (rasterize_coarse.cu:35; axis = R[:, 2])
  float *origin_out;                    // NULL | [B,3]: the camera centres, written by the frame's first kernel
  float *rays_out;                      // NULL | [B,h,W,3]: the ray bundle, written by the sweep for the kernels behind it
};
__host__ __device__ inline CamView no_camera() { return CamView{nullptr, nullptr, nullptr, nullptr, 0, 1, 0, 0, 0, 0, nullptr, nullptr}; }
struct CamK {      // one batch element's camera, as the ray arithmetic wants it
  Mat3 Ri;
  float ifx, ify, px, py;
};
__device__ __forceinline__ CamK cam_load(const CamView &c, const int b) {
  CamK k;
  k.Ri = inv3(c.R + 9 * b);
  k.ifx = 1.0f / c.focal[2 * b]; k.ify = 1.0f / c.focal[2 * b + 1];
  k.px = c.pp[2 * b]; k.py = c.pp[2 * b + 1];
  return k;
}
__device__ __forceinline__ int cam_irow(const CamView &c, const int i) {
  if (i < c.stripe_h) return c.row0 + i;      // (a contiguous band is one stripe: no division on the frame's path)
  // i / stripe_h by a float reciprocal, corrected: exact for any sizes an image has (i < 2^22)
  int k = __float2int_rz(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)c.stripe_h));
  int r = i - k * c.stripe_h;
  if (r < 0) { --k; r += c.stripe_h; } else if (r >= c.stripe_h) { ++k; r -= c.stripe_h; }
  return c.row0 + k * c.pitch + r;
}
// unit world-space direction of image pixel (row ir, column j): Renderer.py:124-128's bundle (rays.hip's header)
__device__ __forceinline__ void cam_ray(const CamK &k, const int ir, const int j, float &x, float &y, float &z) {
  const float vx = (k.px - ((float)j + 0.5f)) * k.ifx;
  const float vy = (k.py - ((float)ir + 0.5f)) * k.ify;
  const float wx = vx * k.Ri.m[0] + vy * k.Ri.m[3] + k.Ri.m[6];
  const float wy = vx * k.Ri.m[1] + vy * k.Ri.m[4] + k.Ri.m[7];
  const float wz = vx * k.Ri.m[2] + vy * k.Ri.m[5] + k.Ri.m[8];
  const float inv = __builtin_amdgcn_rsqf(wx * wx + wy * wy + wz * wz);
  x = wx * inv; y = wy * inv; z = wz * inv;
}
// camera centre C = -T R^-1 (static indices only: a lane-dependent index into Ri makes the compiler park the matrix in LDS)
__device__ __forceinline__ void cam_origin(const CamK &k, const float *t, float &ox, float &oy, float &oz) {
  ox = -(t[0] * k.Ri.m[0] + t[1] * k.Ri.m[3] + t[2] * k.Ri.m[6]);
  oy = -(t[0] * k.Ri.m[1] + t[1] * k.Ri.m[4] + t[2] * k.Ri.m[7]);
  oz = -(t[0] * k.Ri.m[2] + t[1] * k.Ri.m[5] + t[2] * k.Ri.m[8]);
}

// Bounding cone of up to NR rays per thread of a 256-thread workgroup (`has` bit k: ray k exists).  Pass 1: axis =
// direction of the plain vector sum (any axis gives a valid cone; rays of a pinhole camera have near-equal lengths,
// so this is the mean direction).  Pass 2: extrema of the axial cosine and of the SQUARED radial sine -- one v_rsq
// per ray, one sqrt per cone.  red: 4 * 8 floats of LDS.  Every thread returns the cone.
template <int NR>
__device__ __forceinline__ ConeRec block_cone256(const float (&rx)[NR], const float (&ry)[NR], const float (&rz)[NR],
                                                 const unsigned has, const int npx, float *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sx = 0.f, sy = 0.f, sz = 0.f, okf = 1.f;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const float dn2 = fmaf(rz[k], rz[k], fmaf(ry[k], ry[k], rx[k] * rx[k]));
    const bool on = (has >> k) & 1u, fin = dn2 > 1e-30f && dn2 < 1e30f;
    sx += (on && fin) ? rx[k] : 0.f; sy += (on && fin) ? ry[k] : 0.f; sz += (on && fin) ? rz[k] : 0.f;
    okf = (on && !fin) ? 0.f : okf;
  }
  sx = wave_sum_dpp(sx); sy = wave_sum_dpp(sy); sz = wave_sum_dpp(sz); okf = wave_min(okf);
  if (lane == 0) { red[wave * 8 + 0] = sx; red[wave * 8 + 1] = sy; red[wave * 8 + 2] = sz; red[wave * 8 + 3] = okf; }
  __syncthreads();
  sx = red[0] + red[8] + red[16] + red[24];
  sy = red[1] + red[9] + red[17] + red[25];
  sz = red[2] + red[10] + red[18] + red[26];
  okf = fminf(fminf(red[3], red[11]), fminf(red[19], red[27]));
  const float n = sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
  const float ax = sx / n, ay = sy / n, az = sz / n;
  float s2max = 0.f, cmin = 1.f;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const float dn2 = fmaf(rz[k], rz[k], fmaf(ry[k], ry[k], rx[k] * rx[k]));
    const bool on = (has >> k) & 1u, fin = dn2 > 1e-30f && dn2 < 1e30f;
    const float inv = __builtin_amdgcn_rsqf(dn2);
    const float da = fmaf(rz[k], az, fmaf(ry[k], ay, rx[k] * ax));
    const float qx = fmaf(-da, ax, rx[k]), qy = fmaf(-da, ay, ry[k]), qz = fmaf(-da, az, rz[k]);
    const float s2 = fmaf(qz, qz, fmaf(qy, qy, qx * qx)) * (inv * inv);
    s2max = on ? fmaxf(s2max, fin ? s2 : 4.0f) : s2max;
    cmin = on ? fminf(cmin, fin ? da * inv : -1.0f) : cmin;
  }
  s2max = wave_max(s2max); cmin = wave_min(cmin);
  if (lane == 0) { red[wave * 8 + 4] = s2max; red[wave * 8 + 5] = cmin; }
  __syncthreads();
  s2max = fmaxf(fmaxf(red[4], red[12]), fmaxf(red[20], red[28]));
  cmin = fminf(fminf(red[5], red[13]), fminf(red[21], red[29]));
  // v_rsq is good to ~1 ulp: pad the bounds by 4e-7 relative on top of cone_finish's margins
  const Cone c = cone_finish(ax, ay, az, n / (float)max(npx, 1), sqrtf(s2max) * (1.0f + 4e-7f) + 4e-7f, cmin - 4e-7f, okf != 0.f);
  return ConeRec{c.ax, c.ay, c.az, c.cs, c.sn, c.ok ? 1.f : 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------
// Round 5: the whole cone hierarchy of a 32x32-pixel super-tile from ONE pass over its rays -- the super-tile's own cone
// (what block_cone256 returns), its four 16x16-pixel QUAD cones and its sixteen 8x8-pixel TILE cones.  binB used to
// derive the last two itself, per workgroup, from the rays (a memory round trip, two wave reductions and two barriers in
// front of its first useful load: 3.8 us of its 11.5 us mean critical path, profiles/r5_bin_times.txt); now it loads them.
// Layout of the `cones` buffer (voge_cones_floats), per batch element (so that a sub-batch is a pointer offset):
// [nst] super-tile records, then [nst][4] quad records (quad qy * 2 + qx), then [nst][16] tile records (tile ty * 4 + tx,
// local to the super-tile): kConeRecsPerST records per super-tile in all.
// Thread t of the 256 holds four rays of the super-tile (cone_thread_rays): wave ty = t >> 6 is a row of four tiles, and a tile
// is one DPP ROW of 16 lanes (tx = lane >> 4; lane j of the row holds x = 4 (j & 1) .. + 3, y = j >> 1 of the tile), so a
// tile's sums and extrema are four DPP steps on the vector unit -- no LDS crossbar.  (As first written a tile's lanes were
// xor 1, 8, 16, 32 apart: 44 ds_bpermute shuffles per thread, and the ray kernel went from 5 to 9.4 us.)
// ------------------------------------------------------------------------------------------
struct ConeHierLds {
  float4 sum[16];      // per tile: (sum of the unit directions, all finite)
  float cnt[16];       // per tile: rays inside the image
  float ext[16][4];    // per tile: (s2max, cmin) w.r.t. the tile's and the super-tile's axis
};
constexpr int kConeRecsPerST = 21;
// record index of batch element b's super-tile st / its quad q / its tile t (nst = super-tiles per batch element)
__host__ __device__ inline size_t cone_super_at(const size_t b, const size_t nst, const size_t st) { return b * nst * kConeRecsPerST + st; }
__host__ __device__ inline size_t cone_quad_at(const size_t b, const size_t nst, const size_t st, const int q) { return b * nst * kConeRecsPerST + nst + st * 4 + q; }
__host__ __device__ inline size_t cone_tile_at(const size_t b, const size_t nst, const size_t st, const int t) { return b * nst * kConeRecsPerST + nst * 5 + st * 16 + t; }
__host__ __device__ inline size_t cone_records(const size_t B, const size_t nst) { return B * nst * kConeRecsPerST; }
// the super-tile pixel (x of the first of four, y) thread t holds
__device__ __forceinline__ void cone_thread_rays(const int t, int &lx0, int &ly) {
  const int lane = t & 63, j = lane & 15;
  lx0 = (lane >> 4) * 8 + (j & 1) * 4;
  ly = (t >> 6) * 8 + (j >> 1);
}
// reductions over a DPP row of 16 lanes, result in every lane of the row
__device__ __forceinline__ float tile16_sum(float v) {
  v += VOGE_DPP(v, 0xB1); v += VOGE_DPP(v, 0x4E); v += VOGE_DPP(v, 0x141); v += VOGE_DPP(v, 0x140);
  return v;
}
__device__ __forceinline__ float tile16_max(float v) {
  v = fmaxf(v, VOGE_DPP(v, 0xB1)); v = fmaxf(v, VOGE_DPP(v, 0x4E)); v = fmaxf(v, VOGE_DPP(v, 0x141)); v = fmaxf(v, VOGE_DPP(v, 0x140));
  return v;
}
__device__ __forceinline__ float tile16_min(float v) {
  v = fminf(v, VOGE_DPP(v, 0xB1)); v = fminf(v, VOGE_DPP(v, 0x4E)); v = fminf(v, VOGE_DPP(v, 0x141)); v = fminf(v, VOGE_DPP(v, 0x140));
  return v;
}
__device__ __forceinline__ void block_cones_hier256(const float (&rx)[4], const float (&ry)[4], const float (&rz)[4], const unsigned has,
                                                    ConeRec *__restrict__ c_super, ConeRec *__restrict__ c_quad /* [4] */,
                                                    ConeRec *__restrict__ c_tile /* [16] */, ConeHierLds &L) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ltx = lane >> 4, tile = wave * 4 + ltx;
  float sx = 0.f, sy = 0.f, sz = 0.f, okf = 1.f, np = 0.f;
  float inv[4];
  bool fin[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dn2 = fmaf(rz[k], rz[k], fmaf(ry[k], ry[k], rx[k] * rx[k]));
    const bool on = (has >> k) & 1u;
    fin[k] = dn2 > 1e-30f && dn2 < 1e30f;
    inv[k] = __builtin_amdgcn_rsqf(dn2);
    sx += (on && fin[k]) ? rx[k] : 0.f; sy += (on && fin[k]) ? ry[k] : 0.f; sz += (on && fin[k]) ? rz[k] : 0.f;
    okf = (on && !fin[k]) ? 0.f : okf;
    np += on ? 1.f : 0.f;
  }
  sx = tile16_sum(sx); sy = tile16_sum(sy); sz = tile16_sum(sz); okf = tile16_min(okf); np = tile16_sum(np);
  if ((lane & 15) == 0) { L.sum[tile] = make_float4(sx, sy, sz, okf); L.cnt[tile] = np; }      // (the tile's first lane)
  __syncthreads();
  // the two axes this thread's rays are measured against: its tile's and the super-tile's.  (The QUAD cones are the
  // conservative unions of their four tile cones, made by the finishing threads -- binA's child_extrema formula: a third
  // less per-ray work in a kernel the frame waits for; binB filters with the quad cone first and the tile cones after.)
  float ax[2], ay[2], az[2];
  {
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 v = L.sum[i];
      gx += v.x; gy += v.y; gz += v.z;
    }
    const float vx[2] = {sx, gx}, vy[2] = {sy, gy}, vz[2] = {sz, gz};
#pragma unroll
    for (int l = 0; l < 2; ++l) {      // (any axis gives a valid cone: the hardware rsq's last ulp does not matter)
      const float rn = __builtin_amdgcn_rsqf(fmaf(vz[l], vz[l], fmaf(vy[l], vy[l], vx[l] * vx[l])));
      ax[l] = vx[l] * rn; ay[l] = vy[l] * rn; az[l] = vz[l] * rn;
    }
  }
  float s2m[2] = {0.f, 0.f}, cmn[2] = {1.f, 1.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool on = (has >> k) & 1u;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const float da = fmaf(rz[k], az[l], fmaf(ry[k], ay[l], rx[k] * ax[l]));
      const float qx = fmaf(-da, ax[l], rx[k]), qy = fmaf(-da, ay[l], ry[k]), qz = fmaf(-da, az[l], rz[k]);
      const float s2 = fmaf(qz, qz, fmaf(qy, qy, qx * qx)) * (inv[k] * inv[k]);
      s2m[l] = on ? fmaxf(s2m[l], fin[k] ? s2 : 4.0f) : s2m[l];
      cmn[l] = on ? fminf(cmn[l], fin[k] ? da * inv[k] : -1.0f) : cmn[l];
    }
  }
#pragma unroll
  for (int l = 0; l < 2; ++l) { s2m[l] = tile16_max(s2m[l]); cmn[l] = tile16_min(cmn[l]); }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int l = 0; l < 2; ++l) { L.ext[tile][2 * l] = s2m[l]; L.ext[tile][2 * l + 1] = cmn[l]; }
  }
  __syncthreads();
  if (t < 21) {      // threads 0..15: the tiles; 16..19: the quads; 20: the super-tile
    const int level = t < 16 ? 0 : (t < 20 ? 1 : 2);
    const int q0 = (t >= 16 && t < 20) ? ((t - 16) >> 1) * 8 + ((t - 16) & 1) * 2 : 0;
    float vx = 0.f, vy = 0.f, vz = 0.f, ok = 1.f, n_in = 0.f, s2 = 0.f, cm = 1.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool in = level == 0 ? (i == t) : (level == 1 ? ((i == q0) || (i == q0 + 1) || (i == q0 + 4) || (i == q0 + 5)) : true);
      const float4 v = L.sum[i];
      const float c = L.cnt[i];
      vx += in ? v.x : 0.f; vy += in ? v.y : 0.f; vz += in ? v.z : 0.f;
      ok = (in && c > 0.f) ? fminf(ok, v.w) : ok;
      n_in += in ? c : 0.f;
      const float e_s = level == 0 ? L.ext[i][0] : L.ext[i][2];
      const float e_c = level == 0 ? L.ext[i][1] : L.ext[i][3];
      s2 = (in && c > 0.f && level != 1) ? fmaxf(s2, e_s) : s2;
      cm = (in && c > 0.f && level != 1) ? fminf(cm, e_c) : cm;
    }
    // (the SAME axis the extrema above were measured against: the same rsq of the same sums)
    const float n2 = fmaf(vz, vz, fmaf(vy, vy, vx * vx)), rn = __builtin_amdgcn_rsqf(n2);
    const float Ax = vx * rn, Ay = vy * rn, Az = vz * rn;
    float smax = sqrtf(s2) * (1.0f + 4e-7f) + 4e-7f, cmin = cm - 4e-7f;      // (v_rsq is good to ~1 ulp: padded as block_cone256 pads)
    bool all_ok = ok != 0.f;
    if (level == 1) {
      // the quad: union of its tiles' cones about the quad's own axis.  A ray of tile i makes at most alpha_i + theta_i with
      // that axis (alpha_i: angle between the axes):  cos >= cos(alpha_i) cs_i - sin(alpha_i) sn_i,  sin <= sin(alpha_i) + cos(alpha_i) sn_i
      smax = 0.f; cmin = 1.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = q0 + (k >> 1) * 4 + (k & 1);
        if (!(L.cnt[i] > 0.f)) continue;
        const float4 v = L.sum[i];
        const float trn = __builtin_amdgcn_rsqf(fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));      // the tile's axis, as its own record has it
        const float tx = v.x * trn, ty = v.y * trn, tz = v.z * trn;
        const float t_sn = (sqrtf(L.ext[i][0]) * (1.0f + 4e-7f) + 4e-7f) * (1.0f + 1e-5f) + 1e-7f;      // (cone_finish's margins on the tile's bounds)
        const float t_cs = (L.ext[i][1] - 4e-7f) - 1e-6f;
        const float ca = fmaf(tz, Az, fmaf(ty, Ay, tx * Ax));
        const float qx = fmaf(-ca, Ax, tx), qy = fmaf(-ca, Ay, ty), qz = fmaf(-ca, Az, tz);
        const float sa = sqrtf(fmaf(qz, qz, fmaf(qy, qy, qx * qx))) * (1.0f + 1e-6f) + 1e-7f;
        const float cl = fminf(ca, 1.0f) - 1e-7f;
        if (!(cl > 0.0f) || !(t_cs > 0.0f)) all_ok = false;
        cmin = fminf(cmin, fmaf(cl, t_cs, -sa * t_sn));
        smax = fmaxf(smax, fmaf(fminf(ca + 1e-7f, 1.0f), t_sn, sa));
      }
    }
    const Cone c = cone_finish(Ax, Ay, Az, (n2 * rn) / fmaxf(n_in, 1.f), smax, cmin, all_ok);
    // (a tile / quad without a pixel inside the image: ok = -1, "no such tile" -- nothing is ever tested against it)
    const ConeRec rec = {c.ax, c.ay, c.az, c.cs, c.sn, n_in > 0.f ? (c.ok ? 1.f : 0.f) : -1.f, 0.f, 0.f};
    if (level == 0) c_tile[t] = rec; else if (level == 1) c_quad[t - 16] = rec; else *c_super = rec;
  }
}

// Bounding cone of the rays of stacked rows i0 .. i1 (inclusive), columns j0 .. j1 of a band of h rows and W columns, from the
// four corner rays (see CamView above).  Rows of a stacked band are image rows cam_irow(i), monotone in i: the image-space
// rectangle [irow(i0), irow(i1)] contains every row between (a superset when the block straddles two stripes: still a valid
// cone).  A block outside the band: ok = -1 ("no such tile").  The margins are block_cone256's (rsq's ulp) on top of
// cone_finish's; an interior ray's own fp32 rounding (~1e-7) is far inside them and inside the 2e-5 |mu| of every reach.
__device__ __forceinline__ ConeRec cam_rect_cone(const CamK &k, const CamView &c, int j0, int j1, int i0, int i1) {
  if (j0 >= c.W || i0 >= c.h || j1 < j0 || i1 < i0) return ConeRec{0.f, 0.f, 1.f, 1.f, 0.f, -1.f, 0.f, 0.f};
  j1 = min(j1, c.W - 1); i1 = min(i1, c.h - 1);
  const int r0 = cam_irow(c, i0), r1 = cam_irow(c, i1);
  // (the corner rays are unit to an ulp already -- cam_ray normalises with v_rsq -- and hardware sqrt / rsq serve below: every
  //  bound is padded by far more.  As first written -- ray_dir's IEEE 1 / sqrt per corner, IEEE sqrt per extremum, integer
  //  divisions for the rows -- the two cones cost binB's workgroups ~750 instructions per thread in front of their first load:
  //  binB + 3.6 us, more than half of what the ray launch had cost.)
  float ux[4], uy[4], uz[4];
  cam_ray(k, r0, j0, ux[0], uy[0], uz[0]); cam_ray(k, r0, j1, ux[1], uy[1], uz[1]);
  cam_ray(k, r1, j0, ux[2], uy[2], uz[2]); cam_ray(k, r1, j1, ux[3], uy[3], uz[3]);
  const float sx = (ux[0] + ux[1]) + (ux[2] + ux[3]), sy = (uy[0] + uy[1]) + (uy[2] + uy[3]), sz = (uz[0] + uz[1]) + (uz[2] + uz[3]);
  const float n2 = fmaf(sz, sz, fmaf(sy, sy, sx * sx));
  const float rn = __builtin_amdgcn_rsqf(n2);
  const float ax = sx * rn, ay = sy * rn, az = sz * rn;
  float s2max = 0.f, cmin = 1.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float cl = fmaf(uz[q], az, fmaf(uy[q], ay, ux[q] * ax));
    const float qx = fmaf(-cl, ax, ux[q]), qy = fmaf(-cl, ay, uy[q]), qz = fmaf(-cl, az, uz[q]);
    s2max = fmaxf(s2max, fmaf(qz, qz, fmaf(qy, qy, qx * qx)));
    cmin = fminf(cmin, cl);
  }
  const bool fin = n2 > 1e-30f && n2 < 1e30f;      // (NaN / inf corners: a degenerate camera -- nothing may be culled)
  // margins: the corners' |u| = 1 +- 1e-7 (2e-7 on sin and cos), v_sqrt's ulp, the axis' own normalisation -- 1e-6 covers them
  const Cone cn = cone_finish(ax, ay, az, n2 * rn * 0.25f, __builtin_amdgcn_sqrtf(s2max) * (1.0f + 1e-6f) + 1e-6f, cmin - 1e-6f, fin);
  return ConeRec{cn.ax, cn.ay, cn.az, cn.cs, cn.sn, cn.ok ? 1.f : 0.f, 0.f, 0.f};
}

// binB's two cones at once, the work spread over the lanes: lanes 4 g .. 4 g + 3 of a wave each make ONE corner ray of rectangle g
// (g = 0: the wave's tile, g = 1: the workgroup's quad; every group of four lanes beyond does the same again), the four corners
// meet through DPP quad broadcasts, one lane group finishes each cone and the wave reads both out of lanes 0 and 4.  As two
// cam_rect_cone calls every thread of the workgroup made all eight rays and both cones itself: 310 instructions per wave in front
// of binB's first load (SQ_INSTS_VALU 4.59 -> 5.87 M, binB + 0.9 us).  The same cones as cam_rect_cone's.
__device__ __forceinline__ void cam_two_cones(const CamK &k, const CamView &c, const int tj0, const int ti0, const int te,
                                              const int qj0, const int qi0, const int qe, ConeRec &tile, ConeRec &quad) {
  const int lane = threadIdx.x & 63, g = (lane >> 2) & 1, corner = lane & 3;
  int j0 = g ? qj0 : tj0, i0 = g ? qi0 : ti0;
  const int e = g ? qe : te;
  int j1 = j0 + e - 1, i1 = i0 + e - 1;
  const bool none = j0 >= c.W || i0 >= c.h;
  j1 = min(j1, c.W - 1); i1 = min(i1, c.h - 1);
  j0 = min(j0, c.W - 1); i0 = min(i0, c.h - 1);      // (a rectangle outside the band: any ray, the record says ok = -1)
  float rx, ry, rz;
  cam_ray(k, cam_irow(c, (corner & 2) ? i1 : i0), (corner & 1) ? j1 : j0, rx, ry, rz);
  float ux[4], uy[4], uz[4];
  ux[0] = VOGE_DPP(rx, 0x00); uy[0] = VOGE_DPP(ry, 0x00); uz[0] = VOGE_DPP(rz, 0x00);      // quad_perm [q, q, q, q]
  ux[1] = VOGE_DPP(rx, 0x55); uy[1] = VOGE_DPP(ry, 0x55); uz[1] = VOGE_DPP(rz, 0x55);
  ux[2] = VOGE_DPP(rx, 0xAA); uy[2] = VOGE_DPP(ry, 0xAA); uz[2] = VOGE_DPP(rz, 0xAA);
  ux[3] = VOGE_DPP(rx, 0xFF); uy[3] = VOGE_DPP(ry, 0xFF); uz[3] = VOGE_DPP(rz, 0xFF);
  const float sx = (ux[0] + ux[1]) + (ux[2] + ux[3]), sy = (uy[0] + uy[1]) + (uy[2] + uy[3]), sz = (uz[0] + uz[1]) + (uz[2] + uz[3]);
  const float n2 = fmaf(sz, sz, fmaf(sy, sy, sx * sx));
  const float rn = __builtin_amdgcn_rsqf(n2);
  const float ax = sx * rn, ay = sy * rn, az = sz * rn;
  float s2max = 0.f, cmin = 1.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float cl = fmaf(uz[q], az, fmaf(uy[q], ay, ux[q] * ax));
    const float qx = fmaf(-cl, ax, ux[q]), qy = fmaf(-cl, ay, uy[q]), qz = fmaf(-cl, az, uz[q]);
    s2max = fmaxf(s2max, fmaf(qz, qz, fmaf(qy, qy, qx * qx)));
    cmin = fminf(cmin, cl);
  }
  const bool fin = n2 > 1e-30f && n2 < 1e30f;
  const Cone cn = cone_finish(ax, ay, az, n2 * rn * 0.25f, __builtin_amdgcn_sqrtf(s2max) * (1.0f + 1e-6f) + 1e-6f, cmin - 1e-6f, fin);
  const float okf = none ? -1.f : (cn.ok ? 1.f : 0.f);
  tile = ConeRec{VOGE_LANE(cn.ax, 0), VOGE_LANE(cn.ay, 0), VOGE_LANE(cn.az, 0), VOGE_LANE(cn.cs, 0), VOGE_LANE(cn.sn, 0), VOGE_LANE(okf, 0), 0.f, 0.f};
  quad = ConeRec{VOGE_LANE(cn.ax, 4), VOGE_LANE(cn.ay, 4), VOGE_LANE(cn.az, 4), VOGE_LANE(cn.cs, 4), VOGE_LANE(cn.sn, 4), VOGE_LANE(okf, 4), 0.f, 0.f};
}

// Does the cull record carry an ellipsoid record (trace_fwd.hip, prep_one)?
#ifdef VOGE_NO_ELL   // build without the ellipsoid tests (bounding spheres only): for A/B timing
__device__ __forceinline__ bool cull_has_ell(const float4) { return false; }
#else
__device__ __forceinline__ bool cull_has_ell(const float4 c) { return (__float_as_uint(c.w) & 1u) != 0u && c.w > 0.0f && c.w < 3e38f; }
#endif

// Support function of the hit ellipsoid along n (|n| <= ~1), rounded up.  e0 = (M00, M11, M22, M01),
// e1 = (M02, M12, slack of n^T M n, additive slack).
__device__ __forceinline__ float ell_support(const float4 e0, const float4 e1, const float nx, const float ny, const float nz) {
  const float d = fmaf(e0.z * nz, nz, fmaf(e0.y * ny, ny, e0.x * nx * nx));
  const float o = fmaf(e1.y * ny, nz, fmaf(e1.x * nx, nz, e0.w * nx * ny));
  const float h2 = fmaxf(fmaf(2.0f, o, d), 0.0f) + e1.z;
  return fmaf(sqrtf(h2), 1.0f + 1e-5f, e1.w);
}

// Conservative "no line through the origin with a direction inside the cone touches the
// ellipsoid": each nappe of the double cone lies behind the plane through the apex with normal
// n(+-) = cs u -+ sn a (u = unit radial direction of the centre; valid for any cs <= cos, sn >= sin
// of the cone's half angle), so the ellipsoid misses a nappe when n.mu > h(n).  Only called for
// candidates the sphere test kept.
__device__ __forceinline__ bool cone_keep_ell(const float4 c, const float4 e0, const float4 e1, const Cone &k) {
  if (!k.ok) return true;
  const float p = fmaf(c.z, k.az, fmaf(c.y, k.ay, c.x * k.ax));
  const float rx = fmaf(-p, k.ax, c.x), ry = fmaf(-p, k.ay, c.y), rz = fmaf(-p, k.az, c.z);
  const float q2 = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
  if (!(q2 > 1e-24f)) return true;       // centre on the axis
  const float iq = __builtin_amdgcn_rsqf(q2);
  const float q = q2 * iq;
  const float ux = rx * iq, uy = ry * iq, uz = rz * iq;
  const float qc = q * k.cs, ps = p * k.sn;
  const float err = 4e-6f * (q + fabsf(p));       // rounding of q, p and of the two products
  bool sep = true;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float sg = s ? -k.sn : k.sn;
    const float nx = fmaf(-sg, k.ax, k.cs * ux), ny = fmaf(-sg, k.ay, k.cs * uy), nz = fmaf(-sg, k.az, k.cs * uz);
    const float g = (s ? qc + ps : qc - ps) - err;
    sep = sep && (g > ell_support(e0, e1, nx, ny, nz));
  }
  return !sep;
}

// ------------------------------------------------------------------------------------------
// Wave-private accumulation table in LDS: NV4 float4 values per integer key.
// Only the owning wave touches a table, so slots are claimed with a plain write + read-back and
// values are accumulated with plain read-modify-write; lanes of one instruction that carry the
// SAME key elect one writer per round (owner[] write + read-back), so duplicates are safe.
// (LDS float atomics cost ~3 LDS cycles per lane and value on gfx950 -- SQ_LDS_IDX_ACTIVE in
// profiles/ -- the plain b128 path is ~20x cheaper.)
// ------------------------------------------------------------------------------------------
// A volatile access through a GENERIC pointer compiles into flat_load / flat_store ... sc0 sc1 (the address-space inference
// leaves volatile operations alone): an election round of the table below was a system-scope flat store, a vmcnt wait, a flat
// load and another wait.  These types name the LDS address space outright: ds_write_b32 / ds_read_b32.
typedef __attribute__((address_space(3))) volatile int lds_vint;
typedef __attribute__((address_space(3))) volatile unsigned lds_vuint;
__device__ __forceinline__ lds_vint *lds_volatile(int *p) { return (lds_vint *)p; }
__device__ __forceinline__ const lds_vint *lds_volatile(const int *p) { return (const lds_vint *)p; }
__device__ __forceinline__ lds_vuint *lds_volatile(unsigned *p) { return (lds_vuint *)p; }
typedef __attribute__((address_space(3))) volatile float lds_vfloat;
__device__ __forceinline__ const lds_vfloat *lds_volatile(const float *p) { return (const lds_vfloat *)p; }

// Lanes of ONE wave exchanging values through LDS (a wave's LDS operations complete in issue order, so no wait is needed in
// hardware).  llvm.amdgcn.wave.barrier alone is declared IntrNoMem: only the machine scheduler treats it as a barrier, IR
// passes may still hoist a plain LDS load above an earlier plain store of ANOTHER lane's address.  The
// wavefront-scope release / acquire fences are what orders the memory operations for the optimiser; at wavefront scope they
// emit no instruction.  Every cross-lane LDS hand-over in this library goes through this.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NE, int NV4>
struct WaveTable {
  int keys[NE];
  int owner[NE];
  float4 vals[NE * NV4];
};

template <int NE, int NV4>
__device__ __forceinline__ void wt_clear(WaveTable<NE, NV4> &t, const int lane) {
  for (int i = lane; i < NE; i += VOGE_WAVE) t.keys[i] = -1;
  for (int i = lane; i < NE * NV4; i += VOGE_WAVE) t.vals[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Slot of `key` for every lane with want == true (claiming an empty slot if needed), or -1 when
// no slot was found within kWtProbe steps.  Must be called by the whole wave.
constexpr int kWtProbe = 16;
template <int NE, int NV4>
__device__ __forceinline__ int wt_find(WaveTable<NE, NV4> &t, const int key, const bool want) {
  unsigned h = ((unsigned)key * 2654435761u) >> (32 - __builtin_ctz(NE));
  int slot = -1;
  bool pending = want;
#pragma unroll 1
  for (int pr = 0; pr < kWtProbe && __any(pending); ++pr) {
    if (pending) {
      // one LDS compare-and-swap per probe: claims an empty slot or reports who holds it
      const int old = atomicCAS(&t.keys[h], -1, key);
      if (old == -1 || old == key) {
        slot = (int)h;
        pending = false;
      } else {
        h = (h + 1) & (NE - 1);
      }
    }
  }
  return slot;
}

// The same for TWO keys per lane in one probing loop: both compare-and-swaps of a step are in flight together, and the loop
// lasts as long as the longer of a lane's two probe sequences instead of their sum.  Must be called by the whole wave.
template <int NE, int NV4>
__device__ __forceinline__ void wt_find2(WaveTable<NE, NV4> &t, const int key0, const bool want0, const int key1, const bool want1,
                                         int &slot0, int &slot1) {
  unsigned h0 = ((unsigned)key0 * 2654435761u) >> (32 - __builtin_ctz(NE));
  unsigned h1 = ((unsigned)key1 * 2654435761u) >> (32 - __builtin_ctz(NE));
  slot0 = slot1 = -1;
  bool p0 = want0, p1 = want1;
#pragma unroll 1
  for (int pr = 0; pr < kWtProbe && __any(p0 | p1); ++pr) {
    int old0 = -2, old1 = -2;
    if (p0) old0 = atomicCAS(&t.keys[h0], -1, key0);
    if (p1) old1 = atomicCAS(&t.keys[h1], -1, key1);
    if (p0) {
      if (old0 == -1 || old0 == key0) { slot0 = (int)h0; p0 = false; } else h0 = (h0 + 1) & (NE - 1);
    }
    if (p1) {
      if (old1 == -1 || old1 == key1) { slot1 = (int)h1; p1 = false; } else h1 = (h1 + 1) & (NE - 1);
    }
  }
}

// vals[slot] += v for every lane with `on` (slot >= 0).  Must be called by the whole wave.
template <int NE, int NV4>
__device__ __forceinline__ void wt_add(WaveTable<NE, NV4> &t, const int slot, const float4 (&v)[NV4],
                                       const bool on, const int lane) {
  lds_vint *owner = lds_volatile(t.owner);
  bool pending = on;
#pragma unroll 1
  while (__any(pending)) {
    if (pending) owner[slot] = lane;
    if (pending && owner[slot] == lane) {
      float4 *dst = t.vals + slot * NV4;
#pragma unroll
      for (int q = 0; q < NV4; ++q) {
        float4 x = dst[q];
        x.x += v[q].x; x.y += v[q].y; x.z += v[q].z; x.w += v[q].w;
        dst[q] = x;
      }
      pending = false;
    }
  }
}

// Two accumulations per lane in ONE election loop: a lane contends for its first slot until it wins it, then for its
// second -- lanes that are through with one key move on while others still queue for theirs.  (Two wt_add calls in a row
// cost the sum of their election rounds: 3.4 + 3.4 at cfg3; tools/fb_sections.py.)  Must be called by the whole wave.
template <int NE, int NV4>
__device__ __forceinline__ void wt_add2(WaveTable<NE, NV4> &t, const int slot0, const float4 (&v0)[NV4], const bool on0,
                                        const int slot1, const float4 (&v1)[NV4], const bool on1, const int lane) {
  lds_vint *owner = lds_volatile(t.owner);
  bool p0 = on0, p1 = on1;
#pragma unroll 1
  while (__any(p0 | p1)) {
    const bool use0 = p0, pend = p0 | p1;
    const int s = use0 ? slot0 : slot1;
    if (pend) owner[s] = lane;
    if (pend && owner[s] == lane) {
      float4 *dst = t.vals + s * NV4;
#pragma unroll
      for (int q = 0; q < NV4; ++q) {
        float4 x = dst[q];
        x.x += use0 ? v0[q].x : v1[q].x; x.y += use0 ? v0[q].y : v1[q].y;
        x.z += use0 ? v0[q].z : v1[q].z; x.w += use0 ? v0[q].w : v1[q].w;
        dst[q] = x;
      }
      p1 = use0 ? p1 : false;
      p0 = false;
    }
  }
}

// The same two accumulations per lane WITHOUT an election, for callers whose lanes come in groups that cannot collide: the
// lanes of one PIXEL hold distinct Gaussians (a Gaussian sits in a pixel's list once), so the pixels of a round take the table
// one after the other -- `ord` = the lane's pixel ordinal in the round, 0 .. n - 1 -- and inside a step every lane reads,
// adds and writes its two entries with no owner word written, read back and compared.  The fused backward's rounds hold 3-4
// pixels and its elections took 3.4 rounds: as many steps, each a third cheaper.  Must be called by the whole wave.
template <int NE, int NV4>
__device__ __forceinline__ void wt_add2_by_group(WaveTable<NE, NV4> &t, const int slot0, const float4 (&v0)[NV4], const bool on0,
                                                 const int slot1, const float4 (&v1)[NV4], const bool on1, const int ord) {
  bool pend = on0 | on1;
#pragma unroll 1
  for (int g = 0; __any(pend); ++g) {
    if (pend && ord == g) {
      if (on0) {
        float4 *dst = t.vals + slot0 * NV4;
#pragma unroll
        for (int q = 0; q < NV4; ++q) {
          float4 x = dst[q];
          x.x += v0[q].x; x.y += v0[q].y; x.z += v0[q].z; x.w += v0[q].w;
          dst[q] = x;
        }
      }
      if (on1) {
        float4 *dst = t.vals + slot1 * NV4;
#pragma unroll
        for (int q = 0; q < NV4; ++q) {
          float4 x = dst[q];
          x.x += v1[q].x; x.y += v1[q].y; x.z += v1[q].z; x.w += v1[q].w;
          dst[q] = x;
        }
      }
      pend = false;
    }
    wave_lds_sync();      // (the next pixel's reads come after this pixel's writes in the wave's LDS queue)
  }
}

// Compact the occupied slots into t.owner[0..n) (the election array is free once accumulation
// is over) and return n, so the flush issues full-width atomics instead of walking empty slots.
// Must be called by the whole wave.
template <int NE, int NV4>
__device__ __forceinline__ int wt_compact(WaveTable<NE, NV4> &t, const int lane) {
  lds_vint *list = lds_volatile(t.owner);
  int n = 0;
#pragma unroll
  for (int base = 0; base < NE; base += VOGE_WAVE) {
    const bool occ = (base + lane < NE) && (t.keys[base + lane] >= 0);
    const unsigned long long m = __ballot(occ);
    if (occ) list[n + __popcll(m & ((1ull << lane) - 1ull))] = base + lane;
    n += __popcll(m);
  }
  return n;
}

// Sum over each aligned group of 16 lanes (a DPP "row"), result in every lane of the row.
// Four VALU instructions with DPP operands -- no LDS traffic, unlike ds_bpermute shuffles.
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// Segmented sum over runs of `seg` consecutive lanes (seg <= 64, any value); the total of each
// run lands in its first lane.
__device__ __forceinline__ float seg_sum(float x, const int lane, const int seg) {
  const int r = lane % seg;
  for (int o = 1; o < seg; o <<= 1) {
    const float y = __shfl_down(x, o, VOGE_WAVE);
    if (r + o < seg) x += y;
  }
  return x;
}

// Segmented sum by key: lanes holding equal `key` form contiguous runs (key is monotone along
// the lanes); the total of each run lands in its FIRST lane.
__device__ __forceinline__ float seg_sum_key(float x, const int key, const int lane) {
#pragma unroll
  for (int o = 1; o < VOGE_WAVE; o <<= 1) {
    const float y = __shfl_down(x, o, VOGE_WAVE);
    const int k2 = __shfl_down(key, o, VOGE_WAVE);
    if (lane + o < VOGE_WAVE && k2 == key) x += y;
  }
  return x;
}

inline int launch_status() { return (int)hipGetLastError(); }

// Fill `bytes` (a multiple of 4) at p with one 32-bit pattern -- by a kernel of this library, never hipMemsetAsync.
// A memset captured into a HIP graph (torch.cuda.graph around a training iteration) did NOT reliably take effect on
// replay on ROCm 7.2 / gfx950: the fused backward's accumulators kept the previous contents and one replayed SGD step sent
// every vertex to 1e20 (tools/loop_graph_check.py, round 3).  A kernel node has no such problem, and it is one launch where
// the runtime turns a memset into two fill kernels.
static __global__ void __launch_bounds__(256) voge_fill32_kernel(uint32_t *__restrict__ p, const size_t n, const uint32_t value) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n && (reinterpret_cast<uintptr_t>(p + i) & 15) == 0) {
    *reinterpret_cast<uint4 *>(p + i) = make_uint4(value, value, value, value);
  } else {
    for (size_t j = i; j < n && j < i + 4; ++j) p[j] = value;
  }
}
inline hipError_t voge_fill_async(void *p, const int byte_value, const size_t bytes, hipStream_t st) {
  if (bytes == 0 || p == nullptr) return hipSuccess;
  const uint32_t b = (uint32_t)(byte_value & 0xff), v = b | (b << 8) | (b << 16) | (b << 24);
  const size_t n = bytes / 4;      // (every caller fills float / int32 arrays)
  hipLaunchKernelGGL(voge_fill32_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, reinterpret_cast<uint32_t *>(p), n, v);
  return hipGetLastError();
}

// The opt-in for more than 64 KB of dynamic LDS is a per-function, per-device attribute.  Set it when the request
// grows -- not on every call, and not again inside a stream capture once a frame of this size has run.
struct DynLdsCache {
  std::atomic<size_t> set[16];
  DynLdsCache() { for (auto &v : set) v.store(0); }
};
inline int ensure_dynamic_lds(const void *func, const size_t lds, DynLdsCache &cache) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
  if (dev >= 0 && lds <= cache.set[dev].load(std::memory_order_relaxed)) return 0;
  const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0) cache.set[dev].store(lds, std::memory_order_relaxed);
  return 0;
}

}  // namespace voge
