// Round 4: the scalar-sigma sweep, rebuilt around what round 3's per-tile timers and a VALU / LDS issue micro-benchmark
// (tools/valu_bench.hip, profiles/r4_valu_bench.txt) said about the old kernel (trace_fwd_kernel<1, true>):
//   * a tile spends two thirds of its life in the consume loop, ~160 ns per candidate at 1.75 waves per SIMD: ~27 VALU +
//     ~14 SALU + 3 LDS instructions, of which the 64-bit (ord(len) << 32 | id) keys cost the most (key building, two 64-bit
//     compares at ~2x the price of a float compare, register pairs through every select);
//   * the 8-byte keys are also what holds the kernel at 7 waves per CU.
// Here a list entry is the fp32 len itself plus a 16-bit POSITION in the tile's candidate stream (6 bytes: 9 workgroups per
// CU at K = 40).  Order and admission are decided by plain float compares; the Gaussian's id is looked up only when two
// lens tie exactly (the slow path) and in the epilogue, so the result is still "the K lexicographically smallest
// (len, id)" of ray_trace_voge.cu:197-212.  The staged chunk is kept as SoA (x[], y[], z[], a[]): one ds_read_b128 per
// array serves four candidates, addresses are immediates, and no id is read in the loop at all.
// Streams longer than 65536 entries (a pooled list or the stream-everything fallback of a large scene) take the same
// code with 32-bit positions over HALF the rays at a time (two passes: the same LDS bytes hold 32 columns of u32).
#pragma once
#include <type_traits>

#include "voge_common.h"

namespace voge {

#ifndef VOGE_SWEEP_V2
#define VOGE_SWEEP_V2 1
#endif

constexpr int kS2TP = 65;      // floats per row of the len array (odd: the epilogue's transposed reads stay conflict-light)
constexpr int kS2TQ = 66;      // u16 per row of the position array (= 33 u32 in the wide form)
constexpr int kS2Pad = 4;      // never-hit records behind the staged chunk: a trip of four needs no bounds

struct Sweep2Stage {
  float x[64 + kS2Pad], y[64 + kS2Pad], z[64 + kS2Pad], a[64 + kS2Pad];      // the staged chunk's (mu, a), SoA
  union {
    float lb[64];      // its len bounds (the exit test)
    int cnt[64];       // epilogue: hits per ray
  };
  int pos[64 + kS2Pad];      // the staged entries' handles: Gaussian id, or position in the tile's stream (see h_is_id)
};
__host__ __device__ inline size_t sweep2_len_bytes(const int K) { return (sizeof(float) * (size_t)(K + 1) * kS2TP + 15) & ~(size_t)15; }
__host__ __device__ inline size_t sweep2_pos_bytes(const int K) { return (sizeof(uint16_t) * (size_t)(K + 1) * kS2TQ + 15) & ~(size_t)15; }
// GEN (full 3x3 forms, round 5): the staged chunk's eleven further coefficients of pair_eval_gen's record -- s11, s22, s01, s02,
// s12, b (3), k (3); s00 sits in Sweep2Stage::a, where an isotropic candidate keeps its a -- SoA like the rest.  e[0][i]
// (s11) is NaN for a staged ISOTROPIC candidate (and for the padding): that is how the consume loop tells the two apart.
struct Sweep2Gen {
  float e[11][64 + kS2Pad];
};
#ifndef VOGE_S2_QUADS
#define VOGE_S2_QUADS 0      // 1: the scalar kernel, a tile's own list: one candidate stream per 4x4-pixel QUADRANT of the tile (see Sweep2Quad).
                             // Built and measured in round 6 (profiles/r6_quads_ab.txt): bit-identical results, 30 % fewer candidates
                             // evaluated per tile (93.7 -> 65.5), and SLOWER: a trip of four costs 669 instead of 490 ns (four
                             // distinct record addresses per ds_read_b128 instead of one broadcast, 12 v_mov to pair the AoS
                             // records for the packed evaluation), staging + 1.0, prologue + 0.6, epilogue + 1.3 us per tile
                             // (ids through the stream): lean trace 63.1 -> 65.7 us, entry 71.2 -> 75.0 us.  Off.
#endif
// QUADS (round 6; tools/quadrant_sim.py, HISTORY R4: a candidate of an 8x8 tile's list hits 28-37 % of its rays, one stream per
// 4x4 quadrant cuts a tile's trips from 94 to 58.5).  The staged chunk is kept as AoS (x, y, z, a) records in the bytes of
// Sweep2Stage::x..a, every candidate is tested against the four quadrants' bounding cones when it is staged, and the survivors'
// chunk positions are compacted into four byte lists (in the bytes of Sweep2Stage::pos: a handle is the stream position now,
// which needs no look-up).  A trip then evaluates, in every lane, the next four candidates of ITS quadrant's list; the wave
// runs as long as its longest quadrant.  The cones live here.
struct Sweep2Quad {
  float4 c[4];                  // (axis, cos) of quadrant q = (column >> 2) + 2 * (row >> 2)
  float sn[4], ok[4];
};
constexpr int kS2QL = 64 + kS2Pad;      // bytes per quadrant list (padded with 64 = the never-hit record's position)
static_assert(4 * kS2QL == sizeof(int) * (64 + kS2Pad), "the four byte lists take Sweep2Stage::pos's bytes");
// gen: 0 scalar sigmas, 1 general forms (eleven coefficient arrays), 2 a launch of per-axis forms only (five: s11, s22, b)
constexpr int kS2DiagRows = 5;
__host__ __device__ inline size_t sweep2_lds_bytes(const int K, const int gen = 0) {
  return sweep2_len_bytes(K) + sweep2_pos_bytes(K) + sizeof(Sweep2Stage) +
         (gen == 1 ? sizeof(Sweep2Gen) : (gen == 2 ? sizeof(float) * kS2DiagRows * (64 + kS2Pad) : (VOGE_S2_QUADS ? sizeof(Sweep2Quad) : 0)));
}

#ifndef VOGE_S2_EPI_B
#define VOGE_S2_EPI_B 5      // epilogue: items (4 slots each) per thread and batch
#endif
#ifndef VOGE_S2_EPI_R
#define VOGE_S2_EPI_R 3      // ... and how many of them have their (mu, a) gathers in flight together (act / dsd wanted)
#endif
#ifndef VOGE_S2_READLANE
#define VOGE_S2_READLANE 0      // the tile's own list: records out of the lanes' registers (v_readlane) instead of LDS staging
#endif
#ifndef VOGE_S2_PACKED
#define VOGE_S2_PACKED 1        // two candidates per packed-fp32 instruction in the evaluation
#endif
#ifndef VOGE_S2_EXIT_GROUP
#define VOGE_S2_EXIT_GROUP 16   // candidates between two exit tests (a power of two, >= 4)
#endif
constexpr int kExitGroup = VOGE_S2_EXIT_GROUP;
#ifndef VOGE_S2_PREFETCH
#define VOGE_S2_PREFETCH 1      // the next trip's staged records are requested before this trip's commits
#endif
#ifndef VOGE_S2_PUT_AT_CNT
#define VOGE_S2_PUT_AT_CNT 1    // the 16-bit form's commit stores every candidate at row cnt (no row select); see commit()
#endif
#ifndef VOGE_S2_GEN_LEAN
#define VOGE_S2_GEN_LEAN 0      // (experiment) the general kernel takes the scalar kernel's lean_insert / slow_insert instead of deep_insert
#endif
#ifndef VOGE_S2_PRIO_LEN
#define VOGE_S2_PRIO_LEN 0   // (experiment) tiles with at least this many candidates run at raised wave priority; 0: off
#endif

// GEN = 2 (round 6): a launch whose general forms are ALL diagonal (the frame path's per-axis sigmas, gen_kind 1): five of the eleven
// coefficient arrays exist (8 instead of 7 workgroups per CU at K = 40), the trips are the diagonal ones, and the insertion is the
// scalar kernel's (tools/quads_stats.py SLOW=1: the walks of such a scene are as short as a scalar one's -- 304 against 542 ns per event).
// GEN = 0: every Gaussian is A = a I (ms = (mu, a)).  GEN = 1: the general entry points' kernel -- ms = (mu, s00 | NaN),
// NaN sending the reader to evr[3 g .. 3 g + 2], pair_eval_gen's record (VoGE/csrc/ray_trace_voge/ray_trace_voge.cu:11-38: all
// nine entries of isigmas are inputs); an isotropic Gaussian among them is evaluated exactly as the scalar kernel does.
template <int GEN>
__global__ void __launch_bounds__(64)
sweep_iso_kernel(const float4 *__restrict__ cull, const float4 *__restrict__ ms, const float4 *__restrict__ evr, const float *__restrict__ rays,
                 const int *__restrict__ bin_count, const int32_t *__restrict__ bin_id, const float *__restrict__ bin_lb,
                 const int32_t *__restrict__ tl_id, const float *__restrict__ tl_lb, const int32_t *__restrict__ pool_id,
                 const float *__restrict__ pool_lb, const int *__restrict__ tl_off, const int2 *__restrict__ order,
                 const int tiles_per_img, const int nstx, const int nst, const int N, const int H, const int W, const int K,
                 const float thr_act, int32_t *__restrict__ out_idx, float *__restrict__ out_len, float *__restrict__ out_act,
                 float *__restrict__ out_dsd, int32_t *__restrict__ out_cnt,
                 const CamView cam /* R != NULL (round 6): `rays` is NULL, every lane makes its pixel's ray from the camera and the
                                      tile leaves its rays in cam.rays_out for the composite and the backward */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *const Llen = reinterpret_cast<float *>(smem_raw);
  unsigned char *const Lpos_raw = smem_raw + sweep2_len_bytes(K);
  Sweep2Stage &S = *reinterpret_cast<Sweep2Stage *>(smem_raw + sweep2_len_bytes(K) + sweep2_pos_bytes(K));
  Sweep2Gen &G = *reinterpret_cast<Sweep2Gen *>(smem_raw + sweep2_len_bytes(K) + sweep2_pos_bytes(K) + sizeof(Sweep2Stage));      // (GEN only)
  // (GEN = 2: rows s11, s22, bx, by, bz of the eleven -- the others do not exist in that launch's LDS)
  auto grow = [&](const int r) -> float * { return &G.e[GEN == 2 ? (r < 2 ? r : r - 3) : r][0]; };
  auto ghas = [](const int r) { return GEN != 2 || r < 2 || (r >= 5 && r < 8); };
  Sweep2Quad &QC = *reinterpret_cast<Sweep2Quad *>(smem_raw + sweep2_len_bytes(K) + sweep2_pos_bytes(K) + sizeof(Sweep2Stage));   // (!GEN, QUADS)
  float4 *const QR = reinterpret_cast<float4 *>(&S.x[0]);                      // (QUADS) the staged records, AoS
  [[maybe_unused]] unsigned char *const Q8 = reinterpret_cast<unsigned char *>(&S.pos[0]);      // (QUADS) the four position lists

  const int lane = threadIdx.x;
  const int tiles_x = (W + 7) >> 3;
#ifdef VOGE_SWEEP_TIMES
  const unsigned long long ts0 = wall_clock64();
  unsigned long long ts_fill = 0, ts_cons = 0, ts1 = 0, ts2 = 0;
  unsigned st_eval = 0;
  unsigned long long ts_slow = 0; unsigned st_slow = 0, st_lanes = 0, st_moved = 0, st_far = 0;
  int dbg_moved = 0;
#endif
  const int2 slot = order[blockIdx.x];               // (tile [| kPoolFlag], length of its list | -1 = overflowed)
  if (slot.x < 0) return;                            // outside the image
  if (slot.y == 0) {
    // Nothing can hit this tile: its all-sentinel outputs (ray_trace_voge.cu:244-247), 20 KB per array pair at K = 40.
    // The renderer's form (no act / dsd kept: 31 MB per frame at cfg3) writes them HERE: these slots sit behind the loaded
    // ones in the launch order, and their stores go out while the heavy tiles compute (binB 21.1 -> 18.6 us, this kernel
    // + 0.3).  With act / dsd (62 MB) the sweep's own epilogues already keep HBM busy -- the same move costs this kernel
    // 4.7 us for 4.4 saved -- and binB keeps writing them.
    if (out_act != nullptr) return;
    const int lin0 = slot.x & ~kPoolFlag;      // (an empty tile of a pooled quad carries the flag too)
    const int b = lin0 / tiles_per_img, bx = lin0 - b * tiles_per_img;
    const int ftx = bx % tiles_x, fty = bx / tiles_x;
    const int tw = min(8, W - ftx * 8), th = min(8, H - fty * 8);
    const int row_items = tw * K;
    if ((K & 3) == 0) {
      const int ipr = row_items >> 2;
      for (int it = lane; it < th * ipr; it += 64) {
        const int rr = it / ipr, j4 = it - rr * ipr;
        const size_t o = (((size_t)b * H + fty * 8 + rr) * W + (size_t)ftx * 8) * K + (size_t)j4 * 4;
        st16i<(VOGE_NT_STORES & 1) != 0>(out_idx + o, -1, -1, -1, -1);
        st16f<(VOGE_NT_STORES & 1) != 0>(out_len + o, VOGE_SENT_LEN, VOGE_SENT_LEN, VOGE_SENT_LEN, VOGE_SENT_LEN);
      }
    } else {
      for (int it = lane; it < th * row_items; it += 64) {
        const int rr = it / row_items, j = it - rr * row_items;
        const size_t o = (((size_t)b * H + fty * 8 + rr) * W + (size_t)ftx * 8) * K + j;
        out_idx[o] = -1; out_len[o] = VOGE_SENT_LEN;
      }
    }
    if (out_cnt != nullptr) {
      const int rr = lane >> 3, x = lane & 7;
      if (rr < th && x < tw) out_cnt[((size_t)b * H + fty * 8 + rr) * W + (size_t)ftx * 8 + x] = 0;
    }
    if (cam.R != nullptr && cam.rays_out != nullptr) {      // (uniform) the tile's rays, for whoever reads the bundle later
      const int rr = lane >> 3, x = lane & 7;
      if (rr < th && x < tw) {
        const CamK ck = cam_load(cam, b);
        float ex, ey, ez;
        cam_ray(ck, cam_irow(cam, fty * 8 + rr), ftx * 8 + x, ex, ey, ez);
        float *o = cam.rays_out + (((size_t)b * H + fty * 8 + rr) * W + (size_t)ftx * 8 + x) * 3;
        o[0] = ex; o[1] = ey; o[2] = ez;
      }
    }
    return;
  }
  const bool pooled = (slot.x & kPoolFlag) != 0;
  const int lin = slot.x & ~kPoolFlag;
  const int b = lin / tiles_per_img, bx = lin - b * tiles_per_img;
  const int tx = bx % tiles_x, ty = bx / tiles_x;
  // ---- the tile's candidate stream: its own list (binB) -> its quad's ordered list -> every Gaussian of the element
  const int bin = (b * nst + ((ty * 8) / kST) * nstx + (tx * 8) / kST) * 4 + (((ty * 8) / kQuad) & 1) * 2 + (((tx * 8) / kQuad) & 1);
  const int tc = slot.y;
  const int bc = (tc >= 0) ? tc : ((bin_count != nullptr) ? bin_count[bin] : -1);
  const bool binned = bc >= 0;
  const int src_n = binned ? bc : N;
  const size_t list_at = pooled ? (size_t)tl_off[lin] : (size_t)lin * kTileCap;
  const int32_t *src_id = (tc >= 0) ? (pooled ? pool_id : tl_id) + list_at : (binned ? bin_id + (size_t)bin * kQCap : nullptr);
  const float *src_lb = (tc >= 0) ? (pooled ? pool_lb : tl_lb) + list_at : (binned ? bin_lb + (size_t)bin * kQCap : nullptr);
  const bool pref = tc >= 0;      // the tile's own list is already filtered with this tile's cone
#if VOGE_S2_PRIO_LEN > 0
  if (src_n >= VOGE_S2_PRIO_LEN) __builtin_amdgcn_s_setprio(3);
#endif
  const float4 *cullb = cull + (size_t)b * N;
  const float4 *msb = ms + (size_t)b * N;
  const float4 rec_none = make_float4(0.f, 0.f, 0.f, INFINITY);      // act = inf * 0 = NaN: never below the threshold
  const float4 cull_none = make_float4(0.f, 0.f, 0.f, -1.f);
  // What a list entry carries in its 16 (32) bits: the Gaussian's id itself where every id of the batch element fits 16
  // bits -- no look-up anywhere --, else the entry's POSITION in the tile's stream, resolved through the stream's id list
  // on an exact len tie and in the epilogue.
  const bool quads = (VOGE_S2_QUADS != 0) && !GEN && pref && src_n <= 65536;      // (uniform) the per-quadrant form: handles are stream positions
  const bool h_is_id = !quads && ((N <= 65536) || !binned);
  auto id_of = [&](const unsigned h) -> int { return h_is_id ? (int)h : src_id[h]; };
  const float not_full = __uint_as_float(__float_as_uint(VOGE_SENT_LEN) - 1u);      // len <= this  <=>  len < the sentinel
  if (quads) {
    if (lane < kS2Pad) QR[64 + lane] = rec_none;
  } else if (lane < kS2Pad) { S.x[64 + lane] = 0.f; S.y[64 + lane] = 0.f; S.z[64 + lane] = 0.f; S.a[64 + lane] = INFINITY; }
  if (GEN && lane < kS2Pad) G.e[0][64 + lane] = __uint_as_float(0x7fc00000u);      // (padding counts as isotropic)
  const float4 *evrb = GEN ? evr + (size_t)b * N * 3 : nullptr;
  bool tile_gen = false;      // (GEN) a general candidate was staged at some point: the epilogue needs the full records

  // ---- the wave's bounding cone over all 64 rays of the tile (only a cone-filtered stream needs it), unit-ray flag ----
  // (evaluated inside run(), BEHIND the first requests for the tile's list: the flag is needed at the first exit test only,
  // and as a block in front of run() its ray load was waited for -- one memory round trip -- before the list's ids were
  // even asked for: the prologue was order -> rays -> ids -> records, four dependent trips; now rays and ids travel together)
  Cone wcone;
  bool unit_rays = false;
  wcone.ok = false;
  const bool from_cam = cam.R != nullptr;      // (uniform)
  CamK ck;
  if (from_cam) ck = cam_load(cam, b);
  // pixel (x, y) of the band's ray: out of the bundle, or made here (rays_fwd_kernel's operations: the same bits)
  auto ray_of = [&](const int x, const int y, float &rx, float &ry, float &rz) {
    if (from_cam) {
      cam_ray(ck, cam_irow(cam, y), x, rx, ry, rz);
    } else {
      const size_t rid = ((size_t)b * H + y) * W + x;
      rx = rays[3 * rid + 0]; ry = rays[3 * rid + 1]; rz = rays[3 * rid + 2];
    }
  };
  auto ray_setup = [&](const bool mine, const float mdx, const float mdy, const float mdz) {
    // (mine: the caller's (dx, dy, dz) IS lane's ray of the tile -- the 16-bit form; the wide form's lanes hold other rays)
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    float ox_ = mdx, oy_ = mdy, oz_ = mdz;
    if (!mine) ray_of(min(px, W - 1), min(py, H - 1), ox_, oy_, oz_);
    const RayDir u = ray_dir(ox_, oy_, oz_);
    unit_rays = __all(!u.ok || u.unit);
    wcone.ok = false;
    if (!pref) {      // (uniform)
      const bool dirs_ok = __all(u.ok);
      const float wsx = wave_sum_dpp(u.ok ? u.ux : 0.f), wsy = wave_sum_dpp(u.ok ? u.uy : 0.f), wsz = wave_sum_dpp(u.ok ? u.uz : 0.f);
      const float n = sqrtf(fmaf(wsz, wsz, fmaf(wsy, wsy, wsx * wsx)));
      const float ax = wsx / n, ay = wsy / n, az = wsz / n;
      float smax = 0.f, cmin = 1.f;
      cone_partial(u, ax, ay, az, smax, cmin);
      wcone = cone_finish(ax, ay, az, n, wave_max(smax), wave_min(cmin), dirs_ok);
    }
    if (quads) {      // (uniform) the four quadrants' cones: reductions over lane bits 0, 1 (columns) and 3, 4 (rows)
      // (bit 4 = the other 16-lane row of the pair: v_permlane16_swap hands every lane both rows' values -- no LDS round trip)
      auto rows = [&](const float v, float &lo, float &hi) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
      };
      auto qsum = [&](float v) { v += VOGE_DPP(v, 0xB1); v += VOGE_DPP(v, 0x4E); v += VOGE_DPP(v, 0x128); float a, b2; rows(v, a, b2); return a + b2; };
      auto qmax = [&](float v) { v = fmaxf(v, VOGE_DPP(v, 0xB1)); v = fmaxf(v, VOGE_DPP(v, 0x4E)); v = fmaxf(v, VOGE_DPP(v, 0x128)); float a, b2; rows(v, a, b2); return fmaxf(a, b2); };
      const float sx = qsum(u.ok ? u.ux : 0.f), sy = qsum(u.ok ? u.uy : 0.f), sz = qsum(u.ok ? u.uz : 0.f);
      const float n2 = fmaf(sz, sz, fmaf(sy, sy, sx * sx));
      const float rn = __builtin_amdgcn_rsqf(n2);      // (the axis need not be exactly unit: the bounds below are taken against THIS axis, renormalised)
      float ax = sx * rn, ay = sy * rn, az = sz * rn;
      const float fix = __builtin_amdgcn_rsqf(fmaf(az, az, fmaf(ay, ay, ax * ax)));
      ax *= fix; ay *= fix; az *= fix;
      float smax = 0.f, cmin = 1.f;
      {
        const float cl = fmaf(u.uz, az, fmaf(u.uy, ay, u.ux * ax));
        const float rx = fmaf(-cl, ax, u.ux), ry = fmaf(-cl, ay, u.uy), rz = fmaf(-cl, az, u.uz);
        const float sl = __builtin_amdgcn_sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
        smax = u.ok ? sl : 2.0f; cmin = u.ok ? cl : -1.0f;
      }
      smax = qmax(smax) * (1.0f + 1e-6f) + 1e-6f; cmin = -qmax(-cmin) - 1e-6f;      // (hardware sqrt / rsq, ~1 ulp: margins as cam_rect_cone's)
      const int myq = ((lane >> 2) & 1) | ((lane >> 4) & 2);
      const unsigned long long qm = (0x0F0F0F0Full << (4 * (myq & 1))) << (32 * (myq >> 1));
      const bool qok = (__ballot(u.ok) & qm) == qm;
      const Cone c = cone_finish(ax, ay, az, n2 * rn * 0.25f, smax, cmin, qok);
      if ((lane & 0x1B) == 0) { QC.c[myq] = make_float4(c.ax, c.ay, c.az, c.cs); QC.sn[myq] = c.sn; QC.ok[myq] = c.ok ? 1.f : 0.f; }
    }
  };

  // One pass = the whole trace of the rays it covers: all 64 (16-bit positions) or rows [4 pass, 4 pass + 4) of the tile
  // on lanes 0..31 (32-bit positions; lanes 32..63 evaluate along and never commit).
  auto run = [&](auto wide_tag, const int pass) {
    constexpr bool WIDE = decltype(wide_tag)::value;
    const int r = WIDE ? pass * 32 + (lane & 31) : lane;      // the lane's ray inside the tile
    const int col = WIDE ? (lane & 31) : lane;                 // ... and its column of the lists
    const int px = tx * 8 + (r & 7), py = ty * 8 + (r >> 3);
    const bool valid = (px < W) && (py < H) && (!WIDE || lane < 32);
    const size_t ray_id = ((size_t)b * H + min(py, H - 1)) * W + min(px, W - 1);
    // (the list's first two chunks of ids and bounds are requested before anything else -- see ray_setup)
    int id0 = (lane < src_n) ? (binned ? src_id[lane] : lane) : -1;
    float lb0 = (binned && lane < src_n) ? src_lb[lane] : -INFINITY;
    int id1 = (64 + lane < src_n) ? (binned ? src_id[64 + lane] : 64 + lane) : -1;
    float lb1 = (binned && 64 + lane < src_n) ? src_lb[64 + lane] : -INFINITY;
    float dx, dy, dz;
    ray_of(min(px, W - 1), min(py, H - 1), dx, dy, dz);
    if (from_cam && cam.rays_out != nullptr && valid) {      // (the bundle, for the kernels behind the sweep: 12 bytes per lane)
      float *o = cam.rays_out + 3 * ray_id;
      o[0] = dx; o[1] = dy; o[2] = dz;
    }
    // (unconditional 16-byte loads at a clamped index, then a select: the `id >= 0 ? load : constant` form compiled into four
    //  dword loads per record)
    //  dword loads per record, each in an exec-mask region of its own -- and the compiler re-derives that from any select
    //  around a plain load.  Buffer loads: an id of -1 is an offset beyond the range and reads zeros, .w is fixed up.)
    const __amdgpu_buffer_rsrc_t rs_ms = out_rsrc(const_cast<float4 *>(msb), (unsigned)N * 16u),
                                 rs_cull = out_rsrc(const_cast<float4 *>(cullb), (unsigned)N * 16u);
    auto load_ms = [&](const int id) {
      const voge_v4f r = __builtin_amdgcn_raw_buffer_load_b128(rs_ms, id * 16, 0, 0);
      return make_float4(r[0], r[1], r[2], (id >= 0) ? r[3] : INFINITY);
    };
    auto load_cull = [&](const int id) {
      const voge_v4f r = __builtin_amdgcn_raw_buffer_load_b128(rs_cull, id * 16, 0, 0);
      return make_float4(r[0], r[1], r[2], (id >= 0) ? r[3] : -1.f);
    };
    // (GEN: pair_eval_gen's record, three 16-byte loads per candidate, requested with (mu, s00 | NaN) -- for every candidate of
    //  a general launch: whether it is needed is known only when that first load is back)
    const __amdgpu_buffer_rsrc_t rs_ev = out_rsrc(const_cast<float4 *>(GEN ? evrb : msb), GEN ? (unsigned)N * 48u : 0u);
    auto load_ev = [&](const int id, const int k) {
      const voge_v4f r = __builtin_amdgcn_raw_buffer_load_b128(rs_ev, id * 48 + 16 * k, 0, 0);
      return make_float4(r[0], r[1], r[2], r[3]);
    };
    // (the first chunk's records go out as soon as its ids are here; the rays' set-up runs while they travel)
    float4 m0 = load_ms(id0);
    float4 c0 = pref ? cull_none : load_cull(id0);
    float4 ea0 = rec_none, ea1 = rec_none, ea2 = rec_none;
    if (GEN) { ea0 = load_ev(id0, 0); ea1 = load_ev(id0, 1); ea2 = load_ev(id0, 2); }
    if (pass == 0) ray_setup(!WIDE, dx, dy, dz);
    const float rdn2 = __builtin_amdgcn_rcpf((dx * dx + dy * dy) + dz * dz);      // (pair_eval_iso: md * rcp(qxx + qyy + qzz))
    // (24-bit multiplies of 32-bit offsets: a plain `row * stride` index becomes a quarter-rate 64-bit multiply-add)
    float *const mylen = Llen + col;
    uint16_t *const mypos16 = reinterpret_cast<uint16_t *>(Lpos_raw) + col;
    uint32_t *const mypos32 = reinterpret_cast<uint32_t *>(Lpos_raw) + col;
    auto put = [&](const int row, const float len, const unsigned p) {
      mylen[__umul24((unsigned)row, (unsigned)kS2TP)] = len;
      if (WIDE) mypos32[__umul24((unsigned)row, (unsigned)(kS2TQ / 2))] = p; else mypos16[__umul24((unsigned)row, (unsigned)kS2TQ)] = (uint16_t)p;
    };
    auto len_at = [&](const int row) -> float { return mylen[__umul24((unsigned)row, (unsigned)kS2TP)]; };
    auto pos_at = [&](const int row) -> unsigned {
      return WIDE ? mypos32[__umul24((unsigned)row, (unsigned)(kS2TQ / 2))] : (unsigned)mypos16[__umul24((unsigned)row, (unsigned)kS2TQ)];
    };

    int cnt = 0;
    float worstf = valid ? not_full : __uint_as_float(0x7fc00000u);      // (NaN: a ray outside the image takes nothing)
    float tailf = -INFINITY;
    bool wdone = false;
    [[maybe_unused]] unsigned qdone = 0u;      // (QUADS) bit q: quadrant q's stream is over (its exit test fired)

    // a candidate that passed `act < thr && len <= worst` but is not a plain append: exact (len, id) insertion
    auto slow_insert = [&](const float len, const unsigned p) {
      const bool full = (cnt == K);
      int my_id = -1;
      auto key_le = [&](const float pl, const int row) {      // list entry (pl, id(row)) <= (len, my id)
        if (pl != len) return pl < len;
        if (my_id < 0) my_id = id_of(p);
        return id_of(pos_at(row)) <= my_id;
      };
      if (full && len == worstf && key_le(worstf, K - 1)) return;      // not below the K-th entry after all
      int pos = full ? K - 1 : cnt;
      float new_tail = len;      // (full lists) what ends up in row K - 1
      bool first = true;
      while (pos > 0) {      // (four rows per trip with their eight LDS reads in flight together: 0.66 instead of 0.5 us per
        const float pl = len_at(pos - 1);      //  insertion -- the walks are short, the cost is this block's fixed overhead)
        if (key_le(pl, pos - 1)) break;
        if (first) new_tail = pl;
        put(pos, pl, pos_at(pos - 1));
        first = false;
        --pos;
      }
      put(pos, len, p);
#ifdef VOGE_SWEEP_SLOW
      dbg_moved = (full ? K - 1 : cnt) - pos;
#endif
      if (full) {
        tailf = new_tail;
        worstf = new_tail;
      } else {
        if (pos == cnt) tailf = len;      // (a tie with the tail and the larger id: still the last entry)
        if (++cnt == K) worstf = tailf;
      }
    };
    // The insertion as it nearly always is (tools/sweep_stats.py SLOW=1: 14 lanes per event, the longest walk among them one
    // row on average, never four): the new entry lands zero, one or two rows below the top, every comparison that places it
    // strict.  Straight-line: five LDS reads in flight together, at most three row writes.  Anything else -- a len tie (the
    // id rule), a longer walk -- returns false and takes slow_insert.  (slow_insert alone cost 0.5 us per event in fixed
    // overhead: 8-14 events and 5-7 of the 27 us of a heavy tile's candidate loop at cfg3.)
    auto lean_insert = [&](const float len, const unsigned p, const bool slow, const int Kv) -> bool {
      // (branch-free: every lane of the wave runs it, `slow` selects who really writes -- the others hit the spare row)
      const bool full = (cnt == K);
      const int top = full ? Kv - 1 : cnt;      // the row the entry takes if nothing moves (a full list drops its row K - 1)
      const int r1 = max(top - 1, 0), r2 = max(top - 2, 0), r3 = max(top - 3, 0);
      float l1 = len_at(r1), l2 = len_at(r2), l3 = len_at(r3);
      const unsigned h1 = pos_at(r1), h2 = pos_at(r2);
      l1 = (top >= 1) ? l1 : -INFINITY; l2 = (top >= 2) ? l2 : -INFINITY; l3 = (top >= 3) ? l3 : -INFINITY;
      const bool g1 = l1 > len, g2 = g1 & (l2 > len);
      const float below = g2 ? l3 : (g1 ? l2 : l1);      // the entry the new one comes to rest on
      const bool ok = slow & (below < len) & !(full & (len == worstf));
      const int n = (g1 ? 1 : 0) + (g2 ? 1 : 0);      // rows that move up
      put((ok & g1) ? top : Kv, l1, h1);
      put((ok & g2) ? top - 1 : Kv, l2, h2);
      put(ok ? top - n : Kv, len, p);
      tailf = ok ? (g1 ? l1 : len) : tailf;              // (a list that is not full: n >= 1 and l1 is its tail already)
      cnt += (ok & !full) ? 1 : 0;
      worstf = (ok & (cnt == K)) ? tailf : worstf;      // (full before, or filled by this entry)
      return slow & !ok;
    };
    int Kv = K;
    asm volatile("" : "+v"(Kv));      // (K in a VGPR for good: the compiler re-materialised it from its SGPR in front of every select)
    // (GEN) The insertion where it is NOT nearly always an append or one row deep: full 3x3 forms arrive ordered by their centres'
    // depth, their lens along a ray are not -- a third of the candidates of a tile land out of order, four rows deep on average
    // (round 3's counters).  Four rows per round trip: the four entries below the gap are read together, the prefix of them
    // behind the new entry moves up (the list is sorted: it IS a prefix), and only a lane that moved all four goes round again.
    // Exact (len, id) order: an exact len tie is resolved by the ids in a uniform, rare branch.  (lean_insert / slow_insert
    // cost this kernel + 34 % on long thin ellipsoids: 255 against round 3's 187 us.)
    auto deep_insert = [&](const float len, const unsigned p, const bool slow) {
      const bool full = (cnt == K);
      bool go = slow;
      int my_id = -1;
      if (__builtin_expect(__ballot(go & full & (len == worstf)) != 0ull, 0)) {      // a tie with the K-th entry: the id rule says who stays
        if (go & full & (len == worstf)) {
          my_id = id_of(p);
          if (!(my_id < id_of(pos_at(K - 1)))) go = false;
        }
      }
      int pos = full ? K - 1 : cnt;      // the row the entry takes if nothing moves (a full list drops its row K - 1)
      const int pos0 = pos;
      float top = len;                   // (full lists) what ends up in row K - 1
      bool walking = go, first = true;
      while (__ballot(walking) != 0ull) {
        const int r1 = max(pos - 1, 0), r2 = max(pos - 2, 0), r3 = max(pos - 3, 0), r4 = max(pos - 4, 0);
        const float l1 = len_at(r1), l2 = len_at(r2), l3 = len_at(r3), l4 = len_at(r4);
        const unsigned h1 = pos_at(r1), h2 = pos_at(r2), h3 = pos_at(r3), h4 = pos_at(r4);
        bool g1 = l1 > len, g2 = l2 > len, g3 = l3 > len, g4 = l4 > len;
        if (__builtin_expect(__ballot(walking & ((l1 == len) | (l2 == len) | (l3 == len) | (l4 == len))) != 0ull, 0)) {
          if (walking & ((l1 == len) | (l2 == len) | (l3 == len) | (l4 == len))) {      // equal lens: the larger id is behind
            if (my_id < 0) my_id = id_of(p);
            g1 = g1 | ((l1 == len) && id_of(h1) > my_id); g2 = g2 | ((l2 == len) && id_of(h2) > my_id);
            g3 = g3 | ((l3 == len) && id_of(h3) > my_id); g4 = g4 | ((l4 == len) && id_of(h4) > my_id);
          }
        }
        const bool c1 = walking & (pos >= 1) & g1, c2 = c1 & (pos >= 2) & g2, c3 = c2 & (pos >= 3) & g3, c4 = c3 & (pos >= 4) & g4;
        if (first) top = c1 ? l1 : len;
        first = false;
        put(c1 ? pos : Kv, l1, h1);              // (a lane that moves nothing writes the spare row)
        put(c2 ? pos - 1 : Kv, l2, h2);
        put(c3 ? pos - 2 : Kv, l3, h3);
        put(c4 ? pos - 3 : Kv, l4, h4);
        pos -= (c1 ? 1 : 0) + (c2 ? 1 : 0) + (c3 ? 1 : 0) + (c4 ? 1 : 0);
        walking = c4;
      }
      put(go ? pos : Kv, len, p);
      if (go) {
        if (full) {
          tailf = top; worstf = top;
        } else {
          if (pos == pos0) tailf = len;      // (a tie with the tail and the larger id: still the last entry)
          if (++cnt == K) worstf = tailf;
        }
      }
    };

    auto commit = [&](const float len, const float act, const unsigned p) {
      const bool take = (act < thr_act) & (len <= worstf);
      const bool app = take & (len > tailf);      // (a full list has tail == worst: it never appends)
      // Where a candidate that is NOT appended leaves its (len, handle).  Nobody ever reads a row at or behind a list's end
      // (lean / slow / deep_insert read rows < cnt, the epilogue masks slots >= cnt), so the 16-bit form stores every candidate
      // at row cnt -- an append is then just the count moving past it, and a full list (cnt == K) hits the spare row K.  The
      // WIDE form must not: its lanes 32..63 evaluate along with cnt = 0 and share their COLUMN with lanes 0..31 (col = lane &
      // 31), so "row cnt" of such a lane is row 0 of another ray's live list; there the non-appends go to the spare row K, the
      // only row two lanes of a column may both write.  (Round 5 tried put(cnt) for both forms, saw cfg4 -- the wide form --
      // fail, and reverted without finding this; tests/test_gpu_configs.py::test_streams_longer_than_16_bit_handles and ::test_rebuilt_sweep_equals_round_3_sweep_bit_for_bit (its last case) run the wide form.)
#if VOGE_S2_PUT_AT_CNT
      put(WIDE ? (app ? cnt : Kv) : cnt, len, p);
#else
      put(app ? cnt : Kv, len, p);
#endif
      cnt += app ? 1 : 0;
      tailf = app ? len : tailf;
      worstf = (app & (cnt == K)) ? len : worstf;      // the append that fills the list: its len is the admission bound now
      // Everything that is not an append (4 % of the candidates of a tile, any lane) sits OUT of line behind a uniform,
      // predicted-not-taken branch: a divergent `if` is a taken skip-branch for every candidate nobody needs it for.
      const bool slow = take & !app;
      if (__builtin_expect(__ballot(slow) != 0ull, 0)) {
#ifdef VOGE_SWEEP_SLOW      // (-DVOGE_SWEEP_TIMES -DVOGE_SWEEP_SLOW: stamp 7 = time in the insertions << 32 | their number)
        const unsigned long long tss = wall_clock64();
#endif
#ifdef VOGE_SWEEP_SLOW
        dbg_moved = 0;
#endif
        if (GEN == 1 && !(VOGE_S2_GEN_LEAN)) {
          deep_insert(len, p, slow);
        } else {
          const bool hard = lean_insert(len, p, slow, Kv);
          if (__builtin_expect(__ballot(hard) != 0ull, 0)) {
            if (hard) slow_insert(len, p);
          }
        }
#ifdef VOGE_SWEEP_SLOW
        ts_slow += wall_clock64() - tss; ++st_slow;
        {      // (per event: lanes inserting, the longest walk among them)
          const int nl = __popcll(__ballot(slow));
          int mx = slow ? dbg_moved : 0;
          for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
          st_lanes += nl; st_moved += mx; st_far += (mx >= 4) ? 1 : 0;
        }
#endif
      }
    };

    auto load_id = [&](const int g) { return (g < src_n) ? (binned ? src_id[g] : g) : -1; };
    auto load_lb = [&](const int g) { return (binned && g < src_n) ? src_lb[g] : -INFINITY; };

#ifdef VOGE_SWEEP_TIMES
    ts1 = wall_clock64();
#endif
    // two-deep software pipeline: ids two chunks ahead, records one chunk ahead (the first of each: requested at the top of run())
    for (int base = 0; base < src_n && !wdone; base += 64) {
      const int id = id0;
      const float lbv = lb0;
      const float4 mrec = m0, crec = c0;
      const float4 e0 = ea0, e1 = ea1, e2 = ea2;
      id0 = id1; lb0 = lb1;
      m0 = load_ms(id0);
      c0 = pref ? cull_none : load_cull(id0);
      if (GEN) { ea0 = load_ev(id0, 0); ea1 = load_ev(id0, 1); ea2 = load_ev(id0, 2); }
      id1 = load_id(base + 128 + lane);
      lb1 = load_lb(base + 128 + lane);
#ifdef VOGE_SWEEP_TIMES
      const unsigned long long tsa = wall_clock64();
#endif
#if VOGE_S2_READLANE
      if (pref) {
        // ---- the tile's own list: nothing is staged.  Lane i holds entry base + i's record in registers; the wave reads
        // candidate s out of lane s with v_readlane (the values land in SGPRs and feed the evaluation as scalar operands).
        // The old form broadcast every record to all 64 lanes through LDS: a ds_read_b128 per candidate, 1 KB through the
        // LDS crossbar each -- with nine waves per CU the LDS pipe was as busy as the VALUs (tools/valu_bench.hip prices a
        // broadcast ds_read_b64 at ~6 LDS clocks). ----
        const int nb = min(64, src_n - base);
        const int hnd = h_is_id ? id : base + lane;
        int n = nb;
#ifdef VOGE_SWEEP_TIMES
        const unsigned long long tsb = wall_clock64();
        ts_fill += tsb - tsa;
#endif
        for (int s0 = 0; s0 < n; s0 += 4) {
          if ((s0 & (kExitGroup - 1)) == 0 && binned && unit_rays && __all(!valid || cnt == K)) {
            const float wmax = wave_max(valid ? worstf : -INFINITY);
            const unsigned long long ex = __ballot(lane < nb && lbv > wmax);
            if (ex) {
              n = __builtin_ctzll(ex);
              wdone = true;
              if (n <= s0) break;
            }
          }
          float len[4], act[4];
          int pv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {      // (lanes behind the list hold the never-hit record)
            const int l = s0 + q;
            const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mrec.x), l));
            const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mrec.y), l));
            const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mrec.z), l));
            const float sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mrec.w), l));
            pv[q] = __builtin_amdgcn_readlane(hnd, l);
            const float md = fmaf(sz, dz, fmaf(sy, dy, sx * dx));      // pair_eval_iso's operations, bit for bit
            const float t = md * rdn2;
            const float vx = fmaf(-t, dx, sx), vy = fmaf(-t, dy, sy), vz = fmaf(-t, dz, sz);
            len[q] = t;
            act[q] = sa * fmaf(vz, vz, fmaf(vy, vy, vx * vx));
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(len[q]), "+v"(act[q]));
#pragma unroll
          for (int q = 0; q < 4; ++q) commit(len[q], act[q], (unsigned)pv[q]);
        }
#ifdef VOGE_SWEEP_TIMES
        ts_cons += wall_clock64() - tsb;
        st_eval += min(n, nb);
#endif
        continue;
      }
#endif
#if VOGE_S2_QUADS
      if constexpr (!GEN && !WIDE) {
        if (quads) {      // (uniform) ---- the tile's own list, one stream per quadrant (Sweep2Quad) ----
          const int nb = min(64, src_n - base);
          __syncthreads();      // (the previous chunk's readers are done)
          QR[lane] = mrec;      // (behind the list: never-hit records)
          S.lb[lane] = lbv;
          reinterpret_cast<unsigned *>(Q8)[lane] = 0x40404040u;
          if (lane < kS2Pad) reinterpret_cast<unsigned *>(Q8)[64 + lane] = 0x40404040u;
          // (straight-line: the four cones are requested first, the candidate's reach is iso_cull_record's with selects for its
          //  branches, cone_keep's early return is a mask -- as `if`s every quadrant's test sat in exec-mask regions of its own,
          //  each behind its own LDS round trip: 1 us per chunk)
          const float4 c0q = QC.c[0], c1q = QC.c[1], c2q = QC.c[2], c3q = QC.c[3];
          const float4 snq = *reinterpret_cast<const float4 *>(&QC.sn[0]), okq = *reinterpret_cast<const float4 *>(&QC.ok[0]);
          const float nm2 = fmaf(mrec.z, mrec.z, fmaf(mrec.y, mrec.y, mrec.x * mrec.x));
          const float nm = __builtin_amdgcn_sqrtf(nm2);
          float reach = __builtin_amdgcn_sqrtf(fmaxf(thr_act, 0.0f) * __builtin_amdgcn_rcpf(mrec.w * (1.0f - 4e-6f))) * (1.0f + 2e-5f) + 2e-5f * nm + 1e-30f;
          reach = (mrec.w > 0.0f && mrec.w < 3e38f && reach >= 0.0f) ? reach : INFINITY;      // (iso_cull_record, trace_bin.h: conservative)
          int cq[4];
          unsigned kbits = 0u;      // bit q: this lane's staged entry is in quadrant q's list (the exit test re-counts with it)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 c4 = q == 0 ? c0q : (q == 1 ? c1q : (q == 2 ? c2q : c3q));
            const float snv = q == 0 ? snq.x : (q == 1 ? snq.y : (q == 2 ? snq.z : snq.w));
            const float okv = q == 0 ? okq.x : (q == 1 ? okq.y : (q == 2 ? okq.z : okq.w));
            const float pp_ = fmaf(mrec.z, c4.z, fmaf(mrec.y, c4.y, mrec.x * c4.x));
            const float rx = fmaf(-pp_, c4.x, mrec.x), ry = fmaf(-pp_, c4.y, mrec.y), rz = fmaf(-pp_, c4.z, mrec.z);
            const float qq = __builtin_amdgcn_sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx))) * (1.0f - 1e-6f);      // (cone_keep with the hardware root, rounded down)
            const float gap = fmaf(qq, c4.w, -fabsf(pp_) * snv);
            const bool keep = (lane < nb) & !((qdone >> q) & 1u) & ((okv == 0.f) | !(gap > reach));
            const unsigned long long m = __ballot(keep);
            cq[q] = __popcll(m);
            kbits |= keep ? (1u << q) : 0u;
            Q8[q * kS2QL + (keep ? __popcll(m & ((1ull << lane) - 1ull)) : kS2QL - 1)] = (unsigned char)(keep ? lane : 64);
          }
          __syncthreads();
#ifdef VOGE_SWEEP_TIMES
          const unsigned long long tsb = wall_clock64();
          ts_fill += tsb - tsa;
          int st_trips = 0;
#endif
          const float lbl = S.lb[lane];
          const int myq = ((lane >> 2) & 1) | ((lane >> 4) & 2);
          const unsigned char *const myQ = Q8 + myq * kS2QL;
          int trips = (max(max(cq[0], cq[1]), max(cq[2], cq[3])) + 3) >> 2;
          unsigned Icur = *reinterpret_cast<const unsigned *>(myQ), Inx = *reinterpret_cast<const unsigned *>(myQ + 4);
          float4 Rn[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) Rn[k] = QR[(Icur >> (8 * k)) & 0xffu];
          for (int t = 0; t < trips; ++t) {
            if ((t & 3) == 0 && binned && unit_rays) {
              // the exit test, per quadrant: all of its rays hold K hits and the bound of a staged entry lies above every K-th len
              const unsigned long long fullm = __ballot(!valid || cnt == K);
              unsigned ready = 0u;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const unsigned long long qm = (0x0F0F0F0Full << (4 * (q & 1))) << (32 * (q >> 1));
                if (!((qdone >> q) & 1u) && (fullm & qm) == qm) ready |= 1u << q;
              }
              if (ready != 0u) {      // (uniform)
                float v = valid ? worstf : -INFINITY;
                v = fmaxf(v, VOGE_DPP(v, 0xB1)); v = fmaxf(v, VOGE_DPP(v, 0x4E)); v = fmaxf(v, VOGE_DPP(v, 0x128));
                const float wq[4] = {fmaxf(VOGE_LANE(v, 0), VOGE_LANE(v, 16)), fmaxf(VOGE_LANE(v, 4), VOGE_LANE(v, 20)),
                                     fmaxf(VOGE_LANE(v, 32), VOGE_LANE(v, 48)), fmaxf(VOGE_LANE(v, 36), VOGE_LANE(v, 52))};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if ((ready >> q) & 1u) {
                    const unsigned long long ex = __ballot(lane < nb && lbl > wq[q]);
                    if (ex) {
                      cq[q] = __popcll(__ballot((kbits >> q) & 1u) & ((1ull << __builtin_ctzll(ex)) - 1ull));
                      qdone |= 1u << q;
                    }
                  }
                }
                trips = (max(max(cq[0], cq[1]), max(cq[2], cq[3])) + 3) >> 2;
                wdone = qdone == 15u;
                if (t >= trips) break;
              }
            }
            const unsigned I = Icur;
            float len[4], act[4];
            {
              // pair_eval_iso's operations, bit for bit, two candidates per packed instruction (as the 64-ray form below)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const float4 r0 = Rn[2 * h], r1 = Rn[2 * h + 1];
                const v2f x2 = (v2f){r0.x, r1.x}, y2 = (v2f){r0.y, r1.y}, z2 = (v2f){r0.z, r1.z}, a2 = (v2f){r0.w, r1.w};
                const v2f md = pk_fma(z2, splat(dz), pk_fma(y2, splat(dy), x2 * splat(dx)));
                const v2f tt = md * splat(rdn2);
                const v2f vx = pk_fma(-tt, splat(dx), x2), vy = pk_fma(-tt, splat(dy), y2), vz = pk_fma(-tt, splat(dz), z2);
                const v2f a = a2 * pk_fma(vz, vz, pk_fma(vy, vy, vx * vx));
                len[2 * h] = tt.x; len[2 * h + 1] = tt.y;
                act[2 * h] = a.x; act[2 * h + 1] = a.y;
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(len[q]), "+v"(act[q]));
            Icur = Inx;
            Inx = *reinterpret_cast<const unsigned *>(myQ + min(4 * (t + 2), 64));
#pragma unroll
            for (int k = 0; k < 4; ++k) Rn[k] = QR[(Icur >> (8 * k)) & 0xffu];
#pragma unroll
            for (int k = 0; k < 4; ++k) commit(len[k], act[k], (unsigned)base + ((I >> (8 * k)) & 0xffu));
#ifdef VOGE_SWEEP_TIMES
            ++st_trips;
#endif
          }
#ifdef VOGE_SWEEP_TIMES
          ts_cons += wall_clock64() - tsb;
          st_eval += 4 * st_trips;
#endif
          continue;
        }
      }
#endif
      // ---- stage ----
      int nbuf;
      __syncthreads();      // (the previous chunk's readers are done)
      // (GEN) what a staged candidate leaves in the eleven further arrays; an isotropic one (mrec.w is its a, not NaN) only the
      // marker.  A general candidate's s00 takes a's place.
      const bool cgen = GEN && !(mrec.w == mrec.w);
      auto stage_gen = [&](const int sl) {
        const float ev[11] = {cgen ? e0.y : __uint_as_float(0x7fc00000u), e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
#pragma unroll
        for (int r = 0; r < 11; ++r)
          if (ghas(r)) grow(r)[sl] = ev[r];
      };
      const float a_st = cgen ? e0.x : mrec.w;
      if (pref) {
        S.x[lane] = mrec.x; S.y[lane] = mrec.y; S.z[lane] = mrec.z; S.a[lane] = a_st;      // (behind the list: never-hit records)
        S.lb[lane] = lbv;
        S.pos[lane] = h_is_id ? id : base + lane;
        if (GEN) stage_gen(lane);
        nbuf = min(64, src_n - base);
      } else {
        const bool keep = cone_keep(crec, wcone);      // (padding: reach -1, never kept)
        const unsigned long long m = __ballot(keep);
        nbuf = __popcll(m);
        if (keep) {
          const int sl = __popcll(m & ((1ull << lane) - 1ull));
          S.x[sl] = mrec.x; S.y[sl] = mrec.y; S.z[sl] = mrec.z; S.a[sl] = a_st;
          S.lb[sl] = lbv;
          S.pos[sl] = h_is_id ? id : base + lane;
          if (GEN) stage_gen(sl);
        }
        if (lane < kS2Pad) { S.x[nbuf + lane] = 0.f; S.y[nbuf + lane] = 0.f; S.z[nbuf + lane] = 0.f; S.a[nbuf + lane] = INFINITY; }
        if (GEN && lane < kS2Pad) G.e[0][nbuf + lane] = __uint_as_float(0x7fc00000u);
      }
      __syncthreads();
      // (GEN) bit s: staged entry s is a general candidate
      unsigned long long gmask = 0ull;
      // (GEN, round 6) ... and a DIAGONAL one: s01 = s02 = s12 = 0 and k = 0 (the user's per-axis sigmas, Aggregation.py:169-172).
      // A trip of four such candidates skips the chain's ten packed operations on those coefficients -- pk_fma(0, q, x) and
      // pk_fma(t, 0, x) return x for finite operands: the same bits -- and a chunk of nothing else reads five of the eleven
      // coefficient arrays per trip instead of all of them.
      unsigned long long dmask = 0ull;
      if (GEN) {
        const float mk = G.e[0][lane];
        gmask = __ballot(lane < nbuf && mk == mk);
        tile_gen = tile_gen || gmask != 0ull;
        if (GEN == 2) {
          dmask = gmask;      // (a launch of per-axis forms)
        } else if (gmask != 0ull) {      // (uniform)
          const bool dg = G.e[2][lane] == 0.0f && G.e[3][lane] == 0.0f && G.e[4][lane] == 0.0f && G.e[8][lane] == 0.0f &&
                          G.e[9][lane] == 0.0f && G.e[10][lane] == 0.0f;
          dmask = __ballot(lane < nbuf && mk == mk && dg);
        }
      }
      const bool chunk_diag = GEN && gmask != 0ull && dmask == gmask;      // (uniform) every general form of the chunk is diagonal
#ifdef VOGE_SWEEP_TIMES
      const unsigned long long tsb = wall_clock64();
      ts_fill += tsb - tsa;
#endif
      // ---- consume: four candidates per trip, evaluated as one straight-line block, committed in order.  The exit test
      // runs in front of every group of kExitGroup candidates: the bounds are monotone along the stream, so the first
      // staged entry whose bound lies above every ray's K-th len ends the tile (unit rays only: the bound is a depth along
      // a unit direction).  (Round 3's kernel tested once per 64-candidate chunk: a tile whose last ray filled early in a
      // chunk evaluated the rest of it for nothing -- tools/sweep_hwmap.py showed counts of exactly 128 / 192.)  Entries
      // behind n inside the last trip are either real candidates the test just proved too deep (len > worst: rejected by
      // the same compare) or never-hit padding. ----
      const float lbl = S.lb[lane];      // (the bound of staged entry `lane`)
      int n = nbuf;
#if VOGE_S2_PREFETCH
      // The records of trip s0 + 4 are requested BEFORE trip s0's commits (round 5): the loop used to issue its five
      // ds_read_b128 and wait for them at once -- behind the previous trip's eight ds_write still in the wave's LDS queue --
      // so every trip of four candidates began with an exposed LDS round trip (SQ_WAIT_ANY: 42 % of the kernel's wave
      // cycles at 2.2 waves per SIMD, profiles/r5_pmc_sq_counters.txt).  (The staged arrays carry kS2Pad entries of
      // padding: the request behind the last trip reads those.)
      float4 Xn = *reinterpret_cast<const float4 *>(&S.x[0]), Yn = *reinterpret_cast<const float4 *>(&S.y[0]),
             Zn = *reinterpret_cast<const float4 *>(&S.z[0]), An = *reinterpret_cast<const float4 *>(&S.a[0]);
      int4 Pn = *reinterpret_cast<const int4 *>(&S.pos[0]);
#endif
      // (GEN: the eleven further arrays one trip ahead as well -- read at their use they stalled every trip for an LDS round
      //  trip at two waves per SIMD: SQ_WAIT_ANY + 28 % against round 3's general sweep, profiles/r5_pmc_sq_counters_gen.txt)
      float4 Evn[GEN ? 11 : 1];
      if (GEN) {
#pragma unroll
        for (int r = 0; r < 11; ++r) Evn[GEN ? r : 0] = ghas(r) ? *reinterpret_cast<const float4 *>(grow(r)) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (GEN != 2 && chunk_diag) {      // (the six arrays such a chunk never reads again ARE zero: a mixed trip's pair_eval_gen takes them from here)
#pragma unroll
          for (int r = 0; r < 11; ++r)
            if (!(r < 2 || (r >= 5 && r < 8))) Evn[GEN ? r : 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      for (int s0 = 0; s0 < n; s0 += 4) {
        if ((s0 & (kExitGroup - 1)) == 0 && binned && unit_rays && __all(!valid || cnt == K)) {
          const float wmax = wave_max(valid ? worstf : -INFINITY);
          const unsigned long long ex = __ballot(lane < nbuf && lbl > wmax);
          if (ex) {
            n = __builtin_ctzll(ex);
            wdone = true;
            if (n <= s0) break;
          }
        }
#if VOGE_S2_PREFETCH
        const float4 X = Xn, Y = Yn, Z = Zn, A = An;
        const int4 P = Pn;
#else
        const float4 X = *reinterpret_cast<const float4 *>(&S.x[s0]), Y = *reinterpret_cast<const float4 *>(&S.y[s0]),
                     Z = *reinterpret_cast<const float4 *>(&S.z[s0]), A = *reinterpret_cast<const float4 *>(&S.a[s0]);
        const int4 P = *reinterpret_cast<const int4 *>(&S.pos[s0]);      // the entries' handles (id or stream position)
#endif
        const int pv[4] = {P.x, P.y, P.z, P.w};
        float len[4], act[4];
        const unsigned gbits = GEN ? (unsigned)(gmask >> s0) & 15u : 0u;      // (uniform) which of the trip's four are general
        if (GEN && gbits != 0u) {
          // ---- pair_eval_gen's operations, bit for bit (voge_common.h): two candidates per packed instruction when all four
          // of the trip are general, one by one (uniform branches) in a mixed trip ----
          float4 Ev[11];
#pragma unroll
          for (int r = 0; r < 11; ++r) Ev[r] = Evn[GEN ? r : 0];
          const float qxx = dx * dx, qyy = dy * dy, qzz = dz * dz, qxy = dx * dy, qxz = dx * dz, qyz = dy * dz;
          if (gbits == 15u && ((unsigned)(dmask >> s0) & 15u) == 15u) {      // (uniform) four diagonal forms
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              auto half = [&](const float4 v) { return h ? (v2f){v.z, v.w} : (v2f){v.x, v.y}; };
              const v2f x2 = half(X), y2 = half(Y), z2 = half(Z), s00 = half(A), s11 = half(Ev[0]), s22 = half(Ev[1]),
                        bx = half(Ev[5]), by = half(Ev[6]), bz = half(Ev[7]);
              v2f ksk = s00 * splat(qxx);
              ksk = pk_fma(s11, splat(qyy), ksk); ksk = pk_fma(s22, splat(qzz), ksk);
              v2f msk = bx * splat(dx);
              msk = pk_fma(by, splat(dy), msk); msk = pk_fma(bz, splat(dz), msk);
              const v2f t = msk * (v2f){__builtin_amdgcn_rcpf(ksk.x), __builtin_amdgcn_rcpf(ksk.y)} + splat(0.0f);
              const v2f vx = pk_fma(-t, splat(dx), x2), vy = pk_fma(-t, splat(dy), y2), vz = pk_fma(-t, splat(dz), z2);
              v2f a = s00 * (vx * vx);
              a = pk_fma(s11, vy * vy, a); a = pk_fma(s22, vz * vz, a);
              len[2 * h] = t.x; len[2 * h + 1] = t.y;
              act[2 * h] = a.x; act[2 * h + 1] = a.y;
            }
          } else if (GEN != 2 && gbits == 15u) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              auto half = [&](const float4 v) { return h ? (v2f){v.z, v.w} : (v2f){v.x, v.y}; };
              const v2f x2 = half(X), y2 = half(Y), z2 = half(Z), s00 = half(A), s11 = half(Ev[0]), s22 = half(Ev[1]), s01 = half(Ev[2]),
                        s02 = half(Ev[3]), s12 = half(Ev[4]), bx = half(Ev[5]), by = half(Ev[6]), bz = half(Ev[7]), kx = half(Ev[8]),
                        ky = half(Ev[9]), kz = half(Ev[10]);
              v2f ksk = s00 * splat(qxx);
              ksk = pk_fma(s11, splat(qyy), ksk); ksk = pk_fma(s22, splat(qzz), ksk); ksk = pk_fma(s01, splat(qxy), ksk);
              ksk = pk_fma(s02, splat(qxz), ksk); ksk = pk_fma(s12, splat(qyz), ksk);
              v2f msk = bx * splat(dx);
              msk = pk_fma(by, splat(dy), msk); msk = pk_fma(bz, splat(dz), msk);
              const v2f t = msk * (v2f){__builtin_amdgcn_rcpf(ksk.x), __builtin_amdgcn_rcpf(ksk.y)} + splat(0.0f);      // (+0 canonicalises -0, as pair_eval_gen)
              const v2f vx = pk_fma(-t, splat(dx), x2), vy = pk_fma(-t, splat(dy), y2), vz = pk_fma(-t, splat(dz), z2);
              v2f a = s00 * (vx * vx);
              a = pk_fma(s11, vy * vy, a); a = pk_fma(s22, vz * vz, a); a = pk_fma(s01, vx * vy, a);
              a = pk_fma(s02, vx * vz, a); a = pk_fma(s12, vy * vz, a);
              v2f kd = kx * splat(dx);
              kd = pk_fma(ky, splat(dy), kd); kd = pk_fma(kz, splat(dz), kd);
              a = pk_fma(t, kd, a);
              len[2 * h] = t.x; len[2 * h + 1] = t.y;
              act[2 * h] = a.x; act[2 * h + 1] = a.y;
            }
          } else {
            const float mxs[4] = {X.x, X.y, X.z, X.w}, mys[4] = {Y.x, Y.y, Y.z, Y.w}, mzs[4] = {Z.x, Z.y, Z.z, Z.w}, avs[4] = {A.x, A.y, A.z, A.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              auto el = [&](const float4 v) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); };
              if ((gbits >> q) & 1u) {      // (uniform)
                EvalRec e;
                e.s00 = avs[q]; e.s11 = el(Ev[0]); e.s22 = el(Ev[1]); e.s01 = el(Ev[2]); e.s02 = el(Ev[3]); e.s12 = el(Ev[4]);
                e.bx = el(Ev[5]); e.by = el(Ev[6]); e.bz = el(Ev[7]); e.kx = el(Ev[8]); e.ky = el(Ev[9]); e.kz = el(Ev[10]);
                const PairOut o = pair_eval_gen(mxs[q], mys[q], mzs[q], e, dx, dy, dz, qxx, qyy, qzz, qxy, qxz, qyz);
                len[q] = o.len; act[q] = o.act;
              } else {                      // an isotropic candidate (or padding): the scalar kernel's operations
                const float md = fmaf(mzs[q], dz, fmaf(mys[q], dy, mxs[q] * dx));
                const float t = md * rdn2;
                const float vx = fmaf(-t, dx, mxs[q]), vy = fmaf(-t, dy, mys[q]), vz = fmaf(-t, dz, mzs[q]);
                len[q] = t;
                act[q] = avs[q] * fmaf(vz, vz, fmaf(vy, vy, vx * vx));
              }
            }
          }
        } else {
#if VOGE_S2_PACKED
        // pair_eval_iso's operations, bit for bit, two candidates per instruction (v_pk_mul / v_pk_fma_f32: the halves of a
        // ds_read_b128 are aligned register pairs already).  A packed FMA costs 1.7x a plain one on a SIMD that is shared
        // by several waves and the same as one when the wave runs alone (tools/valu_bench.hip) -- the heavy tiles at the
        // end of the launch, which decide its duration.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const v2f x2 = h ? (v2f){X.z, X.w} : (v2f){X.x, X.y}, y2 = h ? (v2f){Y.z, Y.w} : (v2f){Y.x, Y.y},
                    z2 = h ? (v2f){Z.z, Z.w} : (v2f){Z.x, Z.y}, a2 = h ? (v2f){A.z, A.w} : (v2f){A.x, A.y};
          const v2f md = pk_fma(z2, splat(dz), pk_fma(y2, splat(dy), x2 * splat(dx)));
          const v2f t = md * splat(rdn2);      // (no +0: a len of -0 orders, ties and re-evaluates exactly like +0 under float compares)
          const v2f vx = pk_fma(-t, splat(dx), x2), vy = pk_fma(-t, splat(dy), y2), vz = pk_fma(-t, splat(dz), z2);
          const v2f a = a2 * pk_fma(vz, vz, pk_fma(vy, vy, vx * vx));
          len[2 * h] = t.x; len[2 * h + 1] = t.y;
          act[2 * h] = a.x; act[2 * h + 1] = a.y;
        }
#else
        const float mx[4] = {X.x, X.y, X.z, X.w}, my[4] = {Y.x, Y.y, Y.z, Y.w}, mz[4] = {Z.x, Z.y, Z.z, Z.w}, av[4] = {A.x, A.y, A.z, A.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // pair_eval_iso's operations, bit for bit
          const float md = fmaf(mz[q], dz, fmaf(my[q], dy, mx[q] * dx));
          const float t = md * rdn2;      // (no +0: a len of -0 orders, ties and re-evaluates exactly like +0 under float compares)
          const float vx = fmaf(-t, dx, mx[q]), vy = fmaf(-t, dy, my[q]), vz = fmaf(-t, dz, mz[q]);
          len[q] = t;
          act[q] = av[q] * fmaf(vz, vz, fmaf(vy, vy, vx * vx));
        }
#endif
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(len[q]), "+v"(act[q]));      // four interleaved chains, then the commits
#if VOGE_S2_PREFETCH
        Xn = *reinterpret_cast<const float4 *>(&S.x[s0 + 4]); Yn = *reinterpret_cast<const float4 *>(&S.y[s0 + 4]);
        Zn = *reinterpret_cast<const float4 *>(&S.z[s0 + 4]); An = *reinterpret_cast<const float4 *>(&S.a[s0 + 4]);
        Pn = *reinterpret_cast<const int4 *>(&S.pos[s0 + 4]);
#endif
        if (GEN && chunk_diag) {         // (uniform) s11, s22 and b: all a diagonal trip reads
#pragma unroll
          for (int r = 0; r < 11; ++r)
            if (r < 2 || (r >= 5 && r < 8)) Evn[GEN ? r : 0] = *reinterpret_cast<const float4 *>(grow(r) + s0 + 4);
        } else if (GEN == 1 && gmask != 0ull) {      // (uniform: a chunk without a general candidate reads none of them)
#pragma unroll
          for (int r = 0; r < 11; ++r) Evn[GEN ? r : 0] = *reinterpret_cast<const float4 *>(&G.e[r][s0 + 4]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) commit(len[q], act[q], (unsigned)pv[q]);
      }
#ifdef VOGE_SWEEP_TIMES
      ts_cons += wall_clock64() - tsb;
      st_eval += min(n, nbuf);
#endif
    }
#ifdef VOGE_SWEEP_TIMES
    ts2 = wall_clock64();
#endif

    // ---- epilogue: lanes re-mapped to (pixel, 4 slots); ids looked up from the stream, act / dsd re-derived ----
    __syncthreads();
    S.cnt[lane] = valid ? cnt : 0;
    if (out_cnt != nullptr && valid) out_cnt[((size_t)b * H + py) * W + px] = cnt;
    __syncthreads();
    const int row0 = WIDE ? pass * 4 : 0;
    const int th = max(0, min(WIDE ? 4 : 8, H - ty * 8 - row0)), tw = min(8, W - tx * 8);
    const int row_items = tw * K;
    const uint16_t *const P16 = reinterpret_cast<const uint16_t *>(Lpos_raw);
    const uint32_t *const P32 = reinterpret_cast<const uint32_t *>(Lpos_raw);
    auto entry_pos = [&](const int s, const int owner) -> unsigned {
      return WIDE ? P32[s * (kS2TQ / 2) + owner] : (unsigned)P16[s * kS2TQ + owner];
    };
    const bool want_ad = out_act != nullptr;
    const int gofs = b * N;
    if ((K & 3) == 0) {
      // An item = four consecutive slots of one pixel (16-byte stores).  A thread takes kEpiB items per batch -- at K = 40
      // all ten of its items: every len / position comes out of LDS first, then ALL id look-ups of the batch are in flight
      // together (one memory round trip; the 2.5 dependent rounds of a 16-slot batch cost 5.6 us per tile), then the
      // stores; with act / dsd the (mu, a) gathers follow in sub-batches of kEpiR items.
      constexpr int kEpiB = VOGE_S2_EPI_B, kEpiR = VOGE_S2_EPI_R;
      const int ipr = row_items >> 2, nitem = th * ipr;
      const float inv_ipr = 1.0f / (float)max(ipr, 1), invK = 1.0f / (float)K;
      const size_t tile_pix = ((size_t)b * H + ty * 8 + row0) * W + (size_t)tx * 8;      // the pass's first pixel
      int32_t *const t_idx = out_idx + tile_pix * K;
      float *const t_len = out_len + tile_pix * K;
      // Branch-free inside a batch: every load is unconditional (an item beyond the pass re-reads item 0, a slot beyond the
      // pixel's count reads its stale LDS row and a valid dummy record) and selects keep what counts.  The `if (q < count)`
      // form compiled into one exec-mask region per slot and load: hundreds of scalar instructions and branches per batch,
      // 4.7 us of a heavy tile (10-13 us with act / dsd).
      const unsigned h_last = (unsigned)max(src_n - 1, 0);
      const bool wt_ok = (size_t)W * (size_t)K < ((size_t)1 << 26);      // (byte offsets of the pass's eight rows below 2^31)
      // (the descriptor covers the pass's rows INSIDE the image only: on the last tile row of the last batch element a
      //  full eight rows would reach past the end of the allocation, and the hardware's range check would protect nothing)
      const unsigned rs_bytes = wt_ok ? (unsigned)th * (unsigned)W * (unsigned)K * 4u : 0u;
      const __amdgpu_buffer_rsrc_t rs_idx = out_rsrc(t_idx, rs_bytes), rs_len = out_rsrc(t_len, rs_bytes);
      for (int it0 = lane; it0 < nitem; it0 += 64 * kEpiB) {
        int32_t oi[kEpiB][4];
        float ol[kEpiB][4];
        unsigned rel[kEpiB];      // element offset of the item's first slot from the pass's first pixel (8 W K < 2^31)
        int nv[kEpiB];
#pragma unroll
        for (int u = 0; u < kEpiB; ++u) {
          const int it = it0 + u * 64;
          const bool in = it < nitem;
          const int itc = in ? it : 0;
          const int rr = __float2int_rz(((float)itc + 0.5f) * inv_ipr);
          const int j = (itc - rr * ipr) * 4;
          const int x = __float2int_rz(((float)j + 0.5f) * invK);
          const int sl = j - x * K;
          const int owner = rr * 8 + x;
          rel[u] = (unsigned)(rr * W + x) * (unsigned)K + (unsigned)sl;
          nv[u] = in ? max(0, min(4, S.cnt[owner] - sl)) : -1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float l = Llen[(sl + q) * kS2TP + owner];
            const unsigned h = entry_pos(sl + q, owner);      // (the position; the id follows)
            ol[u][q] = (q < nv[u]) ? l : VOGE_SENT_LEN;
            oi[u][q] = (int)h;
          }
        }
        if (h_is_id) {      // (uniform)
#pragma unroll
          for (int u = 0; u < kEpiB; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) oi[u][q] = (q < nv[u]) ? oi[u][q] + gofs : -1;
        } else {
          int idv[kEpiB][4];
#pragma unroll
          for (int u = 0; u < kEpiB; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) idv[u][q] = src_id[min((unsigned)oi[u][q], h_last)];
#pragma unroll
          for (int u = 0; u < kEpiB; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) oi[u][q] = (q < nv[u]) ? idv[u][q] + gofs : -1;
        }
        if (!want_ad) {
#pragma unroll
          for (int u = 0; u < kEpiB; ++u) {
            if (nv[u] < 0) continue;
            if (wt_ok) {      // (uniform) read next by the composite: write-through stores (voge_common.h)
              st16i_wt(rs_idx, rel[u] * 4u, oi[u][0], oi[u][1], oi[u][2], oi[u][3]);
              st16f_wt(rs_len, rel[u] * 4u, ol[u][0], ol[u][1], ol[u][2], ol[u][3]);
            } else {
              st16i<false>(t_idx + rel[u], oi[u][0], oi[u][1], oi[u][2], oi[u][3]);
              st16f<false>(t_len + rel[u], ol[u][0], ol[u][1], ol[u][2], ol[u][3]);
            }
          }
        } else if (GEN && tile_gen) {
          // (GEN) a tile that staged general candidates: centre and full record (four gathers per slot) of one item in flight
          // together; pair_eval's dispatch on the record, so an isotropic entry of such a tile is still exact
          float *const t_act = out_act + tile_pix * K, *const t_dsd = out_dsd + tile_pix * K;
#pragma unroll
          for (int u = 0; u < kEpiB; ++u) {
            const unsigned pr = rel[u] / (unsigned)K;      // pixel offset (rr W + x)
            const float *ry = rays + (tile_pix + pr) * 3;
            const float ex = ry[0], ey = ry[1], ez = ry[2];
            float4 rc[4], g0[4], g1[4], g2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const size_t gi = (size_t)((q < nv[u]) ? oi[u][q] : gofs);
              rc[q] = ms[gi]; g0[q] = evr[gi * 3]; g1[q] = evr[gi * 3 + 1]; g2[q] = evr[gi * 3 + 2];
            }
            float oa[4], od[4];
            const float dn2 = (ex * ex + ey * ey) + ez * ez;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              PairOut o;
              if (rc[q].w == rc[q].w) o = pair_eval_iso_at(rc[q].x, rc[q].y, rc[q].z, rc[q].w, ol[u][q], ex, ey, ez, dn2);
              else o = pair_eval_gen(rc[q].x, rc[q].y, rc[q].z, unpack_eval(g0[q], g1[q], g2[q]), ex, ey, ez, ex * ex, ey * ey, ez * ez,
                                     ex * ey, ex * ez, ey * ez);
              oa[q] = (q < nv[u]) ? o.act : VOGE_SENT_ACT;
              od[q] = (q < nv[u]) ? o.dsd : 0.0f;
            }
            if (nv[u] < 0) continue;
            st16i<(VOGE_NT_STORES & 2) != 0>(t_idx + rel[u], oi[u][0], oi[u][1], oi[u][2], oi[u][3]);
            st16f<(VOGE_NT_STORES & 2) != 0>(t_len + rel[u], ol[u][0], ol[u][1], ol[u][2], ol[u][3]);
            st16f<(VOGE_NT_STORES & 2) != 0>(t_act + rel[u], oa[0], oa[1], oa[2], oa[3]);
            st16f<(VOGE_NT_STORES & 2) != 0>(t_dsd + rel[u], od[0], od[1], od[2], od[3]);
          }
        } else {
          float *const t_act = out_act + tile_pix * K, *const t_dsd = out_dsd + tile_pix * K;
#pragma unroll
          for (int u0 = 0; u0 < kEpiB; u0 += kEpiR) {
            float4 rec[kEpiR][4];
            float ex[kEpiR], ey[kEpiR], ez[kEpiR];
#pragma unroll
            for (int v = 0; v < kEpiR; ++v) {
              const int u = u0 + v;
              if (u >= kEpiB) continue;      // (compile time)
              const unsigned pr = rel[u] / (unsigned)K;      // pixel offset (rr W + x)
              const float *ry = rays + (tile_pix + pr) * 3;
              ex[v] = ry[0]; ey[v] = ry[1]; ez[v] = ry[2];
#pragma unroll
              for (int q = 0; q < 4; ++q) rec[v][q] = ms[(q < nv[u]) ? oi[u][q] : gofs];
            }
#pragma unroll
            for (int v = 0; v < kEpiR; ++v) {
              const int u = u0 + v;
              if (u >= kEpiB) continue;      // (compile time)
              float oa[4], od[4];
              const float dn2 = (ex[v] * ex[v] + ey[v] * ey[v]) + ez[v] * ez[v];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const PairOut o = pair_eval_iso_at(rec[v][q].x, rec[v][q].y, rec[v][q].z, rec[v][q].w, ol[u][q], ex[v], ey[v], ez[v], dn2);
                oa[q] = (q < nv[u]) ? o.act : VOGE_SENT_ACT;
                od[q] = (q < nv[u]) ? o.dsd : 0.0f;
              }
              if (nv[u] < 0) continue;
              st16i<(VOGE_NT_STORES & 2) != 0>(t_idx + rel[u], oi[u][0], oi[u][1], oi[u][2], oi[u][3]);
              st16f<(VOGE_NT_STORES & 2) != 0>(t_len + rel[u], ol[u][0], ol[u][1], ol[u][2], ol[u][3]);
              st16f<(VOGE_NT_STORES & 2) != 0>(t_act + rel[u], oa[0], oa[1], oa[2], oa[3]);
              st16f<(VOGE_NT_STORES & 2) != 0>(t_dsd + rel[u], od[0], od[1], od[2], od[3]);
            }
          }
        }
      }
    } else {
      // K not a multiple of four: one slot per lane and trip over the pass's rows (rows x pixels x slots flattened), kOddU
      // trips in flight; branch-free like the batches above (a trip beyond the pass re-reads item 0, a slot beyond the pixel's
      // count its stale LDS row), 32-bit offsets from the pass's first pixel, reciprocal instead of integer division
      constexpr int kOddU = 8;
      const float inv_ri = 1.0f / (float)max(row_items, 1), invK = 1.0f / (float)K;
      const int nit = th * row_items;
      const size_t tile_pix = ((size_t)b * H + ty * 8 + row0) * W + (size_t)tx * 8;      // the pass's first pixel
      int32_t *const t_idx = out_idx + tile_pix * K;
      float *const t_len = out_len + tile_pix * K;
      const unsigned h_last = (unsigned)max(src_n - 1, 0);
      for (int it0 = lane; it0 < nit; it0 += kOddU * 64) {
        int32_t oi[kOddU];
        float ol[kOddU];
        unsigned rel[kOddU], pr[kOddU];      // slot / pixel offset from the pass's first pixel
        bool in[kOddU], on[kOddU];
#pragma unroll
        for (int u = 0; u < kOddU; ++u) {
          const int it = it0 + u * 64;
          on[u] = it < nit;
          const int itc = on[u] ? it : 0;
          const int rr = __float2int_rz(((float)itc + 0.5f) * inv_ri);
          const int j = itc - rr * row_items;
          const int x = __float2int_rz(((float)j + 0.5f) * invK);
          const int sl = j - x * K;
          const int owner = rr * 8 + x;
          pr[u] = (unsigned)(rr * W + x);
          rel[u] = pr[u] * (unsigned)K + (unsigned)sl;
          in[u] = on[u] && sl < S.cnt[owner];
          ol[u] = Llen[sl * kS2TP + owner];
          oi[u] = (int)entry_pos(sl, owner);
        }
        if (h_is_id) {      // (uniform)
#pragma unroll
          for (int u = 0; u < kOddU; ++u) oi[u] = in[u] ? oi[u] + gofs : -1;
        } else {
          int idv[kOddU];
#pragma unroll
          for (int u = 0; u < kOddU; ++u) idv[u] = src_id[min((unsigned)oi[u], h_last)];
#pragma unroll
          for (int u = 0; u < kOddU; ++u) oi[u] = in[u] ? idv[u] + gofs : -1;
        }
#pragma unroll
        for (int u = 0; u < kOddU; ++u) ol[u] = in[u] ? ol[u] : VOGE_SENT_LEN;
        if (!want_ad) {
#pragma unroll
          for (int u = 0; u < kOddU; ++u)
            if (on[u]) { t_idx[rel[u]] = oi[u]; t_len[rel[u]] = ol[u]; }
        } else {
          float *const t_act = out_act + tile_pix * K, *const t_dsd = out_dsd + tile_pix * K;
          float oa[kOddU], od[kOddU];
#pragma unroll
          for (int u = 0; u < kOddU; ++u) {
            const float *ry = rays + (tile_pix + pr[u]) * 3;
            const float ex = ry[0], ey = ry[1], ez = ry[2];
            const size_t gi = (size_t)(in[u] ? oi[u] : gofs);
            const float4 cc = ms[gi];
            PairOut o;
            if (GEN && tile_gen && !(cc.w == cc.w)) {      // (a general entry: its full record, pair_eval_gen)
              o = pair_eval_gen(cc.x, cc.y, cc.z, unpack_eval(evr[gi * 3], evr[gi * 3 + 1], evr[gi * 3 + 2]), ex, ey, ez, ex * ex, ey * ey,
                                ez * ez, ex * ey, ex * ez, ey * ez);
            } else {
              o = pair_eval_iso_at(cc.x, cc.y, cc.z, cc.w, ol[u], ex, ey, ez, (ex * ex + ey * ey) + ez * ez);
            }
            oa[u] = in[u] ? o.act : VOGE_SENT_ACT; od[u] = in[u] ? o.dsd : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < kOddU; ++u)
            if (on[u]) { t_idx[rel[u]] = oi[u]; t_len[rel[u]] = ol[u]; t_act[rel[u]] = oa[u]; t_dsd[rel[u]] = od[u]; }
        }
      }
    }
  };

  if ((h_is_id ? N : src_n) <= 65536) {      // (every handle fits 16 bits)
    run(std::false_type{}, 0);
  } else {
    run(std::true_type{}, 0);
    run(std::true_type{}, 1);
  }
#ifdef VOGE_SWEEP_TIMES
  if (lane == 0 && b == 0 && bx < 8192) {
    unsigned long long *o = g_sweep_times + 8 * (size_t)bx;
    o[0] = ts0; o[1] = ts1; o[2] = ts_fill; o[3] = ts_cons; o[4] = ts2; o[5] = wall_clock64(); o[6] = ((unsigned long long)blockIdx.x << 32) | st_eval;
#ifdef VOGE_SWEEP_SLOW
    o[7] = (ts_slow << 32) | st_slow;
    o[2] = ((unsigned long long)st_far << 48) | ((unsigned long long)(st_moved & 0xffffu) << 32) | st_lanes;      // (instead of the fill time)
#else
    o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |      // HW_ID
           ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);      // XCC_ID
#endif
  }
#endif
}

}  // namespace voge
