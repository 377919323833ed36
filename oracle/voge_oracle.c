/*
 * voge_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (voge_amd/) never links, imports or calls it and has
 * no CPU fallback.
 *
 * It restates, in plain C, the algorithm of the reference's hot path:
 *   fine ray trace fwd      VoGE/csrc/ray_trace_voge/ray_trace_voge.cu:135-217
 *   fine ray trace bwd      VoGE/csrc/ray_trace_voge/ray_trace_voge.cu:283-332
 *   composite (aggregation) VoGE/Aggregation.py:30-107  (+ analytic backward)
 *   attribute merge         VoGE/Aggregation.py:111-141
 *   background blend        VoGE/Renderer.py:157-171
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - composite / merge / blend (fwd and bwd): PINNED against the imported
 *     reference (tests/golden/make_golden.py ran /root/reference's own
 *     Aggregation.py / Renderer.py in the build container; fixtures committed).
 *   - trace backward chain rule: PINNED by the reference's embedded known-answer
 *     proof (ray_trace_voge.cu:381-448; fixture trace_bwd_known_answer.npz).
 *   - trace forward (forms, threshold, top-K order): the reference ships no test,
 *     fixture or runnable CPU build for it (CUDA only) -> "parity unpinned" for
 *     that row beyond line-by-line restatement + a dense torch cross-check.
 *
 * Two precisions are built from voge_oracle_impl.h:
 *   *_f64 : fp64 evaluation of the reference formulas on the fp32 inputs (truth)
 *   *_f32 : fp32, reference operation order, no FMA contraction (noise-floor demo)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUFFIX f64
#include "voge_oracle_impl.h"
#undef REAL
#undef SUFFIX

#define REAL float
#define SUFFIX f32
#include "voge_oracle_impl.h"
#undef REAL
#undef SUFFIX

int oracle_abi_version(void) { return 1; }
