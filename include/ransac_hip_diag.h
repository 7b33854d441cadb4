/*
 * ransac_hip_diag.h -- entry points that exist in the DIAG build only (libransac_hip_diag.so, -DRH_DIAG): audits of the
 * score kernel's binary32 classifier and box tests against the exact test, and a host-side self-check of the octree
 * sampler's search routines.  Test infrastructure: the product library (libransac_hip.so) does not export them, reads
 * no environment variable and knows none of the A/B switches (ransac.jl_amd/csrc/options.cpp lists them).
 */
#ifndef RANSAC_HIP_DIAG_H
#define RANSAC_HIP_DIAG_H

#include "ransac_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The batched score decides most (candidate, point) pairs with a binary32 evaluation of the reference's
 * compatibles* quantities (plane.jl:114-130, sphere.jl:144-172, cylinder.jl:194-221) and keeps the binary64
 * test for the pairs within a rounding margin of a threshold (csrc/score4_device.h).  This runs every
 * candidate of `shapes` against every point of subset 1 and returns the worst binary32 error in units of
 * the margin width: out[2k], out[2k+1] = max |a32 - a64|, |b32 - b64| for kind k (plane, sphere, cylinder, cone;
 * sound below 1/2), out[8+k] = pairs looked at.  Host shapes, synchronous. */
int rh_dbg_cls_audit(rh_cloud *c, const rh_shape *shapes, int32_t b, const rh_params *p, double *out /* [12] */);
/* diagnostics: the DECISIONS of the score kernel's culling box test and binary32 classifier against the exact test, over
 * every (candidate, point) of shapes x subset 1: per kind k (plane, sphere, cylinder, cone) out[10 k + ...] = 0 pairs
 * (candidate, 64-point group), 1 pairs the box test skips, 2 skipped pairs that hold an exact inlier (must be 0), 3 points,
 * 4 classified surely-in, 5 surely-out, 6 surely-in that the exact test rejects (must be 0), 7 surely-out that it accepts
 * (must be 0), 8 exact inliers, 9 candidates whose classifier takes the all-zero point (a disabled point as staged) for an
 * inlier (must be 0).  The soundness the bit-exact counts rest on, as a count (score4_device.h).  out[40 + k]: pairs whose
 * group holds a BAND point (one whose distance half alone is not surely failed: no test on the group's position box can
 * decide such a pair -- the floor of the necessary pair work, bench.py's frac_necessary); out[44 + k]: pairs with an exact inlier;
 * out[48 + k]: pairs whose SUPER-TILE's box (16 groups: the lists of st_cull) rules the candidate out, out[52 + k]: VIOLATION: such
 * a pair with an exact inlier (must be 0). */
int rh_dbg_cls_soundness(rh_cloud *c, const rh_shape *shapes, int32_t b, const rh_params *p, uint64_t *out /* [56] */);
/* The device's octree sampler finds a point's cell and the r-th enabled point of a cell with a cell directory and
 * bracketed 8-ary searches (csrc/fit_shared.h: cell_bounds_code, lower_bound_in, select_in, select_in_many, select_bit)
 * where the host form uses plain binary searches.  Host-only self-check of those routines against the plain ones on a
 * synthetic Morton order of n points (duplicate codes, random enabled bits, every level): *mismatches = the number of
 * disagreements over `queries` random queries.  Needs no GPU. */
int rh_dbg_oct_search_selftest(int64_t n, uint64_t seed, int64_t queries, int64_t *mismatches);

/* Event counters of the culled score kernel's launches on cloud c (csrc/score4.hip, S4_STAT: chunk visits, box-tested
 * candidates, surviving pairs, batches, pairs of the second pass, undecided points, ring drains, ... per kind; blocks and
 * stagings): what tools/isa_account.py multiplies with the static instruction histogram of the kernel's regions.
 * mode 1: switch on + zero, 0: read into out[128], 2: switch off. */
int rh_dbg_s4_stats(rh_cloud *c, int mode, uint64_t *out /* [128] or NULL */);

#ifdef __cplusplus
}
#endif
#endif
