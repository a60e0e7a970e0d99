// closest_hit.h -- device functions of the path: vector helpers, exact sphere/plane tests, the single-precision
// screens, Fresnel/refraction, ray generation (included by grafthip.hip only; compile with -ffp-contract=off).
#pragma once

#include <type_traits>

#include "trace_types.h"

namespace graft {

// vector.c:57-64: the sum starts from 0.0
__device__ __forceinline__ double dot3(double ax, double ay, double az, double bx, double by, double bz) {
    double acc = 0.0;
    acc += ax * bx;
    acc += ay * by;
    acc += az * bz;
    return acc;
}

// vector.c:74-82: the output is left alone when the length is not positive
__device__ __forceinline__ void unit3(double vx, double vy, double vz, double &ox, double &oy, double &oz) {
    double mag = sqrt(dot3(vx, vy, vz, vx, vy, vz));
    if (mag > 0) {
        ox = vx / mag;
        oy = vy / mag;
        oz = vz / mag;
    }
}

// sqrt(x) for 2^-700 < x < 2^700: the sequence the compiler emits for a double-precision square root (reciprocal square
// root, one coupled Newton step on root and half-reciprocal, two corrections of the root: correctly rounded) without the
// rescaling of arguments below 2^-767 and the special cases 0 and infinity around it, which such an argument never
// takes -- 8 instructions of 22.  Bit for bit the value `sqrt` returns (probed on the device against it).
__device__ __forceinline__ double sqrtInRange(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return g;
}

// vecUnit again, for the hot stages: the three quotients share the divisor, and so does most of the division the
// compiler emits for each (scale, reciprocal, two Newton steps on the divisor alone).  With the squared length between
// 2^-700 and 2^700 and the three components all above 2^-400 nothing is scaled and no special case arises -- neither in
// the square root (sqrtInRange) nor in the quotients --, and the remaining steps per component -- q = x r,
// e = fma(-m, q, x), q' = fma(e, r, q) -- are the division sequence's own: the result is the correctly rounded
// quotient, bit for bit what `/` gives.  Anything else (a zero component among them) takes `sqrt` and `/`.
__device__ __forceinline__ void unit3Shared(double vx, double vy, double vz, double &ox, double &oy, double &oz) {
    const double ss = dot3(vx, vy, vz, vx, vy, vz);
    const double lo = 0x1p-400;
    const double ax = fabs(vx), ay = fabs(vy), az = fabs(vz);
    if (ss > 0x1p-700 && ss < 0x1p700 && fmin(ax, fmin(ay, az)) > lo) {  // (components never exceed the length by more than rounding)
#ifdef GRAFT_PLAIN_SQRT
        const double mag = sqrt(ss);
#else
        const double mag = sqrtInRange(ss);
#endif
        double r = __builtin_amdgcn_rcp(mag);
        r = __builtin_fma(__builtin_fma(-mag, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-mag, r, 1.0), r, r);
        const double qx = vx * r, qy = vy * r, qz = vz * r;
        ox = __builtin_fma(__builtin_fma(-mag, qx, vx), r, qx);
        oy = __builtin_fma(__builtin_fma(-mag, qy, vy), r, qy);
        oz = __builtin_fma(__builtin_fma(-mag, qz, vz), r, qz);
    } else {
        const double mag = sqrt(ss);
        if (mag > 0) {
            ox = vx / mag;
            oy = vy / mag;
            oz = vz / mag;
        }
    }
}

// mainTracing.c:104-114 with the denominator dir.dir hoisted (it only depends on dir)
__device__ __forceinline__ void newPoint(const double start[3], const double dir[3], double den, double dist,
                                         double out[3]) {
    double scalar = sqrt(dist * dist / den);
    if (dist < 0) scalar = -scalar;
    out[0] = start[0] + scalar * dir[0];
    out[1] = start[1] + scalar * dir[1];
    out[2] = start[2] + scalar * dir[2];
}

__device__ __forceinline__ void exactSphere(const double *geom, int slot, int shape, double sx, double sy, double sz,
                                            double dx, double dy, double dz, double &minDist, Hit &h);

// rayIntersectAvoid (mainTracing.c:128-153) over sphereIntersect (shapes.c:47-73) and planeIntersect
// (shapes.c:83-96).  Candidate normals are not formed here: only the winner's is used (mainTracing.c:146).
//
// FUSED: the miss test of the sphere loop is evaluated with fused multiply-adds.  The reference's discriminant
// D_R = fl(fl(b b) - fl(v.v - r r)) may not be contracted (its value decides hits and feeds the hit point), but
// whether it is NEGATIVE can be proven from another rounding of the same expression: with the dot products summed
// by FMAs, T = b b + r r - v.v (1 - k), k = 2^-45 = 256 u (u = 2^-53).  Both evaluations start from the same
// rounded v = s - c; for |d|^2 <= 1.01 each differs from the real value of b b - (v.v - r r) by less than
// 12 u (v.v + r r), and T's own roundings add less than 9 u (v.v + r r).  T < 0 needs v.v > r r, so those errors
// are below 62 u v.v < k v.v: T < 0 implies D_R < 0.  Anything else -- T >= 0 (every origin inside a sphere), NaN,
// infinities, a ray whose direction is longer than that -- goes through the exact test as before.  The bounds are
// relative ones: a job with a sphere whose r r is below 1e-280 (where k v.v could underflow) does not use this
// (TraceParams::fusedMiss).  11 instructions per sphere instead of 16.
template <int AVOID, bool LDSGEOM, bool FUSED = false>
__device__ __forceinline__ void closestHit(const TraceParams &P, double sx, double sy, double sz, double dx,
                                           double dy, double dz, int avoidShape,
                                           const unsigned char *avoidRow, Hit &h) {
    const double *geom = LDSGEOM ? (const double *)ldsGeom : P.geom;
    double minDist = -1;
    h.idx = -1;
    h.slot = 0;
    h.kind = 0;
    h.x = h.y = h.z = 0;
    const double dd = dx * dx + dy * dy + dz * dz;
    const bool fusedOk = FUSED && P.fusedMiss && dd <= 1.01;  // (false for NaN)
    for (int r = 0; r < P.numRuns; ++r) {
        const DevRun run = P.runs[r];
        if (run.kind == GRAFT_SPHERE) {
            // Four spheres per trip (the geometry is followed by never-hit spheres of padding, and what is read past the
            // end of a run is never used).  Per sphere the 16 operations of the discriminant b*b - (v.v - r*r) and
            // nothing else: one integer test on the high words of four discriminants tells whether any of them is
            // something other than a negative finite number (a discriminant is never -0: b*b is not), and only then are
            // the four spheres gone through one by one, in index order.
            const double4v *sp = (const double4v *)(geom + 4 * (size_t)run.slot);
            for (int j = 0; j < run.count; j += 4) {
                int top = !FUSED || fusedOk ? (int)0x80000000 : 0;  // (0: every block goes through the exact tests)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double4v c = sp[j + u];
                    const double vx = sx - c.x, vy = sy - c.y, vz = sz - c.z;
                    if (FUSED) {
                        const double b = __builtin_fma(vz, dz, __builtin_fma(vy, dy, vx * dx));
                        const double vv = __builtin_fma(vz, vz, __builtin_fma(vy, vy, vx * vx));
                        top = max(top, __double2hiint(__builtin_fma(b, b, __builtin_fma(vv, -(1.0 - 0x1p-45), c.w))));
                    } else {
                        // vecDot's leading "0.0 +" only turns a -0 result into +0; it cannot change b*b or v.v
                        const double b = vx * dx + vy * dy + vz * dz;
                        const double vv = vx * vx + vy * vy + vz * vz;
                        top = max(top, __double2hiint(b * b - (vv - c.w)));
                    }
                }
                if (top >= (int)0xFFF00000) {  // not all four negative and finite
#pragma nounroll
                    for (int u = 0; u < 4; ++u) {
                        bool candidate = j + u < run.count;
                        if (AVOID == 1) candidate = candidate && (run.firstShape + j + u != avoidShape);
                        if (AVOID == 2) candidate = candidate && !(avoidRow && avoidRow[run.firstShape + j + u]);
                        if (candidate) exactSphere(geom, run.slot + j + u, run.firstShape + j + u, sx, sy, sz, dx, dy, dz, minDist, h);
                    }
                }
            }
        } else {
            const double *pl = geom + 4 * (size_t)P.numSpheres + 6 * (size_t)run.slot;
            for (int j = 0; j < run.count; ++j) {
                const double px = pl[6 * j], py = pl[6 * j + 1], pz = pl[6 * j + 2];
                const double nx = pl[6 * j + 3], ny = pl[6 * j + 4], nz = pl[6 * j + 5];
                bool skip = false;
                if (AVOID == 1) skip = (run.firstShape + j == avoidShape);
                if (AVOID == 2) skip = avoidRow && avoidRow[run.firstShape + j];
                const double d = dot3(px - sx, py - sy, pz - sz, nx, ny, nz) / dot3(dx, dy, dz, nx, ny, nz);
                if (!skip && !(d < 0)) {
                    const double lx = sx + d * dx, ly = sy + d * dy, lz = sz + d * dz;
                    const double ex = lx - sx, ey = ly - sy, ez = lz - sz;
                    const double dist = sqrt(dot3(ex, ey, ez, ex, ey, ez));
                    if (minDist == -1 || dist < minDist) {
                        minDist = dist;
                        h.idx = run.firstShape + j;
                        h.slot = run.slot + j;
                        h.kind = GRAFT_PLANE;
                        h.x = lx; h.y = ly; h.z = lz;
                    }
                }
            }
        }
    }
}

// One exact sphere test + closest-hit update: sphereIntersect (shapes.c:47-73) and the body of the loop of
// rayIntersectAvoid (mainTracing.c:139-149), for the sphere in geometry slot `slot` (shape index `shape`).
__device__ __forceinline__ void exactSphere(const double *geom, int slot, int shape, double sx, double sy, double sz,
                                            double dx, double dy, double dz, double &minDist, Hit &h) {
    const double4v c4 = *(const double4v *)(geom + 4 * (size_t)slot);
    const double vx = sx - c4.x, vy = sy - c4.y, vz = sz - c4.z;
    const double b = vx * dx + vy * dy + vz * dz;
    const double vv = vx * vx + vy * vy + vz * vz;
    const double disc = b * b - (vv - c4.w);
    if (!(disc < 0)) {
        const double q = sqrt(disc);
        const double bz = b + 0.0;
        const double tFar = -bz + q;
        const double tNear = -bz - q;
        const double t = tNear >= 0 ? tNear : tFar;
        if (!(t < 0)) {
            const double lx = sx + t * dx, ly = sy + t * dy, lz = sz + t * dz;
            const double ex = lx - sx, ey = ly - sy, ez = lz - sz;
            const double dist = sqrt(dot3(ex, ey, ez, ex, ey, ez));
            // "first in index order wins a tie" (mainTracing.c:143-144), written so that it also holds when
            // spheres are visited out of index order (grouped screen); in index order the last clause never fires
            if (minDist == -1 || dist < minDist || (dist == minDist && shape < h.idx)) {
                minDist = dist;
                h.idx = shape;
                h.slot = slot;
                h.kind = GRAFT_SPHERE;
                h.x = lx; h.y = ly; h.z = lz;
            }
        }
    }
}

// Closest-hit bookkeeping of the sphere-only walks (two-level and family screens), which meet spheres out of index
// order.  The reference compares candidates by dist = |loc - s| (mainTracing.c:141-144), a square root per hit; here
// a candidate is kept by its ray parameter t, and dist is only formed when two candidates are too close for t to
// decide.  With loc = s + t d evaluated as the reference does and d of unit length within 1e-2,
// |dist - t |d|| <= 10.2 u t + u |s|_1 (u = 2^-53), so "t_new + slack < t_best - slack" with
// slack(t) = 2e-15 (t + |s|_1) implies dist_new < dist_best, and the mirror image implies the opposite; anything
// else (including NaN and infinities, whose comparisons are false) is decided by the reference's own rule on the
// two distances, ties going to the lower index.  `slackScale` is 2e-15, or +inf for a ray whose direction is not
// of unit length (then every pair is decided by distance).
struct Best {
    double t, dist;  // dist < 0: not formed yet
    int shape;       // -1: no candidate yet
};

__device__ __forceinline__ double hitDistance(double t, double sx, double sy, double sz, double dx, double dy, double dz) {
    const double lx = sx + t * dx, ly = sy + t * dy, lz = sz + t * dz;
    const double ex = lx - sx, ey = ly - sy, ez = lz - sz;
    return sqrt(dot3(ex, ey, ez, ex, ey, ez));
}

// sphereIntersect (shapes.c:47-73) for sphere `shape` + the closest-hit update described above.
// `fresh` (the same for all lanes): no lane has a candidate yet, so a hit is the lane's best without any comparison.
__device__ __forceinline__ void exactSphereLazy(const double4v c4, int shape, double sx, double sy, double sz, double dx,
                                                double dy, double dz, double slackBase, double slackScale, Best &best,
                                                bool fresh = false) {
    const double vx = sx - c4.x, vy = sy - c4.y, vz = sz - c4.z;
    const double b = vx * dx + vy * dy + vz * dz;
    const double vv = vx * vx + vy * vy + vz * vz;
    const double w = vv - c4.w;
    const double bb = b * b;
    const double disc = bb - w;
    // Origin outside the sphere and moving away from its centre: both roots are negative.  disc = fl(b b - w) is
    // at least two units in the last place below fl(b b), so its root is below b and t = -b + q < 0 (shapes.c:62-66).
    // (Few branches on purpose: lanes of a wave disagree on nearly every one of them, and every level of nesting
    // costs copies of the values that live across it.)
    double t = -1;
    if (!(b > 0 && w > bb * 8.9e-16) && !(disc < 0)) {
        const double q = sqrt(disc);
        const double bz = b + 0.0;  // restores the +0 of the reference's dot product
        const double tFar = -bz + q;
        const double tNear = -bz - q;
        t = tNear >= 0 ? tNear : tFar;
    }
    const bool hit = !(t < 0);  // (true for NaN, as in the reference: shapes.c:66 only rejects t < 0)
#ifndef GRAFT_NO_FRESH_ROUND  // (A/B builds)
    if (fresh) {
        best.t = hit ? t : best.t;
        best.shape = hit ? shape : best.shape;
        return;
    }
#endif
    const double slackNew = slackScale * (t + slackBase), slackBest = slackScale * (best.t + slackBase);
    const bool first = best.shape < 0;
    const bool closer = hit && (first || t + slackNew < best.t - slackBest);
    const bool tied = hit && !closer && !(t - slackNew > best.t + slackBest);
    if (tied) {  // rare: too close for the ray parameters to decide; the reference's own comparison of distances
        if (best.dist < 0) best.dist = hitDistance(best.t, sx, sy, sz, dx, dy, dz);
        const double dist = hitDistance(t, sx, sy, sz, dx, dy, dz);
        if (dist < best.dist || (dist == best.dist && shape < best.shape)) {
            best.t = t; best.dist = dist; best.shape = shape;
        }
    }
    best.t = closer ? t : best.t;
    best.dist = closer ? -1.0 : best.dist;
    best.shape = closer ? shape : best.shape;
}

__device__ __forceinline__ void exactSphereLazy(const double *geom, int shape, double sx, double sy, double sz, double dx,
                                                double dy, double dz, double slackBase, double slackScale, Best &best) {
    exactSphereLazy(*(const double4v *)(geom + 4 * (size_t)shape), shape, sx, sy, sz, dx, dy, dz, slackBase, slackScale, best);
}

// the Hit a walk returns for its best candidate: loc = s + t d as sphereIntersect forms it (shapes.c:67-68)
__device__ __forceinline__ void finishBest(const Best &best, double sx, double sy, double sz, double dx, double dy,
                                           double dz, Hit &h) {
    h.idx = best.shape;
    h.slot = best.shape < 0 ? 0 : best.shape;
    h.kind = 0;
    h.x = h.y = h.z = 0;
    if (best.shape >= 0) {
        h.kind = GRAFT_SPHERE;
        h.x = sx + best.t * dx; h.y = sy + best.t * dy; h.z = sz + best.t * dz;
    }
}

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

// Single-precision screening value of one sphere for one ray: an approximation of the discriminant
// b*b - (v.v - r*r) of shapes.c:54 evaluated with fused multiply-adds on operands rounded to float, with the
// sphere's share of the worst-case error already folded into rho (see packScreenTable).  A value below the
// ray's threshold PROVES that the double-precision discriminant of the reference is negative.
// Evaluated for two spheres at once in packed single precision (v_pk_add/mul/fma_f32).
__device__ __forceinline__ float2v screen2(float2v cx, float2v cy, float2v cz, float2v rho, float2v sx, float2v sy,
                                           float2v sz, float2v dx, float2v dy, float2v dz) {
    const float2v vx = sx - cx, vy = sy - cy, vz = sz - cz;
    float2v b = vx * dx;
    b = __builtin_elementwise_fma(vy, dy, b);
    b = __builtin_elementwise_fma(vz, dz, b);
    float2v vv = vx * vx;
    vv = __builtin_elementwise_fma(vy, vy, vv);
    vv = __builtin_elementwise_fma(vz, vz, vv);
    const float2v t = rho - vv;
    return __builtin_elementwise_fma(b, b, t);
}

// screening values of the four spheres of one table block
__device__ __forceinline__ void screenBlock(const float4v c0, const float4v c1, const float4v c2, const float4v c3,
                                            float sxf, float syf, float szf, float dxf, float dyf, float dzf,
                                            float &e0, float &e1, float &e2, float &e3) {
    // block layout: {cx0 cx1 cy0 cy1} {cz0 cz1 rho0 rho1} {cx2 cx3 cy2 cy3} {cz2 cz3 rho2 rho3}
    const float2v S0 = {sxf, sxf}, S1 = {syf, syf}, S2 = {szf, szf};
    const float2v D0 = {dxf, dxf}, D1 = {dyf, dyf}, D2 = {dzf, dzf};
    const float2v ea = screen2(c0.xy, c0.zw, c1.xy, c1.zw, S0, S1, S2, D0, D1, D2);
    const float2v eb = screen2(c2.xy, c2.zw, c3.xy, c3.zw, S0, S1, S2, D0, D1, D2);
    e0 = ea.x; e1 = ea.y; e2 = eb.x; e3 = eb.y;
}

// rayIntersectAvoid with a conservative single-precision screen in front of the exact test.  Every sphere is
// still visited by every ray, in index order; the screen only skips spheres whose exact discriminant is
// provably negative (a miss in shapes.c:55-57), so hit masks and hit points are bit-for-bit unchanged.
// Error budget (u = 2^-24, A = |s|_1 + |c|_1, |d_i| <= 1): rounding s, c, d to float and evaluating in float
// perturbs the discriminant by at most 24 u A^2 + 2.02 u rho <= 48 u |s|_1^2 + 48 u |c|_1^2 + 2.02 u rho; the
// reference's own double rounding adds less than 2^-28 of that.  The ray carries 64 u |s|_1^2 (negThreshold),
// the sphere carries 64 u |c|_1^2 + 2.2 u rho inside rho.  A ray whose coordinates are not finite, huge, or
// whose direction is not inside the unit cube is not screened at all (the caller takes the exact loop), and
// a sphere with such coordinates has rho = +inf and always passes.  With those two rules no screening value
// is ever NaN, so four of them can be merged with max() and tested with one compare.
template <int AVOID, bool LDSGEOM, bool EXACTLDS>
__device__ __forceinline__ void closestHitScreened(const TraceParams &P, double sx, double sy, double sz, double dx,
                                                   double dy, double dz, int avoidShape,
                                                   const unsigned char *avoidRow, float negThreshold, Hit &h) {
    const float4v *tableF = LDSGEOM ? (const float4v *)ldsGeom : (const float4v *)P.geomF;
    const double *geom = EXACTLDS ? (const double *)((const char *)ldsGeom + P.exactLdsOffset) : P.geom;
    double minDist = -1;
    h.idx = -1;
    h.slot = 0;
    h.kind = 0;
    h.x = h.y = h.z = 0;
    const float sxf = (float)sx, syf = (float)sy, szf = (float)sz;
    const float dxf = (float)dx, dyf = (float)dy, dzf = (float)dz;
    auto visit = [&](int slot, int shape) {
        bool skip = false;
        if (AVOID == 1) skip = shape == avoidShape;
        if (AVOID == 2) skip = avoidRow && avoidRow[shape];
        if (!skip) exactSphere(geom, slot, shape, sx, sy, sz, dx, dy, dz, minDist, h);
    };
    for (int r = 0; r < P.numRuns; ++r) {
        const DevRun run = P.runs[r];
        if (run.kind == GRAFT_SPHERE) {
            const float4v *sp = tableF + run.fslot;
            // Four spheres per block; blocks are handled in pairs (A, B) so that each block's registers are
            // refilled from LDS while the other block is being evaluated, without register copies.  Every run's
            // part of the table holds a multiple of 8 entries and the table ends with 8 entries of padding.
            auto block = [&](const float4v c0, const float4v c1, const float4v c2, const float4v c3, int j) {
                float e0, e1, e2, e3;
                screenBlock(c0, c1, c2, c3, sxf, syf, szf, dxf, dyf, dzf, e0, e1, e2, e3);
                const float best = __builtin_fmaxf(__builtin_fmaxf(e0, e1), __builtin_fmaxf(e2, e3));
                if (!(best < negThreshold)) {  // rare: some lane may hit one of the four
                    const int k = run.count - j;  // live spheres from j on (the rest of the run's part is padding)
                    if (k > 0 && !(e0 < negThreshold)) visit(run.slot + j, run.firstShape + j);
                    if (k > 1 && !(e1 < negThreshold)) visit(run.slot + j + 1, run.firstShape + j + 1);
                    if (k > 2 && !(e2 < negThreshold)) visit(run.slot + j + 2, run.firstShape + j + 2);
                    if (k > 3 && !(e3 < negThreshold)) visit(run.slot + j + 3, run.firstShape + j + 3);
                }
            };
            float4v a0 = sp[0], a1 = sp[1], a2 = sp[2], a3 = sp[3];
            for (int j = 0; j < run.fcount; j += 8) {
                const float4v b0 = sp[j + 4], b1 = sp[j + 5], b2 = sp[j + 6], b3 = sp[j + 7];
                block(a0, a1, a2, a3, j);
                a0 = sp[j + 8]; a1 = sp[j + 9]; a2 = sp[j + 10]; a3 = sp[j + 11];
                block(b0, b1, b2, b3, j + 4);
            }
        } else {
            const double *pl = geom + 4 * (size_t)P.numSpheres + 6 * (size_t)run.slot;
            for (int j = 0; j < run.count; ++j) {
                const double px = pl[6 * j], py = pl[6 * j + 1], pz = pl[6 * j + 2];
                const double nx = pl[6 * j + 3], ny = pl[6 * j + 4], nz = pl[6 * j + 5];
                bool skip = false;
                if (AVOID == 1) skip = (run.firstShape + j == avoidShape);
                if (AVOID == 2) skip = avoidRow && avoidRow[run.firstShape + j];
                const double d = dot3(px - sx, py - sy, pz - sz, nx, ny, nz) / dot3(dx, dy, dz, nx, ny, nz);
                if (!skip && !(d < 0)) {
                    const double lx = sx + d * dx, ly = sy + d * dy, lz = sz + d * dz;
                    const double ex = lx - sx, ey = ly - sy, ez = lz - sz;
                    const double dist = sqrt(dot3(ex, ey, ez, ex, ey, ez));
                    if (minDist == -1 || dist < minDist) {
                        minDist = dist;
                        h.idx = run.firstShape + j;
                        h.slot = run.slot + j;
                        h.kind = GRAFT_PLANE;
                        h.x = lx; h.y = ly; h.z = lz;
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ int lane64() { return (int)(threadIdx.x & 63); }


// Two-level screen for sphere-only scenes: the host cuts the spheres into spatially compact groups of 8 (leaves of
// a k-d tree); each group has a bounding ball (centre g, radius R >= |c_k - g| + r_k for its members, plus a
// margin of 0.1).  A ray whose screening value for the bounding ball is below its threshold misses the ball by
// more than the margin, hence (distance to a line is 1-Lipschitz) misses every member's sphere by more than the
// margin, hence every member's double-precision discriminant in the reference is negative: the whole group is
// skipped.  Members of the other groups go through the per-sphere screen and the exact test as before.  The
// margin covers what separates "misses the ball" from "negative computed discriminant": the direction's
// deviation from unit length (|d.d - 1| < 1e-15 after vecUnit) times b*b, and the reference's own rounding;
// both are below 2e-4 for coordinates under 1e5, which is what packGroupedTable and the caller require.
// Spheres are then met out of index order, so ties are broken by index explicitly (exactSphere); without planes
// and with finite coordinates no distance is NaN, so the winner does not depend on the order of the visits.
template <int AVOID, bool LDSGEOM, bool EXACTLDS>
__device__ __forceinline__ void closestHitGrouped(const TraceParams &P, double sx, double sy, double sz, double dx,
                                                  double dy, double dz, int avoidShape,
                                                  const unsigned char *avoidRow, float negThreshold, Hit &h) {
    const float4v *groups = LDSGEOM ? (const float4v *)ldsGeom : (const float4v *)P.geomF;
    const float4v *members = groups + P.memberBase;
    const double *geom = EXACTLDS ? (const double *)((const char *)ldsGeom + P.exactLdsOffset) : P.geom;
    Best closest = {0, -1, -1};
    const double slackBase = fabs(sx) + fabs(sy) + fabs(sz);
    const double dd = dx * dx + dy * dy + dz * dz;
    const double slackScale = dd > 0.99 && dd < 1.01 ? 2e-15 : INFINITY;
    const float sxf = (float)sx, syf = (float)sy, szf = (float)sz;
    const float dxf = (float)dx, dyf = (float)dy, dzf = (float)dz;
    auto members4 = [&](const float4v *mb, int pos) {
        float f0, f1, f2, f3;
        screenBlock(mb[0], mb[1], mb[2], mb[3], sxf, syf, szf, dxf, dyf, dzf, f0, f1, f2, f3);
        const float best = __builtin_fmaxf(__builtin_fmaxf(f0, f1), __builtin_fmaxf(f2, f3));
        if (!(best < negThreshold)) {
#pragma nounroll
            for (int v = 0; v < 4; ++v) {
                const float fv = v == 0 ? f0 : v == 1 ? f1 : v == 2 ? f2 : f3;
                if (!(fv < negThreshold)) {
                    const int shape = P.memberShape[pos + v];
                    bool skip = shape < 0;
                    if (AVOID == 1) skip = skip || shape == avoidShape;
                    if (AVOID == 2) skip = skip || (avoidRow && shape >= 0 && avoidRow[shape]);
#ifdef GRAFT_STAMPS
                    if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 10, 1ull);
#endif
                    if (!skip) exactSphereLazy(geom, shape, sx, sy, sz, dx, dy, dz, slackBase, slackScale, closest);
                }
            }
        }
    };
    auto groups4 = [&](const float4v c0, const float4v c1, const float4v c2, const float4v c3, int j) {
        float e0, e1, e2, e3;
        screenBlock(c0, c1, c2, c3, sxf, syf, szf, dxf, dyf, dzf, e0, e1, e2, e3);
        const float best = __builtin_fmaxf(__builtin_fmaxf(e0, e1), __builtin_fmaxf(e2, e3));
        if (!(best < negThreshold)) {  // some lane's ray comes near one of these four bounding balls
#pragma nounroll
            for (int u = 0; u < 4; ++u) {
                const float eu = u == 0 ? e0 : u == 1 ? e1 : u == 2 ? e2 : e3;
                if (!(eu < negThreshold)) {
#ifdef GRAFT_STAMPS
                    if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 9, 1ull);
#endif
                    const int m0 = (j + u) * P.groupSize;
#pragma nounroll
                    for (int m = 0; m < P.groupSize; m += 4) members4(members + m0 + m, m0 + m);
                }
            }
        }
    };
    float4v a0 = groups[0], a1 = groups[1], a2 = groups[2], a3 = groups[3];
    for (int j = 0; j < P.groupEntries; j += 8) {
        const float4v b0 = groups[j + 4], b1 = groups[j + 5], b2 = groups[j + 6], b3 = groups[j + 7];
        groups4(a0, a1, a2, a3, j);
        a0 = groups[j + 8]; a1 = groups[j + 9]; a2 = groups[j + 10]; a3 = groups[j + 11];
        groups4(b0, b1, b2, b3, j + 4);
    }
    finishBest(closest, sx, sy, sz, dx, dy, dz, h);
}

// pass values of the four entries of one CONE block for a ray direction: (a . d)^2 - kappa2; the table holds
// -kappa2 so that the subtraction is the addend of a fused multiply-add (see packConeTable)
__device__ __forceinline__ void coneBlock(const float4v c0, const float4v c1, const float4v c2, const float4v c3,
                                          float dxf, float dyf, float dzf, float &p0, float &p1, float &p2, float &p3) {
    const float2v D0 = {dxf, dxf}, D1 = {dyf, dyf}, D2 = {dzf, dzf};
    // block layout: {ax0 ax1 ay0 ay1} {az0 az1 -k0 -k1} {ax2 ax3 ay2 ay3} {az2 az3 -k2 -k3}
    float2v xa = c0.xy * D0;
    xa = __builtin_elementwise_fma(c0.zw, D1, xa);
    xa = __builtin_elementwise_fma(c1.xy, D2, xa);
    const float2v pa = __builtin_elementwise_fma(xa, xa, c1.zw);
    float2v xb = c2.xy * D0;
    xb = __builtin_elementwise_fma(c2.zw, D1, xb);
    xb = __builtin_elementwise_fma(c3.xy, D2, xb);
    const float2v pb = __builtin_elementwise_fma(xb, xb, c3.zw);
    p0 = pa.x; p1 = pa.y; p2 = pb.x; p3 = pb.y;
}

// Closest hit with one two-level table per RAY FAMILY (sphere-only scenes).  Every primary ray of a perspective
// view passes through the camera position and every shadow ray through its light: for those families the host
// builds cone tables (packConeTable) whose test costs 4 packed instructions per two entries and culls by
// direction alone; reflected rays use the ball table (family `ballFamily`, as closestHitGrouped).  The tables are
// interleaved block by block (block b of family f, four entries = 64 bytes, at float4 index 4 * (b * families + f)):
// a lane reads a block through one address with four fixed offsets, and lanes of one wave that follow different
// families read different 64-byte slots.  Spheres are met out of index order; see closestHitGrouped for
// why the winner does not depend on it.
// SUPER (scenes whose member sections do not fit in LDS): only the group sections of all families, the ball
// family's member section and the third-level sections are staged in LDS; cone members are read from the table in
// global memory.  The walk has a third level on top: one entry per 8 consecutive groups (an enclosing cone, or
// ball, of their 64 spheres, built like a group's), and only the groups of passing entries are looked at.  The
// argument for skipping is the same as for a group: the entry encloses every member with the member's own margin.
// (With lanes of one wave following different families the groups to open add up, which is why smaller scenes,
// whose tables fit in LDS whole, keep the two-level walk: measured 34 vs 23 Grays/s on 1024 spheres.)
template <int AVOID, bool EXACTLDS, bool SUPER, bool BALL = true>
__device__ __forceinline__ void closestHitFamilies(const TraceParams &P, double sx, double sy, double sz, double dx,
                                                   double dy, double dz, int avoidShape, const unsigned char *avoidRow,
                                                   int family, float negThreshold, Hit &h) {
    const int F = P.families;
    const float4v *table = (const float4v *)ldsGeom + 4 * family;  // this lane's blocks of the interleaved table
    const double *geom = EXACTLDS ? (const double *)((const char *)ldsGeom + P.exactLdsOffset) : P.geom;
    Best closest = {0, -1, -1};
    const double slackBase = fabs(sx) + fabs(sy) + fabs(sz);
    const double dd = dx * dx + dy * dy + dz * dz;
    const double slackScale = dd > 0.99 && dd < 1.01 ? 2e-15 : INFINITY;
    const float sxf = (float)sx, syf = (float)sy, szf = (float)sz;
    const float dxf = (float)dx, dyf = (float)dy, dzf = (float)dz;
    // One walk over the lane's table; instantiated twice (cone / ball entries) and selected per lane OUTSIDE the
    // loops, so that a wave whose lanes all follow cone families never executes a ball instruction.
    auto walk = [&](auto coneTag) {
        constexpr bool CONE = decltype(coneTag)::value;
        const float limit = CONE ? 0.0f : negThreshold;  // an entry is skipped iff its value is below this
        auto evaluate = [&](const float4v c0, const float4v c1, const float4v c2, const float4v c3, float &e0,
                            float &e1, float &e2, float &e3) {
            if (CONE) coneBlock(c0, c1, c2, c3, dxf, dyf, dzf, e0, e1, e2, e3);
            else screenBlock(c0, c1, c2, c3, sxf, syf, szf, dxf, dyf, dzf, e0, e1, e2, e3);
        };
        auto members4 = [&](auto *mb, int pos) {
            float f0, f1, f2, f3;
            evaluate(mb[0], mb[1], mb[2], mb[3], f0, f1, f2, f3);
            const float best = __builtin_fmaxf(__builtin_fmaxf(f0, f1), __builtin_fmaxf(f2, f3));
            if (!(best < limit)) {
#pragma nounroll
                for (int v = 0; v < 4; ++v) {
                    const float fv = v == 0 ? f0 : v == 1 ? f1 : v == 2 ? f2 : f3;
                    if (!(fv < limit)) {
                        const int shape = P.memberShape[(size_t)(pos + v) * F + family];
                        bool skip = shape < 0;
                        if (AVOID == 1) skip = skip || shape == avoidShape;
                        if (AVOID == 2) skip = skip || (avoidRow && shape >= 0 && avoidRow[shape]);
#ifdef GRAFT_STAMPS
                        if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 10, 1ull);
#endif
                        if (!skip) exactSphereLazy(geom, shape, sx, sy, sz, dx, dy, dz, slackBase, slackScale, closest);
                    }
                }
            }
        };
        auto groups4 = [&](const float4v c0, const float4v c1, const float4v c2, const float4v c3, int j) {
            float e0, e1, e2, e3;
            evaluate(c0, c1, c2, c3, e0, e1, e2, e3);
            const float best = __builtin_fmaxf(__builtin_fmaxf(e0, e1), __builtin_fmaxf(e2, e3));
            if (!(best < limit)) {
#pragma nounroll
                for (int u = 0; u < 4; ++u) {
                    const float eu = u == 0 ? e0 : u == 1 ? e1 : u == 2 ? e2 : e3;
                    if (!(eu < limit)) {
#ifdef GRAFT_STAMPS
                        if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 9, 1ull);
#endif
                        if (!SUPER) {
                            const int m0 = (j + u) * P.groupSize;
                            const float4v *mb = table + (size_t)(P.memberBase + m0) * F;
#pragma nounroll
                            for (int m = 0; m < P.groupSize; m += 4) members4(mb + m * F, m0 + m);
                        }
                    }
                }
            }
        };
        if (!SUPER) {  // two levels: scan every group entry
            float4v a0 = table[0], a1 = table[1], a2 = table[2], a3 = table[3];
            const float4v *nb = table + 4 * F;  // the lane's next block: one address, four reads at fixed offsets
            for (int j = 0; j < P.groupEntries; j += 8) {
                const float4v b0 = nb[0], b1 = nb[1], b2 = nb[2], b3 = nb[3];
                groups4(a0, a1, a2, a3, j);
                nb += 4 * F;
                a0 = nb[0]; a1 = nb[1]; a2 = nb[2]; a3 = nb[3];
                nb += 4 * F;
                groups4(b0, b1, b2, b3, j + 4);
            }
        } else {
            // Three levels: scan the entries that stand for 8 groups each and open the groups of those that pass.
            // Control flow below the scan is uniform over the lanes of this walk (branches on ballots, the lane's own
            // condition travels as a predicate), so that the expensive exact test can be deferred: a lane parks its
            // candidate sphere in `pending`, and one exact test runs for ALL lanes with a parked candidate when some
            // lane needs the slot again (and once at the end).  With divergent secondary rays every lane meets
            // different spheres; this way the exact test runs about as often as the busiest lane needs it rather
            // than once per distinct sphere of the wave.
            int pending = -1;
            auto flush = [&]() {
                if (pending >= 0) {
#ifdef GRAFT_STAMPS
                    if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 10, 1ull);
#endif
                    exactSphereLazy(geom, pending, sx, sy, sz, dx, dy, dz, slackBase, slackScale, closest);
                    pending = -1;
                }
            };
            auto membersU = [&](auto *mb, int pos, bool inGroup) {
                float f0, f1, f2, f3;
                evaluate(mb[0], mb[1], mb[2], mb[3], f0, f1, f2, f3);
                const float best = __builtin_fmaxf(__builtin_fmaxf(f0, f1), __builtin_fmaxf(f2, f3));
                if (__ballot(inGroup && !(best < limit))) {
#pragma nounroll
                    for (int v = 0; v < 4; ++v) {
                        const float fv = v == 0 ? f0 : v == 1 ? f1 : v == 2 ? f2 : f3;
                        bool want = inGroup && !(fv < limit);
                        if (__ballot(want)) {
                            int shape = -1;
                            if (want) shape = P.memberShape[(size_t)(pos + v) * F + family];
                            want = want && shape >= 0;
                            if (AVOID == 1) want = want && shape != avoidShape;
                            if (AVOID == 2) want = want && !(avoidRow && shape >= 0 && avoidRow[shape]);
                            if (__ballot(want && pending >= 0)) flush();
                            if (want) pending = shape;
                        }
                    }
                }
            };
            auto groupsU = [&](const float4v c0, const float4v c1, const float4v c2, const float4v c3, int j, bool inSuper) {
                float e0, e1, e2, e3;
                evaluate(c0, c1, c2, c3, e0, e1, e2, e3);
                const float best = __builtin_fmaxf(__builtin_fmaxf(e0, e1), __builtin_fmaxf(e2, e3));
                if (__ballot(inSuper && !(best < limit))) {
#pragma nounroll
                    for (int u = 0; u < 4; ++u) {
                        const float eu = u == 0 ? e0 : u == 1 ? e1 : u == 2 ? e2 : e3;
                        const bool inGroup = inSuper && !(eu < limit);
                        if (__ballot(inGroup)) {
#ifdef GRAFT_STAMPS
                            if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 9, 1ull);
#endif
                            if (CONE) {
                                const int m0 = (j + u) * P.groupSize;
                                const float4v *mb = (const float4v *)P.geomF + (size_t)(P.memberBase + m0) * F + 4 * family;
#pragma nounroll
                                for (int m = 0; m < P.groupSize; m += 4) membersU(mb + m * F, m0 + m, inGroup);
                            } else {
                                const int m0 = (j + u) * P.groupSize;
                                const float4v *mb = (const float4v *)ldsGeom + P.ballMembersLds + m0;
#pragma nounroll
                                for (int m = 0; m < P.groupSize; m += 4) membersU(mb + m, m0 + m, inGroup);
                            }
                        }
                    }
                }
            };
            const float4v *sup = (const float4v *)ldsGeom + P.superLds + 4 * family;
            auto supers4 = [&](const float4v c0, const float4v c1, const float4v c2, const float4v c3, int j) {
                float e0, e1, e2, e3;
                evaluate(c0, c1, c2, c3, e0, e1, e2, e3);
                const float best = __builtin_fmaxf(__builtin_fmaxf(e0, e1), __builtin_fmaxf(e2, e3));
                if (__ballot(!(best < limit))) {
#pragma nounroll
                    for (int u = 0; u < 4; ++u) {
                        const float eu = u == 0 ? e0 : u == 1 ? e1 : u == 2 ? e2 : e3;
                        const bool inSuper = !(eu < limit);
                        if (__ballot(inSuper)) {
#ifdef GRAFT_STAMPS
                            if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 11, 1ull);
#endif
#pragma nounroll
                            for (int half = 0; half < 2; ++half) {
                                const int g0 = (j + u) * 8 + 4 * half;
                                const float4v *gb = table + (size_t)g0 * F;
                                groupsU(gb[0], gb[1], gb[2], gb[3], g0, inSuper);
                            }
                        }
                    }
                }
            };
            float4v a0 = sup[0], a1 = sup[1], a2 = sup[2], a3 = sup[3];
            const float4v *nb = sup + 4 * F;
            for (int j = 0; j < P.superEntries; j += 8) {
                const float4v b0 = nb[0], b1 = nb[1], b2 = nb[2], b3 = nb[3];
                supers4(a0, a1, a2, a3, j);
                nb += 4 * F;
                a0 = nb[0]; a1 = nb[1]; a2 = nb[2]; a3 = nb[3];
                nb += 4 * F;
                supers4(b0, b1, b2, b3, j + 4);
            }
            flush();
        }
    };
    if (!BALL || family != P.ballFamily) walk(std::true_type{});
    else if (BALL) walk(std::false_type{});
    finishBest(closest, sx, sy, sz, dx, dy, dz, h);
}

// Closest hit for rays that pass through no fixed point -- mirror rays and the shadow rays of what they hit -- in
// scenes whose per-family tables carry a uniform grid (packGrid).  Such rays leave a tile in all directions: a walk
// in which all lanes of a wave look at the same table entry costs the union of what the lanes need, so here every
// lane walks ITS OWN cells front to back (3-D DDA) and tests its own cells' spheres; the instructions are shared,
// the addresses are not.  Per sphere the single-precision screen of closestHitScreened (same arithmetic, same error
// budget) decides whether the exact test runs; exact tests are deferred one deep (`pending`) so that the expensive
// code runs when some lane needs the slot again, for all lanes that have a candidate.
//
// What makes it exact: a sphere the ray hits at parameter t has its hit point s + t d on its surface (to rounding),
// hence inside a cell that lists it, and every point within kGroupMargin of that cell's box lies in a cell that lists
// it as well.  The walk's idea of where the ray is differs from the true line by rounding only (well inside the
// margin, which grows with the coordinates: packGrid), so among the cells it visits up to parameter t there is one
// that lists the sphere.  A lane stops when its best hit lies before the exit of the cell just finished by more than a
// slack (no unvisited cell can hold a closer or equally close hit), or when it leaves the grid (every sphere lies
// inside).
//
// tLimit < infinity marks a shadow ray towards a light at that parameter.  The caller only asks whether the closest
// hit lies before or beyond the light (mainTracing.c:211-215: lit iff no hit, or the vector from the hit to the light
// points backwards).  Any hit that is before the light by more than 1e-5 (relative and absolute) settles "before"
// whichever sphere is closest, and nothing that starts beyond the light by that much can change "beyond"; only when
// the best hit falls inside that band is the walk continued to the exact closest hit, which the caller then judges
// with the reference's own arithmetic.
template <int AVOID>
__device__ __forceinline__ void closestHitGrid(const TraceParams &P, double sx, double sy, double sz, double dx, double dy,
                                               double dz, int avoidShape, const unsigned char *avoidRow,
                                               float negThreshold, double tLimit, Hit &h) {
    const ConstDoubles G = (ConstDoubles)(unsigned long long)P.grid;
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    const uint2v *cells = (const uint2v *)((const char *)ldsGeom + P.gridCellsOff);
    const unsigned short *refs = (const unsigned short *)((const char *)ldsGeom + P.gridRefsOff);
    const float4v *aos = (const float4v *)((const char *)ldsGeom + P.gridAosOff);
    const double *geom = P.geom;
    Best closest = {0, -1, -1};
    const double slackBase = fabs(sx) + fabs(sy) + fabs(sz);
    const double dd = dx * dx + dy * dy + dz * dz;
    const double slackScale = dd > 0.99 && dd < 1.01 ? 2e-15 : INFINITY;
    const float sxf = (float)sx, syf = (float)sy, szf = (float)sz;
    const float dxf = (float)dx, dyf = (float)dy, dzf = (float)dz;
    const int nx = P.gridNx, ny = P.gridNy, nz = P.gridNz;

    // The part of the ray inside the grid's box, and per axis: step, parameter of the next cell face, parameter per
    // cell.  Set up in double precision, stepped in single -- as parameters RELATIVE TO THE ENTRY POINT, which never
    // exceed the box's diagonal however far away the ray starts (an orthographic view's screen plane may lie tens
    // of thousands of units from a compact scene): the walk's idea of the ray parameter is then off by at most
    // (steps x half an ulp of a parameter below 3.5 S) ~ 4e-5 of the spheres' coordinate scale S, which the grid's
    // margin (packGrid) covers.
    double tEnter = 0, tLeave = INFINITY;
    bool alive = true;
    float tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ;
    double invX = 0, invY = 0, invZ = 0;
    auto slab = [&](double o, double d, double lo, double hi, double &inv) {
        if (fabs(d) > 1e-30) {  // (a smaller component moves the ray by nothing over any distance in the grid)
            double r = __builtin_amdgcn_rcp(d);
            r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
            inv = r;
            const double t0 = (lo - o) * r, t1 = (hi - o) * r;
            tEnter = fmax(tEnter, fmin(t0, t1));
            tLeave = fmin(tLeave, fmax(t0, t1));
        } else if (o < lo || o > hi) {
            alive = false;
        }
    };
    slab(sx, dx, G[GRID_LO], G[GRID_HI], invX);
    slab(sy, dy, G[GRID_LO + 1], G[GRID_HI + 1], invY);
    slab(sz, dz, G[GRID_LO + 2], G[GRID_HI + 2], invZ);
    alive = alive && tEnter <= tLeave;
    int cx = 0, cy = 0, cz = 0, stepX = 0, stepY = 0, stepZ = 0;
    auto enter = [&](double o, double d, double inv, double lo, double cs, double invCs, int n, int &c, int &step,
                     float &tMax, float &tDelta) {
        const double p = o + tEnter * d;
        double f = floor((p - lo) * invCs);
        f = f < 0 ? 0 : (f > (double)(n - 1) ? (double)(n - 1) : f);  // (false for NaN: cell 0)
        c = (int)f;
        if (inv != 0) {
            step = d > 0 ? 1 : -1;
            const double face = lo + (f + (d > 0 ? 1.0 : 0.0)) * cs;
            tMax = (float)((face - o) * inv - tEnter);
            tDelta = (float)(cs * fabs(inv));
        } else {
            step = 0;
            tMax = INFINITY;
            tDelta = 0;
        }
    };
    enter(sx, dx, invX, G[GRID_LO], G[GRID_CS], G[GRID_INVCS], nx, cx, stepX, tMaxX, tDeltaX);
    enter(sy, dy, invY, G[GRID_LO + 1], G[GRID_CS + 1], G[GRID_INVCS + 1], ny, cy, stepY, tMaxY, tDeltaY);
    enter(sz, dz, invZ, G[GRID_LO + 2], G[GRID_CS + 2], G[GRID_INVCS + 2], nz, cz, stepZ, tMaxZ, tDeltaZ);

    const bool shadow = tLimit < INFINITY;
    const double tSure = tLimit * (1 - 1e-5) - 1e-5;
    const float tStop = shadow ? (float)(tLimit * (1 + 1e-5) + 1e-5 - tEnter) : INFINITY;
    // A lane goes on to the next cell while tExit < goOn (both relative to the entry point): infinity until there is
    // a hit, then the hit's parameter less a slack that covers single precision (or minus infinity once a shadow ray
    // is known to be blocked).
    float goOn = tStop;
    // the lane's cell: entries not yet looked at, the next two of them, and where the ones after those are listed
    int left = 0, kA = 0, kB = 0, ref = 0;
    int pending = -1, last = -1;
    double4v pendingGeom = {0, 0, 0, 0};  // fetched when the candidate is parked: the load's latency hides behind the walk
    auto openCell = [&]() {
        const uint2v w = cells[(cz * ny + cy) * nx + cx];
        ref = (int)(w.x >> 8);
        left = (int)(w.x & 255u);
        kA = (int)(w.y & 0xffffu);
        kB = (int)(w.y >> 16);
    };
    // single-precision screen (closestHitScreened's arithmetic): true = the exact discriminant is negative
    auto miss = [&](const float4v e) {
        const float vx = sxf - e.x, vy = syf - e.y, vz = szf - e.z;
        float b = vx * dxf;
        b = __builtin_fmaf(vy, dyf, b);
        b = __builtin_fmaf(vz, dzf, b);
        float vv = vx * vx;
        vv = __builtin_fmaf(vy, vy, vv);
        vv = __builtin_fmaf(vz, vz, vv);
        return __builtin_fmaf(b, b, e.w - vv) < negThreshold;
    };
    auto avoided = [&](int k) {
        if (AVOID == 1) return k == avoidShape;
        if (AVOID == 2) return avoidRow != nullptr && avoidRow[k] != 0;
        return false;
    };
    if (alive) openCell();
#ifdef GRAFT_STAMPS  // diagnostic build: calls, iterations of the loop below, lanes alive in them, rounds of exact tests
    const unsigned long long stampOn = __ballot(1);
    const bool stampLeader = lane64() == (int)__builtin_ctzll(stampOn);
    if (P.counters && stampLeader) { atomicAdd(P.counters + 14, 1ull); atomicAdd(P.counters + 15, (unsigned long long)__popcll(stampOn)); }
    const unsigned long long stampGrid0 = clock64();
#endif
    while (alive) {
#ifdef GRAFT_STAMPS
        if (P.counters) {
            const unsigned long long on = __ballot(1);
            if (lane64() == (int)__builtin_ctzll(on)) { atomicAdd(P.counters + 12, 1ull); atomicAdd(P.counters + 13, (unsigned long long)__popcll(on)); }
        }
#endif
        // Up to two entries of the lane's cell through the screen, without branches (the entry list is padded, so the
        // reads are always inside it; what lies beyond the cell's entries is masked afterwards).
        const int k0 = kA, k1 = kB;
        const float4v e0 = aos[k0], e1 = aos[k1];
        kA = refs[ref];  // the pair after this one (read in any case: the list is padded; a new cell overwrites it)
        kB = refs[ref + 1];
        const bool m0 = miss(e0), m1 = miss(e1);
        int c0 = (left > 0 && k0 != last && !m0 && !avoided(k0)) ? k0 : -1;  // (a sphere that straddles two cells is
        int c1 = (left > 1 && !m1 && !avoided(k1)) ? k1 : -1;                //  listed in both: skip the last one seen)
        if (left > 0) last = left > 1 ? k1 : k0;
        left -= 2;
        ref += 2;
        if (c0 < 0) { c0 = c1; c1 = -1; }
        const bool cellDone = left <= 0;
        // One candidate is parked per lane.  The exact test runs (for every lane that has one parked) when a lane
        // needs the slot for another candidate, and before a lane decides whether to go on to the next cell.
        if (__ballot(c0 >= 0 || (pending >= 0 && cellDone))) {
#pragma nounroll
            for (int trip = 0; trip < 3; ++trip) {
                if (__ballot(pending >= 0 && (c0 >= 0 || cellDone))) {
#ifdef GRAFT_STAMPS
                    if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 10, 1ull);
#endif
                    if (pending >= 0) {
                        exactSphereLazy(pendingGeom, pending, sx, sy, sz, dx, dy, dz, slackBase, slackScale, closest);
                        pending = -1;
                        if (closest.shape >= 0) {
                            const float guard = (float)(closest.t - tEnter - (1e-5 * (fabs(closest.t) + slackBase) + 1e-6));
                            goOn = shadow && closest.t < tSure ? -INFINITY : fminf(tStop, guard);
                        }
                    }
                }
                if (c0 >= 0) {
                    pending = c0;
                    pendingGeom = *(const double4v *)(geom + 4 * (size_t)c0);
                    c0 = c1;
                    c1 = -1;
                }
                if (!__ballot(c0 >= 0 || (pending >= 0 && cellDone))) break;
            }
        }
        if (cellDone) {
            const float tExit = fminf(tMaxX, fminf(tMaxY, tMaxZ));
            bool stop = !(tExit < goOn);
            if (!stop) {
                if (tMaxX == tExit) {
                    cx += stepX; tMaxX += tDeltaX;
                    stop = (unsigned)cx >= (unsigned)nx;
                } else if (tMaxY == tExit) {
                    cy += stepY; tMaxY += tDeltaY;
                    stop = (unsigned)cy >= (unsigned)ny;
                } else {
                    cz += stepZ; tMaxZ += tDeltaZ;
                    stop = (unsigned)cz >= (unsigned)nz || stepZ == 0;  // (all three infinite: a ray that does not move)
                }
            }
            if (stop) alive = false;
            else openCell();
        }
    }
#ifdef GRAFT_STAMPS
    if (P.counters && stampLeader) atomicAdd(P.counters + 16, (unsigned long long)(clock64() - stampGrid0));
#endif
    finishBest(closest, sx, sy, sz, dx, dy, dz, h);
}

// ---- candidate lists (SCREEN == 6, 7) ---------------------------------------------------------------------------
//
// Sphere-only scenes seen in perspective.  Every ray of the path passes through a point that is known before the ray
// is: a primary ray through the camera, a shadow ray through its light AND through a point of the sphere it starts
// on.  That pins down, ahead of time, the few spheres such a ray can meet:
//   * primary rays: the spheres whose cone, seen from the camera, overlaps the cone that bounds ALL primary rays of
//     the wave's 8x8 tile (every sample of every pixel).  The wave works that list out once per tile, all lanes
//     together, from the camera's two-level cone table (buildTileList), and keeps it in LDS;
//   * shadow rays: the occluder list of (sphere, light), built on the host (packLightLists).
// closestHitLists then runs the exact test (shapes.c:47-73) on the lane's own list, behind the single-precision screen
// of closestHitScreened; lanes of a wave follow different lists, the instructions are shared.  A sphere that is on
// no list of a ray has a provably negative discriminant in the reference (or can only be met beyond the light, which
// lighting() treats like no hit, mainTracing.c:211-215), so hit masks, hit points and colours are unchanged.
// GLOBAL (level 7: scenes with mirrors, whose other rays walk the uniform grid, and scenes whose tables do not fit in
// LDS): the camera table, the occluder lists and the exact doubles are read from global memory; the spheres' screen
// table and the waves' tile lists are in LDS as at level 6.

// The cone that bounds every primary ray of a tile, as (axis t, cos, sin of the half-angle tau), and whether an entry
// (a, kappa2) of a cone table can be met by a line inside it.  The table's kappa2 = cos^2(A) - 16 u rounded down
// (packConeTable; u = 2^-24) stands for an angle A' >= A with cos A' <= cos A - 4.8e-7; the entry is left out iff
// |a . t| + 1e-9 < cos(A' + tau) = cos A' cos tau - sin A' sin tau: then the angle between the axis t and the entry's
// axis (as lines) exceeds A + tau even after the entry's axis was rounded to float (|a . t| is off by at most 1.2e-7),
// and every line within tau of t misses the entry's cone.  Evaluated in double precision.
struct TileCone {
    double ax, ay, az, cosT, sinT;
    bool ok;  // false: no usable bound (non-finite or wide): the tile lists every sphere
};

// entry = {axis, cos A'} with the sine in a table of its own (packLevel6 forms both from the cone table's kappa2, the
// cosine rounded down and the sine up); cos A' = -2 always passes, +inf never does
__device__ __forceinline__ bool coneMeetsTile(const float4v e, float sinA, const TileCone &T) {
    const double c = fabs(__builtin_fma((double)e.z, T.az, __builtin_fma((double)e.y, T.ay, (double)e.x * T.ax)));
    return c + 1e-9 >= __builtin_fma(-(double)sinA, T.sinT, (double)e.w * T.cosT);
}

// The tile list of an ORTHOGRAPHIC view: every primary ray has the direction d = unit(-viewDir) (mainTracing.c:410-413)
// and starts at screenCenter + x lrUnit + y udUnit; in the plane across d through the screen centre its line crosses at
// x P(lrUnit) + y P(udUnit) (P takes the component along d out; the camera block holds the projected vectors), within
// (half width + half height) of the tile's centre w since the projected vectors are no longer than 1.  An entry {q, R}
// of the table (packOrthoTable: R holds the radius, the margin and the rounding to float) can be met iff
// |q - w| <= R + half width + half height (+ a slack for the rounding of the rays' own screen points).
template <bool GLOBAL>
__device__ __forceinline__ int buildTileListOrtho(const TraceParams &P, ConstDoubles C, double xLo, double xHi, double yLo,
                                                  double yHi, unsigned short *list) {
    const int lane = lane64();
    const double xc = 0.5 * (xLo + xHi), yc = 0.5 * (yLo + yHi);
    double w[3];
    for (int a = 0; a < 3; ++a) w[a] = xc * C[CAM_LRUNIT + a] + yc * C[CAM_UDUNIT + a];
    const double reach = (0.5 * (xHi - xLo) + 0.5 * (yHi - yLo)) * (1.0 + 1e-9) + 1e-9 * P.familyScale;
    if (!(reach >= 0 && reach < 1e300) || !(w[0] - w[0] == 0 && w[1] - w[1] == 0 && w[2] - w[2] == 0)) return -1;  // (NaN, infinite)
    auto meets = [&](const float4v e) {
        const double dx = (double)e.x - w[0], dy = (double)e.y - w[1], dz = (double)e.z - w[2];
        const double rr = (double)e.w + reach;  // (-inf for padding entries: never)
        return rr >= 0 && __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)) <= rr * rr * (1.0 + 1e-12);
    };
    const char *image = GLOBAL ? (const char *)P.geomF : (const char *)ldsGeom;
    const float4v *groups = (const float4v *)((const char *)ldsGeom + P.listGroupsOff);
    const float4v *members = (const float4v *)(image + P.listMembersOff);
    const unsigned short *shapeOf = (const unsigned short *)(image + P.listShapeOff);
    const int gs = P.groupSize;
    int count = 0;
    for (int g0 = 0; g0 < P.listGroups; g0 += 64) {
        const int g = g0 + lane;
        unsigned long long open = __ballot(g < P.listGroups && meets(groups[g < P.listGroups ? g : 0]));
        while (open) {
            const int gOpen = g0 + (int)__builtin_ctzll(open);
            open &= open - 1;
            bool take = false;
            unsigned short id = 0xffff;
            if (lane < gs) {
                id = shapeOf[gOpen * gs + lane];
                take = id != 0xffff && meets(members[gOpen * gs + lane]);
            }
            const unsigned long long taken = __ballot(take);
            const int at = count + (int)__popcll(taken & ((1ull << lane) - 1));
            if (take && at < kTileListCap) list[at] = id;
            count += (int)__popcll(taken);
        }
    }
    return count <= kTileListCap ? count : -1;
}

// The wave's tile list: sphere numbers (16 bits) at `list` in LDS; returns the count, or -1 when the list would
// exceed kTileListCap entries or the tile has no usable bounding cone (the caller then lists every sphere).  Called by
// all 64 lanes.  (xLo, xHi) x (yLo, yHi): world offsets on the screen plane that bound every sample of the tile.
template <bool GLOBAL>
__device__ __forceinline__ int buildTileList(const TraceParams &P, ConstDoubles C, double xLo, double xHi, double yLo,
                                             double yHi, unsigned short *list) {
    const int lane = lane64();
    if (P.projectionType == GRAFT_ORTHOGRAPHIC) return buildTileListOrtho<GLOBAL>(P, C, xLo, xHi, yLo, yHi, list);
    auto towards = [&](double x, double y, double v[3]) {
        for (int a = 0; a < 3; ++a) v[a] = C[CAM_TOSCREEN + a] + x * C[CAM_LRUNIT + a] + y * C[CAM_UDUNIT + a];
    };
    TileCone T;
    double t[3], v[3];
    towards(0.5 * (xLo + xHi), 0.5 * (yLo + yHi), t);
    const double tt = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
    const double inv = 1.0 / sqrt(tt);
    T.ax = t[0] * inv; T.ay = t[1] * inv; T.az = t[2] * inv;
    // lane c (mod 4) looks at corner c; a rectangle in a plane lies inside a cone when its corners do
    towards((lane & 1) ? xHi : xLo, (lane & 2) ? yHi : yLo, v);
    const double cx = T.ay * v[2] - T.az * v[1], cy = T.az * v[0] - T.ax * v[2], cz = T.ax * v[1] - T.ay * v[0];
    const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double s2 = (cx * cx + cy * cy + cz * cz) / vv;  // sin^2 of the angle between axis and corner
    const double along = T.ax * v[0] + T.ay * v[1] + T.az * v[2];
    if (!(along > 0) || !(s2 == s2)) s2 = 2.0;
    s2 = fmax(s2, __shfl_xor(s2, 1));
    s2 = fmax(s2, __shfl_xor(s2, 2));
    // generous slack: the rays' screen points are formed by other (bit-exact) operations than the corners here, and
    // differ from them by rounding only
    const double sinT = sqrt(s2) * (1.0 + 1e-6) + 1e-9;
    T.sinT = sinT;
    T.cosT = sqrt(fmax(0.0, 1.0 - sinT * sinT));
    T.ok = tt > 0 && tt < 1e300 && sinT < 0.5;  // (false for NaN)
    if (!T.ok) return -1;

    // (the group entries are in LDS at either level; the member entries of the groups a tile opens come from global
    // memory at level 7)
    const char *image = GLOBAL ? (const char *)P.geomF : (const char *)ldsGeom;
    const float4v *groups = (const float4v *)((const char *)ldsGeom + P.listGroupsOff);
    const float *groupSin = (const float *)((const char *)ldsGeom + P.listSinesOff);
    const float4v *members = (const float4v *)(image + P.listMembersOff);
    const float *memberSin = (const float *)(image + P.listMemberSinesOff);
    const unsigned short *shapeOf = (const unsigned short *)(image + P.listShapeOff);
    const int gs = P.groupSize;
    int count = 0;
    for (int g0 = 0; g0 < P.listGroups; g0 += 64) {
        const int g = g0 + lane;
        const int gRead = g < P.listGroups ? g : 0;
        unsigned long long open = __ballot(g < P.listGroups && coneMeetsTile(groups[gRead], groupSin[gRead], T));
        while (open) {
            const int gOpen = g0 + (int)__builtin_ctzll(open);
            open &= open - 1;
            // lane m looks at member m of the group (a group has at most 64 members)
            bool take = false;
            unsigned short id = 0xffff;
            if (lane < gs) {
                id = shapeOf[gOpen * gs + lane];
                take = id != 0xffff && coneMeetsTile(members[gOpen * gs + lane], memberSin[gOpen * gs + lane], T);
            }
            const unsigned long long taken = __ballot(take);
            const int at = count + (int)__popcll(taken & ((1ull << lane) - 1));
            if (take && at < kTileListCap) list[at] = id;
            count += (int)__popcll(taken);
        }
    }
    return count <= kTileListCap ? count : -1;
}

// Closest hit among the spheres of the lane's list: `len` sphere numbers from chunk `pos` on (a chunk is four 16-bit
// numbers, fetched with one read and used up over four trips) -- of the list pool, or, `tile`, of LDS, where the
// waves' tile lists are (at level 6 the pool is in LDS too and `pos` counts from its start either way).  Lanes of a
// wave follow different lists through one code site.  Spheres are met out of index order: see closestHitGrouped for
// why the winner does not depend on it.
// `self` >= 0: the sphere a shadow ray starts on, which its list leaves out.  The ray starts a hair outside it; when
// it moves away from the centre (the light lies in front of the surface) both roots are negative and the sphere is
// done with (the first clause of exactSphereLazy: shapes.c:62-66 reports a miss); otherwise it is the first candidate.
//
// !GLOBAL (everything in LDS): two loops that take turns.  The screen loop runs the single-precision screen of
// closestHitScreened over the lists and keeps the numbers of the spheres that pass, up to four per lane in one 64-bit
// register; when some lane holds four, or every list has ended, the exact loop takes them one per trip -- the lanes'
// first survivors together, then their second ones.  Both loops have one exit, decided by ballots, and change their
// state in place (loops with several exits cost a dozen register copies a trip).
// GLOBAL (the exact doubles come from global memory): one loop; a sphere that passes the screen is parked with its
// doubles already requested, and the exact test runs, for every lane that has one parked, when some lane needs the
// slot again or has reached the end of its list -- the request is a trip old by then.
template <bool GLOBAL>
__device__ __forceinline__ void closestHitLists(const TraceParams &P, double sx, double sy, double sz, double dx, double dy,
                                                double dz, int pos, int len, bool tile, int self, float negThreshold, Hit &h) {
    const float4v *aos = (const float4v *)ldsGeom;
    const double4v *geom = GLOBAL ? (const double4v *)P.geom : (const double4v *)((const char *)ldsGeom + P.exactLdsOffset);
    const unsigned long long *pool = (const unsigned long long *)((GLOBAL ? (const char *)P.geomF : (const char *)ldsGeom) + P.listPoolOff);
    const unsigned long long *tiles = (const unsigned long long *)ldsGeom;
    unsigned long long chunk = 0;
    int trip = 0;  // (the same for all lanes)
    Best closest = {0, -1, -1};
    const double slackBase = fabs(sx) + fabs(sy) + fabs(sz);
    const double dd = dx * dx + dy * dy + dz * dz;
    const double slackScale = dd > 0.99 && dd < 1.01 ? 2e-15 : INFINITY;
    const float sxf = (float)sx, syf = (float)sy, szf = (float)sz;
    const float dxf = (float)dx, dyf = (float)dy, dzf = (float)dz;
    int pending = -1;
    double4v pendingGeom = {0, 0, 0, 0};
    if (self >= 0) {
        pendingGeom = geom[self];
        const double vx = sx - pendingGeom.x, vy = sy - pendingGeom.y, vz = sz - pendingGeom.z;
        const double b = vx * dx + vy * dy + vz * dz;
        const double w = (vx * vx + vy * vy + vz * vz) - pendingGeom.w;
        if (!(b > 0 && w > b * b * 8.9e-16)) pending = self;
    }
    // single-precision screen (closestHitScreened's arithmetic): below the threshold = exact discriminant negative
    auto passes = [&](const float4v e) {
        const float vx = sxf - e.x, vy = syf - e.y, vz = szf - e.z;
        float b = vx * dxf;
        b = __builtin_fmaf(vy, dyf, b);
        b = __builtin_fmaf(vz, dzf, b);
        float vv = vx * vx;
        vv = __builtin_fmaf(vy, vy, vv);
        vv = __builtin_fmaf(vz, vz, vv);
        return !(__builtin_fmaf(b, b, e.w - vv) < negThreshold);
    };
    if constexpr (!GLOBAL) {
        // numbers + 1 of the spheres that passed the screen and wait for the exact test, 16 bits each, the latest lowest
        unsigned long long kept = pending >= 0 ? (unsigned long long)(pending + 1) : 0;
        bool fresh = true;  // no exact test has run yet for this ray (the same for all lanes: the first trip of the exact loop)
        for (;;) {
            bool screening = (__ballot(len > 0) != 0) & (__ballot((unsigned)(kept >> 32) > 0xffffu) == 0);
            while (screening) {
                if (len > 0) {
                    if ((trip & 3) == 0) {
                        chunk = pool[pos];
                        ++pos;
                    }
                    const int k = (int)(chunk & 0xffffu);
                    chunk >>= 16;
                    if (passes(aos[k])) kept = (kept << 16) | (unsigned)(k + 1);
                }
                trip = __builtin_amdgcn_readfirstlane(trip + 1);
                --len;
                screening = (__ballot(len > 0) != 0) & (__ballot((unsigned)(kept >> 32) > 0xffffu) == 0);
            }
            bool testing = __ballot(kept != 0) != 0;
            while (testing) {
#ifdef GRAFT_STAMPS
                if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 10, 1ull);
#endif
                if (kept != 0) {
                    const int k = (int)(kept & 0xffffu) - 1;
                    kept >>= 16;
                    exactSphereLazy(geom[k], k, sx, sy, sz, dx, dy, dz, slackBase, slackScale, closest, fresh);
                }
                fresh = false;
                testing = __ballot(kept != 0) != 0;
            }
            if (__ballot(len > 0) == 0) break;
        }
    } else {
        while (__ballot(len > 0 || pending >= 0)) {
            int j = -1;
            if ((trip & 3) == 0 && len > 0) {
                chunk = tile ? tiles[pos] : pool[pos];
                ++pos;
            }
            ++trip;
            if (len > 0) {
                const int k = (int)(chunk & 0xffffu);
                chunk >>= 16;
                if (passes(aos[k])) j = k;
            }
            --len;
            if (__ballot(pending >= 0 && (j >= 0 || len <= 0))) {
                if (pending >= 0) {
#ifdef GRAFT_STAMPS
                    if (P.counters && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(P.counters + 10, 1ull);
#endif
                    exactSphereLazy(pendingGeom, pending, sx, sy, sz, dx, dy, dz, slackBase, slackScale, closest);
                    pending = -1;
                }
            }
            if (j >= 0) {
                pending = j;
                pendingGeom = geom[j];
            }
        }
    }
    finishBest(closest, sx, sy, sz, dx, dy, dz, h);
}

// fresnel() of the reference, mainTracing.c:273-294, bug for bug: the two indices live in `float`, sinN is
// rounded to float, and the double result is truncated to int, so the weight is 0 unless it is exactly 1.
// Float division and multiplication are done through double (exact product / correctly rounded quotient, then
// one rounding to float), which equals the IEEE single-precision operation the CPU performs.
__device__ __forceinline__ int fresnelWeight(double refrIndex, const double n[3], const double d[3]) {
    double cosD = dot3(n[0], n[1], n[2], d[0], d[1], d[2]);
    if (cosD < -1) cosD = -1;
    if (cosD > 1) cosD = 1;
    float etaD = 1, etaN = (float)refrIndex;
    if (cosD > 0) {
        etaD = etaN;
        etaN = 1;
    }
    const float ratio = (float)((double)etaD / (double)etaN);
    const double under = (1 - cosD * cosD) < 0 ? 0 : (1 - cosD * cosD);
    const float sinN = (float)((double)ratio * sqrt(under));
    if (sinN >= 1) return 1;
    const float sq = (float)((double)sinN * (double)sinN);
    const float rest = (float)(1.0 - (double)sq);
    const double cosN = sqrt((double)(rest < 0 ? 0.0f : rest));
    cosD = fabs(cosD);
    const double Rs = (((double)etaN * cosD) - ((double)etaD * cosN)) / (((double)etaN * cosD) + ((double)etaD * cosN));
    const double Rp = (((double)etaD * cosD) - ((double)etaN * cosN)) / (((double)etaD * cosD) + ((double)etaN * cosN));
    return (int)((Rs * Rs + Rp * Rp) / 2);
}

// The ray refraction() shoots, mainTracing.c:232-265, bug for bug: the POSITION (not the direction) is scaled by
// eta and serves both in the refracted direction and as the ray's origin.  Returns the child's shouldReflect.
__device__ __forceinline__ int refractionRay(double refrIndex, const double loc[3], const double d[3],
                                             const double nrm[3], double o[3], double dir[3]) {
    int childReflects = 1;
    double s[3] = {loc[0], loc[1], loc[2]};
    double n[3] = {nrm[0], nrm[1], nrm[2]};
    double cosD = dot3(d[0], d[1], d[2], nrm[0], nrm[1], nrm[2]);
    if (cosD < -1) cosD = -1;
    if (cosD > 1) cosD = 1;
    double etaD = 1, etaN = refrIndex;
    if (cosD < 0) {
        cosD = -cosD;
    } else {
        etaD = etaN;
        etaN = 1;
        n[0] = -1 * n[0]; n[1] = -1 * n[1]; n[2] = -1 * n[2];
        childReflects = 0;
    }
    const double eta = etaD / etaN;
    const double k = 1 - eta * eta * (1 - cosD * cosD);
    if (k < 0) {
        dir[0] = d[0]; dir[1] = d[1]; dir[2] = d[2];
    } else {
        s[0] = eta * s[0]; s[1] = eta * s[1]; s[2] = eta * s[2];
        const double f = eta * cosD - sqrt(k);
        n[0] = f * n[0]; n[1] = f * n[1]; n[2] = f * n[2];
        dir[0] = s[0] + n[0]; dir[1] = s[1] + n[1]; dir[2] = s[2] + n[2];
    }
    unit3(dir[0], dir[1], dir[2], dir[0], dir[1], dir[2]);
    o[0] = s[0] + .000001 * dir[0];
    o[1] = s[1] + .000001 * dir[1];
    o[2] = s[2] + .000001 * dir[2];
    return childReflects;
}

// ray through world offsets (xo, yo) of the screen plane: mainTracing.c:405-417 / :425-437
__device__ __forceinline__ ConstDoubles cameraView(const TraceParams &P) {
    return (ConstDoubles)(unsigned long long)P.camera;
}

// the screen point of world offsets (xo, yo): two getNewPoint steps, mainTracing.c:407-408 / :426-427
__device__ __forceinline__ void screenPoint(ConstDoubles C, double xo, double yo, double o[3]) {
    const double centre[3] = {C[CAM_SCREEN], C[CAM_SCREEN + 1], C[CAM_SCREEN + 2]};
    const double lr[3] = {C[CAM_LR], C[CAM_LR + 1], C[CAM_LR + 2]};
    const double ud[3] = {C[CAM_UD], C[CAM_UD + 1], C[CAM_UD + 2]};
    double col[3];
    newPoint(centre, lr, C[CAM_LRDEN], xo, col);
    newPoint(col, ud, C[CAM_UDDEN], yo, o);
}

__device__ __forceinline__ void primaryRay(const TraceParams &P, double xo, double yo, double o[3], double d[3]) {
    const ConstDoubles C = cameraView(P);
    screenPoint(C, xo, yo, o);
    d[0] = d[1] = d[2] = 0;
    if (P.projectionType == GRAFT_ORTHOGRAPHIC) unit3(-C[CAM_VIEW], -C[CAM_VIEW + 1], -C[CAM_VIEW + 2], d[0], d[1], d[2]);
    else unit3(o[0] - C[CAM_POS], o[1] - C[CAM_POS + 1], o[2] - C[CAM_POS + 2], d[0], d[1], d[2]);
}

__device__ __forceinline__ unsigned quant8(double v) {
    double c = v < 0 ? 0 : (v > 1 ? 1 : v);
    if (!(v == v)) c = 0;
    return (unsigned)(c * 255 + 0.5);
}


}  // namespace graft
