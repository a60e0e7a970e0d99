// deferred_mirror.h -- mirror rays of large scenes traced in launches of their own (included by grafthip.hip only;
// compile with -ffp-contract=off).
//
// In the one-launch kernel a lane carries its pixel through shootRay's whole recursion (mainTracing.c:296-372).  The
// rays of a pixel's first level pass through a fixed point (camera, lights) and share table entries across a tile;
// what a mirror reflects does not: those rays walk the grid lane by lane, take very different numbers of cells, and
// the lanes whose pixel saw no mirror wait for them.  Here the first-level kernel only WRITES each mirror ray down
// (MirrorRecord, one slot per pixel and recursion level) and ends the pixel's lane; further launches per recursion
// level then work on dense lists of pixels:
//   walk (closest)  one listed pixel per lane: the grid walk of closestHitGrid for its record; a lane whose ray ends
//                   takes the next pixel at once, so the lanes of a wave stay busy whatever the spread in cells per ray;
//   walk (shadow)   one (hit, light) pair per lane, same loop with the any-hit rule;
//   shade           one hit per lane: mainTracing.c:330-369 for that hit; writes the next level's record or hands
//                   the value up the pixel's records to the frame;
//   compact         between them: turns "this entry goes on / does not" into the next dense list.
// No list is appended to entry by entry: one address takes about 10^8 atomic operations per second, far fewer than
// there are rays.  Producers write a candidate per input position, the compaction adds one count per 4096 of them,
// and the walks draw their work in chunks.
// Values are combined with the reference's operations in the reference's order (innermost level first), so the
// frame is bit for bit the one-launch kernel's.
#pragma once

#include "closest_hit.h"

#ifndef GRAFT_REFILL_LANES
#define GRAFT_REFILL_LANES 16
#endif
#ifndef GRAFT_EXACT_LANES
#define GRAFT_EXACT_LANES 16
#endif

namespace graft {

constexpr unsigned kNoTask = 0xffffffffu;
constexpr int kRefillLanes = GRAFT_REFILL_LANES;  // idle lanes in a wave that trigger a draw of new work
constexpr int kExactLanes = GRAFT_EXACT_LANES;    // lanes with a candidate waiting that trigger a round of exact tests
constexpr int kDrawMax = 1024;      // most tasks a wave draws from a list at once
constexpr int kCountsPerLevel = 8;  // words per level in MirrorQueue::counts
enum { MQ_RECORDS = 0, MQ_HITS = 1, MQ_CURSOR_CLOSEST = 2, MQ_CURSOR_SHADOW = 3 };

struct MirrorRecord {     // 64 bytes: a mirror ray and what its value is needed for
    double o[3], d[3];    // reflection(), mainTracing.c:180-190
    double litW;          // diffuse weight of the hit that shot the ray (mainTracing.c:223-225), with ...
    unsigned shapeLight;  // ... its shape index | (index of its last unshadowed light + 1) << 24
    unsigned pad;
};
static_assert(sizeof(MirrorRecord) == 64, "MirrorRecord layout");

struct MirrorHit {  // closest hit of a pixel's record of the level in work
    double x, y, z;
    int shape, pad;
};

struct MirrorQueue {
    MirrorRecord *records;  // the record of pixel p at level L (1 ..): records[(L - 1) * cap + p]
    MirrorHit *hits;        // [pixel]
    unsigned *cand;         // candidates for the next list, one per input position (kNoTask = dropped)
    unsigned *active;       // pixels with a record at the level in work
    unsigned *hitList;      // those of them whose ray hit something
    unsigned char *lit;     // [position in hitList][light]: 1 = that light reaches the hit
    unsigned *counts;       // kCountsPerLevel words per level, see MQ_*; zeroed before the first-level launch
    unsigned cap;           // pixels of the band
    int level;
};

__device__ __forceinline__ void storePixel(const TraceParams &P, size_t p, const double acc[3]) {
    if (P.format == GRAFT_OUT_F64_RGB) {
        double *f = (double *)P.frame + 3 * p;
        f[0] = acc[0]; f[1] = acc[1]; f[2] = acc[2];
    } else if (P.format == GRAFT_OUT_F32_RGBA) {
        ((float4 *)P.frame)[p] = make_float4((float)acc[0], (float)acc[1], (float)acc[2], 1.0f);
    } else {
        ((unsigned *)P.frame)[p] = quant8(acc[0]) | (quant8(acc[1]) << 8) | (quant8(acc[2]) << 16) | 0xff000000u;
    }
}

// diffuse + ambient part of a shaded hit, mainTracing.c:199-228 and the first terms of :363-369
__device__ __forceinline__ void localPart(const TraceParams &P, const DevMaterial &m, double litW, int litLight, double part[3]) {
    const double rgb[3] = {m.color[0], m.color[1], m.color[2]};
    double light[3] = {0, 0, 0};
    if (litLight >= 0) {
        const DevLight &L = P.lights[litLight];
        for (int c = 0; c < 3; ++c) {
            double t = L.color[c] * rgb[c];
            light[c] = litW * t;
        }
    }
    const double wDiff = (1 - m.ambientLight) * (1 - m.reflectivity) * (1 - m.transparency);
    for (int c = 0; c < 3; ++c) {
        const double amb = m.ambientLight * rgb[c];
        const double lit = wDiff * light[c];
        part[c] = lit + (amb + 0.0);
    }
}

// v = value of pixel `i`'s mirror ray of `level`: combine it into the hit that shot the ray (:338, :363-369),
// that hit's value into the level above, ... and store the pixel.
__device__ __forceinline__ void handUp(const TraceParams &P, const MirrorQueue &Q, int level, unsigned i, double v[3]) {
    for (int lv = level; lv >= 1; --lv) {
        const MirrorRecord *r = Q.records + (size_t)(lv - 1) * Q.cap + i;
        const double litW = r->litW;
        const unsigned sl = r->shapeLight;
        const DevMaterial &m = P.materials[sl & 0xffffffu];
        double part[3];
        localPart(P, m, litW, (int)(sl >> 24) - 1, part);
        const double wRefl = (1 - m.ambientLight) * m.reflectivity;
        for (int c = 0; c < 3; ++c) {
            const double rgb = m.color[c];
            const double t = (1 - .5) * (1.0 - rgb);  // mainTracing.c:334-336
            double mir = (rgb + t) * v[c];
            mir = wRefl * mir;
            const double out = mir + part[c];
            v[c] = (0.0 * 0.0) + out;
        }
    }
    storePixel(P, i, v);
}

// out[0 .. *outCount) = the entries of in[0 .. n) that are not kNoTask (n = *nDev when nDev is given); one atomic
// addition per 4096 entries.  *outCount must be zero beforehand.
__global__ void __launch_bounds__(256) mirrorCompactKernel(const unsigned *in, unsigned nHost, const unsigned *nDev,
                                                           unsigned *out, unsigned *outCount) {
    __shared__ unsigned waveTotal[4];
    __shared__ unsigned blockBase;
    const unsigned n = nDev ? *nDev : nHost;
    const int lane = lane64(), wave = (int)(threadIdx.x >> 6);
    for (unsigned long long span = 4096ull * blockIdx.x; span < n; span += 4096ull * gridDim.x) {
        const unsigned long long base = span + 1024ull * wave;
        unsigned v[16];
        int c = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned long long idx = base + 64 * q + lane;
            v[q] = idx < n ? in[idx] : kNoTask;
            c += v[q] != kNoTask ? 1 : 0;
        }
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) waveTotal[wave] = (unsigned)incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned all = waveTotal[0] + waveTotal[1] + waveTotal[2] + waveTotal[3];
            blockBase = all ? atomicAdd(outCount, all) : 0;
        }
        __syncthreads();
        unsigned o = blockBase + (unsigned)(incl - c);
        for (int w = 0; w < wave; ++w) o += waveTotal[w];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (v[q] != kNoTask) out[o++] = v[q];
        __syncthreads();
    }
}

// Closest hits (SHADOW = false: one listed pixel per task) or light visibility (SHADOW = true: one (hit, light) pair
// per task) of one level.  Persistent waves; each lane walks its own ray through the grid exactly as closestHitGrid
// does (see there for why the result is the reference's); when kRefillLanes lanes of a wave have finished, they write
// their results down and take new tasks together, from a run of consecutive tasks the wave has drawn for itself.
template <int AVOID, bool SHADOW>
__global__ void __launch_bounds__(kLargeBlock) mirrorWalkKernel(const TraceParams P, const MirrorQueue Q) {
    const unsigned total = SHADOW ? Q.counts[kCountsPerLevel * Q.level + MQ_HITS] * (unsigned)P.numLights
                                  : Q.counts[kCountsPerLevel * Q.level + MQ_RECORDS];
    if (total == 0) return;
    {   // the grid's part of the table, at the place the first-level kernel has it
        const int first = P.gridCellsOff / 4;
        const int words = P.ldsHeadFloats + P.ldsTailFloats - first;
        float *dst = (float *)ldsGeom + first;
        const float *src = P.geomF + P.ldsTailSrc + (first - P.ldsHeadFloats);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    unsigned *cursor = Q.counts + kCountsPerLevel * Q.level + (SHADOW ? MQ_CURSOR_SHADOW : MQ_CURSOR_CLOSEST);
    const MirrorRecord *records = Q.records + (size_t)(Q.level - 1) * Q.cap;
    // tasks per draw: at least eight draws per wave (load balance at the end of the list)
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    unsigned perDraw = total / (8 * waves);
    perDraw = perDraw < 64 ? 64 : perDraw > (unsigned)kDrawMax ? (unsigned)kDrawMax : perDraw;
    unsigned drawNext = 0, drawEnd = 0;  // (wave-uniform) what is left of the wave's last draw
    const ConstDoubles G = (ConstDoubles)(unsigned long long)P.grid;
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    const uint2v *cells = (const uint2v *)((const char *)ldsGeom + P.gridCellsOff);
    const unsigned short *refs = (const unsigned short *)((const char *)ldsGeom + P.gridRefsOff);
    const float4v *aos = (const float4v *)((const char *)ldsGeom + P.gridAosOff);
    const double *geom = P.geom;
    const int nx = P.gridNx, ny = P.gridNy, nz = P.gridNz;
    unsigned long long rayCount = 0, testCount = 0;

    // ---- lane state ----
    unsigned task = kNoTask;  // what the lane works on
    unsigned pixel = 0;       // closest: the task's pixel
    bool alive = false;       // its ray is still walking
    bool drained = false;     // (wave-uniform) the cursor has passed the end of the list
    bool exactLoop = false;   // the ray is not one the screens are valid for: the exact loop decides it when it is finished
    double sx = 0, sy = 0, sz = 0, dx = 0, dy = 0, dz = 0, tLimit = INFINITY;
    float sxf = 0, syf = 0, szf = 0, dxf = 0, dyf = 0, dzf = 0, negThreshold = 0;
    float tMaxX = 0, tMaxY = 0, tMaxZ = 0, tDeltaX = 0, tDeltaY = 0, tDeltaZ = 0;
    int cx = 0, cy = 0, cz = 0, stepX = 0, stepY = 0, stepZ = 0;
    float goOn = 0, tStop = 0;
    int left = 0, kA = 0, kB = 0, ref = 0, pending = -1, last = -1;
    double4v pendingGeom = {0, 0, 0, 0};
    Best closest = {0, -1, -1};

    auto openCell = [&]() {
        const uint2v w = cells[(cz * ny + cy) * nx + cx];
        ref = (int)(w.x >> 8);
        left = (int)(w.x & 255u);
        kA = (int)(w.y & 0xffffu);
        kB = (int)(w.y >> 16);
    };
    auto lightOf = [&](unsigned t) { return (int)(t % (unsigned)P.numLights); };

    // the ray of `task` and the start of its walk (closestHitGrid's set-up)
    auto begin = [&]() {
        if (SHADOW) {  // mainTracing.c:200-205
            const unsigned j = task / (unsigned)P.numLights;
            const MirrorHit *hp = Q.hits + Q.hitList[j];
            const double lx = hp->x, ly = hp->y, lz = hp->z;
            const DevLight &L = P.lights[lightOf(task)];
            dx = dy = dz = 0;
            unit3(L.loc[0] - lx, L.loc[1] - ly, L.loc[2] - lz, dx, dy, dz);
            sx = lx + .000001 * dx;
            sy = ly + .000001 * dy;
            sz = lz + .000001 * dz;
            tLimit = (L.loc[0] - sx) * dx + (L.loc[1] - sy) * dy + (L.loc[2] - sz) * dz;
        } else {
            pixel = Q.active[task];
            const MirrorRecord *r = records + pixel;
            sx = r->o[0]; sy = r->o[1]; sz = r->o[2];
            dx = r->d[0]; dy = r->d[1]; dz = r->d[2];
            tLimit = INFINITY;
        }
        closest.t = 0; closest.dist = -1; closest.shape = -1;
        pending = -1; last = -1;
        const double s1 = fabs(sx) + fabs(sy) + fabs(sz);
        const bool tame = s1 < 1e15 && fabs(dx) <= 1.0000001 && fabs(dy) <= 1.0000001 && fabs(dz) <= 1.0000001;
        exactLoop = !(tame && s1 < P.familyScale);
        alive = false;
        if (exactLoop) return;
        negThreshold = -(float)((64.0 * 5.9604644775390625e-08 * s1 * s1 + 1e-30) * 1.000001);
        sxf = (float)sx; syf = (float)sy; szf = (float)sz;
        dxf = (float)dx; dyf = (float)dy; dzf = (float)dz;
        double tEnter = 0, tLeave = INFINITY;
        bool in = true;
        double invX = 0, invY = 0, invZ = 0;
        auto slab = [&](double o, double d, double lo, double hi, double &inv) {
            if (fabs(d) > 1e-30) {
                double r = __builtin_amdgcn_rcp(d);
                r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
                inv = r;
                const double t0 = (lo - o) * r, t1 = (hi - o) * r;
                tEnter = fmax(tEnter, fmin(t0, t1));
                tLeave = fmin(tLeave, fmax(t0, t1));
            } else if (o < lo || o > hi) {
                in = false;
            }
        };
        slab(sx, dx, G[GRID_LO], G[GRID_HI], invX);
        slab(sy, dy, G[GRID_LO + 1], G[GRID_HI + 1], invY);
        slab(sz, dz, G[GRID_LO + 2], G[GRID_HI + 2], invZ);
        in = in && tEnter <= tLeave;
        auto enter = [&](double o, double d, double inv, double lo, double cs, double invCs, int n, int &c, int &step,
                         float &tMax, float &tDelta) {
            const double p = o + tEnter * d;
            double f = floor((p - lo) * invCs);
            f = f < 0 ? 0 : (f > (double)(n - 1) ? (double)(n - 1) : f);
            c = (int)f;
            if (inv != 0) {
                step = d > 0 ? 1 : -1;
                const double face = lo + (f + (d > 0 ? 1.0 : 0.0)) * cs;
                tMax = (float)((face - o) * inv);
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
        tStop = SHADOW ? (float)(tLimit * (1 + 1e-5) + 1e-5) : INFINITY;
        goOn = tStop;
        alive = in;
        if (alive) openCell();
    };

    auto avoided = [&](int k) {
        if (!SHADOW) return false;
        if (AVOID == 1) return k == P.lights[lightOf(task)].avoidFirst;
        if (AVOID == 2) {
            const int li = lightOf(task);
            return P.lights[li].avoidNum > 0 && P.avoidMask[(size_t)li * P.numShapes + k] != 0;
        }
        return false;
    };
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

#ifdef GRAFT_STAMPS  // diagnostic build: per kernel kind 8 counters from P.counters[16] on
    unsigned long long *stat = P.counters ? P.counters + (SHADOW ? 24 : 16) : nullptr;
    auto book = [&](int which, unsigned long long n) {
        if (stat && lane64() == (int)__builtin_ctzll(__ballot(1))) atomicAdd(stat + which, n);
    };
#define MIRROR_BOOK(which, n) book(which, n)
#else
#define MIRROR_BOOK(which, n)
#endif
    // one trip of closestHitGrid's loop for the lanes that are alive
    auto step = [&]() {
        MIRROR_BOOK(0, 1ull);
        MIRROR_BOOK(1, (unsigned long long)__popcll(__ballot(1)));
        const int k0 = kA, k1 = kB;
        const float4v e0 = aos[k0], e1 = aos[k1];
        kA = refs[ref];
        kB = refs[ref + 1];
        const bool m0 = miss(e0), m1 = miss(e1);
        int c0 = (left > 0 && k0 != last && !m0 && !avoided(k0)) ? k0 : -1;
        int c1 = (left > 1 && !m1 && !avoided(k1)) ? k1 : -1;
        if (left > 0) last = left > 1 ? k1 : k0;
        left -= 2;
        ref += 2;
        if (c0 < 0) { c0 = c1; c1 = -1; }
        const bool cellDone = left <= 0;
        // Would the lane stop after this cell as things stand?  (Decided again below, after the exact tests.)
        bool leaving = false;
        if (cellDone) {
            const float tExit = fminf(tMaxX, fminf(tMaxY, tMaxZ));
            leaving = !(tExit < goOn);
            if (!leaving) {
                if (tMaxX == tExit) leaving = (unsigned)(cx + stepX) >= (unsigned)nx;
                else if (tMaxY == tExit) leaving = (unsigned)(cy + stepY) >= (unsigned)ny;
                else leaving = (unsigned)(cz + stepZ) >= (unsigned)nz || stepZ == 0;
            }
        }
        // The exact tests run when a lane cannot do without (it has a second candidate for its one slot, or it is
        // about to stop with a candidate untested), or when kExactLanes lanes have one waiting at the end of a cell: a
        // lane that walks on with an untested candidate only visits cells it might have been spared.
        const bool must = (c0 >= 0 && pending >= 0) || c1 >= 0 || (leaving && (c0 >= 0 || pending >= 0));
        if (__ballot(must) || __popcll(__ballot(c0 >= 0 || (pending >= 0 && cellDone))) >= kExactLanes) {
            const double slackBase = fabs(sx) + fabs(sy) + fabs(sz);
            const double dd = dx * dx + dy * dy + dz * dz;
            const double slackScale = dd > 0.99 && dd < 1.01 ? 2e-15 : INFINITY;
#pragma nounroll
            for (int trip = 0; trip < 3; ++trip) {
                if (__ballot(pending >= 0 && (c0 >= 0 || cellDone))) {
                    MIRROR_BOOK(2, 1ull);
                    MIRROR_BOOK(3, (unsigned long long)__popcll(__ballot(pending >= 0)));
                    if (pending >= 0) {
                        exactSphereLazy(pendingGeom, pending, sx, sy, sz, dx, dy, dz, slackBase, slackScale, closest);
                        pending = -1;
                        if (closest.shape >= 0) {
                            const float guard = (float)(closest.t - (1e-5 * (fabs(closest.t) + slackBase) + 1e-6));
                            const double tSure = tLimit * (1 - 1e-5) - 1e-5;
                            goOn = SHADOW && closest.t < tSure ? -INFINITY : fminf(tStop, guard);
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
        } else if (c0 >= 0) {  // (the slot is free and there is no second candidate, or the tests would have run)
            pending = c0;
            pendingGeom = *(const double4v *)(geom + 4 * (size_t)c0);
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
                    stop = (unsigned)cz >= (unsigned)nz || stepZ == 0;
                }
            }
            if (stop) alive = false;
            else openCell();
        }
    };

    // what the finished ray of `task` found goes where the next launch looks for it
    auto finish = [&]() {
        Hit h;
        int removed = 0;
        if (exactLoop) {
            int avoidShape = -1;
            const unsigned char *avoidRow = nullptr;
            if (SHADOW && AVOID != 0) {
                const int li = lightOf(task);
                const DevLight &L = P.lights[li];
                if (AVOID == 1) avoidShape = L.avoidFirst;
                if (AVOID == 2 && L.avoidNum > 0) avoidRow = P.avoidMask + (size_t)li * P.numShapes;
            }
            closestHit<AVOID, false>(P, sx, sy, sz, dx, dy, dz, avoidShape, avoidRow, h);
        } else {
            finishBest(closest, sx, sy, sz, dx, dy, dz, h);
        }
        if (SHADOW && AVOID != 0) removed = P.lights[lightOf(task)].avoidNum;
        rayCount++;
        testCount += (unsigned)(P.numShapes - removed);
        if (SHADOW) {  // mainTracing.c:210-215
            bool lit = h.idx < 0;
            if (!lit) {
                const DevLight &L = P.lights[lightOf(task)];
                double bx = L.loc[0] - h.x, by = L.loc[1] - h.y, bz = L.loc[2] - h.z;
                unit3(bx, by, bz, bx, by, bz);
                lit = dot3(bx, by, bz, dx, dy, dz) < 0.1;
            }
            Q.lit[task] = lit ? 1 : 0;
        } else if (h.idx < 0) {  // mainTracing.c:314-317
            const ConstDoubles cam = cameraView(P);
            double v[3] = {cam[CAM_BACKGROUND], cam[CAM_BACKGROUND + 1], cam[CAM_BACKGROUND + 2]};
            handUp(P, Q, Q.level, pixel, v);
            Q.cand[task] = kNoTask;
        } else {
            MirrorHit *hp = Q.hits + pixel;
            hp->x = h.x; hp->y = h.y; hp->z = h.z;
            hp->shape = h.idx;
            Q.cand[task] = pixel;
        }
    };

    for (;;) {
        MIRROR_BOOK(6, 1ull);
        const unsigned long long idleMask = __ballot(!alive);
        const int idle = __popcll(idleMask);
        if (idle == 64 || (!drained && idle >= kRefillLanes)) {
            MIRROR_BOOK(4, 1ull);
            MIRROR_BOOK(5, (unsigned long long)idle);
            if (!alive && task != kNoTask) {
                finish();
                task = kNoTask;
            }
            // the idle lanes, in lane order, take the next tasks of the wave's draw; a used-up draw is replaced
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idleMask, 0));
            int served = 0;
            while (served < idle && !drained) {
                if (drawNext == drawEnd) {
                    unsigned base = 0;
                    if (lane64() == 0) base = atomicAdd(cursor, perDraw);
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    if (base >= total) {
                        drained = true;
                        break;
                    }
                    drawNext = base;
                    drawEnd = total - base < perDraw ? total : base + perDraw;
                }
                const int take = min(idle - served, (int)(drawEnd - drawNext));
                if (!alive && rank >= served && rank < served + take) task = drawNext + (unsigned)(rank - served);
                drawNext += (unsigned)take;
                served += take;
            }
            if (!alive && task != kNoTask) begin();
        }
        if (!__ballot(alive)) {
            if (drained && !__ballot(task != kNoTask)) break;
            continue;
        }
        if (alive) step();
    }

    if (P.counters) {
        for (int off = 32; off > 0; off >>= 1) {
            rayCount += __shfl_down(rayCount, off);
            testCount += __shfl_down(testCount, off);
        }
        if (lane64() == 0) {
            atomicAdd(P.counters, rayCount);
            atomicAdd(P.counters + 1, testCount);
        }
    }
}

// mainTracing.c:318-369 for the hits of one level (one hit per lane): normal, diffuse weight of the last light that
// reaches the hit, and then either the next level's mirror ray or the hit's value handed up to its pixel.
__global__ void __launch_bounds__(256) mirrorShadeKernel(const TraceParams P, const MirrorQueue Q) {
    const unsigned hits = Q.counts[kCountsPerLevel * Q.level + MQ_HITS];
    const MirrorRecord *records = Q.records + (size_t)(Q.level - 1) * Q.cap;
    for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < hits; j += gridDim.x * blockDim.x) {
        const unsigned i = Q.hitList[j];  // the pixel
        const MirrorHit *hp = Q.hits + i;
        const double loc[3] = {hp->x, hp->y, hp->z};
        const int shape = hp->shape;
        const double *c = P.geom + 4 * (size_t)shape;  // only spheres reach this path, in shape order
        double nrm[3] = {0, 0, 0};
        unit3(loc[0] - c[0], loc[1] - c[1], loc[2] - c[2], nrm[0], nrm[1], nrm[2]);  // shapes.c:69-71
        int litLight = -1;
        for (int li = 0; li < P.numLights; ++li)
            if (Q.lit[(size_t)j * P.numLights + li]) litLight = li;  // the last unshadowed light overwrites, :223-225
        double litW = 0;
        if (litLight >= 0) {
            const DevLight &L = P.lights[litLight];
            double sdx = 0, sdy = 0, sdz = 0;
            unit3(L.loc[0] - loc[0], L.loc[1] - loc[1], L.loc[2] - loc[2], sdx, sdy, sdz);
            const double w = dot3(sdx, sdy, sdz, nrm[0], nrm[1], nrm[2]);
            litW = w < 0 ? 0 : w;
        }
        const DevMaterial &m = P.materials[shape];
        const bool mirror = m.reflectivity > 0;  // :330 (every ray of this path has shouldReflect = 1)
        if (mirror && Q.level + 1 < P.maxDepth) {  // reflection(), mainTracing.c:180-190
            const double din[3] = {records[i].d[0], records[i].d[1], records[i].d[2]};
            const double k2 = 2 * dot3(din[0], din[1], din[2], nrm[0], nrm[1], nrm[2]);
            double rx = din[0] - k2 * nrm[0], ry = din[1] - k2 * nrm[1], rz = din[2] - k2 * nrm[2];
            unit3(rx, ry, rz, rx, ry, rz);
            MirrorRecord *r = Q.records + (size_t)Q.level * Q.cap + i;
            r->d[0] = rx; r->d[1] = ry; r->d[2] = rz;
            r->o[0] = loc[0] + .000001 * rx;
            r->o[1] = loc[1] + .000001 * ry;
            r->o[2] = loc[2] + .000001 * rz;
            r->litW = litW;
            r->shapeLight = (unsigned)shape | (unsigned)(litLight + 1) << 24;
            Q.cand[j] = i;
        } else {
            double part[3], v[3];
            localPart(P, m, litW, litLight, part);
            const double wRefl = (1 - m.ambientLight) * m.reflectivity;
            for (int ch = 0; ch < 3; ++ch) {
                if (mirror) {  // the mirror child returns black at once (mainTracing.c:305-306)
                    const double rgb = m.color[ch];
                    const double t = (1 - .5) * (1.0 - rgb);
                    double mir = (rgb + t) * 0.0;
                    mir = wRefl * mir;
                    const double out = mir + part[ch];
                    v[ch] = (0.0 * 0.0) + out;
                } else {  // no children: reflectionRGB and refractionRGB stay {0,0,0}
                    const double mir = wRefl * 0.0;
                    double out = mir + part[ch];
                    out = (0.0 * 0.0) + out;
                    v[ch] = out;
                }
            }
            handUp(P, Q, Q.level, i, v);
            Q.cand[j] = kNoTask;
        }
    }
}

}  // namespace graft
